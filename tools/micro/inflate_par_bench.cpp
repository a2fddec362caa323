// Development tool: where the time of ONE gzip member inflated on several threads goes (host/inflate.hpp: inflate_member_parallel) -- block starts, pieces, markers, CRC.
// build: g++ -O3 -march=x86-64-v3 -std=c++17 -DGZ_PHASE_TIMING -o /tmp/inflate_par_bench tools/micro/inflate_par_bench.cpp -lz -pthread ; run: /tmp/inflate_par_bench <file.gz> <threads> [runs]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <cstring>
#include <ctime>
#include <zlib.h>
#include "../../savont_amd/csrc/host/inflate.hpp"
static unsigned g_threads = 8;
static void run_jobs(size_t n, void (*f)(size_t, void*), void* ctx) {
    std::atomic<size_t> next{0}; std::vector<std::thread> th;
    const unsigned T = (unsigned)std::min<size_t>(g_threads, n);
    auto body = [&]() { for (;;) { const size_t j = next.fetch_add(1); if (j >= n) return; f(j, ctx); } };
    for (unsigned t = 1; t < T; t++) th.emplace_back(body);
    body();
    for (auto& t : th) t.join();
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1; fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> src(n); if (fread(src.data(), 1, n, f) != n) return 1; fclose(f);
    g_threads = argc > 2 ? atoi(argv[2]) : 8; const int runs = argc > 3 ? atoi(argv[3]) : 3;
    savont::gz::par_hooks().run = run_jobs; savont::gz::par_hooks().threads = g_threads;
    savont::gz::BigBuf out; size_t len = 0;
    for (int r = 0; r < runs; r++) {
        for (double& x : savont::gz::g_gz_phase) x = 0;
        std::string why; const double t0 = savont::gz::gz_now();
        if (!savont::gz::gunzip_all(src.data(), n, out, len, why, g_threads)) { printf("failed: %s\n", why.c_str()); return 2; }
        const double s = savont::gz::gz_now() - t0; const double* p = savont::gz::g_gz_phase;
        printf("%u threads: %.3f s = block starts %.3f + pieces %.3f + markers %.3f + crc %.3f + rest %.3f  (%zu bytes)\n", g_threads, s, p[0], p[1], p[2], p[3], s - p[0] - p[1] - p[2] - p[3], len);
    }
    return 0;
}
