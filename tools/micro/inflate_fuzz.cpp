// Development tool: host/inflate.hpp under AddressSanitizer + UBSan on corrupted gzip files (truncations, bit flips, garbage runs, broken block headers), every input in an exact-size heap copy.
// build: g++ -O1 -g -march=x86-64-v3 -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/inflate_fuzz tools/micro/inflate_fuzz.cpp -lz -pthread
// run:   /tmp/inflate_fuzz <file.gz> <seed> <iterations> [threads]   (iterations 0: decode the file as it is).  Round 5: 1800 corrupted inputs refused, pristine one- and multi-member files decoded, no report.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstring>
#include <random>
#include <zlib.h>
#include <thread>
#include "../../savont_amd/csrc/host/inflate.hpp"
// round 6: argv[4] = threads (>= 2: one member on several threads, host/inflate.hpp: inflate_member_parallel; the jobs run on plain std::threads here)
static void run_jobs(size_t n, void (*f)(size_t, void*), void* ctx) { std::vector<std::thread> th; for (size_t i = 0; i < n; i++) th.emplace_back([=] { f(i, ctx); }); for (auto& t : th) t.join(); }
int main(int argc, char** argv) {
    const unsigned threads = argc > 4 ? (unsigned)atoi(argv[4]) : 1;
    savont::gz::par_hooks().run = run_jobs; savont::gz::par_hooks().threads = threads;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> src(n); if (fread(src.data(), 1, n, f) != n) return 1; fclose(f);
    std::mt19937_64 rng(atoi(argv[2]));
    int iters = atoi(argv[3]); size_t ok = 0, bad = 0;
    if (iters == 0) { savont::gz::BigBuf out; size_t len = 0; std::string why; unsigned char* h = (unsigned char*)malloc(n); memcpy(h, src.data(), n); bool r = savont::gz::gunzip_all(h, n, out, len, why, threads); printf("pristine: %d len %zu %s\n", r, len, why.c_str()); free(h); return r ? 0 : 2; }
    for (int it = 0; it < iters; it++) {
        std::vector<unsigned char> x(src);
        int mode = rng() % 4;
        if (mode == 0) { size_t cut = rng() % x.size(); x.resize(cut); }                                   // truncation
        else if (mode == 1) { for (int k = 0; k < 1 + (int)(rng() % 8); k++) x[rng() % x.size()] ^= (unsigned char)(1u << (rng() % 8)); }   // bit flips
        else if (mode == 2) { size_t a = rng() % x.size(), len = 1 + rng() % 4096; for (size_t i = a; i < std::min(x.size(), a + len); i++) x[i] = (unsigned char)rng(); }   // a garbage run
        else { size_t a = 10 + rng() % 64; for (size_t i = a; i < std::min(x.size(), a + 8); i++) x[i] = (unsigned char)rng(); }   // the first block header
        // exact-size heap copy: reads past the end are caught by the sanitizer
        unsigned char* h = (unsigned char*)malloc(x.size() ? x.size() : 1); memcpy(h, x.data(), x.size());
        savont::gz::BigBuf out; size_t len = 0; std::string why;
        bool r = savont::gz::gunzip_all(h, x.size(), out, len, why, threads);
        (r ? ok : bad)++;
        free(h);
    }
    printf("seed %s: %zu decoded, %zu refused\n", argv[2], ok, bad);
}
