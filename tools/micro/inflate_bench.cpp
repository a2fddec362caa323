// Development tool: single-thread pace of host/inflate.hpp on one .gz file (best of N runs), for trying decoder variants on the CPU.
// build: g++ -O3 -march=x86-64-v3 -std=c++17 -o /tmp/inflate_bench tools/micro/inflate_bench.cpp -lz -pthread ; run: /tmp/inflate_bench <file.gz> [runs]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstring>
#include <chrono>
#include <zlib.h>
#include "../../savont_amd/csrc/host/inflate.hpp"
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1; fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> src(n); if (fread(src.data(), 1, n, f) != n) return 1; fclose(f);
    const int runs = argc > 2 ? atoi(argv[2]) : 5;
    double best = 1e9; size_t len = 0; unsigned long long dig = 0;
    savont::gz::BigBuf out;
    for (int r = 0; r < runs; r++) {
        std::string why; auto t0 = std::chrono::steady_clock::now();
        if (!savont::gz::gunzip_all(src.data(), n, out, len, why, 1)) { printf("failed: %s\n", why.c_str()); return 2; }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); if (s < best) best = s;
    }
    for (size_t i = 0; i < len; i++) dig = dig * 1099511628211ull + out.p[i];
    printf("%zu -> %zu bytes, best of %d: %.3f s = %.1f MB/s, digest %llx\n", n, len, runs, best, len / best / 1e6, dig);
    return 0;
}
