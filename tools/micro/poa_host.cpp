// Phase timing of the host POA (savont_amd/csrc/host/poa.hpp) on one synthetic cluster, single thread, no GPU:
//   g++ -O3 -std=c++17 -I include -I savont_amd/csrc/host tools/micro/poa_host.cpp -o /tmp/poa_host && /tmp/poa_host
#include <chrono>
#include <cstdio>
#include <random>
#define POA_PHASE_TIMING 1
#include "poa.hpp"
using namespace savont;
int main(int argc, char** argv) {
    const int n = 75, L = 1500; const double err = 0.015;
    std::mt19937_64 rng(5);
    auto U = [&] { return (double)(rng() >> 11) / (double)(1ull << 53); };
    const char ACGT[5] = "ACGT";
    std::vector<uint8_t> hap(L); for (auto& b : hap) b = ACGT[rng() & 3];
    std::vector<std::vector<uint8_t>> seqs(n); std::vector<std::vector<uint32_t>> w(n);
    for (int r = 0; r < n; r++) {
        for (uint8_t b : hap) { double x = U(); if (x < err / 3) continue; if (x < 2 * err / 3) seqs[r].push_back(ACGT[rng() & 3]); seqs[r].push_back(x >= err ? b : (uint8_t)ACGT[rng() & 3]); }
        w[r].resize(seqs[r].size()); for (auto& q : w[r]) q = 5 + (uint32_t)(rng() % 35);
    }
    size_t tot = 0; for (auto& s : seqs) tot += s.size();
    const size_t ref_len = tot / n; uint32_t max_dev = 0;
    for (auto& s : seqs) max_dev = std::max<uint32_t>(max_dev, (uint32_t)std::llabs((long long)ref_len - (long long)s.size()));
    double t_align = 0, t_add = 0, t_cons = 0; uint64_t digest = 0;
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    for (int rep = 0; rep < reps; rep++) {
        PoaGraph g;
        for (int r = 0; r < n; r++) {
            auto t0 = std::chrono::steady_clock::now();
            auto aln = g.align(seqs[r], max_dev, 0.1);
            auto t1 = std::chrono::steady_clock::now();
            g.add_alignment(aln, seqs[r], w[r]);
            auto t2 = std::chrono::steady_clock::now();
            t_align += std::chrono::duration<double>(t1 - t0).count(); t_add += std::chrono::duration<double>(t2 - t1).count();
        }
        auto t0 = std::chrono::steady_clock::now();
        auto c = g.consensus();
        t_cons += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (uint8_t b : c) digest = digest * 1000003u + b;
        if (rep == 0) printf("consensus %zu bases, %zu nodes, cells %.1f M rows %.1f k\n", c.size(), g.nodes.size(), g.cells_done / 1e6, g.rows_done / 1e3);
    }
    printf("per consensus: align %.2f ms, add_alignment+sort %.2f ms, consensus %.3f ms   digest %llx\n", t_align / reps * 1e3, t_add / reps * 1e3, t_cons / reps * 1e3, (unsigned long long)digest);
    if (argc > 2) {   // experiment: every row computed 1 + extra times (identical results): what does a row cost when its inputs are hot?
        for (int extra : {0, 1, 3}) {
            g_poa_exp = extra << 8; g_poa_phase[1] = 0;
            PoaGraph g;
            for (int r = 0; r < n; r++) g.add_alignment(g.align(seqs[r], max_dev, 0.1), seqs[r], w[r]);
            printf("  rows with %d extra passes per row: %.2f ms\n", extra, g_poa_phase[1] * 1e3);
        }
        g_poa_exp = 0;
    }
    printf("  blocks per consensus: fast %llu, general %llu (predecessor visits %llu; %llu of them in multi-predecessor rows)\n", g_poa_cnt[0] / reps, g_poa_cnt[1] / reps, g_poa_cnt[2] / reps, g_poa_cnt[3] / reps);
    printf("  align: set-up %.2f ms, rows %.2f ms, traceback %.2f ms\n", g_poa_phase[0] / reps * 1e3, g_poa_phase[1] / reps * 1e3, g_poa_phase[2] / reps * 1e3);
    return 0;
}
