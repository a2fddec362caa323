// Shape of the partial-order graphs that K12's anti-diagonal engine sweeps (kernels_poa_graph.hip, ENG = 2): in-degree distribution, distance (in rows of the
// topological order) from a row to its predecessors, references that cross a 64-row block, column drift per block.  Host engine only (poa.hpp), no GPU:
//   g++ -O2 -std=c++17 -march=native -I include -I savont_amd/csrc/host tools/micro/poa_graph_stats.cpp -o /tmp/poa_graph_stats && /tmp/poa_graph_stats 0.015 1500
// (error rate, read length; 75 reads).  The numbers of profiles/r04_poa.md section 5 come from it.
#include <cstdio>
#include <random>
#include <map>
#include "poa.hpp"
using namespace savont;
int main(int argc, char** argv) {
    const int n = 75, L = argc > 2 ? atoi(argv[2]) : 1500; const double err = argc > 1 ? atof(argv[1]) : 0.015;
    std::mt19937_64 rng(5);
    auto U = [&] { return (double)(rng() >> 11) / (double)(1ull << 53); };
    const char ACGT[5] = "ACGT";
    std::vector<uint8_t> hap(L); for (auto& b : hap) b = ACGT[rng() & 3];
    std::vector<std::vector<uint8_t>> seqs(n); std::vector<std::vector<uint32_t>> w(n);
    for (int r = 0; r < n; r++) {
        for (uint8_t b : hap) { double x = U(); if (x < err / 3) continue; if (x < 2 * err / 3) seqs[r].push_back(ACGT[rng() & 3]); seqs[r].push_back(x >= err ? b : (uint8_t)ACGT[rng() & 3]); }
        w[r].resize(seqs[r].size()); for (auto& q : w[r]) q = 5 + (uint32_t)(rng() % 35);
    }
    PoaGraph g;
    std::map<int, long> npd, dist; long rows = 0, nonchain_rows = 0, refs = 0, np2 = 0, np3 = 0, np0 = 0, cross = 0, mem_refs = 0, two_mem = 0, three_mem = 0, exported = 0, far64 = 0, far128 = 0;
    long blocks = 0, exp_per_block_max = 0, colspan_sum = 0, colspan_max = 0, nonmono = 0;
    for (int r = 0; r < n; r++) {
        auto aln = g.align(seqs[r], 20, 0.1);
        const int N = (int)g.rank.size();
        if (r >= 1) {
            std::vector<char> is_exp(N + 1, 0);
            for (int i = 1; i <= N; i++) {
                const uint32_t nd = g.rank[i - 1];
                rows++;
                const int np = (int)g.n_in_cnt_[nd];
                npd[std::min(np, 8)]++;
                if (np == 0) np0++;
                int mem = 0;
                for (uint32_t e : g.nodes[nd].in) { const int p = g.w_row_of_[g.edges[e].tail]; const int d = i - p; refs++; dist[std::min(d, 200)]++; if (d != 1 || ((i - 1) & 63) == 0) { mem++; mem_refs++; is_exp[p] = 1; if (((i - 1) >> 6) != ((p - 1) >> 6)) cross++; if (d >= 64) far64++; if (d >= 128) far128++; } }
                if (mem >= 2) two_mem++; if (mem >= 3) three_mem++;
                if (np != 1 || g.w_row_of_[g.edges[g.nodes[nd].in[0]].tail] != i - 1) nonchain_rows++;
            }
            for (int b = 0; b * 64 < N; b++) { long e = 0; int cmin = 1 << 30, cmax = 0; int prevc = -1; for (int i = b * 64 + 1; i <= std::min(N, b * 64 + 64); i++) { e += is_exp[i]; const int c = g.n_col_[g.rank[i - 1]]; cmin = std::min(cmin, c); cmax = std::max(cmax, c); if (c < prevc) nonmono++; prevc = c; } exported += e; exp_per_block_max = std::max(exp_per_block_max, e); blocks++; colspan_sum += cmax - cmin; colspan_max = std::max<long>(colspan_max, cmax - cmin); }
        }
        g.add_alignment(aln, seqs[r], w[r]);
    }
    printf("rows %ld (%.0f per read), np0 %ld; rows not (np==1 && pred==i-1): %.2f %%\n", rows, (double)rows / (n - 1), np0, 100.0 * nonchain_rows / rows);
    printf("np distribution:"); for (auto& kv : npd) printf(" %d:%.2f%%", kv.first, 100.0 * kv.second / rows); printf("\n");
    printf("pred refs %ld, via memory (not lane-1 DPP) %.2f %% of rows-equivalents; rows with >=2 memory preds %.3f %%, >=3 %.3f %%\n", refs, 100.0 * mem_refs / rows, 100.0 * two_mem / rows, 100.0 * three_mem / rows);
    printf("memory refs crossing a 64-row block: %.2f %% of mem refs; dist>=64 %.3f %%, >=128 %.3f %% of mem refs\n", 100.0 * cross / mem_refs, 100.0 * far64 / mem_refs, 100.0 * far128 / mem_refs);
    printf("exported rows per block: mean %.1f max %ld; column span per block mean %.1f max %ld; non-monotone c steps %.2f %% of rows\n", (double)exported / blocks, exp_per_block_max, (double)colspan_sum / blocks, colspan_max, 100.0 * nonmono / rows);
    printf("dist histogram:"); for (auto& kv : dist) if (kv.second * 1000 > refs || kv.first >= 32) printf(" %d:%.2f%%", kv.first, 100.0 * kv.second / refs); printf("\n");
    return 0;
}
