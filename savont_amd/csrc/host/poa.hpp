// poa.hpp -- partial order alignment consensus (host, CPU) for Stage 4a (src/alignment.rs:193-231).
//
// The reference calls spoars 0.1.3 (a Rust port of spoa) with Scoring(3, -8, -6, -6, 0, 0) -- gap open == gap extend, i.e.
// LINEAR gaps --, AlignmentType::Overlap and a band of (max length deviation) + 0.1 * len.  spoars is a third-party crate
// absent from the reference tree, so this is a restatement of the published spoa algorithm (Vaser et al. 2017; Lee 2002):
//   * graph of nodes (base) / weighted edges, "aligned" node sets for mismatching columns;
//   * sequence-to-graph DP over the nodes in topological order, banded around the mean sequence position of the bases fused
//     into the node (band_column below; spoa proper has no band, the band's anchor is this restatement's choice);
//   * overlap mode: leading and trailing overhangs of the sequence AND of the graph are free;
//   * traceback priority: (mis)match, then deletion (graph node without base), then insertion;
//   * add_alignment fuses the path (reuse equal-letter nodes / aligned siblings, else new node), edge weight += w[i-1]+w[i];
//   * consensus = heaviest bundle with branch completion.
// Parity with spoars is UNPINNED; correctness is checked by properties (tests/test_consensus.py).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "savont_hip.h"

namespace savont {
#ifdef POA_PHASE_TIMING
inline double g_poa_phase[4]; inline int g_poa_exp = 0; inline unsigned long long g_poa_cnt[8];   // g_poa_exp: ablation switches of the microbenchmark (wrong results, timing only)      // tools/micro/poa_host.cpp: set-up, rows, traceback (seconds)
inline double poa_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define POA_T(x) const double x = poa_now()
#define POA_ACC(k, a, b) g_poa_phase[k] += (b) - (a)
#else
#define POA_T(x)
#define POA_ACC(k, a, b)
#endif

// inner loop of the sequence-to-graph DP: tmp[j] = max(tmp[j], P[j-1] + sc[j], P[j] + G) for j in [a, b]; runtime-dispatched SIMD clones
__attribute__((target_clones("avx512f", "avx2", "default")))
inline void poa_relax(int* __restrict tmp, const int* __restrict P, const int* __restrict sc, int G, int delta, int a, int b) {
    G += delta;
    for (int j = a; j <= b; j++) {
        const int d = P[j - 1] + sc[j] + delta, u = P[j] + G;
        const int m = d > u ? d : u;
        tmp[j] = tmp[j] > m ? tmp[j] : m;
    }
}
#define SAVONT_RELAX16_BODY  /* int arithmetic: a candidate below the int16 range loses to tmp (>= neg), none exceeds it */ \
    G += delta;                                                                               \
    for (int j = a; j <= b; j++) {                                                            \
        const int d = (int)P[j - 1] + (int)sc[j] + delta, u = (int)P[j] + G;                  \
        const int m = d > u ? d : u;                                                          \
        tmp[j] = (int)tmp[j] > m ? tmp[j] : (int16_t)m;                                       \
    }
__attribute__((target("avx512f,avx512bw,avx512vl"))) inline void poa_relax16_avx512(int16_t* __restrict tmp, const int16_t* __restrict P, const int16_t* __restrict sc, int G, int delta, int a, int b) { SAVONT_RELAX16_BODY }
__attribute__((target("avx2"))) inline void poa_relax16_avx2(int16_t* __restrict tmp, const int16_t* __restrict P, const int16_t* __restrict sc, int G, int delta, int a, int b) { SAVONT_RELAX16_BODY }
inline void poa_relax16_base(int16_t* __restrict tmp, const int16_t* __restrict P, const int16_t* __restrict sc, int G, int delta, int a, int b) { SAVONT_RELAX16_BODY }
#undef SAVONT_RELAX16_BODY
inline void poa_relax(int16_t* __restrict tmp, const int16_t* __restrict P, const int16_t* __restrict sc, int G, int delta, int a, int b) {
    typedef void (*fn_t)(int16_t*, const int16_t*, const int16_t*, int, int, int, int);
    static const fn_t fn = __builtin_cpu_supports("avx512bw") ? (fn_t)poa_relax16_avx512 : __builtin_cpu_supports("avx2") ? (fn_t)poa_relax16_avx2 : (fn_t)poa_relax16_base;
    fn(tmp, P, sc, G, delta, a, b);
}
// Cells are stored in the "ramped" frame R(i, j) = H(i, j) - G*j (G < 0: R = H + 6j).  In that frame the insertion chain
// H[j] = max(tmp[j], H[j-1] + G) is a plain prefix maximum R[j] = max(tmpR[j], R[j-1]), a deletion is still P[j] + G and a
// (mis)match is P[j-1] + (sc[j] - G): the profile carries the -G.  Floors: every candidate is >= neg in the R frame.
// 16-bit rows are additionally stored relative to a per-row base (9 * first column of the row, about R on the diagonal there), so
// that the stored values depend on the band width and not on the sequence length; a predecessor row then contributes with the
// constant delta = base(pred) - base(row) added to both of its candidates.
template <class S> inline void poa_scan(S* __restrict row, const S* __restrict tmp, int first, int a, int b) {
    int m = first;
    for (int j = a; j <= b; j++) { const int u = (int)tmp[j]; m = m > u ? m : u; row[j] = (S)m; }
}

// One row of the DP, 16-bit cells, 32 cells per step (AVX-512BW), candidates and insertion chain in ONE pass:
//   tmp[j] = max(neg, max over the predecessor rows P covering j of (P[j-1] + sc[j], P[j] + G))      (registers only)
//   row[j] = max(tmp[j], row[j-1])                                                                    (prefix maximum, R frame)
// A predecessor row contributes on [a, b] = its band widened by one to the right (P[lo-1], P[hi+1] are sentinels); lanes outside
// stay at neg, exactly as the per-predecessor poa_relax passes over a NEG-initialised tmp leave them.  The adds cannot wrap
// (P >= neg = -30000, sc + delta and G + delta stay small: |delta| <= 16000 is checked by the caller; stored values <= 9 * band width).  The prefix maximum is five in-register lane shifts (valignd moves whole
// dwords: one shuffle uop; the odd shift and the carry broadcast use vpermw), the carry between blocks stays in a vector.  The
// last partial block runs under a lane mask (a prefix maximum is causal: lanes beyond the row never reach the stored ones).
struct PoaPred16 { const int16_t* P; int a, b, delta; };
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw"))) inline void poa_row16_avx512(int16_t* __restrict row, const PoaPred16* __restrict preds, int np, const int16_t* __restrict sc,
                                                                         int first, int G, int neg, int a, int b) {
    const __m512i NEGV = _mm512_set1_epi16((short)-32768);
    alignas(64) static const short SHR1[32] = {0,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30};
    const __m512i floorv = _mm512_set1_epi16((short)neg), gv = _mm512_set1_epi16((short)G), i1 = _mm512_load_si512(SHR1), last = _mm512_set1_epi16(31);
    __m512i carry = _mm512_set1_epi16((short)(first < -32768 ? -32768 : first));
    // prefix maximum of one block (five in-register shifts), then the running maximum of the blocks to the left
#define SAVONT_POA_FINISH(x, STORE)                                                            \
    x = _mm512_max_epi16(x, _mm512_mask_permutexvar_epi16(NEGV, 0xFFFFFFFEu, i1, x));          \
    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 15));                                  \
    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 14));                                  \
    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 12));                                  \
    x = _mm512_max_epi16(x, _mm512_alignr_epi32(x, NEGV, 8));                                   \
    x = _mm512_max_epi16(x, carry);                                                             \
    carry = _mm512_permutexvar_epi16(last, x);                                                  \
    STORE;
    // the columns every predecessor covers: whole blocks inside [in_lo, in_hi] need no masks and no lane bookkeeping
    int in_lo = a, in_hi = b;
    for (int p = 0; p < np; p++) { in_lo = std::max(in_lo, preds[p].a); in_hi = std::min(in_hi, preds[p].b); }
    int j = a;
    // When every predecessor covers the whole row (the usual case: a row's band is its predecessor's shifted by a column), the last block may
    // run past b: its extra lanes read and write the padding behind the rows (align_impl leaves a block of slack and rewrites the right
    // sentinel afterwards), they sit to the RIGHT of every real cell, so they never reach one through the prefix maximum -- and the whole row
    // is the same unmasked block, with no ragged tail.
    const bool overrun = in_lo <= a && b <= in_hi;
    while (j <= b) {
        if (overrun || (j >= in_lo && j + 31 <= in_hi)) {
            const __m512i scv = _mm512_loadu_si512(sc + j);
            __m512i x;
            if (np == 1) {                                                              // six rows of ten
                const int16_t* P = preds[0].P;
                const __m512i d = _mm512_adds_epi16(_mm512_loadu_si512(P + j - 1), scv), u = _mm512_adds_epi16(_mm512_loadu_si512(P + j), gv);
                x = _mm512_max_epi16(floorv, _mm512_adds_epi16(_mm512_max_epi16(d, u), _mm512_set1_epi16((short)preds[0].delta)));
            } else {                                                                    // the node that closes a bubble: one pass per in-edge, same candidates
                x = floorv;
                for (int p = 0; p < np; p++) {
                    const int16_t* P = preds[p].P;
                    const __m512i d = _mm512_adds_epi16(_mm512_loadu_si512(P + j - 1), scv), u = _mm512_adds_epi16(_mm512_loadu_si512(P + j), gv);
                    x = _mm512_max_epi16(x, _mm512_adds_epi16(_mm512_max_epi16(d, u), _mm512_set1_epi16((short)preds[p].delta)));
                }
            }
#ifdef POA_PHASE_TIMING
            g_poa_cnt[0]++;
#endif
            SAVONT_POA_FINISH(x, _mm512_storeu_si512(row + j, x))
            j += 32;
            continue;
        }
#ifdef POA_PHASE_TIMING
        g_poa_cnt[1]++; g_poa_cnt[2] += (unsigned)np; if (np > 1) g_poa_cnt[3]++;
#endif
        // ragged block: lanes of the row up to b, per predecessor the lanes it covers
        const int rem = b - j + 1;
        const __mmask32 k = rem >= 32 ? (__mmask32)0xFFFFFFFFu : (__mmask32)((1u << rem) - 1u);
        const __m512i scv = _mm512_maskz_loadu_epi16(k, sc + j);
        __m512i x = floorv;
        for (int p = 0; p < np; p++) {
            const int l0 = preds[p].a - j, l1 = preds[p].b - j;                       // lanes of this block the predecessor covers
            if (l1 < 0 || l0 > 31) continue;
            __mmask32 kp = k;
            if (l0 > 0) kp &= (__mmask32)(0xFFFFFFFFu << l0);
            if (l1 < 31) kp &= (__mmask32)(0xFFFFFFFFu >> (31 - l1));
            const int16_t* P = preds[p].P;
            // P + sc and P + G cannot saturate (P >= neg, small addends); adding delta afterwards saturates exactly where the sum would
            const __m512i dv = _mm512_set1_epi16((short)preds[p].delta);
            const __m512i d = _mm512_adds_epi16(_mm512_maskz_loadu_epi16(kp, P + j - 1), scv);
            const __m512i u = _mm512_adds_epi16(_mm512_maskz_loadu_epi16(kp, P + j), gv);
            x = _mm512_mask_max_epi16(x, kp, x, _mm512_adds_epi16(_mm512_max_epi16(d, u), dv));
        }
        SAVONT_POA_FINISH(x, _mm512_mask_storeu_epi16(row + j, k, x))               // a block cut short by b is the last one: its carry is never used
        j += 32;
    }
#undef SAVONT_POA_FINISH
}
#endif
inline bool poa_row16_dispatch(int16_t* __restrict row, const PoaPred16* preds, int np, const int16_t* __restrict sc, int first, int G, int neg, int a, int b) {
#if defined(__x86_64__)
    static const bool has512 = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f");
    if (has512 && a <= b) { poa_row16_avx512(row, preds, np, sc, first, G, neg, a, b); return true; }
#endif
    return false;
}

class PoaGraph {
public:
    struct Node { uint8_t code; std::vector<uint32_t> in, out, aligned; };   // in/out hold EDGE ids; the positions of the fused bases live in the flat arrays below
    struct Edge { uint32_t tail, head; int64_t weight; };
    std::vector<Node> nodes;
    std::vector<Edge> edges;
    std::vector<uint32_t> rank;                      // topological order, aligned nodes adjacent
    // what the row loop of align_impl needs of a node, kept flat and up to date by add_node / note_position / add_edge: the per-alignment set-up
    // walked the Node structs (three heap vectors each) twice and divided 64-bit integers per node -- 40 % of an alignment's time once the
    // DP itself ran on AVX-512
    std::vector<int32_t> n_col_;                     // band_column(node)
    std::vector<int32_t> n_first_in_, n_second_in_;  // tail nodes of the first two in-edges (-1: none): nodes with at most two in-edges never touch their Node
    std::vector<uint32_t> n_in_cnt_, n_out_cnt_, n_al_cnt_;   // in-edges, out-edges, aligned siblings
    std::vector<int32_t> n_first_out_head_; std::vector<uint32_t> n_first_out_edge_;   // head node and edge id of the first out-edge (-1: none)
    std::vector<uint8_t> n_ci_;                      // 0..3 for A C G T (anything else counts as T, as idx() below)
    std::vector<uint8_t> n_code_;                    // the node's letter
    std::vector<uint64_t> n_pos_sum_; std::vector<uint32_t> n_pos_n_;   // sum and count of the 1-based sequence positions of the bases fused into the node
    mutable uint64_t cells_done = 0, rows_done = 0;  // DP volume of all align() calls (tracing)
    struct RowMeta { int p0, p1; uint32_t np; uint8_t ci, sink; };   // p0, p1: rows of the first two in-edges' tails
    mutable std::vector<int> w_row_of_, w_coord_, w_lo_, w_hi_, w_base_; mutable std::vector<size_t> w_off_; mutable std::vector<uint64_t> w_prof_, w_tmp_; mutable std::vector<RowMeta> w_meta_;   // align_impl work arrays
    std::vector<int> own_scratch_;
    std::vector<int>* scratch_ = &own_scratch_;      // DP matrix, reused across align() calls; callers may lend a long-lived buffer
    void use_scratch(std::vector<int>* s) { scratch_ = s ? s : &own_scratch_; }
    bool wide_cells = false;                         // Tuning::poa_cells == 32

    // alignment: pairs (node id or -1, sequence position or -1)
    typedef std::vector<std::pair<int32_t, int32_t>> Alignment;

    // 16-bit cells when every reachable score fits (halves the DP traffic, which is what bounds ~100 concurrent clusters)
    Alignment align(const std::vector<uint8_t>& seq, uint32_t band_base, double band_frac) const {
        const bool wide = wide_cells;                                                 // the plain int32 DP (tests: the SIMD 16-bit paths must agree with it)
        // 16-bit cells while the source row (6 per column) and a row's span (9 per band column) fit; the plain int32 DP otherwise
        const long bw = (long)band_base + (long)(band_frac * (double)seq.size()) + 1;
        const bool fits16 = seq.size() <= 5400 && 9 * (2 * bw + 2) < 30000;
        return (fits16 && !wide) ? align_impl<int16_t>(seq, band_base, band_frac, -30000) : align_impl<int>(seq, band_base, band_frac, -(1 << 28));
    }
    template <class S> Alignment align_impl(const std::vector<uint8_t>& seq, uint32_t band_base, double band_frac, const int NEG) const {
        Alignment out;
        const int L = (int)seq.size(), N = (int)rank.size();
        if (N == 0 || L == 0) return out;
        POA_T(t_a);
        const int M = 3, X = -8, G = -6;
        const int bw = (int)band_base + (int)(band_frac * L) + 1;
        // per row, in ONE pass over the rank (a predecessor's row precedes its successors', so row_of is known when meta needs it): the band
        // [lo, hi] around the column the node is expected to align with (band_column: the mean position of the bases already fused into it),
        // the row's offset in the matrix (one NEG sentinel on either side: cell (i, j) lives at H[off[i] + (j - lo[i]) + 1]), the base of
        // its stored values (16-bit rows only) and the flat facts the row loop needs of the node
        std::vector<int>& row_of = w_row_of_;                                           // per-graph work arrays, reused across reads
        std::vector<int>& lo = w_lo_; std::vector<int>& hi = w_hi_; std::vector<size_t>& off = w_off_; std::vector<int>& base = w_base_;
        std::vector<RowMeta>& meta = w_meta_;
        row_of.resize(nodes.size());                                                    // every node is in rank: row_of is fully overwritten
        lo.resize(N + 1); hi.resize(N + 1); off.resize(N + 2); base.resize(N + 1); meta.resize(N + 1);
        lo[0] = 0; hi[0] = L; off[0] = 0; off[1] = (size_t)L + 3; base[0] = 0;
        {
            size_t o = off[1];
            for (int i = 1; i <= N; i++) {
                const uint32_t nd = rank[i - 1];
                row_of[nd] = i;
                const int c = n_col_[nd];
                const int l = std::min(L, std::max(0, c - bw)), h = std::min(L, c + bw);
                lo[i] = l; hi[i] = h; base[i] = sizeof(S) == 2 ? (M - G) * l : 0;
                o += (size_t)(h - l + 3); off[i + 1] = o;
                meta[i] = RowMeta{n_first_in_[nd] < 0 ? 0 : row_of[n_first_in_[nd]], n_in_cnt_[nd] < 2 ? 0 : row_of[n_second_in_[nd]], n_in_cnt_[nd], n_ci_[nd], (uint8_t)(n_out_cnt_[nd] == 0 ? 1 : 0)};
            }
        }
        cells_done += off[N + 1]; rows_done += (uint64_t)N;
        const size_t need = ((off[N + 1] + 64) * sizeof(S) + sizeof(int) - 1) / sizeof(int);     // + a block of slack behind the last row (row kernel overrun)
        if (scratch_->size() < need) { std::vector<int>().swap(*scratch_); scratch_->resize(need + need / 2); }   // grow without copying
        S* H = reinterpret_cast<S*>(scratch_->data());                                                     // every cell of a row is written below; only the sentinels need a value
        H[off[0]] = (S)NEG; H[off[1] - 1] = (S)NEG;                                     // the other rows get their sentinels when they are computed
        for (int j = 0; j <= L; j++) H[off[0] + (size_t)j + 1] = (S)(-G * j);        // free sequence prefix: H = 0
        // score profile in the ramped frame: prof[c][j] = (score of aligning a node with letter c to seq[j-1]) - G
        w_prof_.resize((((size_t)4 * (L + 1) + 64) * sizeof(S) + 7) / 8);              // + a block of slack: the row kernel's last block may read past column L
        S* prof = reinterpret_cast<S*>(w_prof_.data());
        for (size_t x = 0; x < (size_t)4 * (L + 1); x++) prof[x] = (S)(X - G);
        auto idx = [](uint8_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3; };
        for (int j = 1; j <= L; j++) prof[(size_t)idx(seq[j - 1]) * (L + 1) + j] = (S)(M - G);
        w_tmp_.resize((((size_t)L + 2) * sizeof(S) + 7) / 8);
        S* tmp = reinterpret_cast<S*>(w_tmp_.data());
        int best = NEG, bi = 0, bj = 0;
        POA_T(t_b);
        for (int i = 1; i <= N; i++) {
            const RowMeta& rm = meta[i];
            S* row = &H[off[i] + 1] - lo[i];                                          // row[j] addresses cell (i, j)
            H[off[i]] = (S)NEG;
            const S* sc = &prof[(size_t)rm.ci * (L + 1)];
            const int j0 = std::max(lo[i], 1), j1 = hi[i];
            if (lo[i] == 0) row[0] = 0;                                               // free graph prefix
            const int first = (j0 - 1 >= lo[i]) ? (int)row[j0 - 1] : NEG;
            bool fused = false;
            if (sizeof(S) == 2 && rm.np <= 16) {                                      // all candidates and the insertion chain in one AVX-512 pass
                PoaPred16 pr[16]; int np = 0;
                bool small_delta = true;
                auto add_pred = [&](int ip) {
                    const int dl = base[ip] - base[i];
                    if (dl > 16000 || dl < -16000) small_delta = false;               // e.g. the virtual source row under a late node: the scalar path below
                    pr[np++] = PoaPred16{reinterpret_cast<const int16_t*>(&H[off[ip] + 1] - lo[ip]), std::max(j0, lo[ip]), std::min(j1, hi[ip] + 1), dl};
                };
                if (rm.np <= 1) add_pred(rm.p0); else if (rm.np == 2) { add_pred(rm.p0); add_pred(rm.p1); } else for (uint32_t e : nodes[rank[i - 1]].in) add_pred(row_of[edges[e].tail]);
#ifdef POA_PHASE_TIMING
                for (int rep_ = 0; rep_ < (g_poa_exp >> 8); rep_++) poa_row16_dispatch(reinterpret_cast<int16_t*>(row), pr, np, reinterpret_cast<const int16_t*>(sc), first, G, NEG, j0, j1);   // experiment: the same row again (hot caches)
#endif
                fused = small_delta && poa_row16_dispatch(reinterpret_cast<int16_t*>(row), pr, np, reinterpret_cast<const int16_t*>(sc), first, G, NEG, j0, j1);
            }
            if (!fused) {
                for (int j = j0; j <= j1; j++) tmp[j] = (S)NEG;
                auto relax = [&](int ip) {
                    const S* P = &H[off[ip] + 1] - lo[ip];                            // P[lo-1], P[hi+1] are the sentinels
                    const int a = std::max(j0, lo[ip]), b = std::min(j1, hi[ip] + 1);
                    poa_relax(tmp, P, sc, G, base[ip] - base[i], a, b);
                };
                if (rm.np == 0) relax(0); else for (uint32_t e : nodes[rank[i - 1]].in) relax(row_of[edges[e].tail]);
                poa_scan<S>(row, tmp, first, j0, j1);
            }
            H[off[i + 1] - 1] = (S)NEG;                                               // the right sentinel, AFTER the row: the row kernel's last block may have run over it
            if (rm.sink) { for (int j = lo[i]; j <= j1; j++) { const int v = (int)row[j] + base[i] + G * j; if (v > best) { best = v; bi = i; bj = j; } } }   // free trailing overhangs
            else if (j1 == L && (int)row[L] + base[i] + G * L > best) { best = (int)row[L] + base[i] + G * L; bi = i; bj = L; }
        }
        POA_T(t_c);
        POA_ACC(0, t_a, t_b); POA_ACC(1, t_b, t_c);
        if (best <= NEG / 2) return out;
        int i = bi, j = bj;
        out.reserve((size_t)std::min(N, L) + 64);
        // The walk compares STORED values: with true(i, j) = stored(i, j) + base[i] + G*j, "true(p, j-1) + score == true(i, j)" is
        // stored(p, j-1) + (base[p] - base[i]) + prof == stored(i, j) (prof carries score - G), a deletion is stored(p, j) + (base[p] -
        // base[i]) + G == stored(i, j), an insertion stored(i, j-1) == stored(i, j); a cell outside a row's band is no candidate.
        while (i > 0 && j > 0) {
            const RowMeta& rm = meta[i];
            const int32_t node = (int32_t)rank[i - 1];
            const S* ri = &H[off[i] + 1] - lo[i];
            const int v = (int)ri[j], pf = (int)prof[(size_t)rm.ci * (L + 1) + j];
            auto diag_from = [&](int ip) -> bool { return j - 1 >= lo[ip] && j - 1 <= hi[ip] && (int)(&H[off[ip] + 1] - lo[ip])[j - 1] + (base[ip] - base[i]) + pf == v; };
            auto up_from = [&](int ip) -> bool { return j >= lo[ip] && j <= hi[ip] && (int)(&H[off[ip] + 1] - lo[ip])[j] + (base[ip] - base[i]) + G == v; };
            bool moved = false;
            if (rm.np <= 1) {                                                          // the virtual source row 0, or the one predecessor row
                const int ip = rm.p0;
                if (diag_from(ip)) { out.push_back({node, j - 1}); i = ip; j--; moved = true; }
                else if (up_from(ip)) { out.push_back({node, -1}); i = ip; moved = true; }
            } else if (rm.np == 2) {                                                   // both in-edges are in the row's flat facts, in in-edge order
                const int ia = rm.p0, ib = rm.p1;
                if (diag_from(ia)) { out.push_back({node, j - 1}); i = ia; j--; moved = true; }
                else if (diag_from(ib)) { out.push_back({node, j - 1}); i = ib; j--; moved = true; }
                else if (up_from(ia)) { out.push_back({node, -1}); i = ia; moved = true; }
                else if (up_from(ib)) { out.push_back({node, -1}); i = ib; moved = true; }
            } else {
                const Node& nd = nodes[node];
                for (uint32_t e : nd.in) { const int ip = row_of[edges[e].tail]; if (diag_from(ip)) { out.push_back({node, j - 1}); i = ip; j--; moved = true; break; } }
                if (!moved) for (uint32_t e : nd.in) { const int ip = row_of[edges[e].tail]; if (up_from(ip)) { out.push_back({node, -1}); i = ip; moved = true; break; } }
            }
            if (!moved) {
                if (j - 1 >= lo[i] && (int)ri[j - 1] == v) { out.push_back({-1, j - 1}); j--; }
                else break;                                                            // reached a free start (v == 0 at the band edge)
            }
        }
        std::reverse(out.begin(), out.end());
        POA_T(t_d);
        POA_ACC(2, t_c, t_d);
        return out;
    }

    // ---- the final graph of the device-resident engine (K12, svt_poa_graphs_fetch): letters, aligned lists (8 u16 per node: count, ids),
    // edges {tail, head, weight} in creation order.  The lists of every node come back in spoa's list order, so the depth-first sort and the
    // heaviest bundle below read exactly the graph the host engine would have built.
    void import_graph(const uint8_t* code, const uint16_t* aligned8, uint32_t n_nodes, const uint32_t* edges3, uint32_t n_edges) {
        *this = PoaGraph();
        for (uint32_t v = 0; v < n_nodes; v++) add_node(code[v], 0);
        for (uint32_t v = 0; v < n_nodes; v++) { const uint16_t* a = aligned8 + (size_t)8 * v; for (uint16_t x = 0; x < a[0]; x++) { nodes[v].aligned.push_back(a[1 + x]); n_al_cnt_[v]++; } }
        for (uint32_t e = 0; e < n_edges; e++) add_edge(edges3[3 * (size_t)e], edges3[3 * (size_t)e + 1], (int64_t)edges3[3 * (size_t)e + 2]);
        topological_sort();
    }

    void add_alignment(const Alignment& aln, const std::vector<uint8_t>& seq, const std::vector<uint32_t>& w) {
        const int L = (int)seq.size();
        if (L == 0) return;
        std::vector<int> valid;
        for (auto& p : aln) if (p.second != -1) valid.push_back(p.second);
        if (valid.empty()) { add_chain(seq, w, 0, L); topological_sort(); return; }
        int32_t prev = -1; int prev_pos = valid.front() - 1;
        if (add_chain(seq, w, 0, valid.front()) >= 0) prev = (int32_t)nodes.size() - 1;        // unaligned prefix: new chain, remember its last node
        const int32_t tail_first = add_chain(seq, w, valid.back() + 1, L);                      // unaligned suffix: new chain
        for (auto& p : aln) {
            if (p.second == -1) continue;
            const uint8_t letter = seq[p.second];
            int32_t cur;
            if (p.first == -1) cur = add_node(letter, p.second);
            else if (n_code_[p.first] == letter) { cur = p.first; note_position(cur, p.second); }
            else {
                cur = -1;
                if (n_al_cnt_[p.first]) for (uint32_t a : nodes[p.first].aligned) if (n_code_[a] == letter) { cur = (int32_t)a; note_position(cur, p.second); break; }
                if (cur < 0) {
                    cur = add_node(letter, p.second);
                    for (uint32_t a : nodes[p.first].aligned) { nodes[cur].aligned.push_back(a); nodes[a].aligned.push_back((uint32_t)cur); n_al_cnt_[cur]++; n_al_cnt_[a]++; }
                    nodes[cur].aligned.push_back((uint32_t)p.first); nodes[p.first].aligned.push_back((uint32_t)cur); n_al_cnt_[cur]++; n_al_cnt_[p.first]++;
                }
            }
            if (prev >= 0) add_edge((uint32_t)prev, (uint32_t)cur, (int64_t)w[prev_pos] + (int64_t)w[p.second]);
            prev = cur; prev_pos = p.second;
        }
        if (tail_first >= 0) add_edge((uint32_t)prev, (uint32_t)tail_first, (int64_t)w[valid.back()] + (int64_t)w[valid.back() + 1]);
        topological_sort();
    }

    std::vector<uint8_t> consensus() const {
        const int N = (int)rank.size();
        std::vector<uint8_t> out;
        if (N == 0) return out;
        std::vector<int64_t> score(nodes.size(), 0);
        std::vector<int32_t> pred(nodes.size(), -1);
        int32_t mx = -1;
        auto relax = [&](uint32_t v) {
            for (uint32_t e : nodes[v].in) {
                const Edge& ed = edges[e];
                if (score[ed.tail] < 0) continue;
                if (score[v] < ed.weight || (score[v] == ed.weight && pred[v] >= 0 && score[pred[v]] <= score[ed.tail])) { score[v] = ed.weight; pred[v] = (int32_t)ed.tail; }
            }
            if (pred[v] >= 0) score[v] += score[pred[v]];
        };
        for (uint32_t v : rank) { relax(v); if (mx < 0 || score[mx] < score[v]) mx = (int32_t)v; }
        // branch completion: extend the heaviest path to a sink
        std::vector<int> pos(nodes.size(), 0);
        for (int i = 0; i < N; i++) pos[rank[i]] = i;
        while (!nodes[mx].out.empty()) {
            for (uint32_t e : nodes[mx].out) for (uint32_t e2 : nodes[edges[e].head].in) if ((int32_t)edges[e2].tail != mx) score[edges[e2].tail] = -1;
            int32_t nmx = -1; int64_t best = 0;
            for (int i = pos[mx] + 1; i < N; i++) {
                const uint32_t v = rank[i];
                score[v] = -1; pred[v] = -1;
                int64_t sv = -1; int32_t pv = -1;
                for (uint32_t e : nodes[v].in) {
                    const Edge& ed = edges[e];
                    if (score[ed.tail] == -1) continue;
                    if (sv < ed.weight || (sv == ed.weight && pv >= 0 && score[pv] <= score[ed.tail])) { sv = ed.weight; pv = (int32_t)ed.tail; }
                }
                if (pv >= 0) { score[v] = sv + score[pv]; pred[v] = pv; if (nmx < 0 || best < score[v]) { nmx = (int32_t)v; best = score[v]; } }
            }
            if (nmx < 0) break;
            mx = nmx;
        }
        for (int32_t v = mx; v >= 0; v = pred[v]) out.push_back(nodes[v].code);
        std::reverse(out.begin(), out.end());
        return out;
    }

private:
    // Band centre of a node.  spoa proper has no band; the reference's spoars call passes BandConfig{base, frac} (src/alignment.rs:209-221)
    // and the crate is absent, so the band's anchor is this restatement's choice: the rounded mean of the (1-based) positions the
    // node's bases had in their own sequences.  Unlike a longest-path coordinate it does not drift as insertion nodes accumulate
    // (every read's private insertions lengthen the longest path by one; after ~band-width of them the band left the true diagonal
    // and late reads of a deep cluster no longer aligned).
    static int32_t col_of(uint64_t pos_sum, uint32_t pos_n) {                     // band_column: the rounded mean position; the sums stay far below 2^31 (75 reads x 5.4 kb), so the division is a 32-bit one
        if (!pos_n) return 1;
        return pos_sum < (1ull << 30) ? (int32_t)(((uint32_t)(2 * pos_sum) + pos_n) / (2u * pos_n)) : (int32_t)((2 * pos_sum + pos_n) / (2 * (uint64_t)pos_n));
    }
    void note_position(int32_t node, int seq_pos) { n_pos_sum_[node] += (uint64_t)seq_pos + 1; n_pos_n_[node]++; n_col_[node] = col_of(n_pos_sum_[node], n_pos_n_[node]); }
    int32_t add_node(uint8_t code, int seq_pos) {
        nodes.push_back(Node{code, {}, {}, {}}); n_code_.push_back(code); n_pos_sum_.push_back((uint64_t)seq_pos + 1); n_pos_n_.push_back(1);
        n_col_.push_back(col_of((uint64_t)seq_pos + 1, 1)); n_first_in_.push_back(-1); n_second_in_.push_back(-1); n_first_out_head_.push_back(-1); n_first_out_edge_.push_back(0); n_in_cnt_.push_back(0); n_out_cnt_.push_back(0); n_al_cnt_.push_back(0);
        n_ci_.push_back((uint8_t)(code == 'A' ? 0 : code == 'C' ? 1 : code == 'G' ? 2 : 3));
        return (int32_t)nodes.size() - 1;
    }
    void add_edge(uint32_t tail, uint32_t head, int64_t weight) {
        if (n_first_out_head_[tail] == (int32_t)head) { edges[n_first_out_edge_[tail]].weight += weight; return; }     // the usual case: the read follows the node's first out-edge
        for (uint32_t e : nodes[tail].out) if (edges[e].head == head) { edges[e].weight += weight; return; }
        edges.push_back(Edge{tail, head, weight});
        nodes[tail].out.push_back((uint32_t)edges.size() - 1); nodes[head].in.push_back((uint32_t)edges.size() - 1);
        if (n_out_cnt_[tail]++ == 0) { n_first_out_head_[tail] = (int32_t)head; n_first_out_edge_[tail] = (uint32_t)edges.size() - 1; }
        { const uint32_t k_ = n_in_cnt_[head]++; if (k_ == 0) n_first_in_[head] = (int32_t)tail; else if (k_ == 1) n_second_in_[head] = (int32_t)tail; }
    }
    int32_t add_chain(const std::vector<uint8_t>& seq, const std::vector<uint32_t>& w, int begin, int end) {   // new nodes for seq[begin,end)
        if (begin >= end) return -1;
        const int32_t first = add_node(seq[begin], begin);
        for (int i = begin + 1; i < end; i++) { const int32_t n = add_node(seq[i], i); add_edge((uint32_t)n - 1, (uint32_t)n, (int64_t)w[i - 1] + (int64_t)w[i]); }
        return first;
    }
    void topological_sort() {
        rank.clear();
        const size_t n = nodes.size();
        ts_mark_.assign(n, 0); ts_chk_.assign(n, 0); ts_stack_.clear();
        std::vector<uint8_t>& mark = ts_mark_; std::vector<uint8_t>& chk = ts_chk_; std::vector<uint32_t>& st = ts_stack_;
        for (uint32_t s = 0; s < n; s++) {
            if (mark[s]) continue;
            st.push_back(s);
            while (!st.empty()) {
                const uint32_t c = st.back();
                bool valid = true;
                if (mark[c] != 2) {
                    // the flat per-node facts first: nine nodes of ten have one in-edge and no aligned sibling, and never touch their Node
                    const uint32_t ic = n_in_cnt_[c];
                    if (ic == 1) { const uint32_t t = (uint32_t)n_first_in_[c]; if (mark[t] != 2) { st.push_back(t); valid = false; } }
                    else if (ic == 2) {
                        const uint32_t t0 = (uint32_t)n_first_in_[c], t1 = (uint32_t)n_second_in_[c];
                        if (mark[t0] != 2) { st.push_back(t0); valid = false; }
                        if (mark[t1] != 2) { st.push_back(t1); valid = false; }
                    }
                    else if (ic > 2) for (uint32_t e : nodes[c].in) if (mark[edges[e].tail] != 2) { st.push_back(edges[e].tail); valid = false; }
                    const bool has_al = n_al_cnt_[c] != 0;
                    if (has_al && !chk[c]) for (uint32_t a : nodes[c].aligned) if (mark[a] != 2) { st.push_back(a); chk[a] = 1; valid = false; }
                    if (valid) {
                        mark[c] = 2;
                        if (!chk[c]) { rank.push_back(c); if (has_al) for (uint32_t a : nodes[c].aligned) rank.push_back(a); }
                    } else mark[c] = 1;
                }
                if (valid) st.pop_back();
            }
        }
    }
    std::vector<uint8_t> ts_mark_, ts_chk_; std::vector<uint32_t> ts_stack_;   // scratch of topological_sort, reused across reads
};

}  // namespace savont
