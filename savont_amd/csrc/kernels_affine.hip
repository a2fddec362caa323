// kernels_affine.hip -- K8a: minimap2-style `nm` (best LOCAL alignment under map-ont / lr:hq scoring a = 2, b = 4,
// gap(l) = min(4 + 2 l, 24 + l); nm = mismatches + gap bases along it; ties in score -> fewest nm).  Contract: DESIGN.md section 3
// "K8a" (the CPU restatement the tests check it against is align_nm_affine_codes of the test oracle); it replaces what `mapping.alignment.nm` means at
// src/alignment.rs:1848-1862 when the caller asks for the affine contract (svt_align_nm_affine).
//
//   cells (i,j), 0 <= i <= n (query), 0 <= j <= m (target), band |j-i| <= w
//   H = max(0, H(i-1,j-1) + s, E1, E2, F1, F2),  E*(i,j) = max(E*(i,j-1) + x, H(i,j-1) + o),  F*(i,j) = max(F*(i-1,j) + x, H(i-1,j) + o)
//   every value is ONE int32 = score * 4096 - nm, so one max-plus recurrence carries the score and the tie-break.
//   Exactness of the 12-bit nm field: the order of the packed values is the lexicographic order (score, -nm) as long as nm < 4096.  A
//   candidate with nm' >= 4096 can displace one with nm < 4096 in a cell only when its score is strictly higher (score' <= score gives
//   score' * 4096 - nm' <= score * 4096 - 4096 < score * 4096 - nm), which is also the lexicographic order; so every prefix of an optimum
//   whose nm is below 4096 survives, and the result is exact whenever the optimum's nm is below 4096.  The oracle packs with 2^20 in 64 bits
//   and does not share the limit (align_nm_affine_codes of the test oracle).
//
// Mapping (integer max-plus DP, no MFMA): one wavefront per pair, the band's diagonals on the lanes exactly as in K9's wavefront
// kernel (kernels_align.hip): lane l owns P = 4R consecutive diagonals, anti-diagonal steps alternate between the even and the
// odd diagonals, so a cell's left neighbour (diagonal d-1) and upper neighbour (d+1) are the values of the PREVIOUS step and its
// diagonal predecessor is its own register: five int32 registers per diagonal (H, E1, E2, F1, F2), updated in place, three DPP
// moves per step for the lane boundary.  "Not a cell" (outside the matrix, outside the band) is a large negative H: everything
// derived from it stays negative and loses against the local start 0, which is what the oracle's explicit guards do.
//   * cells outside the matrix exist only while the band enters / leaves the matrix: two masked loops around an unmasked steady loop;
//   * diagonals outside the band (the wave always carries 64 P of them) are held down by a per-register ceiling.
// ~26 VALU operations per cell: an order of magnitude above the bit-parallel K8 -- this is the price of the affine contract.
#include <type_traits>
#include "svt_internal.hpp"

#define AFF_NEG (-(1 << 30))
namespace {
__device__ __forceinline__ int aff_from_left(int v) { return __builtin_amdgcn_update_dpp(AFF_NEG, v, 0x138, 0xF, 0xF, false); }    // lane-1, lane 0 gets AFF_NEG
__device__ __forceinline__ int aff_from_right(int v) { return __builtin_amdgcn_update_dpp(AFF_NEG, v, 0x130, 0xF, 0xF, false); }   // lane+1, lane 63 gets AFF_NEG
__device__ __forceinline__ u32 aff_get16(const u32* lds, int nw, int pos) {      // 16 bases from base `pos` (any int), zero outside
    int wi = pos >> 4; u32 o = (u32)(pos & 15) * 2;
    u32 w0 = (wi >= 0 && wi < nw) ? lds[wi] : 0u;
    u32 w1 = (wi + 1 >= 0 && wi + 1 < nw) ? lds[wi + 1] : 0u;
    return o ? ((w0 << o) | (w1 >> (32 - o))) : w0;
}
__device__ __forceinline__ u32 aff_revcomp16(u32 x) {
    u32 y = __brev(~x);
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}
constexpr int AS = 4096;
constexpr int A_MATCH = 2 * AS, A_MISM = -4 * AS - 1;
constexpr int A_O1 = -(4 + 2) * AS - 1, A_X1 = -2 * AS - 1, A_O2 = -(24 + 1) * AS - 1, A_X2 = -1 * AS - 1;
}

template <int R>
__global__ void __launch_bounds__(64) k_align_affine(BatchView Q, BatchView T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                                     const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel, u64 n_sel,
                                                     int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, u32 ldsq, u32 ldst) {
    extern __shared__ __align__(16) unsigned char smem[];
    u32* qw = (u32*)smem;
    u32* tw = qw + ldsq;
    if (blockIdx.x >= n_sel) return;
    const u64 pid = sel ? sel[blockIdx.x] : blockIdx.x;
    const int lane = threadIdx.x;
    const u32 qr = qi[pid], tr = ti[pid];
    const int n = (int)(Q.off[qr + 1] - Q.off[qr]);
    const int m = (int)(T.off[tr + 1] - T.off[tr]);
    const int w = (int)band[pid];
    const int wp = w + (w & 1);
    const int nwq = (n + 15) / 16, nwt = (m + 15) / 16;
    {   // stage both sequences as 2-bit words (the target reverse-complemented when asked)
        const u32* qs = Q.packed + Q.woff[qr];
        const u32* ts = T.packed + T.woff[tr];
        for (int i = lane; i < nwq; i += 64) qw[i] = qs[i];
        if (!(rev && rev[pid])) { for (int i = lane; i < nwt; i += 64) tw[i] = ts[i]; }
        else {
            for (int i = lane; i < nwt; i += 64) {
                int start = m - 16 * i - 16;
                int wi = start >> 4; u32 o = (u32)(start & 15) * 2;
                u32 w0 = (wi >= 0 && wi < nwt) ? ts[wi] : 0u, w1 = (wi + 1 >= 0 && wi + 1 < nwt) ? ts[wi + 1] : 0u;
                u32 x = o ? ((w0 << o) | (w1 >> (32 - o))) : w0;
                tw[i] = aff_revcomp16(x);
            }
        }
    }
    __syncthreads();
    constexpr int P = 4 * R;
    const int d0 = P * lane;
    int H[P], E1[P], E2[P], F1[P], F2[P], CE[P];
    #pragma unroll
    for (int k = 0; k < P; k++) {
        H[k] = E1[k] = E2[k] = F1[k] = F2[k] = AFF_NEG;
        const int dd = d0 + k - wp;
        CE[k] = (dd >= -w && dd <= w) ? 0x7FFFFFFF : AFF_NEG;
    }
    const int I = (wp - d0) / 2;               // i of diagonal d0 at a = 0 (exact: both even)
    const int J = I + d0 - wp;
    u64 QW = 0;                                // q[I-1-x] in bits 62-2x: the query bases of the 2R cells of a step, descending
    #pragma unroll
    for (int x = 0; x < 2 * R; x++) {
        int idx = I - 1 - x;
        u32 b = (idx >= 0 && idx < n) ? ((qw[idx >> 4] >> (30 - 2 * (idx & 15))) & 3u) : 0u;
        QW |= (u64)b << (62 - 2 * x);
    }
    u32 QF = aff_get16(qw, nwq, I);
    u64 TW = ((u64)aff_get16(tw, nwt, J - 1) << 32) | aff_get16(tw, nwt, J + 15);
    int adv = 0;
    int best = 0;
    const int total = n + m;

    auto cell = [&](auto mask_c, int k, int a, u32 differs, int hl, int e1l, int e2l, int hu, int f1u, int f2u) {
        constexpr bool MASK = decltype(mask_c)::value;
        const int hd = H[k] + (differs ? A_MISM : A_MATCH);
        const int e1 = max(e1l + A_X1, hl + A_O1), e2 = max(e2l + A_X2, hl + A_O2);
        const int f1 = max(f1u + A_X1, hu + A_O1), f2 = max(f2u + A_X2, hu + A_O2);
        int h = max(max(max(0, hd), max(e1, e2)), max(f1, f2));
        h = min(h, CE[k]);
        if (MASK) {
            const int dd = d0 + k - wp;
            const int lo = dd < 0 ? -dd : dd, hi = min(2 * n + dd, 2 * m - dd);
            if (a < lo || a > hi) h = AFF_NEG;
        }
        H[k] = h; E1[k] = e1; E2[k] = e2; F1[k] = f1; F2[k] = f2;
        best = max(best, h);
    };
    auto run = [&](auto mask_c, int& a, const int a_end) {
        for (; a < a_end; a += 2) {
            {   // even step a: diagonals d0 + 2x
                const u32 X = (u32)(QW >> 32) ^ (u32)(TW >> 32);
                const u32 y = X | (X << 1);
                const int HL = aff_from_left(H[P - 1]), E1L = aff_from_left(E1[P - 1]), E2L = aff_from_left(E2[P - 1]);
                #pragma unroll
                for (int x = 0; x < 2 * R; x++) {
                    const int k = 2 * x;
                    cell(mask_c, k, a, y & (1u << (31 - 2 * x)), k ? H[k - 1] : HL, k ? E1[k - 1] : E1L, k ? E2[k - 1] : E2L, H[k + 1], F1[k + 1], F2[k + 1]);
                }
            }
            {   // odd step a + 1: diagonals d0 + 2x + 1 (same query bases, targets one further)
                const u32 X = (u32)(QW >> 32) ^ (u32)((TW << 2) >> 32);
                const u32 y = X | (X << 1);
                const int HR = aff_from_right(H[0]), F1R = aff_from_right(F1[0]), F2R = aff_from_right(F2[0]);
                #pragma unroll
                for (int x = 0; x < 2 * R; x++) {
                    const int k = 2 * x + 1;
                    cell(mask_c, k, a + 1, y & (1u << (31 - 2 * x)), H[k - 1], E1[k - 1], E2[k - 1], k + 1 < P ? H[k + 1] : HR, k + 1 < P ? F1[k + 1] : F1R, k + 1 < P ? F2[k + 1] : F2R);
                }
            }
            QW = (QW >> 2) | ((u64)(QF >> 30) << 62);
            QF <<= 2;
            TW <<= 2;
            if (++adv == 16) {
                adv = 0;
                const int s = a / 2 + 1;
                QF = aff_get16(qw, nwq, I + s);
                TW |= (u64)aff_get16(tw, nwt, J + s + 15);
            }
        }
    };
    // masked while the band enters the matrix (a < w) and while it leaves it (a + 1 > min(2n, 2m) - w); unmasked in between
    const int S0 = (w + 1) & ~1;                                   // first even a >= w
    int last = min(2 * n, 2 * m) - w;                              // last step at which every in-band diagonal is inside the matrix
    int S1 = last >= 1 ? ((last - 1) & ~1) + 2 : 0;                // first even a with a + 1 > last
    const int END = total + 1;
    int a = 0;
    run(std::true_type{}, a, min(S0, END));
    run(std::false_type{}, a, min(S1, END));
    run(std::true_type{}, a, END);
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) best = max(best, __shfl_xor(best, s));
    if (lane == 0) {
        const int score = (best + AS - 1) / AS;
        nm_out[pid] = best > 0 ? score * AS - best : 0x7FFFFFFF;
        if (score_out) score_out[pid] = score;
    }
}

int launch_align_affine(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                        const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, int32_t* d_score, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells) {
    if (n_sel == 0) return SVT_OK;
    u32 ldsq = (max_qlen + 15) / 16 + 2, ldst = (max_tlen + 15) / 16 + 2;
    size_t sh = (size_t)(ldsq + ldst) * 4;
    ProfScope ps(c, rclass == 1 ? "k_align_affine_r1" : (rclass == 2 ? "k_align_affine_r2" : "k_align_affine_r4"), algo_bytes, cells);
    BatchView qv = Q->view(), tv = T->view();
    if (rclass == 1) hipLaunchKernelGGL((k_align_affine<1>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, d_score, ldsq, ldst);
    else if (rclass == 2) hipLaunchKernelGGL((k_align_affine<2>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, d_score, ldsq, ldst);
    else hipLaunchKernelGGL((k_align_affine<4>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, d_score, ldsq, ldst);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
