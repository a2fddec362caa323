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
// kernel (kernels_align.hip): lane l owns P consecutive diagonals, anti-diagonal steps alternate between the even and the
// odd diagonals, so a cell's left neighbour (diagonal d-1) and upper neighbour (d+1) are the values of the PREVIOUS step and its
// diagonal predecessor is its own register: five int32 registers per diagonal (H, E1, E2, F1, F2), updated in place; the lane - 1 / lane + 1 values of a step
// half arrive through four DPP-modified adds (shift + gap cost in one instruction).  Outside the MATRIX "not a cell" is a large negative H: everything derived
// from it stays negative and loses against the local start 0, which is what the oracle's explicit guards do.
//   * (band check of a 16-lane / 8-lane group: 32 P / G - 1 >= w, the wave always carries 64 P / G diagonals per pair)
//   * cells outside the matrix exist only while the band enters / leaves the matrix: two masked loops around an unmasked steady loop;
//   * diagonals outside the BAND (the wave always carries 64 P / G of them per pair) are held at H = 0 by the ceiling operand of the cell's v_med3: a local start
//     whose gap states are negative, which no in-band cell can tell from "not a cell" (see the cell).
// 18-21 VALU operations per cell (round 4; 28 in round 3): an order of magnitude above the bit-parallel K8 -- this is the price of the affine contract.  The cell is
// written for the instruction RATES of gfx950 (profiles/r04_valu_rates.txt): two-operand add / and issue in ~2.7 cycles, every max / min / med3 / select / DPP /
// three-operand form in ~4.2, so selects and separate moves are what it avoids.
#include <type_traits>
#include "svt_internal.hpp"

#define AFF_NEG (-(1 << 30))
namespace {
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
// (lane - 1 / lane + 1 of the pair's registers) + constant in ONE instruction each: the shift rides on the add (DPP).  Lanes the shift has no source for get
// 0 + c, and a pair's first / last lane gets its neighbour pair's value: the cell that takes these in caps its E / F (FIXL / FIXR in the kernel).  One block
// for the four sums of a step half, behind ONE s_nop 1: a DPP read needs two wait states after the VALU write of its source, and the compiler does not
// look into inline assembly for that hazard.
#define AFF_DPP4(SH) "s_nop 1\n" \
    "\tv_add_u32_dpp %0, %4, %7 " SH " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_add_u32_dpp %1, %5, %8 " SH " row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "\tv_add_u32_dpp %2, %6, %9 " SH " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_add_u32_dpp %3, %5, %10 " SH " row_mask:0xf bank_mask:0xf bound_ctrl:0"
// -> g1 + x1, h + o1, g2 + x2, h + o2 of the lane below (LEFT) / above
template <int G, bool LEFT> __device__ __forceinline__ void aff_add_shifted(int g1, int h, int g2, int x1, int o1, int x2, int o2, int& r1a, int& r1b, int& r2a, int& r2b) {
    if (G >= 4 && LEFT)  asm(AFF_DPP4("row_shr:1")  : "=&v"(r1a), "=&v"(r1b), "=&v"(r2a), "=&v"(r2b) : "v"(g1), "v"(h), "v"(g2), "v"(x1), "v"(o1), "v"(x2), "v"(o2));
    if (G >= 4 && !LEFT) asm(AFF_DPP4("row_shl:1")  : "=&v"(r1a), "=&v"(r1b), "=&v"(r2a), "=&v"(r2b) : "v"(g1), "v"(h), "v"(g2), "v"(x1), "v"(o1), "v"(x2), "v"(o2));
    if (G < 4 && LEFT)   asm(AFF_DPP4("wave_shr:1") : "=&v"(r1a), "=&v"(r1b), "=&v"(r2a), "=&v"(r2b) : "v"(g1), "v"(h), "v"(g2), "v"(x1), "v"(o1), "v"(x2), "v"(o2));
    if (G < 4 && !LEFT)  asm(AFF_DPP4("wave_shl:1") : "=&v"(r1a), "=&v"(r1b), "=&v"(r2a), "=&v"(r2b) : "v"(g1), "v"(h), "v"(g2), "v"(x1), "v"(o1), "v"(x2), "v"(o2));
}
__device__ __forceinline__ int aff_max(int a, int b) {       // v_max3_i32 issues faster than the two-operand v_max_i32 (tools/micro/valu_rates.hip: 4.17 against 4.58 cycles)
    int r; asm("v_max3_i32 %0, %1, %2, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ int aff_flag(u32 y, int b) {      // bit b of y as 0 / -1; in assembly: the compiler turns the C form into and + compare + select per cell
    int m; asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(y), "s"(b)); return m;
}
constexpr int AS = 4096;
constexpr int A_MATCH = 2 * AS, A_MISM = -4 * AS - 1;
constexpr int A_O1 = -(4 + 2) * AS - 1, A_X1 = -2 * AS - 1, A_O2 = -(24 + 1) * AS - 1, A_X2 = -1 * AS - 1;
}

// G pairs per wavefront, each on 64 / G lanes owning P consecutive diagonals (P even): 64 P / G diagonals per pair, band <= 32 P / G - 1.
// The anti-diagonal step `a` is shared by the groups; every group masks its own cells while ANY group's band is entering or leaving its matrix.
template <int P, int G>
__global__ void __launch_bounds__(64) k_align_affine(BatchView Q, BatchView T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                                     const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel, u64 n_sel,
                                                     int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, u32 ldsq, u32 ldst) {
    static_assert(P % 2 == 0 && P <= 32, "an even and an odd diagonal per step half; the query bases of a step live in one 64-bit word");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int LG = 64 / G;                 // lanes per pair
    const int lane = threadIdx.x, grp = lane / LG, gl = lane % LG;
    u32* qw = (u32*)smem + (size_t)grp * (ldsq + ldst);
    u32* tw = qw + ldsq;
    const u64 slot = (u64)blockIdx.x * G + grp;
    const bool live = slot < n_sel;            // a wave's last groups may have no pair: n = m = 0, no cell ever unmasked
    const u64 pid = live ? (sel ? sel[slot] : slot) : 0;
    const u32 qr = live ? qi[pid] : 0, tr = live ? ti[pid] : 0;
    const int n = live ? (int)(Q.off[qr + 1] - Q.off[qr]) : 0;
    const int m = live ? (int)(T.off[tr + 1] - T.off[tr]) : 0;
    const int w = live ? (int)band[pid] : 0;
    const int wp = w + (w & 1);
    const int nwq = (n + 15) / 16, nwt = (m + 15) / 16;
    {   // stage both sequences as 2-bit words (the target reverse-complemented when asked)
        const u32* qs = Q.packed + Q.woff[qr];
        const u32* ts = T.packed + T.woff[tr];
        for (int i = gl; i < nwq; i += LG) qw[i] = qs[i];
        if (!(rev && live && rev[pid])) { for (int i = gl; i < nwt; i += LG) tw[i] = ts[i]; }
        else {
            for (int i = gl; i < nwt; i += LG) {
                int start = m - 16 * i - 16;
                int wi = start >> 4; u32 o = (u32)(start & 15) * 2;
                u32 w0 = (wi >= 0 && wi < nwt) ? ts[wi] : 0u, w1 = (wi + 1 >= 0 && wi + 1 < nwt) ? ts[wi + 1] : 0u;
                u32 x = o ? ((w0 << o) | (w1 >> (32 - o))) : w0;
                tw[i] = aff_revcomp16(x);
            }
        }
    }
    __syncthreads();
    const int d0 = P * gl;
    int H[P], E1[P], E2[P], F1[P], F2[P], CE[P];
    #pragma unroll
    for (int k = 0; k < P; k++) {
        H[k] = E1[k] = E2[k] = F1[k] = F2[k] = AFF_NEG;
        const int dd = d0 + k - wp;
        CE[k] = (live && dd >= -w && dd <= w) ? 0x7FFFFFFF : 0;
    }
    const int I = (wp - d0) / 2;               // i of diagonal d0 at a = 0 (exact: both even)
    const int J = I + d0 - wp;
    u64 QW = 0;                                // q[I-1-x] in bits 62-2x: the query bases of the P / 2 cells of a step, descending
    #pragma unroll
    for (int x = 0; x < P / 2; x++) {
        int idx = I - 1 - x;
        u32 b = (idx >= 0 && idx < n) ? ((qw[idx >> 4] >> (30 - 2 * (idx & 15))) & 3u) : 0u;
        QW |= (u64)b << (62 - 2 * x);
    }
    u32 QF = aff_get16(qw, nwq, I);
    u64 TW = ((u64)aff_get16(tw, nwt, J - 1) << 32) | aff_get16(tw, nwt, J + 15);
    int adv = 0;
    int best = 0;
    int VX1 = A_X1, VO1 = A_O1, VX2 = A_X2, VO2 = A_O2;            // the DPP form of v_add takes its second operand from a VGPR
    asm volatile("" : "+v"(VX1), "+v"(VO1), "+v"(VX2), "+v"(VO2));

    // G == 8: a pair's first / last lane receives its neighbour pair's registers through the row shifts; what they would feed -- E of the lane's first
    // diagonal, F of its last -- is capped instead (two v_min per step half against three selects + three constants moved into the shift's destination)
    const int FIXL = gl == 0 ? AFF_NEG : 0x7FFFFFFF, FIXR = gl == LG - 1 ? AFF_NEG : 0x7FFFFFFF;
    // a cell from the four sums that feed its gap states: e1a = E1(left) + x1, e1b = H(left) + o1, ... (the lane's first / last diagonal gets them through aff_add_*)
    auto cell = [&](auto mask_c, int k, int a, int differs, int e1a, int e1b, int e2a, int e2b, int f1a, int f1b, int f2a, int f2b) {
        constexpr bool MASK = decltype(mask_c)::value;
        const int hd = H[k] + A_MATCH + (differs & (A_MISM - A_MATCH));      // differs: 0 / -1 (one v_bfe_i32; a compare + select costs two half-rate instructions and a constant in a VGPR)
        int e1 = aff_max(e1a, e1b), e2 = aff_max(e2a, e2b);
        int f1 = aff_max(f1a, f1b), f2 = aff_max(f2a, f2b);
        if (k == 0) { e1 = min(e1, FIXL); e2 = min(e2, FIXL); }
        if (k == P - 1) { f1 = min(f1, FIXR); f2 = min(f2, FIXR); }
        // floor 0 and band ceiling in one v_med3: CE is 0 outside the band, so a diagonal the wave carries beyond the band holds H = 0 -- a local start whose
        // E / F (<= the gap-open cost < 0) lose against the floor of every cell they reach, exactly like the "not a cell" of the oracle's guards
        int h; { const int x = max(max(hd, e1), max(max(e2, f1), f2)); asm("v_med3_i32 %0, %1, 0, %2" : "=v"(h) : "v"(x), "v"(CE[k])); }
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
                int L1a, L1b, L2a, L2b; aff_add_shifted<G, true>(E1[P - 1], H[P - 1], E2[P - 1], VX1, VO1, VX2, VO2, L1a, L1b, L2a, L2b);
                #pragma unroll
                for (int x = 0; x < P / 2; x++) {
                    const int k = 2 * x;
                    cell(mask_c, k, a, aff_flag(y, 31 - 2 * x), k ? E1[k - 1] + A_X1 : L1a, k ? H[k - 1] + A_O1 : L1b, k ? E2[k - 1] + A_X2 : L2a, k ? H[k - 1] + A_O2 : L2b,
                         F1[k + 1] + A_X1, H[k + 1] + A_O1, F2[k + 1] + A_X2, H[k + 1] + A_O2);
                }
            }
            {   // odd step a + 1: diagonals d0 + 2x + 1 (same query bases, targets one further)
                const u32 X = (u32)(QW >> 32) ^ (u32)((TW << 2) >> 32);
                const u32 y = X | (X << 1);
                int R1a, R1b, R2a, R2b; aff_add_shifted<G, false>(F1[0], H[0], F2[0], VX1, VO1, VX2, VO2, R1a, R1b, R2a, R2b);
                #pragma unroll
                for (int x = 0; x < P / 2; x++) {
                    const int k = 2 * x + 1;
                    const bool in = k + 1 < P;
                    cell(mask_c, k, a + 1, aff_flag(y, 31 - 2 * x), E1[k - 1] + A_X1, H[k - 1] + A_O1, E2[k - 1] + A_X2, H[k - 1] + A_O2,
                         in ? F1[in ? k + 1 : 0] + A_X1 : R1a, in ? H[in ? k + 1 : 0] + A_O1 : R1b, in ? F2[in ? k + 1 : 0] + A_X2 : R2a, in ? H[in ? k + 1 : 0] + A_O2 : R2b);
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
    // masked while a band enters its matrix (a < w) and while one leaves it (a + 1 > min(2n, 2m) - w); unmasked in between
    int S0 = (w + 1) & ~1;                                         // first even a >= w
    const int last = min(2 * n, 2 * m) - w;                        // last step at which every in-band diagonal is inside the matrix
    int S1 = last >= 1 ? ((last - 1) & ~1) + 2 : 0;                // first even a with a + 1 > last
    int END = n + m + 1;
    if (G > 1) {                                                   // the groups of a wave share the loops: latest entry, earliest exit, longest pair
        #pragma unroll
        for (int s = LG; s < 64; s <<= 1) { S0 = max(S0, __shfl_xor(S0, s)); S1 = min(S1, __shfl_xor(S1, s)); END = max(END, __shfl_xor(END, s)); }
    }
    int a = 0;
    run(std::true_type{}, a, min(S0, END));
    run(std::false_type{}, a, min(S1, END));
    run(std::true_type{}, a, END);
    #pragma unroll
    for (int s = LG / 2; s >= 1; s >>= 1) best = max(best, __shfl_xor(best, s));
    if (gl == 0 && live) {
        const int score = (best + AS - 1) / AS;
        nm_out[pid] = best > 0 ? score * AS - best : 0x7FFFFFFF;
        if (score_out) score_out[pid] = score;
    }
}

// Band classes (round 4).  A pair occupies 64 / G lanes x P diagonals: what a wave carries beyond 2 w + 1 is lost work, and the three DPP moves +
// the sequence-window shifts of a step are shared by the P / 2 cells a lane updates in it.  So: eight pairs per wave with 6 / 8 / 10 / 12 diagonals per
// lane for the bands Stage 7 produces (w = |e| + d + 8 ~ 20-30: 48 or 64 diagonals instead of 64 on 16 lanes x 4), and P = 8-16 above (round 3 ran
// every class at P = 4 or 6: 3 DPP + 8 shift instructions per TWO cells).
//   cls   0      1      2      3       4       5      6       7       8       9
//   P,G   4,8    6,8    8,8    10,8    12,8    8,4    12,4    16,4    16,2    16,1
//   w <=  15     23     31     39      47      63     95      127     255     511
int affine_class_of(u32 w) { for (int cls = 0; cls < AFF_NCLS; cls++) if ((int)w <= 32 * AFF_P[cls] / AFF_G[cls] - 1) return cls; return AFF_NCLS - 1; }
int launch_align_affine(svt_ctx* c, hipStream_t on, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                        const u32* d_sel, u64 n_sel, int cls, int32_t* d_nm, int32_t* d_score, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells) {
    if (n_sel == 0) return SVT_OK;
    static const char* names[AFF_NCLS] = {"k_align_affine_p4g8", "k_align_affine_p6g8", "k_align_affine_p8g8", "k_align_affine_p10g8", "k_align_affine_p12g8",
                                          "k_align_affine_p8g4", "k_align_affine_p12g4", "k_align_affine_p16g4", "k_align_affine_p16g2", "k_align_affine_p16g1"};
    const int G = AFF_G[cls];
    u32 ldsq = (max_qlen + 15) / 16 + 2, ldst = (max_tlen + 15) / 16 + 2;
    size_t sh = (size_t)(ldsq + ldst) * 4 * G;
    ProfScope ps(c, names[cls], algo_bytes, cells, on);
    BatchView qv = Q->view(), tv = T->view();
    const dim3 grid((u32)((n_sel + G - 1) / G));
#define SVT_K8A(PP, GG) hipLaunchKernelGGL((k_align_affine<PP, GG>), grid, dim3(64), sh, on, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, d_score, ldsq, ldst)
    switch (cls) {
        case 0: SVT_K8A(4, 8); break; case 1: SVT_K8A(6, 8); break; case 2: SVT_K8A(8, 8); break; case 3: SVT_K8A(10, 8); break; case 4: SVT_K8A(12, 8); break;
        case 5: SVT_K8A(8, 4); break; case 6: SVT_K8A(12, 4); break; case 7: SVT_K8A(16, 4); break; case 8: SVT_K8A(16, 2); break; default: SVT_K8A(16, 1); break;
    }
#undef SVT_K8A
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
