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
// The pairs of the wave are sel[first .. first + count), count <= G (a wave's last groups may have no pair).
template <int P, int G>
__device__ __forceinline__ void aff_pairs(const BatchView& Q, const BatchView& T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                          const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel, const u64 first, const u32 count,
                                          int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, const u32 ldsq, const u32 ldst, unsigned char* smem) {
    static_assert(P % 2 == 0 && P <= 32, "an even and an odd diagonal per step half; the query bases of a step live in one 64-bit word");
    constexpr int LG = 64 / G;                 // lanes per pair
    const int lane = threadIdx.x, grp = lane / LG, gl = lane % LG;
    u32* qw = (u32*)smem + (size_t)grp * (ldsq + ldst);
    u32* tw = qw + ldsq;
    const u64 slot = first + grp;
    const bool live = (u32)grp < count;        // a wave's last groups may have no pair: n = m = 0, no cell ever unmasked
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

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// K8a, packed cell (round 6): TWO pairs per lane group, one in each 16-bit half of every state register, so that every add and every maximum of the cell serves two
// cell updates (v_pk_add_i16 / v_pk_max_i16 / v_pk_mad_i16; each issues like one 32-bit max: tools/micro/valu_rates.hip).  19 slow + 1 fast instructions per TWO cells
// against 9.5 slow + 7 fast per cell of the 32-bit kernel above.
//   value   v = (score - a) * 128 - nm   with a = i + j, the cell's anti-diagonal.  score <= 2 min(i, j) <= a, so v <= 0; along an alignment v never rises (a match keeps it,
//           a mismatch costs 6 * 128 + 1, a gap base 2..7 * 128 + 1); the local start (score 0) is v = -128 a, a scalar per step half; -32768 is "not a cell": every add
//           saturates (clamp), so it stays there.
//   exact   for every alignment whose value stays above -32768 on its whole way, the packed recurrence holds the true (score, nm) order: two candidates of a cell compare as
//           the 32-bit kernel's do whenever the winner's nm is below 128, and a candidate with nm >= 128 wins only with a strictly higher score (the argument of the 12-bit nm
//           field above, with 7 bits).  An alignment that does NOT stay above -32768 has 128 (a_end - score) + nm >= 32768 with nm <= (a_end - score) / 2, hence
//           score <= a_end - 255 <= n + m - 255.  So a result with  score >= n + m - 254  is the optimum of the 32-bit kernel: nothing the packed cell cannot hold reaches
//           that score.  The kernel checks this CERTIFICATE per pair; a pair without it is appended to `redo` and runs through the 32-bit cell (the queue kernel's waves take those between their tasks).
//           For full-length 16S pairs the certificate asks for 6 d + |n - m| below ~250: the callers send pairs with |n - m| <= 64 and bands <= 39 here.
//   layout  a wave carries 32 pairs: group g (four lanes x P diagonals, as the sixteen-pair classes) holds pair first + g in the low halves and pair first + 16 + g in the
//           high halves.  Bands, sequence windows, in-band ceilings and matrix masks are per half; the anti-diagonal counter and the loop bounds are the wave's.  The
//           sequences are read from HBM / L2 (two dwords per 16 bases and sequence, asked for one refill ahead): no LDS, so the wave count is not bound by it.
//   flags   Z1 = (yA >> 16) | (yB & 0xFFFF0000) puts the mismatch bit of cell x of pair A at bit 15 - 2x and that of pair B sixteen above it: one shift + one and give the
//           packed 0 / 1 multiplier of v_pk_mad_i16(w, MISM, H).
namespace {
constexpr int S16 = 128;
constexpr int V16_MISM = -6 * S16 - 1, V16_X1 = -3 * S16 - 1, V16_O1 = -7 * S16 - 1, V16_X2 = -2 * S16 - 1, V16_O2 = -26 * S16 - 1;
constexpr u32 NEG2 = 0x80008000u, TOP2 = 0x7FFF7FFFu;
__host__ __device__ constexpr u32 pk2c(int v) { return ((u32)v & 0xFFFFu) * 0x10001u; }
// the packed operations as vector builtins (the compiler knows what it issues: no wait states around them, free scheduling); saturating add = v_pk_add_i16 ... clamp.
// Only the multiply-add keeps its assembly form: there is no builtin for its saturating variant.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s2(u32 x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ u32 as_u(s16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return as_u(__builtin_elementwise_add_sat(as_s2(a), as_s2(b))); }
__device__ __forceinline__ u32 pk_add_s(u32 a, u32 s) { return as_u(__builtin_elementwise_add_sat(as_s2(a), as_s2(s))); }
__device__ __forceinline__ u32 pk_mad_s(u32 w, u32 s, u32 c) { u32 r; asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(w), "s"(s), "v"(c)); return r; }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return as_u(__builtin_elementwise_max(as_s2(a), as_s2(b))); }
__device__ __forceinline__ u32 pk_max_s(u32 a, u32 s) { return as_u(__builtin_elementwise_max(as_s2(a), as_s2(s))); }
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) { return as_u(__builtin_elementwise_min(as_s2(a), as_s2(b))); }
#define AFF16_DPP3(SH) "s_nop 1\n\tv_mov_b32_dpp %0, %3 " SH " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %4 " SH " row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "\tv_mov_b32_dpp %2, %5 " SH " row_mask:0xf bank_mask:0xf bound_ctrl:0"
}

template <int P, int LG>     // LG lanes x P diagonals per pair: bands <= LG P / 2 - 1; 2 * 64 / LG pairs per wave
__device__ __forceinline__ void aff16_pairs(const BatchView& Q, const BatchView& T, const u32* __restrict__ qi, const u32* __restrict__ ti, const u8* __restrict__ rev,
                                            const u32* __restrict__ band, const u32* __restrict__ sel, const u32 first, const u32 count,
                                            int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, u32* __restrict__ redo) {
    static_assert(P % 2 == 0 && P <= 20 && (LG == 4 || LG == 8), "the flags of a step half live in two 16-bit fields; a lane group lies inside a row of sixteen lanes");
    constexpr int NG = 64 / LG;                // lane groups per wave; pair first + g in the low halves of group g, pair first + NG + g in the high halves
    const int lane = threadIdx.x, grp = lane / LG, gl = lane % LG;
    const int d0 = P * gl;
    int n[2], m[2], w[2], wp[2], nwq[2], nwt[2], I[2], J[2]; bool live[2], rc[2]; u64 pid[2]; const u32* qs[2]; const u32* ts[2];
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        const u32 idx = (u32)grp + (u32)NG * h;
        live[h] = idx < count;
        pid[h] = live[h] ? (sel ? sel[first + idx] : (u64)first + idx) : 0;
        const u32 qr = live[h] ? qi[pid[h]] : 0, tr = live[h] ? ti[pid[h]] : 0;
        n[h] = live[h] ? (int)(Q.off[qr + 1] - Q.off[qr]) : 0;
        m[h] = live[h] ? (int)(T.off[tr + 1] - T.off[tr]) : 0;
        w[h] = live[h] ? (int)band[pid[h]] : 0;
        wp[h] = w[h] + (w[h] & 1);
        nwq[h] = (n[h] + 15) / 16; nwt[h] = (m[h] + 15) / 16;
        qs[h] = Q.packed + Q.woff[qr]; ts[h] = T.packed + T.woff[tr];
        rc[h] = rev && live[h] && rev[pid[h]];
        I[h] = (wp[h] - d0) / 2;               // i of diagonal d0 at a = 0 (exact: both even)
        J[h] = I[h] + d0 - wp[h];
    }
    // 16 target bases from base `pos` of the (possibly reverse-complemented) target of half h
    auto tget = [&](int h, int pos) -> u32 { return rc[h] ? aff_revcomp16(aff_get16(ts[h], nwt[h], m[h] - 16 - pos)) : aff_get16(ts[h], nwt[h], pos); };
    u32 H[P], E1[P], E2[P], F1[P], F2[P], CE[P];
    #pragma unroll
    for (int k = 0; k < P; k++) {
        H[k] = E1[k] = E2[k] = F1[k] = F2[k] = NEG2;
        u32 ce = 0;
        #pragma unroll
        for (int h = 0; h < 2; h++) { const int dd = d0 + k - wp[h]; ce |= ((live[h] && dd >= -w[h] && dd <= w[h]) ? 0x7FFFu : 0x8000u) << (16 * h); }
        CE[k] = ce;
    }
    u64 QW[2], TW[2]; u32 QF[2], nQF[2], nTW[2];
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        u64 qw = 0;                            // q[I-1-x] in bits 62-2x: the query bases of the P / 2 cells of a step, descending
        #pragma unroll
        for (int x = 0; x < P / 2; x++) {
            const int idx = I[h] - 1 - x;
            const u32 b = (idx >= 0 && idx < n[h]) ? ((qs[h][idx >> 4] >> (30 - 2 * (idx & 15))) & 3u) : 0u;
            qw |= (u64)b << (62 - 2 * x);
        }
        QW[h] = qw;
        QF[h] = aff_get16(qs[h], nwq[h], I[h]);
        TW[h] = ((u64)tget(h, J[h] - 1) << 32) | tget(h, J[h] + 15);
        nQF[h] = aff_get16(qs[h], nwq[h], I[h] + 16); nTW[h] = tget(h, J[h] + 16 + 15);     // the words of the first refill (s = 16)
    }
    int adv = 0;
    int best32[2] = {-(1 << 30), -(1 << 30)};
    const u32 FIXL = gl == 0 ? NEG2 : TOP2, FIXR = gl == LG - 1 ? NEG2 : TOP2;
    u32 cX1 = pk2c(V16_X1), cO1 = pk2c(V16_O1), cX2 = pk2c(V16_X2), cO2 = pk2c(V16_O2), cMM = pk2c(V16_MISM), cS = pk2c(S16);
    asm volatile("" : "+s"(cX1), "+s"(cO1), "+s"(cX2), "+s"(cO2), "+s"(cMM), "+s"(cS));
    u32 bm;                                    // maximum over the cells of a step half (both pairs)
    auto cell = [&](auto mask_c, const int k, const int a, const u32 wf, const u32 e1a, const u32 e1b, const u32 e2a, const u32 e2b,
                    const u32 f1a, const u32 f1b, const u32 f2a, const u32 f2b, const u32 floor2, const bool first_cell) {
        constexpr bool MASK = decltype(mask_c)::value;
        const u32 hd = pk_mad_s(wf, cMM, H[k]);
        u32 e1 = pk_max(e1a, e1b), e2 = pk_max(e2a, e2b), f1 = pk_max(f1a, f1b), f2 = pk_max(f2a, f2b);
        if (k == 0) { e1 = pk_min(e1, FIXL); e2 = pk_min(e2, FIXL); }
        if (k == P - 1) { f1 = pk_min(f1, FIXR); f2 = pk_min(f2, FIXR); }
        const u32 x = pk_max(pk_max(hd, e1), pk_max(pk_max(e2, f1), f2));
        u32 h = pk_min(pk_max_s(x, floor2), CE[k]);                  // local start, then the band ceiling: a diagonal outside the band is "not a cell"
        if (MASK) {
            u32 keep = 0;
            #pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int dd = d0 + k - wp[hh];
                const int lo = dd < 0 ? -dd : dd, hi = min(2 * n[hh] + dd, 2 * m[hh] - dd);
                if (!(a < lo || a > hi)) keep |= 0xFFFFu << (16 * hh);
            }
            h = (h & keep) | (NEG2 & ~keep);
        }
        H[k] = h; E1[k] = e1; E2[k] = e2; F1[k] = f1; F2[k] = f2;
        bm = first_cell ? h : pk_max(bm, h);
    };
    auto run = [&](auto mask_c, int& a, const int a_end) {
        for (; a < a_end; a += 2) {
            u32 fl_e, fl_o;                                         // local start of the two step halves: -128 a, not a cell once that leaves 16 bits
            { const int fa = a < 256 ? -S16 * a : -32768, fb = a + 1 < 256 ? -S16 * (a + 1) : -32768;
              fl_e = (u32)__builtin_amdgcn_readfirstlane((int)pk2c(fa)); fl_o = (u32)__builtin_amdgcn_readfirstlane((int)pk2c(fb)); }
            u32 bm_e;
            {   // even step a: diagonals d0 + 2x
                const u32 XA = (u32)(QW[0] >> 32) ^ (u32)(TW[0] >> 32), XB = (u32)(QW[1] >> 32) ^ (u32)(TW[1] >> 32);
                const u32 yA = XA | (XA << 1), yB = XB | (XB << 1);
                const u32 Z1 = (yA >> 16) | (yB & 0xFFFF0000u), Z2 = (yA & 0xFFFFu) | (yB << 16);
                u32 lE1, lH, lE2;
                asm(AFF16_DPP3("row_shr:1") : "=&v"(lE1), "=&v"(lH), "=&v"(lE2) : "v"(E1[P - 1]), "v"(H[P - 1]), "v"(E2[P - 1]));
                u32 pe1a = pk_add_s(lE1, cX1), pe1b = pk_add_s(lH, cO1), pe2a = pk_add_s(lE2, cX2), pe2b = pk_add_s(lH, cO2);   // the sums the E side of the next even cell takes
                #pragma unroll
                for (int x = 0; x < P / 2; x++) {
                    const int k = 2 * x;
                    const u32 hO1 = pk_add_s(H[k + 1], cO1), hO2 = pk_add_s(H[k + 1], cO2);
                    const u32 wf = ((x < 8 ? Z1 : Z2) >> (15 - 2 * (x & 7))) & 0x00010001u;
                    cell(mask_c, k, a, wf, pe1a, pe1b, pe2a, pe2b, pk_add_s(F1[k + 1], cX1), hO1, pk_add_s(F2[k + 1], cX2), hO2, fl_e, x == 0);
                    if (x + 1 < P / 2) { pe1a = pk_add_s(E1[k + 1], cX1); pe1b = hO1; pe2a = pk_add_s(E2[k + 1], cX2); pe2b = hO2; }
                }
                bm_e = bm;
            }
            {   // odd step a + 1: diagonals d0 + 2x + 1 (same query bases, targets one further)
                const u32 XA = (u32)(QW[0] >> 32) ^ (u32)((TW[0] << 2) >> 32), XB = (u32)(QW[1] >> 32) ^ (u32)((TW[1] << 2) >> 32);
                const u32 yA = XA | (XA << 1), yB = XB | (XB << 1);
                const u32 Z1 = (yA >> 16) | (yB & 0xFFFF0000u), Z2 = (yA & 0xFFFFu) | (yB << 16);
                u32 rF1, rH, rF2;
                asm(AFF16_DPP3("row_shl:1") : "=&v"(rF1), "=&v"(rH), "=&v"(rF2) : "v"(F1[0]), "v"(H[0]), "v"(F2[0]));
                u32 pe1a = pk_add_s(E1[0], cX1), pe1b = pk_add_s(H[0], cO1), pe2a = pk_add_s(E2[0], cX2), pe2b = pk_add_s(H[0], cO2);
                #pragma unroll
                for (int x = 0; x < P / 2; x++) {
                    const int k = 2 * x + 1;
                    const bool in = k + 1 < P;
                    const u32 hn = in ? H[in ? k + 1 : 0] : rH;
                    const u32 hO1 = pk_add_s(hn, cO1), hO2 = pk_add_s(hn, cO2);
                    const u32 wf = ((x < 8 ? Z1 : Z2) >> (15 - 2 * (x & 7))) & 0x00010001u;
                    cell(mask_c, k, a + 1, wf, pe1a, pe1b, pe2a, pe2b, pk_add_s(in ? F1[in ? k + 1 : 0] : rF1, cX1), hO1, pk_add_s(in ? F2[in ? k + 1 : 0] : rF2, cX2), hO2, fl_o, x == 0);
                    if (in) { pe1a = pk_add_s(E1[k + 1], cX1); pe1b = hO1; pe2a = pk_add_s(E2[k + 1], cX2); pe2b = hO2; }
                }
            }
            {   // the best cell so far, in absolute units score * 128 - nm: the odd half lifted to the even half's anti-diagonal, one unpack per double step.  A value that
                // saturated (-32768, or -32640 after the lift) is no result; nothing below -32639 can carry the certificate either
                const u32 b2 = pk_max(bm_e, pk_add_s(bm, cS));
                const int vA = (int)(b2 << 16) >> 16, vB = (int)b2 >> 16;
                const int base = S16 * a;
                if (vA >= -32639) best32[0] = max(best32[0], vA + base);
                if (vB >= -32639) best32[1] = max(best32[1], vB + base);
            }
            #pragma unroll
            for (int h = 0; h < 2; h++) { QW[h] = (QW[h] >> 2) | ((u64)(QF[h] >> 30) << 62); QF[h] <<= 2; TW[h] <<= 2; }
            if (++adv == 16) {
                adv = 0;
                const int s = a / 2 + 1;
                #pragma unroll
                for (int h = 0; h < 2; h++) {
                    QF[h] = nQF[h]; TW[h] |= (u64)nTW[h];
                    nQF[h] = aff_get16(qs[h], nwq[h], I[h] + s + 16); nTW[h] = tget(h, J[h] + s + 16 + 15);     // asked for one refill ahead
                }
            }
        }
    };
    // masked while a band enters its matrix (a < w) and while one leaves it (a + 1 > min(2n, 2m) - w); unmasked in between.  A half without a pair never holds a cell
    // (its ceilings are "not a cell"), so it does not bound the loops
    int S0 = 0, S1 = 0x7FFFFFFF, END = 0;
    #pragma unroll
    for (int h = 0; h < 2; h++) if (live[h]) {
        S0 = max(S0, (w[h] + 1) & ~1);
        const int last = min(2 * n[h], 2 * m[h]) - w[h];
        S1 = min(S1, last >= 1 ? ((last - 1) & ~1) + 2 : 0);
        END = max(END, n[h] + m[h] + 1);
    }
    #pragma unroll
    for (int s = LG; s < 64; s <<= 1) { S0 = max(S0, __shfl_xor(S0, s)); S1 = min(S1, __shfl_xor(S1, s)); END = max(END, __shfl_xor(END, s)); }
    int a = 0;
    run(std::true_type{}, a, min(S0, END));
    run(std::false_type{}, a, min(S1, END));
    run(std::true_type{}, a, END);
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        int b = best32[h];
        #pragma unroll
        for (int sft = LG / 2; sft >= 1; sft >>= 1) b = max(b, __shfl_xor(b, sft));
        if (gl == 0 && live[h]) {
            const int score = b > 0 ? (b + S16 - 1) / S16 : 0;
            if (b > 0 && score >= n[h] + m[h] - 254) { nm_out[pid[h]] = score * S16 - b; if (score_out) score_out[pid[h]] = score; }
            else { nm_out[pid[h]] = 0x7FFFFFFF; __hip_atomic_store(&redo[4 + atomicAdd(redo, 1u)], (u32)pid[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // no certificate: the 32-bit cell decides (the queue kernel's waves pick these up between their tasks)
        }
    }
}
template <int P, int LG>
__device__ __noinline__ void aff16_pairs_fn(const BatchView* Q, const BatchView* T, const u32* __restrict__ qi, const u32* __restrict__ ti, const u8* __restrict__ rev, const u32* __restrict__ band,
                                            const u32* __restrict__ sel, u32 first, u32 count, int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, u32* __restrict__ redo) {
    aff16_pairs<P, LG>(*Q, *T, qi, ti, rev, band, sel, first, count, nm_out, score_out, redo);
}
// One pair-class per kernel: the launch of round 4 (svt_set_option "k8a_queue" = 0; also what tools/k8a_isa_mix.py counts the steady loops in)
template <int P, int G>
__global__ void __launch_bounds__(64) k_align_affine(BatchView Q, BatchView T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                                     const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel, u64 n_sel,
                                                     int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, u32 ldsq, u32 ldst) {
    extern __shared__ __align__(16) unsigned char smem[];
    const u64 first = (u64)blockIdx.x * G;
    aff_pairs<P, G>(Q, T, qi, ti, rev, band, sel, first, (u32)min((u64)G, n_sel - first), nm_out, score_out, ldsq, ldst, smem);
}

// Band classes (round 5).  A pair occupies 64 / G lanes x P diagonals: what a wave carries beyond 2 w + 1 is lost work, and the DPP moves + the
// sequence-window shifts of a step (~30 instructions) are shared by the P cells a lane updates in it.  Round 4 ran eight pairs per wave on 6-12
// diagonals per lane: classes 16 diagonals apart (11 % of the lanes' cells outside the band) at 19-21.5 instructions per cell.  Now SIXTEEN pairs per
// wave on four lanes x 8-20 diagonals for the bands Stage 7 produces: classes 8 diagonals apart, 18-19 instructions per cell; the eight-pair classes stay
// for sequences whose 2-bit words do not fit sixteen to a wave in LDS (AFF_LDS_BUDGET) and for bands 40-63.
//   cls    0     1     2     3     4     5     6   |  7    8    9    10    11    12  |  13    14    15  |  16    17
//   P,G    8,16 10,16 12,16 14,16 16,16 18,16 20,16 | 6,8  8,8  10,8  12,8  14,8  16,8 | 10,4  12,4  16,4 | 16,2  16,1
//   w <=   15    19    23    27    31    35    39   | 23   31   39    47    55    63   | 79    95    127  | 255   511
// The class of a pair: the fewest diagonals carried among the classes that hold its band and fit the LDS budget with `lds_words` words per pair.
int affine_class_of(u32 w, u32 lds_words, int max_g) {
    int best = AFF_NCLS - 1;
    for (int cls = AFF_NCLS - 1; cls >= 0; cls--) {
        if ((int)w > 32 * AFF_P[cls] / AFF_G[cls] - 1 || AFF_G[cls] > max_g) continue;
        if (AFF_G[cls] > 1 && (size_t)AFF_G[cls] * lds_words * 4 > AFF_LDS_BUDGET) continue;
        if (64 * AFF_P[cls] / AFF_G[cls] <= 64 * AFF_P[best] / AFF_G[best]) best = cls;      // ties: the earlier class (more pairs per wave)
    }
    return best;
}
// the packed cell's class of a band (-1: none holds it): the fewest diagonals among four lanes x 8..20 and eight lanes x 12..16; no LDS, so the sequence lengths do not matter
int affine16_class_of(u32 w) {
    for (int k = 0; k < AFF16_NCLS; k++) if ((int)w <= AFF16_LG[k] * AFF16_P[k] / 2 - 1) return AFF_NCLS + k;
    return -1;
}
double affine_task_cost(int cls, u32 steps) {
    if (cls >= AFF_NCLS) return (double)steps * (20.0 * AFF16_P[cls - AFF_NCLS] + 90.0) * 1.15;    // the packed cell: 20 instructions per diagonal and double step for two pairs, nearly all of the slow kind
    return (double)steps * (16.5 * AFF_P[cls] + 30.0);
}   // VALU instructions of a wave that walks `steps` double steps (ISA counts: profiles/r05_k8a_isa_mix.json)

// the packed cell: (id, P, LG): LG lanes x P diagonals per pair, 128 / LG pairs per wave; bands <= 15 .. 39 on four lanes, <= 47 / 55 / 63 on eight (ids follow the 32-bit classes)
#define SVT_K8A16_CLASSES(X) X(0, 8, 4) X(1, 10, 4) X(2, 12, 4) X(3, 14, 4) X(4, 16, 4) X(5, 18, 4) X(6, 20, 4) X(7, 12, 8) X(8, 14, 8) X(9, 16, 8)
#define SVT_K8A_CLASSES(X) X(0, 8, 16) X(1, 10, 16) X(2, 12, 16) X(3, 14, 16) X(4, 16, 16) X(5, 18, 16) X(6, 20, 16) X(7, 6, 8) X(8, 8, 8) X(9, 10, 8) X(10, 12, 8) \
    X(11, 14, 8) X(12, 16, 8) X(13, 10, 4) X(14, 12, 4) X(15, 16, 4) X(16, 16, 2) X(17, 16, 1)

// the class bodies are FUNCTIONS of the queue kernel, not inlined into it: inlined, the register allocator works on eighteen bodies at once and spills
template <int P, int G>
__device__ __noinline__ void aff_pairs_fn(const BatchView* Q, const BatchView* T, const u32* __restrict__ qi, const u32* __restrict__ ti, const u8* __restrict__ rev, const u32* __restrict__ band,
                                          const u32* __restrict__ sel, u32 first, u32 count, int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, u32 ldsq, u32 ldst, unsigned char* smem) {
    aff_pairs<P, G>(*Q, *T, qi, ti, rev, band, sel, first, count, nm_out, score_out, ldsq, ldst, smem);
}

// ONE launch for all classes (round 5): a task is one wave's pairs -- up to G pairs of ONE class, neighbours in length -- and the waves of a grid that just fills the chip
// draw tasks from a counter until none is left.  The host orders the tasks by falling cost (long waves of the wide classes first, the short waves of the narrow classes
// last), so that the last round of waves is the cheapest the call has: round 4 launched every class on a stream of its own and lost ~1 ms per call to the classes' tails.
// tasks[t] = {first position in sel, count | class << 8}; counter[0] = tasks drawn so far beyond the grid's first round (zeroed with the upload).
__global__ void __launch_bounds__(64, 2) k_align_affine_q(BatchView Q, BatchView T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                                          const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel,
                                                          const uint2* __restrict__ tasks, u32 n_tasks, u32* __restrict__ counter,
                                                          int32_t* __restrict__ nm_out, int32_t* __restrict__ score_out, u32 ldsq, u32 ldst, u32* __restrict__ redo, u32 n_packed_tasks) {
    extern __shared__ __align__(16) unsigned char smem[];
    // The pairs the packed cell gives no certificate for are rerun through the 32-bit cell BY THIS KERNEL: redo[0] = pairs appended, redo[1] = pairs claimed, redo[2] = packed
    // tasks finished, entries from redo[4] (0xFFFFFFFF until written).  Between two tasks a wave takes eight of them when eight are waiting; once every packed task has
    // finished whoever comes by takes what is left.  (A launch of its own behind this one cost the time of a whole wave -- ~1.5 ms -- for a handful of pairs.)
    auto redo_chunk = [&](const bool final_) -> bool {          // claim up to eight appended pairs and align them; false: nothing to claim
        u32 c0 = 0, cn = 0;
        if (threadIdx.x == 0) {
            for (;;) {
                const u32 app = __hip_atomic_load(&redo[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), cl = __hip_atomic_load(&redo[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 avail = app > cl ? app - cl : 0;
                if (avail == 0 || (!final_ && avail < 8)) break;
                const u32 take = avail < 8 ? avail : 8;
                if (atomicCAS(&redo[1], cl, cl + take) == cl) { c0 = cl; cn = take; break; }
            }
        }
        c0 = (u32)__builtin_amdgcn_readfirstlane((int)c0); cn = (u32)__builtin_amdgcn_readfirstlane((int)cn);
        if (cn == 0) return false;
        if (threadIdx.x < cn) while (__hip_atomic_load(&redo[4 + c0 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xFFFFFFFFu) __builtin_amdgcn_s_sleep(1);   // counted before written
        __syncthreads();
        aff_pairs_fn<16, 8>(&Q, &T, qi, ti, rev, band, redo + 4, c0, cn, nm_out, score_out, ldsq, ldst, smem);     // bands <= 63, eight pairs
        __syncthreads();
        return true;
    };
    u32 t = blockIdx.x;
    while (t < n_tasks) {
        const uint2 tk = tasks[t];
        const u32 first = (u32)__builtin_amdgcn_readfirstlane((int)tk.x), cc = (u32)__builtin_amdgcn_readfirstlane((int)tk.y);
        switch (cc >> 8) {
#define X(ID, PP, GG) case ID: aff_pairs_fn<PP, GG>(&Q, &T, qi, ti, rev, band, sel, first, cc & 0xFF, nm_out, score_out, ldsq, ldst, smem); break;
            SVT_K8A_CLASSES(X)
#undef X
#define X(ID, PP, LL) case AFF_NCLS + ID: aff16_pairs_fn<PP, LL>(&Q, &T, qi, ti, rev, band, sel, first, cc & 0xFF, nm_out, score_out, redo); break;
            SVT_K8A16_CLASSES(X)
#undef X
            default: break;
        }
        __syncthreads();                                   // the next task's staging overwrites the sequences in LDS
        if ((cc >> 8) >= AFF_NCLS) { __threadfence(); if (threadIdx.x == 0) atomicAdd(&redo[2], 1u); }     // this task's appends are out before it counts as finished
        if (n_packed_tasks) redo_chunk(__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&redo[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= (int)n_packed_tasks);
        u32 nx = 0;
        if (threadIdx.x == 0) nx = atomicAdd(counter, 1u);
        t = gridDim.x + (u32)__builtin_amdgcn_readfirstlane((int)nx);
    }
    // Out of tasks: take what is waiting, and leave.  NO wave waits for another one (with several samples on the chip the blocks of a launch are not all resident: a wave
    // that waited for a block that has not started yet would hold the slot that block needs).  Nothing is lost: the wave that finishes the LAST packed task counts it, sees
    // the count complete and takes all that is left, here or after its remaining tasks.
    if (n_packed_tasks) {
        for (;;) {
            const bool all_done = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&redo[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= (int)n_packed_tasks;
            if (!redo_chunk(all_done)) break;
        }
    }
}

const char* affine_class_name(int cls) {
    static const char* names[AFF_NCLS + AFF16_NCLS] = {
#define X(ID, PP, GG) "k_align_affine_p" #PP "g" #GG,
        SVT_K8A_CLASSES(X)
#undef X
#define X(ID, PP, LL) "k_align_affine16_p" #PP "l" #LL,
        SVT_K8A16_CLASSES(X)
#undef X
    };
    return names[cls];
}
int launch_align_affine(svt_ctx* c, hipStream_t on, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                        const u32* d_sel, u64 n_sel, int cls, int32_t* d_nm, int32_t* d_score, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells) {
    if (n_sel == 0) return SVT_OK;
    const int G = AFF_G[cls];
    u32 ldsq = (max_qlen + 15) / 16 + 2, ldst = (max_tlen + 15) / 16 + 2;
    size_t sh = (size_t)(ldsq + ldst) * 4 * G;
    ProfScope ps(c, affine_class_name(cls), algo_bytes, cells, on);
    BatchView qv = Q->view(), tv = T->view();
    const dim3 grid((u32)((n_sel + G - 1) / G));
    switch (cls) {
#define X(ID, PP, GG) case ID: hipLaunchKernelGGL((k_align_affine<PP, GG>), grid, dim3(64), sh, on, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, d_score, ldsq, ldst); break;
        SVT_K8A_CLASSES(X)
#undef X
        default: return svt_fail(c, SVT_ERR_ARG, "K8a: no such band class");
    }
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
// the queue launch: d_tasks[n_tasks] and the zeroed d_counter are on the device; max_g = the most pairs a task of this call holds (sizes the LDS of a wave)
int launch_align_affine_queue(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band, const u32* d_sel,
                              const void* d_tasks, u32 n_tasks, u32* d_counter, int max_g, int32_t* d_nm, int32_t* d_score, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells,
                              u32* d_redo, u64 n_packed, u32 n_packed_tasks) {
    if (n_tasks == 0) return SVT_OK;
    u32 ldsq = (max_qlen + 15) / 16 + 2, ldst = (max_tlen + 15) / 16 + 2;
    size_t sh = (size_t)(ldsq + ldst) * 4 * max_g;
    static int cus = 0;
    if (!cus) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, c->device) == hipSuccess) cus = pr.multiProcessorCount; if (cus <= 0) cus = 256; }
    // the grid that fills the chip: 4 SIMDs x 2 waves per CU by the kernel's registers (the widest class body takes ~200), fewer when a wave's sequences take more than an eighth of the CU's LDS
    u32 per_cu = 8;
    if (sh * per_cu > (size_t)160 * 1024) per_cu = (u32)std::max<size_t>(1, (size_t)160 * 1024 / sh);
    const u32 grid = std::min<u32>(n_tasks, (u32)cus * per_cu);
    ProfScope ps(c, "k_align_affine_span", algo_bytes, cells);
    BatchView qv = Q->view(), tv = T->view();
    if (n_packed) {                                               // redo: four header words (appended, claimed, packed tasks finished), then the entries, 0xFFFFFFFF until written
        HIPCHK(c, hipMemsetAsync(d_redo, 0xFF, (size_t)(n_packed + 4) * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d_redo, 0, 16, c->stream));
    }
    hipLaunchKernelGGL(k_align_affine_q, dim3(grid), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, (const uint2*)d_tasks, n_tasks, d_counter, d_nm, d_score, ldsq, ldst, d_redo, n_packed_tasks);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
