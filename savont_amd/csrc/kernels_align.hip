// kernels_align.hip -- K8: banded OVERLAP edit distance (the build's replacement for minimap2 `nm`
// at src/alignment.rs:1848-1862, :2135-2148; contract in DESIGN.md section 3 and the oracle header).
//
//   cells (i,j), 0<=i<=n (query = ASV), 0<=j<=m (target = read or its reverse complement),
//   band |j-i| <= w, D(0,j) = D(i,0) = 0, unit costs, result = min over the last row / last column.
//
// Mapping (MI355X, integer DP -- no MFMA): one wavefront per pair, the BAND's diagonals live on the
// lanes.  Lane l owns P = 4R consecutive diagonals d = P*l .. P*l+P-1 (d = j - i + w', w' = w
// rounded up to even).  Anti-diagonal a = i + j advances one step at a time; on even steps the even
// diagonals are active, on odd steps the odd ones.  Two diagonals of equal parity share one VGPR as
// packed u16 (v_pk_add_u16 / v_pk_min_u16 / v_pk_max_u16), so one step of a 256-diagonal band is
// ~10 VALU ops per lane.  The recurrence in (d,a) coordinates:
//     D[d][a] = min( D[d][a-2] + (q[i-1] != t[j-1]),  D[d-1][a-1] + 1,  D[d+1][a-1] + 1 )
// needs only the neighbouring diagonals: in-register for inner ones, one DPP wave_shr/wave_shl for
// the lane boundary (no LDS traffic for the DP state).  Sequences are staged ONCE per pair into LDS
// as 2-bit words (the target is reverse-complemented while staging); each lane keeps sliding 2-bit
// windows of q (descending) and t (ascending) in registers and refills them from LDS every 16 bases.
#include <type_traits>
#include "svt_internal.hpp"

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define INF16 0x3FFFu
#define INFPK (INF16 | (INF16 << 16))
#define ONEPK 0x00010001u

__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { us2 r = __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b); return __builtin_bit_cast(u32, r); }
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) { us2 r = __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)); return __builtin_bit_cast(u32, r); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { us2 r = __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)); return __builtin_bit_cast(u32, r); }
// value of lane-1 (lane 0 gets INFPK) / lane+1 (lane 63 gets INFPK)
__device__ __forceinline__ u32 from_left(u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)INFPK, (int)v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ u32 from_right(u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)INFPK, (int)v, 0x130, 0xF, 0xF, false); }

__device__ __forceinline__ u32 get16(const u32* lds, int nw, int pos) {      // 16 bases from base `pos` (any int), zero outside
    int wi = pos >> 4; u32 o = (u32)(pos & 15) * 2;
    u32 w0 = (wi >= 0 && wi < nw) ? lds[wi] : 0u;
    u32 w1 = (wi + 1 >= 0 && wi + 1 < nw) ? lds[wi + 1] : 0u;
    return o ? ((w0 << o) | (w1 >> (32 - o))) : w0;
}
__device__ __forceinline__ u32 revcomp16(u32 x) {                            // reverse complement of 16 packed bases
    u32 y = __brev(~x);
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}

__device__ __forceinline__ u32 fix_half(u32 v, int d, int lo, int hi) {
    if (d < lo || d > hi) return INF16;
    if (d == lo || d == hi) return 0;
    return v;
}
__device__ __forceinline__ u32 fix_pk(u32 v, int dlo, int dhi, int lo, int hi) {
    return fix_half(v & 0xFFFF, dlo, lo, hi) | (fix_half(v >> 16, dhi, lo, hi) << 16);
}
__device__ __forceinline__ void extract_pk(u32 v, int dlo, int dhi, int dr, int dc, u32& best) {
    u32 a = v & 0xFFFF, b = v >> 16;
    if (dlo == dr || dlo == dc) best = min(best, a);
    if (dhi == dr || dhi == dc) best = min(best, b);
}

// traceback side outputs of the K9 variant (TB = true)
struct TbOut {
    u32* tb;                  // direction bits: per pair a slab of tb_dwords*64 dwords, [step group][lane]
    u64 tb_stride;            // dwords per pair
    u64* cells;               // pile-up rows (one u64 per query position), row of pair p starts at cell_off[p]
    const u64* cell_off;
    u32* span;                // 4 per pair: q_start, q_end, t_start, t_end
    const u8* qualbins;       // target batch quality bins (two per byte) or nullptr
    const u64* qb_off;
    const u8* tag_qual;       // per-base tags of the target batch (svt_batch_set_tags: homopolymer-compressed reads): quality byte and run
    const u8* tag_hp;         // length of every base at T.off[read] + position; when present they replace the quality bins, and a Base
                              // entry carries its run length in bits 56-63 (src/alignment.rs:480,532-538)
};

// K9: 2 direction bits per cell (0 diagonal, 1 up = deletion in the read, 2 left = insertion in the read), priority in that order
__device__ __forceinline__ u32 dir_half(u32 v, u32 cd, u32 cu) { return v == cd ? 0u : (v == cu ? 1u : 2u); }
__device__ __forceinline__ u32 dir_bits(u32 v, u32 cdiag, u32 cup) {
    return dir_half(v & 0xFFFF, cdiag & 0xFFFF, cup & 0xFFFF) | (dir_half(v >> 16, cdiag >> 16, cup >> 16) << 2);
}
__device__ __forceinline__ void extract_key(u32 v, int dlo, int dhi, int dr, int dc, int a, u64& best) {
    if (dlo == dr || dlo == dc) { u64 k = ((u64)(v & 0xFFFF) << 40) | ((u64)a << 16) | (u64)dlo; best = k < best ? k : best; }
    if (dhi == dr || dhi == dc) { u64 k = ((u64)(v >> 16) << 40) | ((u64)a << 16) | (u64)dhi; best = k < best ? k : best; }
}

template <int R, bool TB>
__global__ void __launch_bounds__(64) k_align(BatchView Q, BatchView T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                              const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel, u64 n_sel,
                                              int32_t* __restrict__ nm_out, u32 ldsq, u32 ldst, TbOut tbo) {
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
    {   // stage sequences
        const u32* qs = Q.packed + Q.woff[qr];
        const u32* ts = T.packed + T.woff[tr];
        for (int i = lane; i < nwq; i += 64) qw[i] = qs[i];
        if (!(rev && rev[pid])) { for (int i = lane; i < nwt; i += 64) tw[i] = ts[i]; }
        else {
            for (int i = lane; i < nwt; i += 64) {
                int start = m - 16 * i - 16;                      // t'[16i .. 16i+15] = revcomp(t[start .. start+15])
                int wi = start >> 4; u32 o = (u32)(start & 15) * 2;
                u32 w0 = (wi >= 0 && wi < nwt) ? ts[wi] : 0u, w1 = (wi + 1 >= 0 && wi + 1 < nwt) ? ts[wi + 1] : 0u;
                u32 x = o ? ((w0 << o) | (w1 >> (32 - o))) : w0;
                tw[i] = revcomp16(x);
            }
        }
    }
    __syncthreads();
    constexpr int P = 4 * R;
    const int d0 = P * lane;
    u32 E[R], O[R], FE[R], FO[R];
    #pragma unroll
    for (int r = 0; r < R; r++) {
        E[r] = INFPK; O[r] = INFPK;
        auto fl = [&](int d) -> u32 { return (d < wp - w || d > wp + w) ? INF16 : 0u; };
        FE[r] = fl(d0 + 4 * r) | (fl(d0 + 4 * r + 2) << 16);
        FO[r] = fl(d0 + 4 * r + 1) | (fl(d0 + 4 * r + 3) << 16);
    }
    int I = (wp - d0) / 2;                 // exact: both even
    int J = I + d0 - wp;
    u64 QW = 0;
    #pragma unroll
    for (int x = 0; x < 2 * R; x++) {
        int idx = I - 1 - x;
        u32 b = (idx >= 0 && idx < n) ? ((qw[idx >> 4] >> (30 - 2 * (idx & 15))) & 3u) : 0u;
        QW |= (u64)b << (62 - 2 * x);
    }
    u32 QF = get16(qw, nwq, I);
    u64 TW = ((u64)get16(tw, nwt, J - 1) << 32) | get16(tw, nwt, J + 15);
    int adv = 0;
    u32 best = INF16;
    u64 best_key = ~0ull;                          // TB: value<<40 | a<<16 | d  (ties: smallest anti-diagonal, then smallest diagonal)
    constexpr int SPD = 32 / (4 * R);              // TB: steps packed per dword (2 bits per cell, 2R cells per lane and step)
    u32 tb_acc = 0;
    u32* tb_base = TB ? tbo.tb + (u64)blockIdx.x * tbo.tb_stride : nullptr;
    const int total = n + m;
    int tail_start = 2 * min(n, m) - w; if (tail_start < 0) tail_start = 0;
    // Three specialised loops over the even anti-diagonals a (each iteration = steps a and a+1):
    //   FORCE   (a <= w+1): free-start edges D(0,j)=D(i,0)=0 and cells outside the matrix are imposed after each update;
    //   EXTRACT (a+1 >= tail_start): cells of the last row / last column are harvested into `best`;
    //   the steady state in between has neither, so its body is ~25 VALU per step pair and register.
    auto run = [&](auto force_c, auto extract_c, int& a, const int a_end) {
        constexpr bool FORCE = decltype(force_c)::value, EXTRACT = decltype(extract_c)::value;
        for (; a < a_end; a += 2) {
            // ---- even step a: even diagonals d0+4r, d0+4r+2
            {
                u32 X = (u32)(QW >> 32) ^ (u32)(TW >> 32);
                u32 y = X | (X << 1);
                u32 OL = from_left(O[R - 1]);
                #pragma unroll
                for (int r = 0; r < R; r++) {
                    u32 neq = ((y >> (31 - 4 * r)) & 1u) | (((y >> (29 - 4 * r)) & 1u) << 16);
                    u32 L = __builtin_amdgcn_alignbit(O[r], r == 0 ? OL : O[r - 1], 16);      // (o[d-1]) pairs
                    u32 mn = pk_min(O[r], L);
                    const u32 cdiag = pk_add(E[r], neq);
                    u32 v = pk_min(cdiag, pk_add(mn, ONEPK));
                    if (TB) tb_acc |= dir_bits(v, cdiag, pk_add(O[r], ONEPK)) << (4 * R * (a % SPD) + 4 * r);      // up = diagonal d+1 = O[r]
                    if (FORCE) v = fix_pk(v, d0 + 4 * r, d0 + 4 * r + 2, wp - a, wp + a);
                    v = pk_max(v, FE[r]);
                    E[r] = v;
                    if (EXTRACT) { if (TB) extract_key(v, d0 + 4 * r, d0 + 4 * r + 2, a + wp - 2 * n, 2 * m + wp - a, a, best_key);
                                   else extract_pk(v, d0 + 4 * r, d0 + 4 * r + 2, a + wp - 2 * n, 2 * m + wp - a, best); }
                }
            }
            // ---- odd step a+1: odd diagonals d0+4r+1, d0+4r+3
            {
                const int a1 = a + 1;
                u32 X = (u32)(QW >> 32) ^ (u32)((TW << 2) >> 32);
                u32 y = X | (X << 1);
                u32 ER = from_right(E[0]);
                #pragma unroll
                for (int r = 0; r < R; r++) {
                    u32 neq = ((y >> (31 - 4 * r)) & 1u) | (((y >> (29 - 4 * r)) & 1u) << 16);
                    u32 Rr = __builtin_amdgcn_alignbit(r == R - 1 ? ER : E[r + 1], E[r], 16);  // (e[d+1]) pairs
                    u32 mn = pk_min(E[r], Rr);
                    const u32 cdiag = pk_add(O[r], neq);
                    u32 v = pk_min(cdiag, pk_add(mn, ONEPK));
                    if (TB) tb_acc |= dir_bits(v, cdiag, pk_add(Rr, ONEPK)) << (4 * R * (a1 % SPD) + 4 * r);        // up = diagonal d+1 = Rr
                    if (FORCE) v = fix_pk(v, d0 + 4 * r + 1, d0 + 4 * r + 3, wp - a1, wp + a1);
                    v = pk_max(v, FO[r]);
                    O[r] = v;
                    if (EXTRACT) { if (a1 <= total) { if (TB) extract_key(v, d0 + 4 * r + 1, d0 + 4 * r + 3, a1 + wp - 2 * n, 2 * m + wp - a1, a1, best_key);
                                                      else extract_pk(v, d0 + 4 * r + 1, d0 + 4 * r + 3, a1 + wp - 2 * n, 2 * m + wp - a1, best); } }
                }
            }
            if (TB) { if (((a + 1) % SPD) == SPD - 1) { tb_base[(u64)(a / SPD) * 64 + lane] = tb_acc; tb_acc = 0; } }
            // ---- advance one base
            QW = (QW >> 2) | ((u64)(QF >> 30) << 62);
            QF <<= 2;
            TW <<= 2;
            if (++adv == 16) {
                adv = 0;
                const int s = a / 2 + 1;
                QF = get16(qw, nwq, I + s);
                TW |= (u64)get16(tw, nwt, J + s + 15);
            }
        }
    };
    const int A1 = (w + 3) & ~1;                                   // first even a with a > w+1
    int T0 = (tail_start - 1) & ~1; if (T0 < 0) T0 = 0;           // first even a with a+1 >= tail_start
    const int END = total + 1;
    int a = 0;
    run(std::true_type{}, std::false_type{}, a, min(min(A1, T0), END));
    run(std::true_type{}, std::true_type{}, a, min(A1, END));      // only when the tail starts inside the prologue (short sequences)
    run(std::false_type{}, std::false_type{}, a, min(T0, END));
    run(std::false_type{}, std::true_type{}, a, END);
    if (!TB) {
        #pragma unroll
        for (int s = 32; s >= 1; s >>= 1) best = min(best, (u32)__shfl_xor((int)best, s));
        if (lane == 0) nm_out[pid] = best >= INF16 ? 0x7FFFFFFF : (int32_t)best;
        return;
    }
    // ---- K9 epilogue: flush the partial direction dword, pick the end cell, walk back.
    // Round 5: the walk no longer chases its direction words through HBM one dependent load per step (a block's launch time WAS that chain: ~1500 steps x
    // ~0.8 us against 0.2 ms for the DP; merge and chimera, a few hundred pairs each, took 1.7 ms per launch).  The 64 lanes fetch a WINDOW of direction
    // words ahead of the walk in one go -- 21 step groups x the three lane columns around the current diagonal -- into LDS, every lane then takes the same
    // steps from there (uniform control flow, lane 0 stores), and a new window is fetched when the walk leaves this one: one memory latency per ~80 moves.
    // The row is written whole, once: "not covered" by all lanes first, then the walk's cells, each composed in registers (no read-modify-write), with the
    // quality bins of the target staged in LDS.
    if (((total | 1) % SPD) != SPD - 1) tb_base[(u64)((total | 1) / SPD) * 64 + lane] = tb_acc;   // last, partial group (a full one was stored in the loop)
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { u64 o = __shfl_xor(best_key, s); best_key = o < best_key ? o : best_key; }
    u64* cells = tbo.cells + tbo.cell_off[blockIdx.x];
    for (int x = lane; x < n; x += 64) cells[x] = 7;
    const u32 val = (u32)(best_key >> 40);
    int ta = (int)((best_key >> 16) & 0xFFFFFF), d = (int)(best_key & 0xFFFF);
    int i = (ta - (d - wp)) / 2, j = i + d - wp;
    u32* sp = tbo.span + (u64)blockIdx.x * 4;
    if (lane == 0) nm_out[pid] = val >= INF16 ? 0x7FFFFFFF : (int32_t)val;
    if (val >= INF16) { if (lane == 0) sp[0] = sp[1] = sp[2] = sp[3] = 0; return; }
    if (lane == 0) { sp[1] = (u32)i; sp[3] = (u32)j; }
    const bool rv = rev && rev[pid];
    const u8* qb = tbo.qualbins ? tbo.qualbins + tbo.qb_off[tr] : nullptr;
    __shared__ u32 win[64];
    __shared__ u8 qlds[2048];                                      // 4-bit quality bins of the target, two per byte, one per 4 bases: 16000 bases = 2000 bytes
    const u8* tgq = tbo.tag_hp ? tbo.tag_qual + T.off[tr] : nullptr; const u8* tgh = tbo.tag_hp ? tbo.tag_hp + T.off[tr] : nullptr;
    if (qb && !tgq) for (int x = lane; x < (m + 7) / 8 && x < 2048; x += 64) qlds[x] = qb[x];
    auto tbase = [&](int x) -> u64 { return (tw[x >> 4] >> (30 - 2 * (x & 15))) & 3u; };
    auto tqual = [&](int x) -> u64 { int src = rv ? m - 1 - x : x; if (tgq) return tgq[src]; if (!qb) return 33; u32 bin = (qlds[(src >> 2) >> 1] >> (4 * ((src >> 2) & 1))) & 15u; return bin * 3 + 33; };
    auto thp = [&](int x) -> u64 { return tgh ? (u64)tgh[rv ? m - 1 - x : x] << 56 : 0ull; };
    int ins_run = 0;
    auto ins_bits = [&](int first_j) -> u64 {                      // the insertion that follows the cell about to be written: bases t[first_j ..), the first two kept
        u64 cc = 0;
        if (ins_run > 0) {
            const int keep = ins_run < 2 ? ins_run : 2;
            cc |= (u64)keep << 16; cc |= (u64)(ins_run < 255 ? ins_run : 255) << 18;
            for (int x = 0; x < keep; x++) { cc |= tbase(first_j + x) << (32 + 2 * x); cc |= tqual(first_j + x) << (40 + 8 * x); }
        }
        ins_run = 0;
        return cc;
    };
    constexpr int NG = 21;                                         // step groups per window: 21 x 3 lane columns = 63 words
    while (i > 0 && j > 0) {                                       // i, j, ins_run are the same in every lane
        const int gq = (i + j) / SPD, ln0 = (j - i + wp) / P;
        __syncthreads();                                           // the walk of the window before is done with `win` (and, the first time: qlds is staged, the direction words are stored)
        {
            const int g = gq - lane / 3, l = ln0 - 1 + lane % 3;
            win[lane] = (lane < 3 * NG && g >= 0 && l >= 0 && l < 64) ? tb_base[(u64)g * 64 + l] : 0u;
        }
        __syncthreads();
        while (i > 0 && j > 0) {
            d = j - i + wp; ta = i + j;
            const int ln = d / P, x = (d - ln * P) >> 1;               // lane that owns diagonal d, cell index inside its step
            const int dg = gq - ta / SPD, dl = ln - ln0 + 1;
            if (dg >= NG || (u32)dl > 2u) break;                       // outside the window: fetch the next one
            const u32 dr = (win[dg * 3 + dl] >> (4 * R * (ta % SPD) + 2 * x)) & 3u;
            if (dr == 2) { ins_run++; j--; continue; }
            u64 cc = ins_bits(j);
            if (dr == 0) { cc |= tbase(j - 1) | (tqual(j - 1) << 8) | thp(j - 1); j--; }
            else cc |= 4;
            i--;
            if (lane == 0) cells[i] = cc;
        }
    }
    if (lane == 0) {
        if (i >= 1) cells[i - 1] = 7 | ins_bits(j);
        sp[0] = (u32)i; sp[2] = (u32)j;
    }
}

int launch_align(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                 const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells) {
    if (n_sel == 0) return SVT_OK;
    u32 ldsq = (max_qlen + 15) / 16 + 2, ldst = (max_tlen + 15) / 16 + 2;
    size_t sh = (size_t)(ldsq + ldst) * 4;
    ProfScope ps(c, rclass == 1 ? "k_align_r1" : (rclass == 2 ? "k_align_r2" : "k_align_r4"), algo_bytes, cells);
    BatchView qv = Q->view(), tv = T->view();
    if (rclass == 1) hipLaunchKernelGGL((k_align<1, false>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, ldsq, ldst, TbOut{});
    else if (rclass == 2) hipLaunchKernelGGL((k_align<2, false>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, ldsq, ldst, TbOut{});
    else hipLaunchKernelGGL((k_align<4, false>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, ldsq, ldst, TbOut{});
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// K9 launcher: pairs must be pre-grouped by band class; pair p of THIS launch is sel[p]; slabs/rows are indexed by the launch-local p
u64 align_tb_dwords(int rclass, u32 max_qlen, u32 max_tlen) { const u64 spd = 32 / (4 * rclass); return ((u64)(max_qlen + max_tlen + 2) / spd + 2) * 64; }
int launch_align_tb(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                    const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, u32 max_qlen, u32 max_tlen, u32* d_tb, u64* d_cells, const u64* d_cell_off, u32* d_span) {
    if (n_sel == 0) return SVT_OK;
    u32 ldsq = (max_qlen + 15) / 16 + 2, ldst = (max_tlen + 15) / 16 + 2;
    size_t sh = (size_t)(ldsq + ldst) * 4;
    TbOut tbo; tbo.tb = d_tb; tbo.tb_stride = align_tb_dwords(rclass, max_qlen, max_tlen); tbo.cells = d_cells; tbo.cell_off = d_cell_off; tbo.span = d_span;
    tbo.qualbins = T->seeds.valid ? T->seeds.qualbins : nullptr; tbo.qb_off = T->seeds.valid ? T->seeds.qb_off : nullptr;
    tbo.tag_qual = T->d_tag_qual; tbo.tag_hp = T->d_tag_hp;
    ProfScope ps(c, rclass == 1 ? "k_align_tbw_r1" : (rclass == 2 ? "k_align_tbw_r2" : "k_align_tbw_r4"), (double)n_sel * ((max_qlen + max_tlen) / 4.0 + 8.0 * max_qlen + 24.0), (double)n_sel);   // tbw: the wave-per-pair kernel (small launches: merge, chimera); tb: the lane-per-pair one
    BatchView qv = Q->view(), tv = T->view();
    if (rclass == 1) hipLaunchKernelGGL((k_align<1, true>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, ldsq, ldst, tbo);
    else if (rclass == 2) hipLaunchKernelGGL((k_align<2, true>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, ldsq, ldst, tbo);
    else hipLaunchKernelGGL((k_align<4, true>), dim3((u32)n_sel), dim3(64), sh, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, ldsq, ldst, tbo);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// =====================================================================================================================
// K8 bit-parallel: the SAME banded overlap edit distance, one pair per LANE, Myers/Hyyro bit-vectors.
// Column j of the DP holds the band rows [max(1, j-w), min(n, j+w)] as vertical deltas (+1 / -1 / 0) in NW 64-bit words whose
// bit 0 is the top band row: the window slides down one row per column (a 1-bit shift), blocks of 64 rows are chained top to
// bottom through their horizontal deltas.  The band edges need no infinities: an out-of-band neighbour is replaced by
// "previous cell + 1" (horizontal delta +1 above the window, vertical delta +1 for the row that enters at the bottom), which can
// never beat the diagonal candidate that is always inside the band -- so the values equal the INF-banded recurrence cell by cell.
// The value of the top band row is carried along; the last row / last column cells (free trailing overhang) are read off it
// with popcounts.  ~200 VALU ops per column for 255 band cells instead of ~2.5 per cell.
// =====================================================================================================================
namespace {
struct SeqReader {                                  // 2-bit bases of one sequence, forward or reverse-complemented, one word cached
    const u32* w; int len; bool rc; int cw; u32 cur;
    __device__ __forceinline__ void init(const u32* words, int n, bool reverse) { w = words; len = n; rc = reverse; cw = -1; cur = 0; }
    __device__ __forceinline__ u32 base(int x) {    // x-th base of the (possibly reverse-complemented) sequence
        const int p = rc ? len - 1 - x : x;
        const int wi = p >> 4;
        if (wi != cw) { cw = wi; cur = w[wi]; }
        const u32 b = (cur >> (30 - 2 * (p & 15))) & 3u;
        return rc ? 3u - b : b;
    }
};
// Sequential access, 16 bases per group, WAVE-UNIFORM refill points.  The per-base reader above reloads a word whenever a lane crosses
// a 16-base boundary; the lanes of a wave (64 different pairs) cross at different columns, so nearly every column had some lane
// loading and the whole wave waited for that load -- two L2 round trips per DP column, far more than the ~250 VALU operations of the
// column itself (rocprof: VALUBusy 36 %).  Here every lane fetches the two raw words of its NEXT group at the same loop trip
// (issue), 16 columns before they are combined into one register holding that group's 16 bases (combine): the latency is hidden
// behind a whole group of columns, and a column's base is a shift of a register.
struct Seq16 {
    const u32* w; int len; bool rc; int last_word; u32 A, B;
    __device__ __forceinline__ void init(const u32* words, int n, bool reverse) { w = words; len = n; rc = reverse; last_word = (n + 15) / 16; A = B = 0; }   // w[last_word + 1] is still a zero pad word
    __device__ __forceinline__ int first_pos(int x0) const { return rc ? len - 16 - x0 : x0; }      // lowest packed position of the group x0 .. x0+15
    __device__ __forceinline__ void issue(int x0) {                                                  // raw words of the group that starts at base x0
        const int pc = max(first_pos(x0), 0), a = min(pc >> 4, last_word);
        A = w[a]; B = w[a + 1];
    }
    __device__ __forceinline__ u32 combine(int x0) const {                                           // bases x0 .. x0+15, base x0 in the top bit pair; bases outside the sequence read 0 (forward)
        const int p0 = first_pos(x0), pc = max(p0, 0);
        const u32 o = (u32)(pc & 15) * 2;
        u32 f = (pc >> 4) > last_word ? 0u : (o ? __builtin_amdgcn_alignbit(A, B, 32 - o) : A);      // packed positions pc .. pc+15, pc in the top pair
        if (!rc) {
            const int valid = len - x0;                                                              // bases of the group that exist
            return valid >= 16 ? f : (valid <= 0 ? 0u : (f & (~0u << (32 - 2 * valid))));
        }
        f = __brev(f); f = ((f & 0x55555555u) << 1) | ((f >> 1) & 0x55555555u);                      // reverse the order of the 16 pairs: position pc+15 on top
        if (p0 < 0) f = p0 <= -16 ? 0u : (f << (2 * (-p0)));                                          // the group hangs over the sequence start: its first base is position len-1-x0 = 15 + p0
        return ~f;                                                                                   // complement (pairs beyond the sequence are never consumed)
    }
};
}  // namespace

// N 32-bit words (32-bit ALU ops and v_alignbit funnel shifts are full rate on CDNA; 64-bit shifts are not).  The registers hold the
// rows [top, top + 32N - 1]: the band rows [top, bot] plus "not yet entered" rows below whose state is pinned to Pv = 1 / Mv = 0 by
// the mask Bm (bits beyond the band bottom), so a row that enters the band at the bottom edge already carries the +1 convention
// and nothing has to be inserted at a lane-dependent bit position.  The query bases of all register rows sit in two bit planes.
template <int N>
__global__ void __launch_bounds__(64, N <= 8 ? 5 : (N <= 16 ? 3 : 1)) k_align_bp(BatchView Q, BatchView T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                                 const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel, u64 n_sel,
                                                 int32_t* __restrict__ nm_out) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_sel) return;
    const u64 pid = sel ? sel[g] : g;
    const u32 qr = qi[pid], tr = ti[pid];
    const int n = (int)(Q.off[qr + 1] - Q.off[qr]);
    const int m = (int)(T.off[tr + 1] - T.off[tr]);
    const int w = (int)band[pid];
    if (n == 0 || m == 0 || m <= w || n <= w) { nm_out[pid] = 0; return; }           // a zero boundary cell lies on the last row / column
    SeqReader qs, ts;
    qs.init(Q.packed + Q.woff[qr], n, false);
    ts.init(T.packed + T.woff[tr], m, rev && rev[pid]);
    u32 Pv[N], Mv[N], Lo[N], Hi[N], Bm[N];
    int bot = min(n, w);                             // column 0: band rows 1..min(n, w), all D(i,0) = 0
    #pragma unroll
    for (int k = 0; k < N; k++) {
        u32 lo = 0, hi = 0;
        #pragma unroll 1
        for (int b = 0; b < 32; b++) { const int r = 32 * k + b + 1; const u32 q = r <= n ? qs.base(r - 1) : 0u; lo |= (q & 1u) << b; hi |= (q >> 1) << b; }
        Lo[k] = lo; Hi[k] = hi; Mv[k] = 0;
        const int first_beyond = bot - 32 * k;       // bit index (inside this word) of the first row below the band
        Bm[k] = first_beyond <= 0 ? ~0u : (first_beyond >= 32 ? 0u : (~0u << first_beyond));
        Pv[k] = Bm[k];
    }
    int top_row = 1, top_val = 0, best = 0x7FFFFFFF;
    const int jend = min(m, n + w);
    // the target base of column j is base j-1; the query row that enters the registers when column j slides is row j - w + 32N - 1
    Seq16 tg, qg;
    tg.init(T.packed + T.woff[tr], m, rev && rev[pid]); qg.init(Q.packed + Q.woff[qr], n, false);
    const int xq0 = 32 * N - 2 - w;                  // query base index of column j: j + xq0
    tg.issue(0); qg.issue(1 + xq0);
    for (int jg = 1; jg <= jend; jg += 16) {
    const u32 T16 = tg.combine(jg - 1), Q16 = qg.combine(jg + xq0);
    tg.issue(jg + 15); qg.issue(jg + 16 + xq0);      // the next group's words: needed 16 columns from now
    const int je = min(jend, jg + 15);
    #pragma unroll 1
    for (int j = jg; j <= je; j++) {
        const u32 sh = 30u - 2u * (u32)(j - jg);
        const bool slide = max(1, j - w) > top_row, grow = j + w <= n;
        int vtmp = 0;
        if (slide) {                                 // the window slides down one row: everything moves one bit towards bit 0
            top_row++;
            const u32 q = (Q16 >> sh) & 3u;          // row top_row + 32N - 1 (0 beyond row n)
            #pragma unroll
            for (int k = 0; k < N; k++) {
                Pv[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Pv[(k + 1) % N] : 1u, Pv[k], 1);
                Mv[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Mv[(k + 1) % N] : 0u, Mv[k], 1);
                Lo[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Lo[(k + 1) % N] : (q & 1u), Lo[k], 1);
                Hi[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Hi[(k + 1) % N] : (q >> 1), Hi[k], 1);
            }
            vtmp = (int)(Pv[0] & 1) - (int)(Mv[0] & 1);           // D(top, j-1) - D(top-1, j-1)
            if (!grow) {                             // the band bottom stays at row n: one more register row is below the band
                #pragma unroll
                for (int k = 0; k < N; k++) Bm[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Bm[(k + 1) % N] : 1u, Bm[k], 1);
            }
        } else if (grow) {                           // the band grows at the bottom while its top rests on row 1
            #pragma unroll
            for (int k = N - 1; k >= 0; k--) Bm[k] = __builtin_amdgcn_alignbit(Bm[k], k > 0 ? Bm[(k + N - 1) % N] : 0u, 31);
        }
        if (grow) bot = j + w;
        const u32 c = (T16 >> sh) & 3u;
        const u32 clo = (c & 1) ? ~0u : 0u, chi = (c >> 1) ? ~0u : 0u;
        u32 hp = top_row == 1 ? 0u : 1u, hn = 0u;   // row 0 is all zeros; an out-of-band row above counts as +1
        int h0 = 0;
        #pragma unroll
        for (int k = 0; k < N; k++) {
            const u32 Eq = ~(Lo[k] ^ clo) & ~(Hi[k] ^ chi);
            const u32 pv = Pv[k], mv = Mv[k];
            const u32 Xv = Eq | mv;
            const u32 Eh = Eq | hn;                  // a -1 coming down from the block above acts like a match in bit 0
            const u32 Xh = (((Eh & pv) + pv) ^ pv) | Eh;
            const u32 Ph = mv | ~(Xh | pv);
            const u32 Mh = pv & Xh;
            if (k == 0) h0 = (int)(Ph & 1) - (int)(Mh & 1);
            const u32 Phs = (Ph << 1) | hp, Mhs = (Mh << 1) | hn;
            hp = Ph >> 31; hn = Mh >> 31;
            Pv[k] = (Mhs | ~(Xv | Phs)) | Bm[k];    // rows below the band keep +1 / 0
            Mv[k] = (Phs & Xv) & ~Bm[k];
        }
        top_val = (top_row == 1) ? (int)(Pv[0] & 1) - (int)(Mv[0] & 1) : top_val + vtmp + h0;
        if (bot == n) {                              // cell (n, j): top value + vertical deltas of the band rows below the top
            int v = top_val;
            #pragma unroll
            for (int k = 0; k < N; k++) { const u32 mask = ~Bm[k] & (k == 0 ? ~1u : ~0u); v += __popc(Pv[k] & mask) - __popc(Mv[k] & mask); }
            best = min(best, v);
        }
        if (j == m) {                                // last column: every band row
            int v = top_val; best = min(best, v);
            const int nb = bot - top_row;
            #pragma unroll
            for (int k = 0; k < N; k++) {
                const u32 pv = Pv[k], mv = Mv[k];
                #pragma unroll 1
                for (int b = (k == 0 ? 1 : 0); b < 32; b++) {
                    if (32 * k + b > nb) break;
                    v += (int)((pv >> b) & 1) - (int)((mv >> b) & 1);
                    best = min(best, v);
                }
            }
        }
    }
    }
    nm_out[pid] = best;
}

int launch_align_bp(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                    const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, double algo_bytes, double cells) {
    if (n_sel == 0) return SVT_OK;
    ProfScope ps(c, rclass == 1 ? "k_align_r1" : (rclass == 2 ? "k_align_r2" : "k_align_r4"), algo_bytes, cells);
    BatchView qv = Q->view(), tv = T->view();
    const dim3 grid((u32)((n_sel + 63) / 64));
    if (rclass == 1) hipLaunchKernelGGL((k_align_bp<8>), grid, dim3(64), 0, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm);
    else if (rclass == 2) hipLaunchKernelGGL((k_align_bp<16>), grid, dim3(64), 0, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm);
    else hipLaunchKernelGGL((k_align_bp<32>), grid, dim3(64), 0, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// =====================================================================================================================
// K9 bit-parallel: the forward pass of k_align_bp also derives, for every band cell, the move the oracle's traceback would take
// there -- diagonal, then deletion, then insertion, strict improvements only -- as two bit vectors per column:
//   Dg = Eq | ~D0      the diagonal is optimal: a match, or a mismatch whose value is the diagonal neighbour's + 1
//                      (D0 = Xh | Mv is Hyyro's "diagonal delta is zero" vector);
//   Up = ~Dg & Pv'     otherwise the deletion is optimal iff the new vertical delta is +1; otherwise the insertion.
// Out-of-band neighbours never win: the top band row sees a +1 horizontal carry (its Pv' bit is 0) and the bottom row's left
// neighbour carries the pinned +1 vertical delta.  The 2N dwords of a column are stored [column][word][lane], so a wave's store is
// one coalesced 256-byte row, and then EVERY LANE walks its own alignment back -- 64 tracebacks per wave instead of one.  The walk
// prefetches the direction words of the next PF columns along its diagonal in one batch (one memory latency per PF steps, not per
// step).  The end cell follows the oracle's order: smallest value over the last row / column, then smallest i + j, then j - i.
// =====================================================================================================================
// WIN = false: every direction word of every column is stored (2N dwords per pair-column: 96 KB per 1.5 kb pair, 8x what the row it yields weighs).
// WIN = true, the default: a walk only reads the bits around ITS diagonal, and a diagonal keeps its bit position while the band window slides,
// so each column keeps one 64-bit window of Dg and of Up (one dwordx4 per pair-column, 24 KB per 1.5 kb pair, coalesced over the 64 lanes)
// around the diagonal the walk is expected on: first launch, the line from (0, 0) to (n, m) -- right for every end-to-end pair whose indels
// stay within +-32 bases of it; a walk that leaves its window (or starts outside: overhangs) puts the pair on `redo`, and the caller runs the
// list again with the window centred on the now known end diagonal (use_keys; the forward pass leaves every end cell in `keys`), and what
// still drifts, once more with the full slab.  The rows are the same whichever launch writes them: the window only decides what is kept.
// 3 waves per SIMD: the three captured words of Dg and Up cost ~10 registers, and at 4 waves (128 VGPRs) 16 of them spill inside the column loop
#ifndef SVT_K9_WIN_OCC
#define SVT_K9_WIN_OCC 3
#endif
// MODE 0 full slab | 1 windowed slab | 2 no walk at all: only the end cell's key (value << 40 | (i + j) << 20 | (j - i + 2048)) into `keys`
// WB (MODE 1): bits of the direction window kept per column: 64 (round 3: +-32 diagonals around the expected one, 16 bytes per pair-column) or 32 (round 4: +-16, 8 bytes:
// 12 KB instead of 24 KB per 1.5 kb pair; more walks leave the window and run again)
template <int N, int MODE, int WB = 64>
__global__ void __launch_bounds__(64, N <= 8 ? (MODE == 1 ? SVT_K9_WIN_OCC : 4) : (N <= 16 ? 2 : 1)) k_align_bp_tb(BatchView Q, BatchView T, const u32* __restrict__ qi, const u32* __restrict__ ti,
                                                                    const u8* __restrict__ rev, const u32* __restrict__ band, const u32* __restrict__ sel, u64 n_sel,
                                                                    int32_t* __restrict__ nm_out, u32 max_cols, TbOut tbo, u64* __restrict__ keys, u32* __restrict__ redo, const u32* __restrict__ remap, int use_keys) {
    constexpr bool WIN = MODE == 1, KEYS = MODE == 2;
    constexpr int CW = 2 * N;                        // dwords per stored column (full slab)
    constexpr int PF = 8;                            // columns of direction words fetched per batch
    const u64 g0 = (u64)blockIdx.x * 64 + threadIdx.x;
    if (g0 >= n_sel) return;
    const u64 g = remap ? remap[g0] : g0;            // the redo launch: position of the pair in the chunk (sel, cell_off, span are per position)
    const u32 lane = threadIdx.x;
    u32* slab = tbo.tb + (u64)blockIdx.x * (u64)(max_cols + 1) * CW * 64 + lane;      // column j, word x at slab[(j * CW + x) * 64]
    const u64 pid = sel ? sel[g] : g;
    const u32 qr = qi[pid], tr = ti[pid];
    const int n = (int)(Q.off[qr + 1] - Q.off[qr]);
    const int m = (int)(T.off[tr + 1] - T.off[tr]);
    const int w = (int)band[pid];
    const bool rv = rev && rev[pid];
    if (KEYS && (n == 0 || m == 0)) { keys[g] = 2048; return; }                      // value 0 at (0, 0)
    u64* cells = KEYS ? nullptr : tbo.cells + tbo.cell_off[g];
    u32* sp = KEYS ? nullptr : tbo.span + g * 4;
    if (n == 0 || m == 0) { for (int x = 0; x < n; x++) cells[x] = 7; nm_out[pid] = 0; sp[0] = sp[1] = sp[2] = sp[3] = 0; return; }
    // WIN: expected diagonal j - i at column j = d0 + slope * j (16.16 fixed point), and the first bit of the 64-bit window kept for that column
    int d0 = 0, slope = 0;
    if (WIN) {
        if (use_keys) { const u64 kq = keys[g]; d0 = kq == ~0ull ? 0 : (int)(kq & 0xFFFFF) - 2048; }
        else slope = (min(max(m - n, -w), w) * 65536) / m;
    }
    auto win_start = [&](int jj) -> int { return min(max((jj - (d0 + ((slope * jj) >> 16))) - max(1, jj - w) - WB / 2, 0), 32 * N - (WB + 32)); };
    uint4* wslab = (uint4*)tbo.tb + (u64)blockIdx.x * (u64)(max_cols + 1) * 64 + lane;     // WB = 64: column j at wslab[j * 64]: {Dg lo, Dg hi, Up lo, Up hi}
    uint2* wslab2 = (uint2*)tbo.tb + (u64)blockIdx.x * (u64)(max_cols + 1) * 64 + lane;    // WB = 32: {Dg, Up}
    SeqReader qs, ts;
    qs.init(Q.packed + Q.woff[qr], n, false);
    ts.init(T.packed + T.woff[tr], m, rv);
    const u64 NOKEY = ~0ull;
    auto key_of = [&](int v, int i, int j) -> u64 { return ((u64)(u32)v << 40) | ((u64)(u32)(i + j) << 20) | (u64)(u32)(j - i + 2048); };
    u64 best = NOKEY;
    if (n <= w) best = min(best, key_of(0, n, 0));   // boundary zeros on the last row / column
    if (m <= w) best = min(best, key_of(0, 0, m));
    u32 Pv[N], Mv[N], Lo[N], Hi[N], Bm[N];
    int bot = min(n, w);
    #pragma unroll
    for (int k = 0; k < N; k++) {
        u32 lo = 0, hi = 0;
        #pragma unroll 1
        for (int b = 0; b < 32; b++) { const int r = 32 * k + b + 1; const u32 q = r <= n ? qs.base(r - 1) : 0u; lo |= (q & 1u) << b; hi |= (q >> 1) << b; }
        Lo[k] = lo; Hi[k] = hi; Mv[k] = 0;
        const int first_beyond = bot - 32 * k;
        Bm[k] = first_beyond <= 0 ? ~0u : (first_beyond >= 32 ? 0u : (~0u << first_beyond));
        Pv[k] = Bm[k];
    }
    int top_row = 1, top_val = 0;
    const int jend = min(m, n + w);
    Seq16 tg, qg;                                    // wave-uniform 16-base groups (see Seq16): no global load inside a column
    tg.init(T.packed + T.woff[tr], m, rv); qg.init(Q.packed + Q.woff[qr], n, false);
    const int xq0 = 32 * N - 2 - w;
    tg.issue(0); qg.issue(1 + xq0);
    for (int jg = 1; jg <= jend; jg += 16) {
    const u32 T16 = tg.combine(jg - 1), Q16 = qg.combine(jg + xq0);
    tg.issue(jg + 15); qg.issue(jg + 16 + xq0);
    const int je = min(jend, jg + 15);
    #pragma unroll 1
    for (int j = jg; j <= je; j++) {
        const u32 sh = 30u - 2u * (u32)(j - jg);
        const bool slide = max(1, j - w) > top_row, grow = j + w <= n;
        int vtmp = 0;
        if (slide) {
            top_row++;
            const u32 q = (Q16 >> sh) & 3u;
            #pragma unroll
            for (int k = 0; k < N; k++) {
                Pv[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Pv[(k + 1) % N] : 1u, Pv[k], 1);
                Mv[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Mv[(k + 1) % N] : 0u, Mv[k], 1);
                Lo[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Lo[(k + 1) % N] : (q & 1u), Lo[k], 1);
                Hi[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Hi[(k + 1) % N] : (q >> 1), Hi[k], 1);
            }
            vtmp = (int)(Pv[0] & 1) - (int)(Mv[0] & 1);
            if (!grow) {
                #pragma unroll
                for (int k = 0; k < N; k++) Bm[k] = __builtin_amdgcn_alignbit(k + 1 < N ? Bm[(k + 1) % N] : 1u, Bm[k], 1);
            }
        } else if (grow) {
            #pragma unroll
            for (int k = N - 1; k >= 0; k--) Bm[k] = __builtin_amdgcn_alignbit(Bm[k], k > 0 ? Bm[(k + N - 1) % N] : 0u, 31);
        }
        if (grow) bot = j + w;
        const u32 c = (T16 >> sh) & 3u;
        const u32 clo = (c & 1) ? ~0u : 0u, chi = (c >> 1) ? ~0u : 0u;
        u32 hp = top_row == 1 ? 0u : 1u, hn = 0u;
        int h0 = 0;
        u32* col = slab + (u64)j * CW * 64;
        int kwin = 0, wsh = 0; u32 a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;   // WIN: words kwin .. kwin + 2 of Dg (a) and Up (b), shifted in as they are made
        if (WIN) { const int ws = win_start(j); kwin = ws >> 5; wsh = ws & 31; }
        #pragma unroll
        for (int k = 0; k < N; k++) {
            const u32 Eq = ~(Lo[k] ^ clo) & ~(Hi[k] ^ chi);
            const u32 pv = Pv[k], mv = Mv[k];
            const u32 Xv = Eq | mv;
            const u32 Eh = Eq | hn;
            const u32 Xh = (((Eh & pv) + pv) ^ pv) | Eh;
            const u32 Ph = mv | ~(Xh | pv);
            const u32 Mh = pv & Xh;
            if (k == 0) h0 = (int)(Ph & 1) - (int)(Mh & 1);
            const u32 Phs = (Ph << 1) | hp, Mhs = (Mh << 1) | hn;
            hp = Ph >> 31; hn = Mh >> 31;
            Pv[k] = (Mhs | ~(Xv | Phs)) | Bm[k];
            Mv[k] = (Phs & Xv) & ~Bm[k];
            const u32 Dg = Eq | ~(Xh | mv);
            if (MODE == 0) { col[(u64)k * 64] = Dg; col[(u64)(N + k) * 64] = ~Dg & Pv[k]; }
            if (WIN) { const u32 Up = ~Dg & Pv[k]; const bool cap = k <= kwin + WB / 32; a0 = cap ? a1 : a0; a1 = cap ? a2 : a1; a2 = cap ? Dg : a2; b0 = cap ? b1 : b0; b1 = cap ? b2 : b1; b2 = cap ? Up : b2; }
        }
        if (WIN && WB == 64) wslab[(u64)j * 64] = make_uint4(__builtin_amdgcn_alignbit(a1, a0, wsh), __builtin_amdgcn_alignbit(a2, a1, wsh), __builtin_amdgcn_alignbit(b1, b0, wsh), __builtin_amdgcn_alignbit(b2, b1, wsh));
        if (WIN && WB == 32) wslab2[(u64)j * 64] = make_uint2(__builtin_amdgcn_alignbit(a2, a1, wsh), __builtin_amdgcn_alignbit(b2, b1, wsh));
        top_val = (top_row == 1) ? (int)(Pv[0] & 1) - (int)(Mv[0] & 1) : top_val + vtmp + h0;
        if (bot == n) {
            int v = top_val;
            #pragma unroll
            for (int k = 0; k < N; k++) { const u32 mask = ~Bm[k] & (k == 0 ? ~1u : ~0u); v += __popc(Pv[k] & mask) - __popc(Mv[k] & mask); }
            best = min(best, key_of(v, n, j));
        }
        if (j == m) {
            int v = top_val; best = min(best, key_of(v, top_row, m));
            const int nb = bot - top_row;
            #pragma unroll
            for (int k = 0; k < N; k++) {
                const u32 pv = Pv[k], mv = Mv[k];
                #pragma unroll 1
                for (int b = (k == 0 ? 1 : 0); b < 32; b++) {
                    if (32 * k + b > nb) break;
                    v += (int)((pv >> b) & 1) - (int)((mv >> b) & 1);
                    best = min(best, key_of(v, top_row + 32 * k + b, m));
                }
            }
        }
    }
    }
    if (WIN || KEYS) keys[g] = best;
    if (KEYS) return;
    if (best == NOKEY) { for (int x = 0; x < n; x++) cells[x] = 7; nm_out[pid] = 0x7FFFFFFF; sp[0] = sp[1] = sp[2] = sp[3] = 0; return; }
    int i, j;
    { const int a = (int)((best >> 20) & 0xFFFFF), d = (int)(best & 0xFFFFF) - 2048; i = (a - d) / 2; j = i + d; }
    nm_out[pid] = (int)(best >> 40);
    const int i_end = i;
    sp[1] = (u32)i; sp[3] = (u32)j;
    // target qualities: one cached dword of 4-bit bins (8 bins = 32 bases)
    const u32* qw = tbo.qualbins ? (const u32*)tbo.qualbins : nullptr;
    const u64 qb0 = qw ? tbo.qb_off[tr] : 0;
    u64 q_cached = ~0ull; u32 q_word = 0;
    auto tbase = [&](int x) -> u64 { return (u64)ts.base(x); };
    const u8* tgq = tbo.tag_hp ? tbo.tag_qual + T.off[tr] : nullptr; const u8* tgh = tbo.tag_hp ? tbo.tag_hp + T.off[tr] : nullptr;
    auto thp = [&](int x) -> u64 { return tgh ? (u64)tgh[rv ? m - 1 - x : x] << 56 : 0ull; };
    auto tqual = [&](int x) -> u64 {
        const int src = rv ? m - 1 - x : x;
        if (tgq) return tgq[src];
        if (!qw) return 33;
        const u64 byte = qb0 + (u64)(src >> 3);
        if ((byte >> 2) != q_cached) { q_cached = byte >> 2; q_word = qw[q_cached]; }
        const u32 bin = (q_word >> (8 * (u32)(byte & 3) + 4 * ((src >> 2) & 1))) & 15u;
        return bin * 3 + 33;
    };
    int ins_run = 0;
    auto ins_bits = [&](int first_j) -> u64 {
        u64 cc = 0;
        if (ins_run > 0) {
            const int keep = ins_run < 2 ? ins_run : 2;
            cc |= (u64)keep << 16; cc |= (u64)(ins_run < 255 ? ins_run : 255) << 18;
            for (int x = 0; x < keep; x++) { cc |= tbase(first_j + x) << (32 + 2 * x); cc |= tqual(first_j + x) << (40 + 8 * x); }
        }
        ins_run = 0;
        return cc;
    };
    // Row cells are queued per lane in LDS and written in runs of 8 (64 bytes, four 16-byte stores) when the walk crosses a multiple of 8: one
    // 8-byte store per step touched 64 different lines per wave instruction, and the lines went to HBM in pieces (profiles/r03_pmc_hbm.md)
    __shared__ u64 stg[8 * 64];
    int flushed_to = i_end;                           // rows [flushed_to, i_end) are in HBM; rows [i, flushed_to) of the open group are queued
    auto emit = [&](int idx, u64 cc) {
        stg[(idx & 7) * 64 + lane] = cc;
        if ((idx & 7) == 0) {
            const int hi = min(idx + 8, i_end);
            if (hi - idx == 8) {
                u64 v[8];
                #pragma unroll
                for (int x = 0; x < 8; x++) v[x] = stg[x * 64 + lane];
                ulonglong2* dst = (ulonglong2*)(cells + idx);       // 8-byte aligned is all a row start guarantees: 16-byte stores need no more on gfx950
                #pragma unroll
                for (int x = 0; x < 4; x++) dst[x] = make_ulonglong2(v[2 * x], v[2 * x + 1]);
            } else for (int x = idx; x < hi; x++) cells[x] = stg[(x & 7) * 64 + lane];
            flushed_to = idx;
        }
    };
    if constexpr (WIN) {
        bool drifted = false;
        while (i > 0 && j > 0 && !drifted) {
            uint4 x[PF]; int ws[PF];
            #pragma unroll
            for (int c = 0; c < PF; c++) {           // the windows of columns j, j-1, ...
                const int jc = j - c;
                x[c] = make_uint4(0, 0, 0, 0); ws[c] = 0;
                if (jc >= 1) { if (WB == 64) x[c] = wslab[(u64)jc * 64]; else { const uint2 y2 = wslab2[(u64)jc * 64]; x[c] = make_uint4(y2.x, 0, y2.y, 0); } ws[c] = win_start(jc); }
            }
            #pragma unroll
            for (int c = 0; c < PF; c++) {
                if (drifted || i <= 0 || j <= 0) continue;                              // here j == (batch's first column) - c
                for (;;) {
                    const int rel = i - max(1, j - w) - ws[c];
                    if ((u32)rel >= (u32)WB) { drifted = true; break; }                     // the cell's bit is not in the window kept for column j
                    const u32 bit = 1u << (rel & 31), dgw = rel < 32 ? x[c].x : x[c].y, upw = rel < 32 ? x[c].z : x[c].w;
                    if (dgw & bit) { const u64 cc = ins_bits(j) | tbase(j - 1) | (tqual(j - 1) << 8) | thp(j - 1); emit(--i, cc); j--; break; }
                    if (upw & bit) { const u64 cc = ins_bits(j) | 4; emit(--i, cc); if (i == 0) break; continue; }
                    ins_run++; j--; break;
                }
            }
        }
        if (drifted) { redo[1 + atomicAdd(redo, 1u)] = (u32)g; return; }               // redo[0] counts; the next launch rewrites every output of the pair
    } else {
    while (i > 0 && j > 0) {
        u32 dg[PF], up[PF]; int kw[PF];
        #pragma unroll
        for (int c = 0; c < PF; c++) {               // the direction words of columns j, j-1, ... at the rows of the current diagonal
            const int jc = j - c;
            kw[c] = -1; dg[c] = up[c] = 0;
            if (jc >= 1) {
                const int p = max(0, (i - c) - max(1, jc - w));
                kw[c] = p >> 5;
                const u32* col = slab + (u64)jc * CW * 64;
                dg[c] = col[(u64)kw[c] * 64]; up[c] = col[(u64)(N + kw[c]) * 64];
            }
        }
        bool reload = false;
        #pragma unroll
        for (int c = 0; c < PF; c++) {
            if (reload || i <= 0 || j <= 0) continue;                                  // here j == (batch's first column) - c
            for (;;) {
                const int p = i - max(1, j - w);
                if ((p >> 5) != kw[c]) { reload = true; break; }
                const u32 bit = 1u << (p & 31);
                if (dg[c] & bit) { const u64 cc = ins_bits(j) | tbase(j - 1) | (tqual(j - 1) << 8) | thp(j - 1); emit(--i, cc); j--; break; }
                if (up[c] & bit) { const u64 cc = ins_bits(j) | 4; emit(--i, cc); if (i == 0) break; continue; }
                ins_run++; j--; break;
            }
        }
    }
    }
    for (int x = i; x < min(flushed_to, i_end); x++) cells[x] = stg[(x & 7) * 64 + lane];   // the open group
    sp[0] = (u32)i; sp[2] = (u32)j;
    for (int x = 0; x + 1 < i; x++) cells[x] = 7;
    if (i >= 1) cells[i - 1] = 7 | ins_bits(j);
    for (int x = i_end; x < n; x++) cells[x] = 7;
}

u64 align_tb_dwords_bp(int rclass, u32 max_tlen, bool full, int win_bits) { const int N = rclass == 1 ? 8 : (rclass == 2 ? 16 : 24); return (u64)(max_tlen + 1) * (full ? 2 * N : (win_bits == 32 ? 2 : 4)); }   // per pair (slabs are per 64 pairs)
// mode 0: full slab; mode 1: windowed slab around the line (0,0)-(n,m); mode 2: windowed around the end diagonals left in d_keys by mode 1;
// mode 3: forward pass only, end-cell keys into d_keys (d_tb, d_cells, d_cell_off, d_span unused).
// d_remap (optional): the launch covers the pairs at these positions of the chunk (a redo list); d_redo[0] counts, d_redo[1..] lists drifted walks
int launch_align_tb_bp(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                       const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, u32 max_tlen, u32* d_tb, u64* d_cells, const u64* d_cell_off, u32* d_span,
                       int mode, u64* d_keys, u32* d_redo, const u32* d_remap, double band_cells, int win_bits) {
    if (n_sel == 0) return SVT_OK;
    TbOut tbo; tbo.tb = d_tb; tbo.tb_stride = 0; tbo.cells = d_cells; tbo.cell_off = d_cell_off; tbo.span = d_span;
    tbo.qualbins = T->seeds.valid ? T->seeds.qualbins : nullptr; tbo.qb_off = T->seeds.valid ? T->seeds.qb_off : nullptr;
    tbo.tag_qual = T->d_tag_qual; tbo.tag_hp = T->d_tag_hp;
    static const char* names[3][4] = {{"k_align_tb_r1_full", "k_align_tb_r1", "k_align_tb_r1_again", "k_align_end_r1"}, {"k_align_tb_r2_full", "k_align_tb_r2", "k_align_tb_r2_again", "k_align_end_r2"},
                                      {"k_align_tb_r3_full", "k_align_tb_r3", "k_align_tb_r3_again", "k_align_end_r3"}};   // r1 / r2 / r3: 8 / 16 / 24 words of band rows per column (bands <= 127 / 255 / 383)
    // algorithmic bytes per pair: both packed sequences + the 8-byte row per query base + descriptors
    ProfScope ps(c, names[rclass - 1][d_remap || mode ? mode : 1], (double)n_sel * ((Q->max_len + max_tlen) / 4.0 + (mode == 3 ? 0.0 : 8.0 * Q->max_len) + 24.0), band_cells > 0 ? band_cells : (double)n_sel);   // units: band cells when the caller counted them, else pairs
    BatchView qv = Q->view(), tv = T->view();
    const dim3 grid((u32)((n_sel + 63) / 64));
#define SVT_K9(NN, MM) hipLaunchKernelGGL((k_align_bp_tb<NN, MM>), grid, dim3(64), 0, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, max_tlen, tbo, d_keys, d_redo, d_remap, mode == 2 ? 1 : 0)
#define SVT_K9W(NN) hipLaunchKernelGGL((k_align_bp_tb<NN, 1, 32>), grid, dim3(64), 0, c->stream, qv, tv, d_q, d_t, d_rev, d_band, d_sel, n_sel, d_nm, max_tlen, tbo, d_keys, d_redo, d_remap, mode == 2 ? 1 : 0)
    if (rclass == 1) { if (mode == 0) SVT_K9(8, 0); else if (mode == 3) SVT_K9(8, 2); else if (win_bits == 32) SVT_K9W(8); else SVT_K9(8, 1); }
    else if (rclass == 2) { if (mode == 0) SVT_K9(16, 0); else if (mode == 3) SVT_K9(16, 2); else if (win_bits == 32) SVT_K9W(16); else SVT_K9(16, 1); }
    else { if (mode == 0) SVT_K9(24, 0); else if (mode == 3) SVT_K9(24, 2); else if (win_bits == 32) SVT_K9W(24); else SVT_K9(24, 1); }   // round 5: bands 256-383 (4.3 kb rRNA-operon pairs: w = 331), 768 band rows per column
#undef SVT_K9W
#undef SVT_K9
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
