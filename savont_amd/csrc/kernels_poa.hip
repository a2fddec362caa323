// kernels_poa.hip -- K11: banded sequence-to-graph alignment with traceback, the inner loop of the Stage-4 POA
// (src/alignment.rs:193-231: spoars engine.align(sequence, graph) with Scoring(3,-8,-6,-6) linear gaps, overlap mode, band).
//
// The host (savont_amd/csrc/host/poa.hpp) owns the partial-order graph and its bookkeeping (add_alignment, topological sort):
// pointer-chasing, tens of microseconds per read.  What costs the CPU seconds per step is the DP: rows (graph nodes in
// topological order) x band (2*(base + 0.1*L) + 1 columns) cells per read, 75 reads per cluster, ~100 clusters.  This kernel
// aligns ONE sequence to ONE graph per workgroup (a single wave64; ~100 graphs run side by side):
//   * a lane owns C consecutive columns of the row (C = ceil(width/64) <= 8): candidates from every predecessor row
//     (match/mismatch from (p, j-1), deletion from (p, j)) are register work; the insertion chain row[j] = max(., row[j-1]+G)
//     is a prefix maximum of (value - j*G): lane-local scan + one DPP wave scan;
//   * the last 64 rows live in an LDS ring (predecessors are almost always a few rows back); older rows are read back from
//     their HBM spill copy;
//   * every cell stores a 16-bit back-pointer (move, row distance to the predecessor used); the traceback is one lane walking
//     them back from the best end cell (free trailing overhangs: sink nodes, or column L).
// Results are bit-identical to PoaGraph::align_impl<int16_t> (same candidate order, same tie-breaks, same floor at NEG).
// Bound: not HBM, not MFMA -- a latency chain of ~1 row per few hundred cycles per wave (LDS + DPP); HBM traffic is the
// back-pointer / spill rows, 4 B per cell.
#include "svt_internal.hpp"

struct PoaRowDev { u32 lohi, info, pred01, pred_start; };   // lo | hi<<16 ; code | sink<<8 | npred<<16 ; p0 | p1<<16 ; offset of the full list
struct PoaJobDev { u64 row_base, pred_base, seq_base, cell_base, path_base; u32 n_rows, seq_len, stride, pad; };

namespace {

// inclusive prefix maximum over the 64 lanes (gfx9 DPP: row shifts, then row broadcasts)
__device__ __forceinline__ int wave_prefix_max(int v, const int ident) {
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xF, 0xF, false));   // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2 and 3
    return v;
}

template <int C>
__global__ __launch_bounds__(64) void k_poa_align(const PoaJobDev* __restrict__ jobs, const PoaRowDev* __restrict__ rows, const u16* __restrict__ preds,
                                                  const u8* __restrict__ seqs, int16_t* __restrict__ Hs, u16* __restrict__ Ds,
                                                  int32_t* __restrict__ path_row, int32_t* __restrict__ path_pos, u32* __restrict__ path_len, int32_t* __restrict__ score,
                                                  const int SM, const int SX, const int SG, const int NEG) {
    constexpr int RS = 64 * C + 2;                 // LDS row stride (int16 elements)
    constexpr int IDENT = -(1 << 29);
    extern __shared__ int16_t lds[];
    int16_t* ring = lds;                            // [64][RS]
    u16* ring_lo = (u16*)(ring + 64 * RS);          // [64]
    u16* ring_hi = ring_lo + 64;                    // [64]
    PoaRowDev* stage = (PoaRowDev*)(ring_hi + 64);  // [64]
    u8* sq = (u8*)(stage + 64);                     // [seq_len]
    const PoaJobDev job = jobs[blockIdx.x];
    const int lane = threadIdx.x, L = (int)job.seq_len, N = (int)job.n_rows;
    const PoaRowDev* jr = rows + job.row_base;
    const u16* jp = preds + job.pred_base;
    int16_t* H = Hs + job.cell_base;
    u16* D = Ds + job.cell_base;
    for (int x = lane; x < L; x += 64) sq[x] = seqs[job.seq_base + x];
    __syncthreads();
    int best_v = NEG, best_i = 0, best_j = 0;       // lane-local first maximum in (row, column) order
    for (int rbase = 1; rbase <= N; rbase += 64) {
        if (rbase + lane <= N) stage[lane] = jr[rbase - 1 + lane];
        __syncthreads();
        const int rend = min(64, N - rbase + 1);
        for (int r = 0; r < rend; r++) {
            const int i = rbase + r;
            const PoaRowDev m = stage[r];
            const int lo = (int)(m.lohi & 0xFFFF), hi = (int)(m.lohi >> 16);
            const int code = (int)(m.info & 0xFF), sink = (int)((m.info >> 8) & 1), np = (int)(m.info >> 16);
            const int j0 = max(lo, 1);
            const int jf = lo + lane * C;           // first column of this lane
            int sc[C], dmax[C], umax[C], dd[C], du[C];
            #pragma unroll
            for (int c = 0; c < C; c++) {
                const int j = jf + c;
                sc[c] = (j >= 1 && j <= L && (int)sq[j - 1] == code) ? SM : SX;
                dmax[c] = NEG; umax[c] = NEG; dd[c] = 0; du[c] = 0;
            }
            const int npe = np == 0 ? 1 : np;       // no in-edges: the virtual source row 0
            for (int k = 0; k < npe; k++) {
                int p = 0;
                if (np > 0) p = k == 0 ? (int)(m.pred01 & 0xFFFF) : (k == 1 ? (int)(m.pred01 >> 16) : (int)jp[m.pred_start + k]);
                const int delta = i - p;            // p == 0 -> delta == i
                int lop, hip; int val[C + 1];
                if (p == 0) {
                    lop = 0; hip = L;
                    #pragma unroll
                    for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= 0 && x <= L) ? 0 : NEG; }
                } else if (delta < 64) {
                    const int slot = p & 63;
                    lop = (int)ring_lo[slot]; hip = (int)ring_hi[slot];
                    const int16_t* src = ring + slot * RS;
                    #pragma unroll
                    for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= lop && x <= hip) ? (int)src[x - lop] : NEG; }
                } else {
                    const u32 lh = jr[p - 1].lohi;
                    lop = (int)(lh & 0xFFFF); hip = (int)(lh >> 16);
                    const int16_t* src = H + (size_t)p * job.stride;
                    #pragma unroll
                    for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= lop && x <= hip) ? (int)src[x - lop] : NEG; }
                }
                const int ra = max(j0, lop), rb = min(hi, hip + 1);
                #pragma unroll
                for (int c = 0; c < C; c++) {
                    const int j = jf + c;
                    if (j >= ra && j <= rb) {
                        const int d = val[c] + sc[c], u = val[c + 1] + SG;
                        if (d > dmax[c]) { dmax[c] = d; dd[c] = delta; }
                        if (u > umax[c]) { umax[c] = u; du[c] = delta; }
                    }
                }
            }
            // insertion chain: prefix maximum of (tmp - j*G) over j >= j0, seeded by row[j0-1] (0 when the row starts at column 0)
            int run[C]; int acc = IDENT;
            #pragma unroll
            for (int c = 0; c < C; c++) {
                const int j = jf + c;
                if (j >= j0 && j <= hi) { const int t = max(dmax[c], umax[c]) - j * SG; acc = max(acc, t); }
                run[c] = acc;
            }
            int incl = wave_prefix_max(acc, IDENT);
            int excl = __builtin_amdgcn_update_dpp(IDENT, incl, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps IDENT)
            const int first = (lo == 0 ? 0 : NEG) - (j0 - 1) * SG;
            excl = max(excl, first);
            int16_t* dst = ring + (i & 63) * RS;
            u16* drow = D + (size_t)i * job.stride;
            int16_t* hrow = H + (size_t)i * job.stride;
            #pragma unroll
            for (int c = 0; c < C; c++) {
                const int j = jf + c;
                if (j > hi) continue;
                int v; u16 e;
                if (j < j0) { v = 0; e = 3; }                                   // column 0: free graph prefix
                else {
                    const int mm = max(excl, run[c]);
                    v = max(mm + j * SG, NEG);
                    e = dmax[c] == v ? (u16)(0 | (dd[c] << 2)) : (umax[c] == v ? (u16)(1 | (du[c] << 2)) : (u16)2);
                }
                dst[j - lo] = (int16_t)v; drow[j - lo] = e; hrow[j - lo] = (int16_t)v;
                if ((sink || j == L) && v > best_v) { best_v = v; best_i = i; best_j = j; }
            }
#ifdef POA_DEBUG
            if (i <= 3 && lane < 3) printf("row %d lane %d: lo %d hi %d code %d sink %d np %d | dmax %d %d umax %d %d run %d %d excl %d first %d sc %d %d L %d N %d SM %d SX %d SG %d NEG %d\n", i, lane, lo, hi, code, sink, np, dmax[0], dmax[1], umax[0], umax[1], run[0], run[1], excl, first, sc[0], sc[1], L, N, SM, SX, SG, NEG);
#endif
            if (lane == 0) { ring_lo[i & 63] = (u16)lo; ring_hi[i & 63] = (u16)hi; }
            __syncthreads();
        }
    }
    // best end cell: maximum value, then the smallest row, then the smallest column
    ull key = ((ull)(u32)(best_v + 32768) << 32) | (ull)(0xFFFFFFFFu - (((u32)best_i << 12) | (u32)best_j));
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const ull o = __shfl_xor(key, s); key = o > key ? o : key; }
    const int bv = (int)(u32)(key >> 32) - 32768;
    const u32 bij = 0xFFFFFFFFu - (u32)(key & 0xFFFFFFFFu);
    // the traceback looks up every visited row's first column: stage them in LDS (the ring is free now)
    u16* lo_all = (u16*)ring;
    const bool lo_in_lds = N + 1 <= 64 * RS;
    __syncthreads();
    if (lo_in_lds) for (int x = lane + 1; x <= N; x += 64) lo_all[x] = (u16)(jr[x - 1].lohi & 0xFFFF);
    __threadfence_block();
    __syncthreads();
    if (lane != 0) return;
    score[blockIdx.x] = bv;
    u32 n = 0;
    if (bv > NEG / 2) {
        int i = (int)(bij >> 12), j = (int)(bij & 0xFFF);
        int32_t* pr = path_row + job.path_base; int32_t* pp = path_pos + job.path_base;
        while (i > 0 && j > 0) {
            const int lo = lo_in_lds ? (int)lo_all[i] : (int)(jr[i - 1].lohi & 0xFFFF);
            const u16 e = D[(size_t)i * job.stride + (j - lo)];
            const int mv = e & 3, dl = e >> 2;
            if (mv == 0) { pr[n] = i; pp[n] = j - 1; n++; i -= dl; j--; }
            else if (mv == 1) { pr[n] = i; pp[n] = -1; n++; i -= dl; }
            else if (mv == 2) { pr[n] = 0; pp[n] = j - 1; n++; j--; }
            else break;
        }
    }
    path_len[blockIdx.x] = n;
}

}  // namespace

size_t poa_lds_bytes(int C, u32 max_seq_len) { return (size_t)(64 * (64 * C + 2)) * 2 + 128 * 2 + 64 * sizeof(PoaRowDev) + ((max_seq_len + 15) & ~15u); }

int launch_poa_align(svt_ctx* c, int C, u32 n_jobs, u32 max_seq_len, const void* d_jobs, const void* d_rows, const u16* d_preds, const u8* d_seqs,
                     int16_t* d_H, u16* d_D, int32_t* d_path_row, int32_t* d_path_pos, u32* d_path_len, int32_t* d_score,
                     int sm, int sx, int sg, int neg, double cells) {
    if (n_jobs == 0) return SVT_OK;
    C = C <= 2 ? 2 : (C <= 4 ? 4 : (C <= 6 ? 6 : 8));                             // the instantiated widths
    const size_t sh = poa_lds_bytes(C, max_seq_len);
    ProfScope ps(c, "k_poa_align", cells * 4.0, cells);
    #define POA_LAUNCH(CC) do { \
        HIPCHK(c, hipFuncSetAttribute((const void*)k_poa_align<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        hipLaunchKernelGGL((k_poa_align<CC>), dim3(n_jobs), dim3(64), sh, c->stream, (const PoaJobDev*)d_jobs, (const PoaRowDev*)d_rows, d_preds, d_seqs, d_H, d_D, \
                           d_path_row, d_path_pos, d_path_len, d_score, sm, sx, sg, neg); } while (0)
    if (C <= 2) POA_LAUNCH(2); else if (C <= 4) POA_LAUNCH(4); else if (C <= 6) POA_LAUNCH(6); else POA_LAUNCH(8);
    #undef POA_LAUNCH
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
