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
// Bound: not HBM, not MFMA -- a latency chain of one row per few hundred cycles per wave (LDS + DPP); HBM traffic is the
// back-pointer / spill rows, 4 B per cell.
// Round 2: a row no longer waits for its own HBM stores.  The workgroup is ONE wave, so the LDS ring needs no s_barrier (LDS
// operations of a wave complete in order): `__syncthreads()` -- s_waitcnt vmcnt(0) + s_barrier, i.e. a full HBM write round trip per
// row, 2.6 us of the 2.6 us a row took -- became an lgkmcnt wait.  A lane's C cells go to the ring as one ds_write and to the two
// HBM rows (back-pointers, spill copy) as one vector store each (row stride = 64*C elements, so every lane's slice is aligned and
// the wave's stores are contiguous).  The spill copy is read back only for predecessors more than 64 rows up (a long insertion
// bubble): that rare path first drains the store queue (vmcnt(0)); a 128-byte line never holds two rows, so no stale line exists.
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

__device__ __forceinline__ void lds_sync() {      // single-wave workgroup: order the wave's LDS traffic, never wait on HBM stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// C consecutive 16-bit values of one lane -> one vector store (dst is 2*C-byte aligned by construction)
template <int C, class T> __device__ __forceinline__ void store_cells(T* dst, const int (&v)[C]) {
    u32 w[C / 2];
    #pragma unroll
    for (int c = 0; c < C / 2; c++) w[c] = ((u32)v[2 * c] & 0xFFFFu) | ((u32)v[2 * c + 1] << 16);
    if constexpr (C == 8) *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    else if constexpr (C == 4) *reinterpret_cast<uint2*>(dst) = make_uint2(w[0], w[1]);
    else if constexpr (C == 2) *reinterpret_cast<u32*>(dst) = w[0];
    else { u32* d = reinterpret_cast<u32*>(dst); d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; }
}

template <int C>
__global__ __launch_bounds__(64) void k_poa_align(const PoaJobDev* __restrict__ jobs, const PoaRowDev* __restrict__ rows, const u16* __restrict__ preds,
                                                  const u8* __restrict__ seqs, int16_t* __restrict__ Hs, u16* __restrict__ Ds,
                                                  int32_t* __restrict__ path_row, int32_t* __restrict__ path_pos, u32* __restrict__ path_len, int32_t* __restrict__ score,
                                                  const int SM, const int SX, const int SG, const int NEG) {
    constexpr int RS = 64 * C + 8;                 // LDS row stride (int16 elements): rows stay 16-byte aligned
    constexpr int IDENT = -(1 << 29);
    extern __shared__ __attribute__((aligned(16))) int16_t lds[];
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
    const size_t stride = job.stride;               // = 64 * C
    for (int x = lane; x < L; x += 64) sq[x] = seqs[job.seq_base + x];
    lds_sync();
    int best_v = NEG, best_i = 0, best_j = 0;       // lane-local first maximum in (row, column) order
    for (int rbase = 1; rbase <= N; rbase += 64) {
        if (rbase + lane <= N) stage[lane] = jr[rbase - 1 + lane];
        lds_sync();
        const int rend = min(64, N - rbase + 1);
        for (int r = 0; r < rend; r++) {
            const int i = rbase + r;
            const PoaRowDev m = stage[r];
            const int lo = (int)(m.lohi & 0xFFFF), hi = (int)(m.lohi >> 16);
            const int code = (int)(m.info & 0xFF), sink = (int)((m.info >> 8) & 1), np = (int)(m.info >> 16);
            const int j0 = max(lo, 1);
            const int jf = lo + lane * C;           // first column of this lane
            // Straight-line code: every LDS read is unconditional on a clamped index and a select applies the range test afterwards, so
            // the reads of a row issue back to back and are waited for once (as predicated reads each one was its own exec-masked branch
            // with its own s_waitcnt: ~15 serialized LDS round trips per row).
            constexpr int MININT = -2147483647 - 1;
            int sc[C], dmax[C], umax[C], dd[C], du[C];
            #pragma unroll
            for (int c = 0; c < C; c++) {
                const int j = jf + c;
                const int sv = (int)sq[min(max(j - 1, 0), L - 1)];
                sc[c] = (j >= 1 && j <= L && sv == code) ? SM : SX;
                dmax[c] = NEG; umax[c] = NEG; dd[c] = 0; du[c] = 0;
            }
            // candidates from one predecessor row p (0 = the virtual source row of a node without in-edges)
            auto relax = [&](const int p) {
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
                    int raw[C + 1];
                    #pragma unroll
                    for (int c = 0; c <= C; c++) raw[c] = (int)src[min(max(jf - 1 + c - lop, 0), RS - 1)];
                    #pragma unroll
                    for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= lop && x <= hip) ? raw[c] : NEG; }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the spill row was stored by this wave >= 64 rows ago: make sure it has landed
                    const u32 lh = jr[p - 1].lohi;
                    lop = (int)(lh & 0xFFFF); hip = (int)(lh >> 16);
                    const volatile int16_t* src = H + (size_t)p * stride;
                    #pragma unroll
                    for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= lop && x <= hip) ? (int)src[min(max(x - lop, 0), (int)stride - 1)] : NEG; }
                }
                const int ra = max(j0, lop), rb = min(hi, hip + 1);
                #pragma unroll
                for (int c = 0; c < C; c++) {
                    const int j = jf + c;
                    const bool in = j >= ra && j <= rb;
                    const int d = in ? val[c] + sc[c] : MININT, u = in ? val[c + 1] + SG : MININT;
                    const bool bd = d > dmax[c], bu = u > umax[c];
                    dmax[c] = bd ? d : dmax[c]; dd[c] = bd ? delta : dd[c];
                    umax[c] = bu ? u : umax[c]; du[c] = bu ? delta : du[c];
                }
            };
            // the first two predecessors come with the row descriptor (LDS); only a row with more than two reads the job's list in HBM --
            // kept out of the common path, whose loop would otherwise wait for every outstanding HBM store at its join point
            const int p0 = __builtin_amdgcn_readfirstlane((int)(m.pred01 & 0xFFFF)), p1 = __builtin_amdgcn_readfirstlane((int)(m.pred01 >> 16));
            relax(np == 0 ? 0 : p0);
            if (np >= 2) relax(p1);
            if (np > 2) for (int k = 2; k < np; k++) relax(__builtin_amdgcn_readfirstlane((int)jp[m.pred_start + k]));
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
            int vv[C], ee[C];
            #pragma unroll
            for (int c = 0; c < C; c++) {
                const int j = jf + c;
                int v = 0, e = 3;                                               // column 0: free graph prefix; beyond hi: never read
                if (j >= j0 && j <= hi) {
                    const int mm = max(excl, run[c]);
                    v = max(mm + j * SG, NEG);
                    e = dmax[c] == v ? (0 | (dd[c] << 2)) : (umax[c] == v ? (1 | (du[c] << 2)) : 2);
                }
                vv[c] = v; ee[c] = e;
                if (j <= hi && (sink || j == L) && v > best_v) { best_v = v; best_i = i; best_j = j; }
            }
            store_cells<C>(ring + (i & 63) * RS + lane * C, vv);               // one ds_write per lane
            if (jf <= hi) {                                                     // lanes beyond the band store nothing
                store_cells<C>(H + (size_t)i * stride + lane * C, vv);         // spill copy, read back only by far predecessors
                store_cells<C>(D + (size_t)i * stride + lane * C, ee);         // back-pointers for the traceback
            }
            if (lane == 0) { ring_lo[i & 63] = (u16)lo; ring_hi[i & 63] = (u16)hi; }
            lds_sync();
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
    lds_sync();
    if (lo_in_lds) for (int x = lane + 1; x <= N; x += 64) lo_all[x] = (u16)(jr[x - 1].lohi & 0xFFFF);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the back-pointer rows of this wave have landed before lane 0 walks them
    lds_sync();
    if (lane != 0) return;
    score[blockIdx.x] = bv;
    u32 n = 0;
    if (bv > NEG / 2) {
        int i = (int)(bij >> 12), j = (int)(bij & 0xFFF);
        int32_t* pr = path_row + job.path_base; int32_t* pp = path_pos + job.path_base;
        const volatile u16* Dv = D;
        while (i > 0 && j > 0) {
            const int lo = lo_in_lds ? (int)lo_all[i] : (int)(jr[i - 1].lohi & 0xFFFF);
            const u16 e = Dv[(size_t)i * stride + (j - lo)];
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

size_t poa_lds_bytes(int C, u32 max_seq_len) { return (size_t)(64 * (64 * C + 8)) * 2 + 128 * 2 + 64 * sizeof(PoaRowDev) + ((max_seq_len + 15) & ~15u); }

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
