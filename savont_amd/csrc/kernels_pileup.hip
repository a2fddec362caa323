// kernels_pileup.hip -- K10: per-column statistics over device-resident pile-up rows (the output of K9, kernels_align.hip).
//
// Reference: src/alignment.rs:663-786 (estimate_quality_error_rates: which columns count, per-quality error histogram) and
// :893-1029 (analyze_pileup_consensuses: per-column depth and the two log-likelihood sums).  The reference folds every CIGAR
// into Vec<PileupBase> per position on the CPU; here the rows written by K9 never leave HBM: one thread owns one consensus
// column and walks the cluster's rows top to bottom -- adjacent threads read adjacent u64 cells (coalesced, HBM-streaming),
// and the walk order IS the reference's push order (read after read; base/deletion entry, then the insertion entry), so the
// f64 sums are the same additions in the same order (no contraction: the TU is built with -ffp-contract=off).
// log() is NOT evaluated on the device (a different libm would break bit-parity): the caller passes ln tables.
//
// Bound: HBM.  Algorithmic bytes: 8 B per (row, column) cell per pass; k_pileup_stats reads the qualifying columns twice.
#include "svt_internal.hpp"

namespace {

struct ColTile { u32 group, col0; };   // 256 consecutive columns of one group

__device__ __forceinline__ u32 q_base(const BatchView& q, u32 read, u32 pos) {
    const u32 w = q.packed[q.woff[read] + (pos >> 4)];
    return (w >> (30 - 2 * (pos & 15))) & 3;
}

// depth = entries in the column, err = entries that are not "Base equal to the consensus base" (:701-719).
// For selected groups and columns with err/depth < 0.05 (exactly 20*err < depth, depth > 0) every Base entry adds to
// total[qual] and, when it differs from the consensus base, to errs[qual] (:726-734).
__global__ __launch_bounds__(256) void k_pileup_stats(BatchView Q, const u64* __restrict__ cells, const u64* __restrict__ cell_off, const u32* __restrict__ pair_q,
                                                      const u64* __restrict__ grp_off, const u64* __restrict__ col_off, const u8* __restrict__ grp_sel,
                                                      const ColTile* __restrict__ tiles, u32* __restrict__ depth, u32* __restrict__ err,
                                                      ull* __restrict__ g_total, ull* __restrict__ g_err) {
    __shared__ u32 h_tot[256], h_err[256];
    h_tot[threadIdx.x] = 0; h_err[threadIdx.x] = 0;
    __syncthreads();
    const ColTile t = tiles[blockIdx.x];
    const u64 r0 = grp_off[t.group], r1 = grp_off[t.group + 1];
    const u32 ncol = (u32)(col_off[t.group + 1] - col_off[t.group]);
    const u32 p = t.col0 + threadIdx.x;
    const bool sel = grp_sel && grp_sel[t.group];
    if (p < ncol && r1 > r0) {
        const u32 ref = q_base(Q, pair_q[r0], p);
        u32 d = 0, e = 0;
        for (u64 r = r0; r < r1; r++) {
            const u64 c = cells[cell_off[r] + p];
            const u32 code = (u32)(c & 7);
            if (code < 4) { d++; e += code != ref; } else if (code == 4) { d++; e++; }
            if ((c >> 16) & 3) { d++; e++; }
        }
        depth[col_off[t.group] + p] = d; err[col_off[t.group] + p] = e;
        if (sel && d > 0 && 20u * e < d) {
            for (u64 r = r0; r < r1; r++) {
                const u64 c = cells[cell_off[r] + p];
                const u32 code = (u32)(c & 7);
                if (code < 4) { const u32 q = (u32)(c >> 8) & 0xFF; atomicAdd(&h_tot[q], 1u); if (code != ref) atomicAdd(&h_err[q], 1u); }
            }
        }
    } else if (p < ncol) { depth[col_off[t.group] + p] = 0; err[col_off[t.group] + p] = 0; }
    __syncthreads();
    if (h_tot[threadIdx.x]) atomicAdd(&g_total[threadIdx.x], (ull)h_tot[threadIdx.x]);
    if (h_err[threadIdx.x]) atomicAdd(&g_err[threadIdx.x], (ull)h_err[threadIdx.x]);
}

// lr = sum of ln P(entry | consensus base is right), ln = sum of ln P(entry | it is wrong) (:946-987), entry by entry.
// tab[q] = {ln(1 - er(q)), ln(er(q))}; indel = {ln(indel_error_rate), ln(1 - indel_error_rate)}.
__global__ __launch_bounds__(256) void k_pileup_loglik(BatchView Q, const u64* __restrict__ cells, const u64* __restrict__ cell_off, const u32* __restrict__ pair_q,
                                                       const u64* __restrict__ grp_off, const u64* __restrict__ col_off, const ColTile* __restrict__ tiles,
                                                       const double* __restrict__ g_tab, double indel_lr, double indel_ln, double* __restrict__ out_lr, double* __restrict__ out_ln) {
    __shared__ double tab[512];
    tab[threadIdx.x] = g_tab[threadIdx.x]; tab[256 + threadIdx.x] = g_tab[256 + threadIdx.x];
    __syncthreads();
    const ColTile t = tiles[blockIdx.x];
    const u64 r0 = grp_off[t.group], r1 = grp_off[t.group + 1];
    const u32 ncol = (u32)(col_off[t.group + 1] - col_off[t.group]);
    const u32 p = t.col0 + threadIdx.x;
    if (p >= ncol) return;
    double lr = 0.0, ln = 0.0;
    if (r1 > r0) {
        const u32 ref = q_base(Q, pair_q[r0], p);
        for (u64 r = r0; r < r1; r++) {
            const u64 c = cells[cell_off[r] + p];
            const u32 code = (u32)(c & 7);
            if (code < 4) {
                const u32 q = (u32)(c >> 8) & 0xFF;
                const double la = tab[2 * q], le = tab[2 * q + 1];
                if (code == ref) { lr += la; ln += le; } else { lr += le; ln += la; }
            } else if (code == 4) { lr += indel_lr; ln += indel_ln; }
            if ((c >> 16) & 3) {
                const u32 q = (u32)(c >> 40) & 0xFF;
                ln += tab[2 * q]; lr += tab[2 * q + 1];
            }
        }
    }
    out_lr[col_off[t.group] + p] = lr; out_ln[col_off[t.group] + p] = ln;
}

// --use-hpc (src/alignment.rs:586-625): the median homopolymer run length of the Base entries of a column (hp = bits 56-63 of a
// cell, written by K9 for a tagged target batch); even counts average the two middle values (integer division), no Base entry -> 1.
// The k-th smallest of <= a few hundred bytes is found by bisecting the VALUE range [min, max] (run lengths: a handful of distinct
// values) with one counting pass per step; the rows of a group stay in L2 between the passes.
__global__ __launch_bounds__(256) void k_pileup_hp_median(const u64* __restrict__ cells, const u64* __restrict__ cell_off, const u64* __restrict__ grp_off,
                                                          const u64* __restrict__ col_off, const ColTile* __restrict__ tiles, u8* __restrict__ out) {
    const ColTile t = tiles[blockIdx.x];
    const u64 r0 = grp_off[t.group], r1 = grp_off[t.group + 1];
    const u32 ncol = (u32)(col_off[t.group + 1] - col_off[t.group]);
    const u32 p = t.col0 + threadIdx.x;
    if (p >= ncol) return;
    u32 n = 0, mn = 255, mx = 0;
    for (u64 r = r0; r < r1; r++) {
        const u64 c = cells[cell_off[r] + p];
        if ((c & 7) < 4) { const u32 hp = (u32)(c >> 56); n++; mn = min(mn, hp); mx = max(mx, hp); }
    }
    if (n == 0) { out[col_off[t.group] + p] = 1; return; }
    auto kth = [&](u32 k) -> u32 {                       // smallest v with #(hp <= v) > k
        u32 lo = mn, hi = mx;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            u32 cnt = 0;
            for (u64 r = r0; r < r1; r++) { const u64 c = cells[cell_off[r] + p]; cnt += ((c & 7) < 4) && (u32)(c >> 56) <= mid; }
            if (cnt > k) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    const u32 mid = n / 2;
    out[col_off[t.group] + p] = (u8)((n & 1) ? kth(mid) : (kth(mid - 1) + kth(mid)) / 2);
}

}  // namespace

// host-side launchers ---------------------------------------------------------------------------------------------------
int launch_pileup_stats(svt_ctx* c, const svt_batch* Q, const u64* d_cells, const u64* d_cell_off, const u32* d_pair_q, const u64* d_grp_off, const u64* d_col_off,
                        const u8* d_grp_sel, const void* d_tiles, u32 n_tiles, u64 n_cells, u32* d_depth, u32* d_err, ull* d_total, ull* d_errs) {
    if (n_tiles == 0) return SVT_OK;
    ProfScope ps(c, "k_pileup_stats", (double)n_cells * 8.0, (double)n_cells);
    hipLaunchKernelGGL(k_pileup_stats, dim3(n_tiles), dim3(256), 0, c->stream, Q->view(), d_cells, d_cell_off, d_pair_q, d_grp_off, d_col_off, d_grp_sel,
                       (const ColTile*)d_tiles, d_depth, d_err, d_total, d_errs);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
int launch_pileup_loglik(svt_ctx* c, const svt_batch* Q, const u64* d_cells, const u64* d_cell_off, const u32* d_pair_q, const u64* d_grp_off, const u64* d_col_off,
                         const void* d_tiles, u32 n_tiles, u64 n_cells, const double* d_tab, double indel_lr, double indel_ln, double* d_lr, double* d_ln) {
    if (n_tiles == 0) return SVT_OK;
    ProfScope ps(c, "k_pileup_loglik", (double)n_cells * 8.0, (double)n_cells);
    hipLaunchKernelGGL(k_pileup_loglik, dim3(n_tiles), dim3(256), 0, c->stream, Q->view(), d_cells, d_cell_off, d_pair_q, d_grp_off, d_col_off,
                       (const ColTile*)d_tiles, d_tab, indel_lr, indel_ln, d_lr, d_ln);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
int launch_pileup_hp_median(svt_ctx* c, const u64* d_cells, const u64* d_cell_off, const u64* d_grp_off, const u64* d_col_off, const void* d_tiles, u32 n_tiles, u64 n_cells, u8* d_out) {
    if (n_tiles == 0) return SVT_OK;
    ProfScope ps(c, "k_pileup_hp_median", (double)n_cells * 8.0, (double)n_cells);
    hipLaunchKernelGGL(k_pileup_hp_median, dim3(n_tiles), dim3(256), 0, c->stream, d_cells, d_cell_off, d_grp_off, d_col_off, (const ColTile*)d_tiles, d_out);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
