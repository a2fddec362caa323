// kernels_pairs.hip -- K5/K7 (distinct-minimizer set intersections) and K6 (SNPmer bitset tiles).
//
// Reference semantics:
//   K5  src/asv_cluster.rs:131-143   count = |set(read minimizers) ∩ rep minimizer list|
//   K7  src/alignment.rs:1798-1799   |set(read) ∩ set(ASV)|
//   K6  src/asv_cluster.rs:356-383   (matches, mismatches) over shared split k-mers;
//       :481-496 compatible filter; :1057-1097 best cluster per read.
//
// K6 is bitset algebra: matches = popc(Pr & Pc & ~(Ar ^ Ac)), mismatches = popc(Pr & Pc & (Ar ^ Ac)).
// A read carries <= ~100 SNPmers out of M = 1e3..3e4 sites, so its presence row is almost empty: the ROW
// operand is kept SPARSE (the list of its non-zero 64-bit words, built once by k_snp_bits) and only those
// words of the column operand are touched.  Column operands are gathered once per call into a TRANSPOSED
// [word][column] matrix of interleaved {presence, allele} pairs, so a wavefront's 64 columns are one
// coalesced 1 KiB line per touched word; the row's word list is wave-uniform.
#include <algorithm>
#include "svt_internal.hpp"

// ------------------------------------------------------------------------------------------------
// K5 / K7: one wavefront per pair; B's set is staged in LDS, A's elements binary-search it
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_set_intersect(SeedsDev A, SeedsDev B, const u32* __restrict__ ai, const u32* __restrict__ bi, u64 n,
                                                      u32* __restrict__ shared_out, u32* __restrict__ same_out, u32 cap_lds) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64* sb = (u64*)smem;
    const u32 lane = threadIdx.x;
    for (u64 pid = blockIdx.x; pid < n; pid += gridDim.x) {      // persistent workgroups: no per-pair dispatch cost
    __syncthreads();                                             // the LDS copy of the previous pair is dead
    const u32 a = ai[pid], b = bi[pid];
    const u32 na = A.set_cnt[a], nb = B.set_cnt[b];
    const u64* pa = A.set_kmer + A.mini_base[a];
    const u64* pb = B.set_kmer + B.mini_base[b];
    u32 sh = 0, sm = 0;
    const u64 KM = ~(1ull << 63);
    if (nb <= cap_lds) {
        for (u32 i = lane; i < nb; i += 64) sb[i] = pb[i];
        __syncthreads();
        for (u32 i = lane; i < na; i += 64) {
            u64 va = pa[i], ka = va & KM;
            int lo = 0, hi = (int)nb - 1;
            while (lo <= hi) {
                int mid = (lo + hi) >> 1;
                u64 vb = sb[mid], kb = vb & KM;
                if (kb == ka) { sh++; sm += ((va >> 63) == (vb >> 63)); break; }
                if (kb < ka) lo = mid + 1; else hi = mid - 1;
            }
        }
    } else {                                                     // a set larger than the LDS copy (the copy is sized by the batch's largest capacity: unreachable unless a caller mixes batches): search it in HBM
        __syncthreads();
        for (u32 i = lane; i < na; i += 64) {
            u64 va = pa[i], ka = va & KM;
            int lo = 0, hi = (int)nb - 1;
            while (lo <= hi) {
                int mid = (lo + hi) >> 1;
                u64 vb = pb[mid], kb = vb & KM;
                if (kb == ka) { sh++; sm += ((va >> 63) == (vb >> 63)); break; }
                if (kb < ka) lo = mid + 1; else hi = mid - 1;
            }
        }
    }
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { sh += __shfl_xor(sh, s); sm += __shfl_xor(sm, s); }
    if (lane == 0) { shared_out[pid] = sh; if (same_out) same_out[pid] = sm; }
    }
}

int launch_set_intersect(svt_ctx* c, const svt_batch* A, const svt_batch* B, const u32* d_a, const u32* d_b, u64 n, u32* d_shared, u32* d_same) {
    if (n == 0) return SVT_OK;
    // The LDS copy of B's set decides how many pairs a CU works on at once (the kernel is a chain of dependent loads and LDS probes: ~5 us per pair whatever
    // the clock).  Round 3 sized it by the longest READ (16 KB at 2 kb: 10 waves per CU); no set is larger than the batch's largest minimizer capacity
    // (len / 6 + 2: 2.7 KB at 2 kb -> 32 waves per CU).  A larger set (impossible for seeds of this batch) is searched in HBM by the kernel.
    u32 cap = B->seeds.max_set ? B->seeds.max_set + 8 : B->max_len + 8;
    if (cap < 128) cap = 128;
    if ((size_t)cap * 8 > 64 * 1024) cap = 8192;
    ProfScope ps(c, "k_set_intersect", (double)n * (6.0 * 270.0 + 4.0), (double)n);   // ~1.6 KB/pair
    hipLaunchKernelGGL(k_set_intersect, dim3((u32)std::min<u64>(n, 256 * 32)), dim3(64), (size_t)cap * 8, c->stream, A->seeds, B->seeds, d_a, d_b, n, d_shared, d_same, cap);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------------
// K6
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_cols_t(const u64* __restrict__ srcP, const u64* __restrict__ srcA, const u32* __restrict__ idx, u32 n, u32 words,
                                ulonglong2* __restrict__ dst) {
    // dst[w * n + col] = {P[idx[col]][w], A[idx[col]][w]}; thread = (col, w) with w fastest on the READ side
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)n * words) return;
    u32 col = (u32)(t / words), w = (u32)(t % words);
    u64 src = (u64)(idx ? idx[col] : col) * words + w;
    ulonglong2 v; v.x = srcP[src]; v.y = srcA[src];
    dst[(u64)w * n + col] = v;
}
int launch_gather_cols_t(svt_ctx* c, const u64* srcP, const u64* srcA, const u32* d_idx, u32 n, u32 words, ulonglong2* dstPA) {
    u64 tot = (u64)n * words;
    if (tot == 0) return SVT_OK;
    ProfScope ps(c, "k_gather_cols_t", 32.0 * (double)tot, (double)n);
    hipLaunchKernelGGL(k_gather_cols_t, dim3((u32)((tot + 255) / 256)), dim3(256), 0, c->stream, srcP, srcA, d_idx, n, words, dstPA);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

#define ROWS_PER_BLOCK 8
// (matches, mismatches) of sparse row `read` against column j
__device__ __forceinline__ void sparse_row_dot(const SeedsDev& R, int view, u32 read, const ulonglong2* __restrict__ colPA, u32 n_cols, u32 j, bool jv,
                                               u32& m, u32& x) {
    m = 0; x = 0;
    const u64 base = R.snp_base[read];
    const u32 cnt = R.nz_cnt[read];
    const u64* rpv = view == SVT_VIEW_FILTERED ? R.nz_pf : R.nz_pa;
    for (u32 t = 0; t < cnt; t++) {
        const u32 w = R.nz_idx[base + t];            // wave-uniform
        const u64 rp = rpv[base + t], ra = R.nz_a[base + t];
        if (rp == 0) continue;
        ulonglong2 cv; cv.x = 0; cv.y = 0;
        if (jv) cv = colPA[(u64)w * n_cols + j];
        const u64 both = rp & cv.x, d = ra ^ cv.y;
        m += __popcll(both & ~d); x += __popcll(both & d);
    }
}

__global__ void __launch_bounds__(256) k_compat_lists(SeedsDev R, int row_view, const u32* __restrict__ row_idx, u32 n_rows,
                                                      const ulonglong2* __restrict__ colPA, u32 n_cols,
                                                      int filter, int triangular, u32 tri_base, const u32* __restrict__ row_max_x,
                                                      u32* __restrict__ o_row, u32* __restrict__ o_col, u32* __restrict__ o_mm, u64 cap, ull* __restrict__ counter) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;       // column
    const u32 lane = threadIdx.x & 63;
    const u32 r0 = blockIdx.y * ROWS_PER_BLOCK;
    const bool jv = j < n_cols;
    const u64* rpv = row_view == SVT_VIEW_FILTERED ? R.nz_pf : R.nz_pa;
    // Row operands are wave-uniform.  Reading them with scalar loads inside the word loop made every launch a chain of ~500
    // dependent loads (~50 us however small the tile); instead lanes 0..7 fetch the 8 row descriptors, then lane t fetches entry t
    // of a row (coalesced) and the loop broadcasts entries with v_readlane.
    u64 my_base = 0; u32 my_cnt = 0, my_maxx = 0xFFFFFFFFu;
    if (lane < ROWS_PER_BLOCK && r0 + lane < n_rows) {
        const u32 read = row_idx[r0 + lane]; my_base = R.snp_base[read]; my_cnt = R.nz_cnt[read];
        if (row_max_x) my_maxx = row_max_x[r0 + lane];
    }
    u32 mm[ROWS_PER_BLOCK]; ull masks[ROWS_PER_BLOCK]; u32 total = 0;
    #pragma unroll
    for (int r = 0; r < ROWS_PER_BLOCK; r++) {
        const u32 ri = r0 + r;
        const u64 base = ((u64)(u32)__builtin_amdgcn_readlane((int)(my_base >> 32), r) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)my_base, r);
        const u32 cnt = (u32)__builtin_amdgcn_readlane((int)my_cnt, r);
        const u32 maxx = (u32)__builtin_amdgcn_readlane((int)my_maxx, r);
        u32 m = 0, x = 0;
        for (u32 c0 = 0; c0 < cnt; c0 += 64) {                  // wave-uniform
            const u32 nn = min(64u, cnt - c0);
            u32 e_idx = 0; u64 e_p = 0, e_a = 0;
            if (lane < nn) { e_idx = R.nz_idx[base + c0 + lane]; e_p = rpv[base + c0 + lane]; e_a = R.nz_a[base + c0 + lane]; }
            for (u32 t = 0; t < nn; t++) {
                const u32 plo = (u32)__builtin_amdgcn_readlane((int)(u32)e_p, t), phi = (u32)__builtin_amdgcn_readlane((int)(e_p >> 32), t);
                if ((plo | phi) == 0) continue;
                const u64 rp = ((u64)phi << 32) | plo;
                const u64 ra = ((u64)(u32)__builtin_amdgcn_readlane((int)(e_a >> 32), t) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)e_a, t);
                const u32 w = (u32)__builtin_amdgcn_readlane((int)e_idx, t);
                ulonglong2 cv; cv.x = 0; cv.y = 0;
                if (jv) cv = colPA[(u64)w * n_cols + j];
                const u64 both = rp & cv.x, d = ra ^ cv.y;
                m += __popcll(both & ~d); x += __popcll(both & d);
            }
        }
        bool keep = false;
        if (ri < n_rows) {                                       // wave-uniform
            keep = jv && (filter == SVT_LIST_COMPATIBLE ? (x == 0 && m > 0) : (m + x > 0));
            if (triangular && j >= tri_base) keep = keep && (j - tri_base < ri);     // in-block columns: only EARLIER rows
            keep = keep && (x <= maxx);
        }
        mm[r] = (m << 16) | (x & 0xFFFF);
        masks[r] = __ballot(keep);
        total += __popcll(masks[r]);
    }
    if (total == 0) return;                                      // wave-uniform
    ull pos = 0;
    if (lane == 0) pos = atomicAdd(counter, (ull)total);         // ONE append per wave for all 8 rows
    pos = __shfl(pos, 0);
    #pragma unroll
    for (int r = 0; r < ROWS_PER_BLOCK; r++) {
        const ull mk = masks[r];
        if ((mk >> lane) & 1) {
            const u64 d = pos + d_rank(mk);
            if (d < cap) { o_row[d] = r0 + r; o_col[d] = j; o_mm[d] = mm[r]; }
        }
        pos += __popcll(mk);
    }
}
// Same result, column operands staged in LDS: one wave owns a tile of 64 columns ({P,A} of all `words` words: words KB of LDS) and
// walks ROWS_PER_TILE rows over it.  The plain kernel re-reads 16 B from L2 per (row, column, non-zero row word): ~320 MB of L2
// traffic for a 1024 x 2000 tile whose operands are 1 MB; here L2 sees every column word once per 128 rows.
#define ROWS_PER_TILE 128
__global__ void __launch_bounds__(256) k_compat_lists_lds(SeedsDev R, int row_view, const u32* __restrict__ row_idx, u32 n_rows,
                                                         const ulonglong2* __restrict__ colPA, u32 n_cols, u32 words,
                                                         int filter, int triangular, u32 tri_base, const u32* __restrict__ row_max_x,
                                                         u32* __restrict__ o_row, u32* __restrict__ o_col, u32* __restrict__ o_mm, u64 cap, ull* __restrict__ counter) {
    extern __shared__ ulonglong2 tile[];                          // [words][64]
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // 4 waves share the tile, each walks a quarter of the rows
    const u32 j = blockIdx.x * 64 + lane;
    const bool jv = j < n_cols;
    for (u32 w = wave; w < words; w += 4) { ulonglong2 v; v.x = 0; v.y = 0; if (jv) v = colPA[(u64)w * n_cols + j]; tile[w * 64 + lane] = v; }
    __syncthreads();
    const u64* rpv = row_view == SVT_VIEW_FILTERED ? R.nz_pf : R.nz_pa;
    const u32 rbeg = blockIdx.y * ROWS_PER_TILE + wave * (ROWS_PER_TILE / 4), rend = min(n_rows, rbeg + ROWS_PER_TILE / 4);
    for (u32 r0 = rbeg; r0 < rend; r0 += ROWS_PER_BLOCK) {
        // row descriptors of the 8 rows by lanes 0..7 (two dependent vector loads instead of 24 dependent scalar loads)
        u64 my_base = 0; u32 my_cnt = 0;
        if (lane < ROWS_PER_BLOCK && r0 + lane < rend) { const u32 read = row_idx[r0 + lane]; my_base = R.snp_base[read]; my_cnt = R.nz_cnt[read]; }
        u32 mm[ROWS_PER_BLOCK]; ull masks[ROWS_PER_BLOCK]; u32 total = 0;
        #pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; r++) {
            const u32 ri = r0 + r;
            const u64 base = ((u64)(u32)__builtin_amdgcn_readlane((int)(my_base >> 32), r) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)my_base, r);
            const u32 cnt = (u32)__builtin_amdgcn_readlane((int)my_cnt, r);
            u32 m = 0, x = 0;
            // the row's non-zero words: lane t holds entry t (one coalesced load per array), the loop broadcasts them with readlane
            for (u32 c0 = 0; c0 < cnt; c0 += 64) {                    // wave-uniform
                const u32 nn = min(64u, cnt - c0);
                u32 e_idx = 0; u64 e_p = 0, e_a = 0;
                if (lane < nn) { e_idx = R.nz_idx[base + c0 + lane]; e_p = rpv[base + c0 + lane]; e_a = R.nz_a[base + c0 + lane]; }
                for (u32 t = 0; t < nn; t++) {
                    const u32 plo = (u32)__builtin_amdgcn_readlane((int)(u32)e_p, t), phi = (u32)__builtin_amdgcn_readlane((int)(e_p >> 32), t);
                    if ((plo | phi) == 0) continue;
                    const u64 rp = ((u64)phi << 32) | plo;
                    const u64 ra = ((u64)(u32)__builtin_amdgcn_readlane((int)(e_a >> 32), t) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)e_a, t);
                    const ulonglong2 cv = tile[(u32)__builtin_amdgcn_readlane((int)e_idx, t) * 64 + lane];
                    const u64 both = rp & cv.x, d = ra ^ cv.y;
                    m += __popcll(both & ~d); x += __popcll(both & d);
                }
            }
            bool keep = false;
            if (ri < rend) {
                keep = jv && (filter == SVT_LIST_COMPATIBLE ? (x == 0 && m > 0) : (m + x > 0));
                if (triangular && j >= tri_base) keep = keep && (j - tri_base < ri);
                if (row_max_x) keep = keep && (x <= row_max_x[ri]);
            }
            mm[r] = (m << 16) | (x & 0xFFFF);
            masks[r] = __ballot(keep);
            total += __popcll(masks[r]);
        }
        if (total == 0) continue;                                 // wave-uniform
        ull pos = 0;
        if (lane == 0) pos = atomicAdd(counter, (ull)total);
        pos = __shfl(pos, 0);
        #pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; r++) {
            const ull mk = masks[r];
            if ((mk >> lane) & 1) {
                const u64 d = pos + d_rank(mk);
                if (d < cap) { o_row[d] = r0 + r; o_col[d] = j; o_mm[d] = mm[r]; }
            }
            pos += __popcll(mk);
        }
    }
}
// ---- column-sparse form (columns are reads / ASVs of a batch) ---------------------------------------------------------------
// The tile kernels above keep the ROW sparse and the columns dense, so every (row, non-zero row word) re-reads 16 B per column
// from L2.  When the columns are sequences of a batch their sparse lists exist too: here a block builds the DENSE {P,A} rows of
// RT consecutive rows in LDS (RT * words * 16 B), and every thread walks the non-zero words of ITS column once, looking the RT
// rows up in LDS.  L2 sees a column's ~15 entries once per RT rows instead of 15 words per row; no column gather pass at all.
template <int RT>
__global__ void __launch_bounds__(256) k_compat_lists_cs(SeedsDev R, int row_view, const u32* __restrict__ row_idx, u32 n_rows,
                                                         SeedsDev C, int col_view, const u32* __restrict__ col_idx, u32 n_cols, u32 words,
                                                         int filter, int triangular, u32 tri_base, const u32* __restrict__ row_max_x,
                                                         u32* __restrict__ o_row, u32* __restrict__ o_col, u32* __restrict__ o_mm, u64 cap, ull* __restrict__ counter,
                                                         u32 col_lo, u32* __restrict__ row_has, const u32* __restrict__ sel_list, const u32* __restrict__ sel_count) {
    extern __shared__ ulonglong2 rows_lds[];                      // [RT][words]
    const u32 lane = threadIdx.x & 63;
    const u32 r0 = blockIdx.y * RT;
    const u32 max_row = min(r0 + RT, n_rows) - 1;
    // columns of this launch: positions [col_lo, n_cols) of col_idx, or -- sel_list -- a device-made list of positions (the in-tile
    // columns whose own row found no compatible column among the first tri_base ones: only those can become representatives in the
    // caller's greedy loop; usually none or a handful, so ONE column block per row tile walks the list in chunks of 256)
    const u32 n_eff = sel_list ? *sel_count : n_cols - col_lo;
    const u32 cb0 = blockIdx.x * 256, cstep = gridDim.x * 256;
    bool any = false;
    for (u32 cc = cb0 + threadIdx.x; cc < n_eff; cc += cstep) {
        const u32 jj = sel_list ? sel_list[cc] : col_lo + cc;
        if (!(triangular && jj >= tri_base && jj - tri_base >= max_row)) any = true;   // an in-tile column only meets LATER rows
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;                  // nothing to compare in this (column block, row tile): skip the LDS build
    for (u32 x = threadIdx.x; x < RT * words; x += 256) { ulonglong2 z; z.x = 0; z.y = 0; rows_lds[x] = z; }
    __syncthreads();
    {
        const u64* rpv = row_view == SVT_VIEW_FILTERED ? R.nz_pf : R.nz_pa;
        for (u32 r = threadIdx.x >> 4; r < RT; r += 16) {         // 16 threads per row
            if (r0 + r >= n_rows) continue;
            const u32 read = row_idx[r0 + r];
            const u64 base = R.snp_base[read]; const u32 cnt = R.nz_cnt[read];
            for (u32 t = threadIdx.x & 15; t < cnt; t += 16) { ulonglong2 v; v.x = rpv[base + t]; v.y = R.nz_a[base + t]; rows_lds[r * words + R.nz_idx[base + t]] = v; }
        }
    }
    __syncthreads();
    __shared__ u32 wave_tot[4]; __shared__ ull blk_base;
    for (u32 cbase = cb0; cbase < n_eff; cbase += cstep) {       // block-uniform trip count (one trip unless sel_list is long)
        const u32 cc = cbase + threadIdx.x;
        bool jv = cc < n_eff;
        const u32 j = jv ? (sel_list ? sel_list[cc] : col_lo + cc) : 0;
        if (jv && triangular && j >= tri_base && j - tri_base >= max_row) jv = false;
        u32 m[RT], x[RT];
        #pragma unroll
        for (int r = 0; r < RT; r++) { m[r] = 0; x[r] = 0; }
        if (jv) {
            const u32 col = col_idx[j];
            const u64 base = C.snp_base[col]; const u32 cnt = C.nz_cnt[col];
            const u64* cpv = col_view == SVT_VIEW_FILTERED ? C.nz_pf : C.nz_pa;
            for (u32 t = 0; t < cnt; t++) {
                const u64 cp = cpv[base + t];
                if (cp == 0) continue;
                const u64 ca = C.nz_a[base + t];
                const u32 w = C.nz_idx[base + t];
                #pragma unroll
                for (int r = 0; r < RT; r++) {
                    const ulonglong2 rv = rows_lds[r * words + w];
                    const u64 both = rv.x & cp, d = rv.y ^ ca;
                    m[r] += __popcll(both & ~d); x[r] += __popcll(both & d);
                }
            }
        }
        // appends: ONE atomic per block and chunk (a counter hit by thousands of wave-level atomics per launch serialises the whole launch)
        ull masks[RT]; u32 total = 0;
        #pragma unroll
        for (int r = 0; r < RT; r++) {
            const u32 ri = r0 + r;
            bool keep = false;
            if (ri < n_rows) {                                    // wave-uniform
                keep = jv && (filter == SVT_LIST_COMPATIBLE ? (x[r] == 0 && m[r] > 0) : (m[r] + x[r] > 0));
                if (triangular && j >= tri_base) keep = keep && (j - tri_base < ri);
                if (row_max_x) keep = keep && (x[r] <= row_max_x[ri]);
            }
            masks[r] = __ballot(keep);
            total += __popcll(masks[r]);
            if (row_has && masks[r] != 0 && lane == 0) row_has[ri] = 1;
        }
        const u32 wave = threadIdx.x >> 6;
        __syncthreads();                                          // the previous chunk's readers of wave_tot / blk_base are done
        if (lane == 0) wave_tot[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) { const u32 all = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3]; blk_base = all ? atomicAdd(counter, (ull)all) : 0; }
        __syncthreads();
        ull pos = blk_base;
        for (u32 w = 0; w < wave; w++) pos += wave_tot[w];
        #pragma unroll
        for (int r = 0; r < RT; r++) {
            const ull mk = masks[r];
            if ((mk >> lane) & 1) {
                const u64 d = pos + d_rank(mk);
                if (d < cap) { o_row[d] = r0 + r; o_col[d] = j; o_mm[d] = (m[r] << 16) | (x[r] & 0xFFFF); }
            }
            pos += __popcll(mk);
        }
    }
}
// in-tile columns (positions tri_base + i) whose own row i has no flag -> list + count
__global__ void k_unflagged_cols(const u32* __restrict__ flags, u32 n, u32 tri_base, u32* __restrict__ sel, u32* __restrict__ count) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !flags[i]) sel[atomicAdd(count, 1u)] = tri_base + i;
}
int launch_unflagged_cols(svt_ctx* c, const u32* d_flags, u32 n, u32 tri_base, u32* d_sel, u32* d_count) {
    if (n == 0) return SVT_OK;
    hipLaunchKernelGGL(k_unflagged_cols, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_flags, n, tri_base, d_sel, d_count);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
// ---- the same tiles for MANY k-mer clusters in one launch (Stage 3, one launch per wave of blocks) ----------------------------------
// A segment = the block of reads one k-mer cluster's greedy loop looks at next: its rows are rows [row_begin, +n_rows) of row_idx, its columns
// col_idx[col_begin ..] = the cluster's n_reps representatives followed by the rows themselves (in-block columns, local index = position in the
// block).  A tile = RT consecutive rows of ONE segment.  phase 0: every tile against its segment's representatives (sets row_has[row]);
// phase 1: every tile against its segment's list of in-block columns whose own row found no compatible representative (sel / sel_count, made by
// k_unflagged_cols_seg), an in-block column meeting only LATER rows.  o_row = global row, o_col = column index inside the segment
// (< n_reps: representative, else n_reps + local index): what svt_snpmer_compat_lists reports for one cluster at a time.
struct SegDesc { u32 row_begin, n_rows, col_begin, n_reps; };
struct SegTile { u32 seg, row0; };
template <int RT>
__global__ void __launch_bounds__(256) k_compat_lists_seg(SeedsDev R, int row_view, const u32* __restrict__ row_idx, const SegTile* __restrict__ tiles, const SegDesc* __restrict__ segs,
                                                          SeedsDev C, int col_view, const u32* __restrict__ col_idx, u32 words, int filter, int phase,
                                                          u32* __restrict__ o_row, u32* __restrict__ o_col, u32* __restrict__ o_mm, u64 cap, ull* __restrict__ counter,
                                                          u32* __restrict__ row_has, const u32* __restrict__ sel, const u32* __restrict__ sel_count) {
    extern __shared__ ulonglong2 rows_lds[];                      // [RT][words]
    const u32 lane = threadIdx.x & 63;
    const SegTile tile = tiles[blockIdx.y];
    const SegDesc sg = segs[tile.seg];
    const u32 r0 = tile.row0, rend = min(tile.row0 + RT, sg.row_begin + sg.n_rows);     // global rows [r0, rend)
    const u32 max_local = rend - 1 - sg.row_begin;                // local index of the tile's last row
    const u32 n_eff = phase == 0 ? sg.n_reps : sel_count[tile.seg];
    const u32* sl = sel + sg.row_begin;                           // the segment's slice of the selection list (capacity n_rows)
    const u32 cb0 = blockIdx.x * 256, cstep = gridDim.x * 256;
    bool any = false;
    for (u32 cc = cb0 + threadIdx.x; cc < n_eff; cc += cstep) {
        if (phase == 0 || sl[cc] < max_local) any = true;         // an in-block column only meets LATER rows
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;
    for (u32 x = threadIdx.x; x < RT * words; x += 256) { ulonglong2 z; z.x = 0; z.y = 0; rows_lds[x] = z; }
    __syncthreads();
    {
        const u64* rpv = row_view == SVT_VIEW_FILTERED ? R.nz_pf : R.nz_pa;
        for (u32 r = threadIdx.x >> 4; r < RT; r += 16) {         // 16 threads per row
            if (r0 + r >= rend) continue;
            const u32 read = row_idx[r0 + r];
            const u64 base = R.snp_base[read]; const u32 cnt = R.nz_cnt[read];
            for (u32 t = threadIdx.x & 15; t < cnt; t += 16) { ulonglong2 v; v.x = rpv[base + t]; v.y = R.nz_a[base + t]; rows_lds[r * words + R.nz_idx[base + t]] = v; }
        }
    }
    __syncthreads();
    __shared__ u32 wave_tot[4]; __shared__ ull blk_base;
    for (u32 cbase = cb0; cbase < n_eff; cbase += cstep) {       // block-uniform trip count
        const u32 cc = cbase + threadIdx.x;
        bool jv = cc < n_eff;
        const u32 jl = jv ? (phase == 0 ? cc : sl[cc]) : 0;       // phase 0: representative index; phase 1: local index of the in-block column
        if (jv && phase == 1 && jl >= max_local) jv = false;
        u32 m[RT], x[RT];
        #pragma unroll
        for (int r = 0; r < RT; r++) { m[r] = 0; x[r] = 0; }
        if (jv) {
            const u32 col = col_idx[sg.col_begin + (phase == 0 ? jl : sg.n_reps + jl)];
            const u64 base = C.snp_base[col]; const u32 cnt = C.nz_cnt[col];
            const u64* cpv = col_view == SVT_VIEW_FILTERED ? C.nz_pf : C.nz_pa;
            for (u32 t = 0; t < cnt; t++) {
                const u64 cp = cpv[base + t];
                if (cp == 0) continue;
                const u64 ca = C.nz_a[base + t];
                const u32 w = C.nz_idx[base + t];
                #pragma unroll
                for (int r = 0; r < RT; r++) {
                    const ulonglong2 rv = rows_lds[r * words + w];
                    const u64 both = rv.x & cp, d = rv.y ^ ca;
                    m[r] += __popcll(both & ~d); x[r] += __popcll(both & d);
                }
            }
        }
        ull masks[RT]; u32 total = 0;
        #pragma unroll
        for (int r = 0; r < RT; r++) {
            const u32 ri = r0 + r;
            bool keep = false;
            if (ri < rend) {                                      // wave-uniform
                keep = jv && (filter == SVT_LIST_COMPATIBLE ? (x[r] == 0 && m[r] > 0) : (m[r] + x[r] > 0));
                if (phase == 1) keep = keep && (jl < ri - sg.row_begin);
            }
            masks[r] = __ballot(keep);
            total += __popcll(masks[r]);
            if (phase == 0 && masks[r] != 0 && lane == 0) row_has[ri] = 1;
        }
        const u32 wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) wave_tot[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) { const u32 all = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3]; blk_base = all ? atomicAdd(counter, (ull)all) : 0; }
        __syncthreads();
        ull pos = blk_base;
        for (u32 w = 0; w < wave; w++) pos += wave_tot[w];
        #pragma unroll
        for (int r = 0; r < RT; r++) {
            const ull mk = masks[r];
            if ((mk >> lane) & 1) {
                const u64 d = pos + d_rank(mk);
                if (d < cap) { o_row[3 * d] = r0 + r; o_col[3 * d] = phase == 0 ? jl : sg.n_reps + jl; o_mm[3 * d] = (m[r] << 16) | (x[r] & 0xFFFF); }   // (row, col, mm) records
            }
            pos += __popcll(mk);
        }
    }
}
// in-block columns whose own row found no compatible representative -> the segment's selection list (local indices) + count
__global__ void k_unflagged_cols_seg(const u32* __restrict__ flags, const u32* __restrict__ row_seg, const SegDesc* __restrict__ segs, u32 n, u32* __restrict__ sel, u32* __restrict__ count) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || flags[i]) return;
    const u32 s = row_seg[i]; const u32 rb = segs[s].row_begin;
    sel[rb + atomicAdd(&count[s], 1u)] = i - rb;
}
size_t seg_desc_bytes() { return sizeof(SegDesc); }
size_t seg_tile_bytes() { return sizeof(SegTile); }
int compat_seg_rt(u32 words) { const size_t per_row = (size_t)words * sizeof(ulonglong2); return per_row * 16 <= 150 * 1024 ? 16 : (per_row * 8 <= 150 * 1024 ? 8 : 0); }
int launch_compat_lists_seg(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const void* d_tiles, u32 n_tiles, const void* d_segs, u32 max_reps,
                            const SeedsDev& cols, int col_view, const u32* d_col_idx, u32 n_cols, u32 words, int filter, int phase,
                            u32* o_row, u32* o_col, u32* o_mm, u64 cap, ull* d_counter, u32* d_row_has, const u32* d_sel, const u32* d_sel_count) {
    if (n_tiles == 0) return SVT_OK;
    if (phase == 0 && max_reps == 0) return SVT_OK;
    const int RT = compat_seg_rt(words);
    if (RT == 0) return 1;
    const size_t per_row = (size_t)words * sizeof(ulonglong2);
    const u32 nc = phase == 0 ? max_reps : 256;                     // phase 1: one column block per tile walks its segment's list
    double bytes = 20.0 * 16.0 * ((double)n_rows + (double)n_cols);
    ProfScope ps(c, "k_compat_lists", bytes, (double)n_rows * (double)(phase == 0 ? max_reps : 1));
    const size_t sh = per_row * RT;
    dim3 grid((nc + 255) / 256, n_tiles);
    if (RT == 16) {
        DYN_LDS_ONCE(c, 0, k_compat_lists_seg<16>, 150 * 1024);
        hipLaunchKernelGGL((k_compat_lists_seg<16>), grid, dim3(256), sh, c->stream, rows, row_view, d_row_idx, (const SegTile*)d_tiles, (const SegDesc*)d_segs, cols, col_view, d_col_idx, words, filter, phase, o_row, o_col, o_mm, cap, d_counter, d_row_has, d_sel, d_sel_count);
    } else {
        DYN_LDS_ONCE(c, 1, k_compat_lists_seg<8>, 150 * 1024);
        hipLaunchKernelGGL((k_compat_lists_seg<8>), grid, dim3(256), sh, c->stream, rows, row_view, d_row_idx, (const SegTile*)d_tiles, (const SegDesc*)d_segs, cols, col_view, d_col_idx, words, filter, phase, o_row, o_col, o_mm, cap, d_counter, d_row_has, d_sel, d_sel_count);
    }
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
int launch_unflagged_cols_seg(svt_ctx* c, const u32* d_flags, const u32* d_row_seg, const void* d_segs, u32 n, u32* d_sel, u32* d_count) {
    if (n == 0) return SVT_OK;
    hipLaunchKernelGGL(k_unflagged_cols_seg, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_flags, d_row_seg, (const SegDesc*)d_segs, n, d_sel, d_count);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ---- (row, col, mm) records -> rows: what the greedy loop of a wave wants is "the entries of row r" (src/asv_cluster.rs:596-660 walks a read's compatible
// representatives), and building that on the host was three passes over ~10^6 records per 100k-read step.  k_rec_row_count counts the records of every row
// (the count of the launch that made them is still on the device: d_count, clamped to the capacity), the host turns the counts into offsets while it reads
// the total anyway, k_rec_fill drops every record into its row's slots (order inside a row: as the atomics fall, as unordered as the records were).
__global__ void k_rec_row_count(const u32* __restrict__ rec, const ull* __restrict__ d_count, u64 n_host, u64 cap, u32* __restrict__ row_cnt) {
    const u64 n = d_count ? min((u64)*d_count, cap) : n_host;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) atomicAdd(&row_cnt[rec[3 * i]], 1u);
}
__global__ void k_rec_fill(const u32* __restrict__ rec, u64 n, const u32* __restrict__ off, u32* __restrict__ cur, u32* __restrict__ o_col, u32* __restrict__ o_mm) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 r = rec[3 * i];
        const u32 p = off[r] + atomicAdd(&cur[r], 1u);
        o_col[p] = rec[3 * i + 1]; o_mm[p] = rec[3 * i + 2];
    }
}
int launch_rec_row_count(svt_ctx* c, const u32* d_rec, const ull* d_count, u64 n_host, u64 cap, u32* d_row_cnt) {
    const u64 n = d_count ? cap : n_host;
    if (n == 0) return SVT_OK;
    hipLaunchKernelGGL(k_rec_row_count, dim3((u32)std::min<u64>((n + 255) / 256, 2048)), dim3(256), 0, c->stream, d_rec, d_count, n_host, cap, d_row_cnt);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
int launch_rec_fill(svt_ctx* c, const u32* d_rec, u64 n, const u32* d_off, u32* d_cur, u32* d_col, u32* d_mm) {
    if (n == 0) return SVT_OK;
    hipLaunchKernelGGL(k_rec_fill, dim3((u32)std::min<u64>((n + 255) / 256, 2048)), dim3(256), 0, c->stream, d_rec, n, d_off, d_cur, d_col, d_mm);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// returns SVT_OK after launching, or 1 when the dense rows do not fit LDS (the caller falls back to the dense-column kernels)
int launch_compat_lists_cs(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const SeedsDev& cols, int col_view, const u32* d_col_idx, u32 n_cols,
                           u32 words, int filter, int triangular, u32 tri_base, const u32* d_row_max_x, u32* o_row, u32* o_col, u32* o_mm, u64 cap, ull* d_counter,
                           u32 col_lo, u32* d_row_has, const u32* d_sel_list, const u32* d_sel_count) {
    if (n_rows == 0 || n_cols <= col_lo) return SVT_OK;
    const size_t per_row = (size_t)words * sizeof(ulonglong2);
    const int RT = per_row * 16 <= 150 * 1024 ? 16 : (per_row * 8 <= 150 * 1024 ? 8 : 0);
    if (RT == 0) return 1;
    const u32 nc = d_sel_list ? 256 : n_cols - col_lo;             // a device-made column list: one column block per row tile walks it
    // algorithmic bytes of the column-sparse kernel: the SPARSE operands once (20 B per non-zero word: index + P_all + P_filt + A, ~16 non-zero
    // words per read) -- the emitted triples (12 B each) are added by the caller once the count is known (prof_add_bytes)
    double bytes = 20.0 * 16.0 * ((double)n_rows + (double)nc);
    ProfScope ps(c, "k_compat_lists", bytes, (double)n_rows * (double)nc);
    const size_t sh = per_row * RT;
    dim3 grid((nc + 255) / 256, (n_rows + RT - 1) / RT);
    if (RT == 16) {
        DYN_LDS_ONCE(c, 2, k_compat_lists_cs<16>, 150 * 1024);
        hipLaunchKernelGGL((k_compat_lists_cs<16>), grid, dim3(256), sh, c->stream, rows, row_view, d_row_idx, n_rows, cols, col_view, d_col_idx, n_cols, words, filter, triangular, tri_base, d_row_max_x, o_row, o_col, o_mm, cap, d_counter, col_lo, d_row_has, d_sel_list, d_sel_count);
    } else {
        DYN_LDS_ONCE(c, 3, k_compat_lists_cs<8>, 150 * 1024);
        hipLaunchKernelGGL((k_compat_lists_cs<8>), grid, dim3(256), sh, c->stream, rows, row_view, d_row_idx, n_rows, cols, col_view, d_col_idx, n_cols, words, filter, triangular, tri_base, d_row_max_x, o_row, o_col, o_mm, cap, d_counter, col_lo, d_row_has, d_sel_list, d_sel_count);
    }
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

int launch_compat_lists(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const ulonglong2* colPA, u32 n_cols, u32 words,
                        int filter, int triangular, u32 tri_base, const u32* d_row_max_x, u32* o_row, u32* o_col, u32* o_mm, u64 cap, ull* d_counter) {
    if (n_rows == 0 || n_cols == 0) return SVT_OK;
    if (words <= 96) {                                            // the column tile fits LDS (words KB of the CU's 160 KB)
        double bytes = 20.0 * 16.0 * (double)n_rows + 16.0 * words * (double)n_cols;     // sparse rows + dense columns once; emitted triples added by the caller
        ProfScope ps(c, "k_compat_lists", bytes, (double)n_rows * (double)n_cols);
        const size_t sh = (size_t)words * 64 * sizeof(ulonglong2);
        DYN_LDS_ONCE(c, 4, k_compat_lists_lds, 96 * 64 * sizeof(ulonglong2));
        dim3 grid((n_cols + 63) / 64, (n_rows + ROWS_PER_TILE - 1) / ROWS_PER_TILE);
        hipLaunchKernelGGL(k_compat_lists_lds, grid, dim3(256), sh, c->stream, rows, row_view, d_row_idx, n_rows, colPA, n_cols, words, filter, triangular, tri_base, d_row_max_x,
                           o_row, o_col, o_mm, cap, d_counter);
        HIPCHK(c, hipGetLastError());
        return SVT_OK;
    }
    // SURVEY 8d K6 charges a dense 4 B per pair; the kernel emits only the listed pairs (12 B each, added by the caller)
    double bytes = 20.0 * 16.0 * (double)n_rows + 16.0 * words * (double)n_cols;
    ProfScope ps(c, "k_compat_lists", bytes, (double)n_rows * (double)n_cols);
    dim3 grid((n_cols + 255) / 256, (n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL(k_compat_lists, grid, dim3(256), 0, c->stream, rows, row_view, d_row_idx, n_rows, colPA, n_cols, filter, triangular, tri_base, d_row_max_x,
                       o_row, o_col, o_mm, cap, d_counter);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// per row: FIRST column of [lo, hi) with the smallest (mismatches, -matches)   (asv_cluster.rs:1058-1097; the initial
// best (usize::MAX, 0) loses to the first column, so "first column wins ties" is the whole rule).  One workgroup per row,
// thread = column: key = mismatches<<48 | (0xFFFF-matches)<<32 | column, block-wide min.
__global__ void __launch_bounds__(256) k_best_column(SeedsDev R, int row_view, const u32* __restrict__ row_idx, u32 n_rows,
                                                     const ulonglong2* __restrict__ colPA, u32 n_cols, const u32* __restrict__ lo_, const u32* __restrict__ hi_,
                                                     u32* __restrict__ best_col, u32* __restrict__ best_score) {
    // Round 5: a WAVE per row (four rows per workgroup).  A row's column range [lo, hi) is the handful of SNPmer clusters of its k-mer cluster -- rarely more than 64 --
    // and a workgroup per row made four waves walk the row's non-zero words (a chain of dependent loads) for columns only the first wave had: 1.1 ms per 100k-read step.
    const u32 ri = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (ri >= n_rows) return;
    const u32 lane = threadIdx.x & 63;
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(lo_ ? lo_[ri] : 0)), hi = (u32)__builtin_amdgcn_readfirstlane((int)(hi_ ? hi_[ri] : n_cols));
    const u32 read = (u32)__builtin_amdgcn_readfirstlane((int)row_idx[ri]);
    u64 best = ~0ull;
    for (u32 cb = lo; cb < hi; cb += 64) {                       // wave-uniform trip count
        const u32 j = cb + lane; const bool jv = j < hi;
        u32 m, x;
        sparse_row_dot(R, row_view, read, colPA, n_cols, j, jv, m, x);
        if (jv) { u64 key = ((u64)x << 48) | ((u64)(0xFFFFu - m) << 32) | j; best = key < best ? key : best; }
    }
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { u64 o = __shfl_xor(best, s); best = o < best ? o : best; }
    if (lane == 0) {
        if (best == ~0ull) { best_col[ri] = lo; if (best_score) best_score[ri] = 0xFFFFu; }
        else { best_col[ri] = (u32)best; if (best_score) best_score[ri] = ((0xFFFFu - (u32)((best >> 32) & 0xFFFF)) << 16) | (u32)(best >> 48); }
    }
}
int launch_best_column(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const ulonglong2* colPA, u32 n_cols, u32 words,
                       const u32* d_lo, const u32* d_hi, u32* best_col, u32* best_score) {
    if (n_rows == 0) return SVT_OK;
    double bytes = 16.0 * words * ((double)n_rows + (double)n_cols) + 8.0 * (double)n_rows;
    ProfScope ps(c, "k_best_column", bytes, (double)n_rows);
    hipLaunchKernelGGL(k_best_column, dim3((n_rows + 3) / 4), dim3(256), 0, c->stream, rows, row_view, d_row_idx, n_rows, colPA, n_cols, d_lo, d_hi, best_col, best_score);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// build_consensus_snpmers (src/asv_cluster.rs:840-894) for many clusters in one launch: workgroup = (cluster, word).
// Lane = member; rows are sparse, so per 64 members only the bit positions that any of them sets are visited: one ballot
// per (bit, allele) over the wave, lane b accumulates the counts of bit b; the 4 waves are combined through LDS; lane b
// then decides site b: allele = (count1 > count0) (tie -> 0 = smaller mid base), kept iff best count >= max(1, len/6)
// (:878).  The two result words are ballots.  (A coalesced word-per-lane variant was tried: one huge cluster becomes a
// serial chain per word chunk and runs 3x slower than this member-parallel form.)
__global__ void __launch_bounds__(256) k_consensus(SeedsDev R, const u64* __restrict__ cl_off, const u32* __restrict__ members, u32 n_clusters, u32 words,
                                                   u64* __restrict__ out_p, u64* __restrict__ out_a) {
    __shared__ u32 s0[4][64], s1[4][64];
    const u32 cl = blockIdx.x, w = blockIdx.y;
    const u64 a = cl_off[cl], e = cl_off[cl + 1];
    const u32 lane = d_lane(), wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // explicitly wave-uniform
    u32 c0 = 0, c1 = 0;
    for (u64 i = a + wave * 64; i < e; i += 256) {               // wave-uniform trip count
        const u64 mi = i + lane;
        u64 p = 0, al = 0;
        if (mi < e) { const u64 row = (u64)members[mi] * words + w; p = R.p_filt[row]; if (p) al = R.allele[row]; }
        u64 anyv = p;
        #pragma unroll
        for (int s = 32; s >= 1; s >>= 1) anyv |= __shfl_xor(anyv, s);
        // make the uniformity explicit (SGPR loop counter): the ballots below must run with all lanes active
        u64 any = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(anyv >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)anyv);
        const u64 one = p & al, zero = p & ~al;
        while (any) {                                            // wave-uniform (scalar)
            const u32 b = (u32)__builtin_ctzll(any); any &= any - 1;
            const u32 n1 = __popcll(__ballot((one >> b) & 1)), n0 = __popcll(__ballot((zero >> b) & 1));
            if (lane == b) { c1 += n1; c0 += n0; }
        }
    }
    s0[wave][lane] = c0; s1[wave][lane] = c1;
    __syncthreads();
    if (wave == 0) {
        c0 = s0[0][lane] + s0[1][lane] + s0[2][lane] + s0[3][lane];
        c1 = s1[0][lane] + s1[1][lane] + s1[2][lane] + s1[3][lane];
        const u64 len = e - a;
        const u32 thr = (u32)(len / 6) > 1 ? (u32)(len / 6) : 1;
        const bool one = c1 > c0;
        const u32 best = one ? c1 : c0;
        const bool keep = best >= thr && best > 0;
        const ull pm = __ballot(keep), am = __ballot(keep && one);
        if (lane == 0) { out_p[(u64)cl * words + w] = pm; out_a[(u64)cl * words + w] = am; }
    }
}
// The same consensus from the SPARSE rows (the dense-row kernel above reads one 8-byte word per member per (cluster, word) block:
// a 64-byte sector each, ~37 KB per member; a member's sparse row is ~15 entries of 18 bytes).  A block takes up to `chunk` (256) members
// of ONE cluster and one range of <= CONS_WORDS site words, 16 lanes per member (lane = sparse entry), and counts the set bits per
// site in LDS (one u32 per site: allele-0 count | allele-1 count << 16, both <= chunk), then adds the non-zero sites to the
// cluster's global counters; a second kernel applies the rule of :878 per site.
#define CONS_WORDS 512u                                           // 512 words = 32768 sites = 128 KB of LDS counters
__global__ void __launch_bounds__(256) k_consensus_count(SeedsDev R, const u64* __restrict__ cl_off, const u32* __restrict__ members, u32 words, ull* __restrict__ g_cnt, u32 chunk) {
    extern __shared__ u32 site_cnt[];                             // [(whi - wlo) * 64]
    const u32 cl = blockIdx.x;
    const u64 a = cl_off[cl] + (u64)blockIdx.y * chunk, e = min(cl_off[cl + 1], a + chunk);
    if (a >= cl_off[cl + 1]) return;
    const u32 wlo = blockIdx.z * CONS_WORDS, whi = min(words, wlo + CONS_WORDS), nsites = (whi - wlo) * 64;
    for (u32 x = threadIdx.x; x < nsites; x += 256) site_cnt[x] = 0;
    __syncthreads();
    const u32 grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    for (u64 i = a + grp; i < e; i += 16) {
        const u32 read = members[i];
        const u64 base = R.snp_base[read]; const u32 cnt = R.nz_cnt[read];
        for (u32 t = l16; t < cnt; t += 16) {
            const u32 w = R.nz_idx[base + t];
            if (w < wlo || w >= whi) continue;
            u64 p = R.nz_pf[base + t];
            if (p == 0) continue;
            const u64 al = R.nz_a[base + t];
            while (p) { const u32 b = (u32)__builtin_ctzll(p); p &= p - 1; atomicAdd(&site_cnt[(w - wlo) * 64 + b], ((al >> b) & 1) ? 0x10000u : 1u); }
        }
    }
    __syncthreads();
    ull* g = g_cnt + ((u64)cl * words + wlo) * 64;
    for (u32 x = threadIdx.x; x < nsites; x += 256) { const u32 v = site_cnt[x]; if (v) atomicAdd(&g[x], (ull)(v & 0xFFFFu) | ((ull)(v >> 16) << 32)); }
}
__global__ void __launch_bounds__(256) k_consensus_decide(const ull* __restrict__ g_cnt, const u64* __restrict__ cl_off, u32 words, u64* __restrict__ out_p, u64* __restrict__ out_a) {
    const u32 cl = blockIdx.x, w = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave = one site word
    if (w >= words) return;
    const ull v = g_cnt[((u64)cl * words + w) * 64 + lane];
    const u32 c0 = (u32)v, c1 = (u32)(v >> 32);
    const u64 len = cl_off[cl + 1] - cl_off[cl];
    const u32 thr = (u32)(len / 6) > 1 ? (u32)(len / 6) : 1;
    const bool one = c1 > c0;
    const u32 best = one ? c1 : c0;
    const bool keep = best >= thr && best > 0;
    const ull pm = __ballot(keep), am = __ballot(keep && one);
    if (lane == 0) { out_p[(u64)cl * words + w] = pm; out_a[(u64)cl * words + w] = am; }
}
size_t consensus_counter_bytes(u32 n_clusters, u32 words) { return (size_t)n_clusters * words * 64 * 8; }
int launch_consensus(svt_ctx* c, const SeedsDev& rows, const u64* d_cl_off, const u32* d_members, u32 n_clusters, u64 n_members, u32 words, u64* d_p, u64* d_a,
                     ull* d_counters, u64 max_cluster) {
    if (n_clusters == 0 || words == 0) return SVT_OK;
    if (d_counters) {
        const size_t sh = (size_t)std::min(words, CONS_WORDS) * 256;
        DYN_LDS_ONCE(c, 5, k_consensus_count, 150 * 1024);
        ProfScope ps(c, "k_consensus", 20.0 * 16.0 * (double)n_members + 8.0 * (double)n_members + 16.0 * words * (double)n_clusters, (double)n_clusters);
        HIPCHK(c, hipMemsetAsync(d_counters, 0, (size_t)n_clusters * words * 64 * 8, c->stream));
        const int ce = c->opt().consensus_chunk;                                  // svt_set_option("consensus_chunk"): small chunks exercise the multi-block path on small clusters (tests)
        const u32 chunk = ce > 0 ? std::min(32768u, (u32)ce) : 256u;
        const u32 chunks = (u32)std::max<u64>(1, (max_cluster + chunk - 1) / chunk);
        hipLaunchKernelGGL(k_consensus_count, dim3(n_clusters, chunks, (words + CONS_WORDS - 1) / CONS_WORDS), dim3(256), sh, c->stream, rows, d_cl_off, d_members, words, d_counters, chunk);
        hipLaunchKernelGGL(k_consensus_decide, dim3(n_clusters, (words + 3) / 4), dim3(256), 0, c->stream, (const ull*)d_counters, d_cl_off, words, d_p, d_a);
        HIPCHK(c, hipGetLastError());
        return SVT_OK;
    }
    ProfScope ps(c, "k_consensus", 16.0 * words * ((double)n_clusters + (double)n_members), (double)n_clusters);
    hipLaunchKernelGGL(k_consensus, dim3(n_clusters, words), dim3(256), 0, c->stream, rows, d_cl_off, d_members, n_clusters, words, d_p, d_a);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ---- Stage 7 candidate scoring on the device (src/alignment.rs:1797-1846) ---------------------------------------------------
// K6 leaves (row, asv, mismatches) triples and K7 their shared / same-strand minimizer counts in HBM; these two passes apply the
// reference's f64 filters where the data is (IEEE division, same operand order as the host code they replace) and keep, per
// read, the ASVs that tie on the lowest mismatch count.  Only those ties (~1.5 per read) cross PCIe.
__global__ void k_pair_rows_to_reads(const u32* __restrict__ row_idx, const u32* __restrict__ o_row, u64 n, u32* __restrict__ a_idx) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a_idx[i] = row_idx[o_row[i]];
}
__global__ void k_tie_lowest(const u32* __restrict__ o_row, const u32* __restrict__ o_col, const u32* __restrict__ o_mm, const u32* __restrict__ a_idx,
                             const u32* __restrict__ shared, const u32* __restrict__ r_unique, const u32* __restrict__ a_unique, u64 n,
                             double min_frac, double cpar, u32* __restrict__ lowest, u8* __restrict__ keep, u8* __restrict__ done) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 mm = shared[i], mism = o_mm[i] & 0xFFFF;
    u8 k = 0;
    if (mm != 0) {                                                                       // :1801
        const u32 den = min(r_unique[a_idx[i]], a_unique[o_col[i]]);
        if (!((double)mm / (double)den < min_frac)) {                                    // :1805-1808
            const double ratio = (double)mism / (double)mm / cpar;                       // :1811
            if (ratio <= 0.0050) { k = 1; atomicMin(&lowest[o_row[i]], mism); if (done) done[o_row[i]] = 1; }   // :1829-1843
        }
    }
    keep[i] = k;
}
__global__ void __launch_bounds__(1024) k_tie_emit(const u32* __restrict__ o_row, const u32* __restrict__ o_col, const u32* __restrict__ o_mm, const u32* __restrict__ shared,
                           const u32* __restrict__ same, const u8* __restrict__ keep, const u32* __restrict__ lowest, u64 n,
                           u32* __restrict__ t_row, u32* __restrict__ t_col, u8* __restrict__ t_rev, u32* __restrict__ t_mm, u64 cap, ull* __restrict__ counter) {
    __shared__ u32 wave_tot[16]; __shared__ ull blk_base;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = i < n && keep[i] && (o_mm[i] & 0xFFFF) == lowest[o_row[i]];
    const ull mask = __ballot(hit);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_tot[wave] = (u32)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) { u32 all = 0; for (int w = 0; w < 16; w++) all += wave_tot[w]; blk_base = all ? atomicAdd(counter, (ull)all) : 0; }   // one atomic per 1024 pairs
    __syncthreads();
    if (hit) {
        ull o = blk_base + __popcll(mask & ((1ull << lane) - 1));
        for (u32 w = 0; w < wave; w++) o += wave_tot[w];
        if (o < cap) { t_row[o] = o_row[i]; t_col[o] = o_col[i]; t_rev[o] = (shared[i] - same[i]) > same[i] ? 1 : 0; if (t_mm) t_mm[o] = o_mm[i] & 0xFFFF; }
    }
}
// Two-phase candidate evaluation: most reads are settled by their candidates at the read's LOWEST mismatch count (a survivor
// there is by definition a lowest-mismatch survivor), so K7 runs on those first (~1.5 pairs per read instead of ~38); only reads
// without a survivor at that level have their remaining candidates evaluated.
__global__ void k_row_min_mism(const u32* __restrict__ o_row, const u32* __restrict__ o_mm, u64 n, u32* __restrict__ rmin) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMin(&rmin[o_row[i]], o_mm[i] & 0xFFFF);
}
// mode 0: candidates at their row's minimum; mode 1: the other candidates of rows that are not done.  Compacted copies of
// (row, col, mm) in list order; one append atomic per 1024 candidates.
__global__ void __launch_bounds__(1024) k_select_candidates(const u32* __restrict__ o_row, const u32* __restrict__ o_col, const u32* __restrict__ o_mm, u64 n,
                                                            const u32* __restrict__ rmin, const u8* __restrict__ done, int mode,
                                                            u32* __restrict__ s_row, u32* __restrict__ s_col, u32* __restrict__ s_mm, ull* __restrict__ counter) {
    __shared__ u32 wave_tot[16]; __shared__ ull blk_base;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (i < n) { const u32 r = o_row[i]; const bool at_min = (o_mm[i] & 0xFFFF) == rmin[r]; hit = mode == 0 ? at_min : (!at_min && !done[r]); }
    const ull mask = __ballot(hit);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_tot[wave] = (u32)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) { u32 all = 0; for (int w = 0; w < 16; w++) all += wave_tot[w]; blk_base = all ? atomicAdd(counter, (ull)all) : 0; }
    __syncthreads();
    if (hit) {
        ull o = blk_base + __popcll(mask & ((1ull << lane) - 1));
        for (u32 w = 0; w < wave; w++) o += wave_tot[w];
        s_row[o] = o_row[i]; s_col[o] = o_col[i]; s_mm[o] = o_mm[i];
    }
}
int launch_candidate_select(svt_ctx* c, const u32* o_row, const u32* o_col, const u32* o_mm, u64 n, u32* rmin, const u8* done, int mode,
                            u32* s_row, u32* s_col, u32* s_mm, ull* counter) {
    if (n == 0) return SVT_OK;
    ProfScope ps(c, "k_tie_passes", (double)n * 16.0, (double)n);
    if (mode == 0) hipLaunchKernelGGL(k_row_min_mism, dim3((u32)((n + 255) / 256)), dim3(256), 0, c->stream, o_row, o_mm, n, rmin);
    hipLaunchKernelGGL(k_select_candidates, dim3((u32)((n + 1023) / 1024)), dim3(1024), 0, c->stream, o_row, o_col, o_mm, n, rmin, done, mode, s_row, s_col, s_mm, counter);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

int launch_tie_passes(svt_ctx* c, const u32* d_row_idx, const u32* o_row, const u32* o_col, const u32* o_mm, u64 n, u32* a_idx, int phase,
                      const u32* shared, const u32* same, const u32* r_unique, const u32* a_unique, double min_frac, double cpar,
                      u32* lowest, u8* keep, u32* t_row, u32* t_col, u8* t_rev, u32* t_mm, u64 cap, ull* counter, u8* done) {
    if (n == 0) return SVT_OK;
    const u32 blocks = (u32)((n + 255) / 256);
    if (phase == 0) { ProfScope ps(c, "k_pair_rows_to_reads", (double)n * 12.0, (double)n); hipLaunchKernelGGL(k_pair_rows_to_reads, dim3(blocks), dim3(256), 0, c->stream, d_row_idx, o_row, n, a_idx); }
    else {
        ProfScope ps(c, "k_tie_passes", (double)n * 45.0, (double)n);
        hipLaunchKernelGGL(k_tie_lowest, dim3(blocks), dim3(256), 0, c->stream, o_row, o_col, o_mm, a_idx, shared, r_unique, a_unique, n, min_frac, cpar, lowest, keep, done);
        hipLaunchKernelGGL(k_tie_emit, dim3((u32)((n + 1023) / 1024)), dim3(1024), 0, c->stream, o_row, o_col, o_mm, shared, same, keep, lowest, n, t_row, t_col, t_rev, t_mm, cap, counter);
    }
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ---- Stage 2 candidate lists, brute force (src/asv_cluster.rs:303-337 query_read_against_bucket_index + the list rule :111-125) -----------------------
// The reference walks 20 hash buckets per read and counts, per representative, in how many tables the signatures agree.  The representatives of a
// block are few (hundreds to a few thousand), so the device compares every (query, reference) pair of a block directly: one wave per query, a
// reference per lane, 20 64-bit compares.  References with at least one equal signature are appended to a per-wave LDS list (ballot + rank: ascending
// reference position); mode 0 orders them (hits, position) descending and keeps the maxima or the first top_n, whichever is longer -- the list rule;
// mode 1 keeps all of them in ascending position with their counts (the pairs of a block's later reads with its potential new representatives).
// The lists go to ONE flat array (a cursor hands out room per query: offset + count); a query whose list outgrows `cap`, or that finds the array full, gets
// the count 0xFFFFFFFF: the caller builds that one on the host.
#define LSHC_CAP 256
#define LSHC_WAVES 16      // queries per workgroup: ONE atomic on the cursor per workgroup -- returning atomics on a single address cost ~20 ns each on this chip, and a block of Stage 2 has 32k queries
__global__ void __launch_bounds__(64 * LSHC_WAVES) k_lsh_candidates(const u64* __restrict__ lsh, const u8* __restrict__ lsh_valid, const u32* __restrict__ q_idx, u32 n_q,
                                                        const u32* __restrict__ r_idx, u32 n_ref, const u32* __restrict__ ref_limit, u32 mode, u32 top_n, u32 cap,
                                                        u32 capacity, u32* __restrict__ cursor, u32* __restrict__ out_cnt, u32* __restrict__ out_off, u32* __restrict__ out) {
    __shared__ u64 s_list[LSHC_WAVES][LSHC_CAP];
    __shared__ u32 s_keep[LSHC_WAVES], s_base;
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 i = blockIdx.x * LSHC_WAVES + wv;
    u64* list = s_list[wv];
    // what the wave found: 0 nothing to write (no query, no signature, no reference shares one), 1 a list of `keep` entries, 2 the list outgrew its room (the caller builds it)
    u32 state = 0, T = 0, keep = 0, max_hits = 0;
    if (i < n_q) {
        const u32 qo = q_idx[i];
        if (lsh_valid[qo]) {
            u64 qs[SVT_LSH_TABLES];
            #pragma unroll
            for (u32 t = 0; t < SVT_LSH_TABLES; t++) qs[t] = lsh[(u64)qo * SVT_LSH_TABLES + t];      // wave-uniform address: scalar loads
            const u32 lim = ref_limit ? min(ref_limit[i], n_ref) : n_ref;
            bool over = false;
            for (u32 j0 = 0; j0 < lim; j0 += 64) {
                const u32 j = j0 + lane;
                u32 hits = 0;
                if (j < lim) {
                    const u64* rs = lsh + (u64)r_idx[j] * SVT_LSH_TABLES;
                    #pragma unroll
                    for (u32 t = 0; t < SVT_LSH_TABLES; t++) hits += (rs[t] == qs[t]) ? 1u : 0u;
                }
                const u64 bal = __ballot(hits > 0);
                if (bal) {
                    const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0));
                    const u32 cnt = (u32)__popcll(bal);
                    if (T + cnt > cap || T + cnt > LSHC_CAP) { over = true; break; }
                    if (hits > 0) list[T + before] = ((u64)hits << 32) | j;
                    T += cnt;
                    u32 mh = hits;
                    #pragma unroll
                    for (int s = 32; s >= 1; s >>= 1) mh = max(mh, (u32)__shfl_xor((int)mh, s));
                    max_hits = max(max_hits, mh);
                }
            }
            if (over) state = 2;
            else if (T) {
                state = 1;
                // how many entries stay: all (mode 1), or the maxima / the first top_n, whichever is longer (mode 0: the maxima form a prefix of the order)
                keep = T;
                if (mode == 0) {
                    u32 M = 0;
                    for (u32 e0 = 0; e0 < T; e0 += 64) { const u32 e = e0 + lane; M += (u32)__popcll(__ballot(e < T && (u32)(list[min(e, T - 1)] >> 32) == max_hits)); }
                    keep = min(T, max(M, top_n));
                }
            }
        }
    }
    if (lane == 0) s_keep[wv] = state == 1 ? keep : 0;
    __syncthreads();
    if (threadIdx.x == 0) { u32 all = 0; for (u32 x = 0; x < LSHC_WAVES; x++) all += s_keep[x]; s_base = all ? atomicAdd(cursor, all) : 0; }
    __syncthreads();
    if (i >= n_q) return;
    u32 off = s_base;
    for (u32 x = 0; x < wv; x++) off += s_keep[x];
    if (state == 1 && off + keep > capacity) state = 2;                           // the flat output is full: the caller builds this list itself
    if (state != 1) { if (lane == 0) { out_cnt[i] = state == 2 ? 0xFFFFFFFFu : 0; out_off[i] = 0; } return; }
    u32* o = out + (u64)off * 2;
    if (mode == 1) {                                                            // ascending position: {position, hits}
        for (u32 e = lane; e < T; e += 64) { const u64 v = list[e]; o[2 * e] = (u32)v; o[2 * e + 1] = (u32)(v >> 32); }
    } else {                                                                    // rank = number of entries that precede in (hits, position) descending order: {hits, position}
        for (u32 e = lane; e < T; e += 64) {
            const u64 v = list[e];
            u32 rank = 0;
            for (u32 x = 0; x < T; x++) rank += (list[x] > v) ? 1u : 0u;
            if (rank < keep) { o[2 * rank] = (u32)(v >> 32); o[2 * rank + 1] = (u32)v; }
        }
    }
    if (lane == 0) { out_cnt[i] = keep; out_off[i] = off; }
}
int launch_lsh_candidates(svt_ctx* c, const svt_batch* B, const u32* d_q, u32 n_q, const u32* d_r, u32 n_ref, const u32* d_lim, u32 mode, u32 top_n, u32 cap, u32 capacity, u32* d_cursor,
                          u32* d_cnt, u32* d_off, u32* d_out) {
    if (n_q == 0) return SVT_OK;
    ProfScope ps(c, "k_lsh_candidates", (double)n_q * 160.0 + (double)n_ref * 160.0 + (double)n_q * 8.0, (double)n_q * (double)n_ref);
    HIPCHK(c, hipMemsetAsync(d_cursor, 0, 4, c->stream));
    hipLaunchKernelGGL(k_lsh_candidates, dim3((n_q + LSHC_WAVES - 1) / LSHC_WAVES), dim3(64 * LSHC_WAVES), 0, c->stream, B->seeds.lsh, B->seeds.lsh_valid, d_q, n_q, d_r, n_ref, d_lim, mode, top_n, cap, capacity, d_cursor, d_cnt, d_off, d_out);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
