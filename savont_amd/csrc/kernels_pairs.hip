// kernels_pairs.hip -- K5/K7 (distinct-minimizer set intersections) and K6 (SNPmer bitset tiles).
//
// Reference semantics:
//   K5  src/asv_cluster.rs:131-143   count = |set(read minimizers) ∩ rep minimizer list|
//   K7  src/alignment.rs:1798-1799   |set(read) ∩ set(ASV)|
//   K6  src/asv_cluster.rs:356-383   (matches, mismatches) over shared split k-mers;
//       :481-496 compatible filter; :1057-1097 best cluster per read.
//
// K6 is bitset algebra: matches = popc(Pr & Pc & ~(Ar ^ Ac)), mismatches = popc(Pr & Pc & (Ar ^ Ac)).
// Column operands are gathered once per call into a TRANSPOSED [word][column] matrix so a
// wavefront's 64 columns are one coalesced 512-byte row per word; row operands are wave-uniform
// and come through the scalar cache.
#include "svt_internal.hpp"

// ------------------------------------------------------------------------------------------------
// K5 / K7: one wavefront per pair; B's set is staged in LDS, A's elements binary-search it
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_set_intersect(SeedsDev A, SeedsDev B, const u32* __restrict__ ai, const u32* __restrict__ bi, u64 n,
                                                      u32* __restrict__ shared_out, u32* __restrict__ same_out, u32 cap_lds) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64* sb = (u64*)smem;
    const u64 pid = blockIdx.x;
    if (pid >= n) return;
    const u32 lane = threadIdx.x;
    const u32 a = ai[pid], b = bi[pid];
    const u32 na = A.set_cnt[a], nb = B.set_cnt[b];
    const u64* pa = A.set_kmer + A.mini_base[a];
    const u64* pb = B.set_kmer + B.mini_base[b];
    const u32 nbl = nb < cap_lds ? nb : cap_lds;
    for (u32 i = lane; i < nbl; i += 64) sb[i] = pb[i];
    __syncthreads();
    u32 sh = 0, sm = 0;
    const u64 KM = ~(1ull << 63);
    for (u32 i = lane; i < na; i += 64) {
        u64 va = pa[i], ka = va & KM;
        int lo = 0, hi = (int)nbl - 1;
        while (lo <= hi) {
            int mid = (lo + hi) >> 1;
            u64 vb = sb[mid], kb = vb & KM;
            if (kb == ka) { sh++; sm += ((va >> 63) == (vb >> 63)); break; }
            if (kb < ka) lo = mid + 1; else hi = mid - 1;
        }
    }
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { sh += __shfl_xor(sh, s); sm += __shfl_xor(sm, s); }
    if (lane == 0) { shared_out[pid] = sh; if (same_out) same_out[pid] = sm; }
}

int launch_set_intersect(svt_ctx* c, const svt_batch* A, const svt_batch* B, const u32* d_a, const u32* d_b, u64 n, u32* d_shared, u32* d_same) {
    if (n == 0) return SVT_OK;
    u32 cap = B->max_len / 2 + 8;                        // >= any set size (spacing >= 1 is handled by mini caps <= len)
    if (cap < 512) cap = 512;
    u32 maxcap = B->max_len + 8; if (cap < maxcap && (size_t)maxcap * 8 <= 64 * 1024) cap = maxcap;
    ProfScope ps(c, "k_set_intersect", (double)n * (6.0 * 270.0 + 4.0), (double)n);   // ~1.6 KB/pair
    hipLaunchKernelGGL(k_set_intersect, dim3((u32)n), dim3(64), (size_t)cap * 8, c->stream, A->seeds, B->seeds, d_a, d_b, n, d_shared, d_same, cap);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------------
// K6
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_rows(const u64* __restrict__ srcP, const u64* __restrict__ srcA, const u32* __restrict__ idx, u32 n, u32 words,
                              u64* __restrict__ dstP, u64* __restrict__ dstA, int transpose) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)n * words) return;
    u32 row, w;
    if (transpose) { w = (u32)(t / n); row = (u32)(t % n); } else { row = (u32)(t / words); w = (u32)(t % words); }
    u64 src = (u64)(idx ? idx[row] : row) * words + w;
    dstP[t] = srcP[src]; dstA[t] = srcA[src];
}
int launch_gather_rows(svt_ctx* c, const u64* srcP, const u64* srcA, const u32* d_idx, u32 n, u32 words, u64* dstP, u64* dstA, bool transpose) {
    u64 tot = (u64)n * words;
    if (tot == 0) return SVT_OK;
    ProfScope ps(c, "k_gather_rows", 32.0 * (double)tot, (double)n);
    hipLaunchKernelGGL(k_gather_rows, dim3((u32)((tot + 255) / 256)), dim3(256), 0, c->stream, srcP, srcA, d_idx, n, words, dstP, dstA, transpose ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

#define ROWS_PER_BLOCK 8
__global__ void __launch_bounds__(256) k_compat_lists(const u64* __restrict__ rowP, const u64* __restrict__ rowA, u32 n_rows,
                                                      const u64* __restrict__ colPT, const u64* __restrict__ colAT, u32 n_cols, u32 words,
                                                      int filter, int triangular, u32 tri_base,
                                                      u32* __restrict__ o_row, u32* __restrict__ o_col, u32* __restrict__ o_mm, u64 cap, ull* __restrict__ counter) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;       // column
    const u32 r0 = blockIdx.y * ROWS_PER_BLOCK;
    u32 m[ROWS_PER_BLOCK], x[ROWS_PER_BLOCK];
    #pragma unroll
    for (int r = 0; r < ROWS_PER_BLOCK; r++) { m[r] = 0; x[r] = 0; }
    const bool jv = j < n_cols;
    for (u32 w = 0; w < words; w++) {
        u64 cp = jv ? colPT[(u64)w * n_cols + j] : 0, ca = jv ? colAT[(u64)w * n_cols + j] : 0;
        #pragma unroll
        for (int r = 0; r < ROWS_PER_BLOCK; r++) {
            u32 ri = r0 + r;
            if (ri < n_rows) {                                   // wave-uniform
                u64 rp = rowP[(u64)ri * words + w], ra = rowA[(u64)ri * words + w];
                u64 both = rp & cp, d = ra ^ ca;
                m[r] += __popcll(both & ~d); x[r] += __popcll(both & d);
            }
        }
    }
    #pragma unroll
    for (int r = 0; r < ROWS_PER_BLOCK; r++) {
        u32 ri = r0 + r;
        if (ri >= n_rows) break;
        bool keep = jv && (filter == SVT_LIST_COMPATIBLE ? (x[r] == 0 && m[r] > 0) : (m[r] + x[r] > 0));
        if (triangular && j >= tri_base) keep = keep && (j - tri_base < ri);     // in-block columns: only EARLIER rows
        ull mk = __ballot(keep);
        ull pos = 0;
        if (mk) {
            if (d_lane() == (u32)(__ffsll((long long)mk) - 1)) pos = atomicAdd(counter, (ull)__popcll(mk));
            pos = __shfl(pos, __ffsll((long long)mk) - 1);
            if (keep) {
                u64 d = pos + d_rank(mk);
                if (d < cap) { o_row[d] = ri; o_col[d] = j; o_mm[d] = (m[r] << 16) | (x[r] & 0xFFFF); }
            }
        }
    }
}
int launch_compat_lists(svt_ctx* c, const u64* rowP, const u64* rowA, u32 n_rows, const u64* colPT, const u64* colAT, u32 n_cols, u32 words,
                        int filter, int triangular, u32 tri_base, u32* o_row, u32* o_col, u32* o_mm, u64 cap, ull* d_counter) {
    if (n_rows == 0 || n_cols == 0) return SVT_OK;
    // T x T tile bytes = 2*T*ceil(M/4) + 4*T^2 ; here rows x cols
    double bytes = 16.0 * words * ((double)n_rows + (double)n_cols) + 4.0 * (double)n_rows * (double)n_cols;
    ProfScope ps(c, "k_compat_lists", bytes, (double)n_rows * (double)n_cols);
    dim3 grid((n_cols + 255) / 256, (n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL(k_compat_lists, grid, dim3(256), 0, c->stream, rowP, rowA, n_rows, colPT, colAT, n_cols, words, filter, triangular, tri_base,
                       o_row, o_col, o_mm, cap, d_counter);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// per row: first column with the smallest (mismatches, -matches)   (asv_cluster.rs:1058-1097; the initial
// best (usize::MAX, 0) loses to column 0, so "first column wins ties" is the whole rule)
__global__ void __launch_bounds__(256) k_best_column(const u64* __restrict__ rowP, const u64* __restrict__ rowA, u32 n_rows,
                                                     const u64* __restrict__ colP, const u64* __restrict__ colA, u32 n_cols, u32 words,
                                                     u32* __restrict__ best_col, u32* __restrict__ best_score) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    u32 bc = 0, bm = 0, bx = 0xFFFFFFFFu;
    for (u32 cidx = 0; cidx < n_cols; cidx++) {
        u32 m = 0, x = 0;
        for (u32 w = 0; w < words; w++) {
            u64 rp = rowP[(u64)i * words + w], ra = rowA[(u64)i * words + w];
            u64 cp = colP[(u64)cidx * words + w], ca = colA[(u64)cidx * words + w];   // wave-uniform
            u64 both = rp & cp, d = ra ^ ca;
            m += __popcll(both & ~d); x += __popcll(both & d);
        }
        if (x < bx || (x == bx && m > bm)) { bx = x; bm = m; bc = cidx; }
    }
    best_col[i] = bc; best_score[i] = (bm << 16) | (bx & 0xFFFF);
}
int launch_best_column(svt_ctx* c, const u64* rowP, const u64* rowA, u32 n_rows, const u64* colP, const u64* colA, u32 n_cols, u32 words, u32* best_col, u32* best_score) {
    if (n_rows == 0) return SVT_OK;
    double bytes = 16.0 * words * ((double)n_rows + (double)n_cols) + 8.0 * (double)n_rows;
    ProfScope ps(c, "k_best_column", bytes, (double)n_rows * (double)n_cols);
    hipLaunchKernelGGL(k_best_column, dim3((n_rows + 255) / 256), dim3(256), 0, c->stream, rowP, rowA, n_rows, colP, colA, n_cols, words, best_col, best_score);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
