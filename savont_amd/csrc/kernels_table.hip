// kernels_table.hip -- Stage 1a/1b on the device side of the count table: canonical order of the kept k-mers and the two short
// lists Stage 1b (kmer_comp::get_snpmers_inplace_sort, src/kmer_comp.rs:454-642) actually needs from it.
//
//   * order: (masked k-mer, mid base), the sort key of src/kmer_comp.rs:480.  Keys are unique, so a radix sort over
//     key' = masked << 2 | mid (2k + 2 bits) gives the order of the reference's comparison sort.  The sort and the two ordered
//     selections are rocPRIM device primitives (plain library sort / select; no hand-written kernel would do them differently);
//     the key build, gather and flag kernels around them are below.
//   * candidates: entries whose masked k-mer is shared with a neighbour (groups of >= 2 alleles, :507-519) -- the only entries the
//     binomial / Fisher tests of :543-623 look at;
//   * heavy: entries with total count > 100.  thresh = max(q-th largest total, 100) with q = n / 100000 + 1 (:474) and the
//     high-frequency k-mers (total > thresh, :494-496) are functions of this list alone.
// The full sorted table stays in HBM (svt_count_fetch copies it out on demand: tests, multi-GPU merge).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "svt_internal.hpp"

namespace {
__global__ void k_table_keys(const u64* __restrict__ km, u64 n, u32 k, u64* __restrict__ key, u32* __restrict__ idx) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 sm = 3ull << (k - 1);
    const u64 v = km[i];
    key[i] = ((v & ~sm) << 2) | ((v & sm) >> (k - 1));
    idx[i] = (u32)i;
}
__global__ void k_table_gather(const u32* __restrict__ idx, u64 n, const u64* __restrict__ km, const u32* __restrict__ rv, const u32* __restrict__ fw,
                               u64* __restrict__ okm, u32* __restrict__ orv, u32* __restrict__ ofw) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 s = idx[i];
    okm[i] = km[s]; orv[i] = rv[s]; ofw[i] = fw[s];
}
// flags over the SORTED table: bit 0 = in a group of >= 2 alleles, bit 1 = total > 100
__global__ void k_table_flags(const u64* __restrict__ km, const u32* __restrict__ rv, const u32* __restrict__ fw, u64 n, u32 k, u8* __restrict__ fl_grp, u8* __restrict__ fl_heavy) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 sm = 3ull << (k - 1);
    const u64 m = km[i] & ~sm;
    const bool g = (i > 0 && (km[i - 1] & ~sm) == m) || (i + 1 < n && (km[i + 1] & ~sm) == m);
    fl_grp[i] = g ? 1 : 0;
    fl_heavy[i] = ((u64)rv[i] + fw[i] > 100) ? 1 : 0;
}
// ---- Stage 1c on the device: which reads become twin reads and in which order (src/kmer_comp.rs:117,185,248, src/main.rs:538) ----
// flag = the read passes the length window, produced seeds (status 0), keeps enough solid minimizers (n_solid >= len / c / 20, integer divisions) and its
// estimated identity is not below the cut-off (reads without an estimate pass); key = the identity as a DESCENDING radix key (100.0 stands in for "no estimate")
__device__ __forceinline__ u64 desc_key_of(double e) {
    u64 b = (u64)__double_as_longlong(e);
    b = (b >> 63) ? ~b : (b | (1ull << 63));            // ascending order of the doubles
    return ~b;                                          // descending
}
__global__ void k_twin_flags(const u64* __restrict__ off, const u8* __restrict__ status, const u32* __restrict__ n_solid, const double* __restrict__ est, const u8* __restrict__ est_valid,
                             u32 n, u32 min_len, u32 max_len, u32 cpar, double cutoff, u8* __restrict__ flag, u64* __restrict__ key_all) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 len = off[i + 1] - off[i];
    const bool ev = est_valid[i] != 0; const double e = est[i];
    const bool pass = len >= min_len && len <= max_len && status[i] == 0 && (u64)n_solid[i] >= len / cpar / 20 && (!ev || e >= cutoff);
    flag[i] = pass ? 1 : 0;
    key_all[i] = desc_key_of(ev ? e : 100.0);
}
__global__ void k_twin_keys(const u32* __restrict__ idx, u32 n, const u64* __restrict__ key_all, u64* __restrict__ key) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) key[t] = key_all[idx[t]];
}
// the per-twin records in twin order: one thread per twin read
__global__ void k_twin_gather(const u32* __restrict__ order, u32 n, const u64* __restrict__ off, SeedsDev sd,
                              u32* __restrict__ o_len, u32* __restrict__ o_nmini, u32* __restrict__ o_nuniq, u32* __restrict__ o_nsnpf, double* __restrict__ o_est, u8* __restrict__ o_ev, u8* __restrict__ o_lv) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 i = order[t];
    o_len[t] = (u32)(off[i + 1] - off[i]); o_nmini[t] = sd.mini_cnt[i]; o_nuniq[t] = sd.set_cnt[i];
    u32 f = 0; const u64 sb = sd.snp_base[i]; const u32 sc = sd.snp_cnt[i];
    for (u32 x = 0; x < sc; x++) f += sd.snp_flags[sb + x] & 1;                     // SNPmers that are not high-frequency (src/kmer_comp.rs:190-202)
    o_nsnpf[t] = f; o_est[t] = sd.est_id[i]; o_ev[t] = sd.est_valid[i]; o_lv[t] = sd.lsh_valid[i];
}
__global__ void k_twin_gather_lsh(const u32* __restrict__ order, u32 n, const u64* __restrict__ lsh, u64* __restrict__ o_lsh) {
    const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;                      // one thread per signature: 20 consecutive threads read one read's 160 bytes
    if (x >= (u64)n * SVT_LSH_TABLES) return;
    const u32 t = (u32)(x / SVT_LSH_TABLES), s = (u32)(x % SVT_LSH_TABLES);
    o_lsh[x] = lsh[(u64)order[t] * SVT_LSH_TABLES + s];
}
}  // namespace

// Stage 1c order on the device.  d_idx_a / d_idx_b / d_key_a / d_key_b: n entries each; d_flag: n bytes; d_count: one u32.  After the call d_idx_b[0 .. *count) is the
// twin order (identity descending, input order among equal identities: the radix sort is stable) and d_key_b the keys in that order.
int launch_twin_order(svt_ctx* c, const svt_batch* b, u32 min_len, u32 max_len, u32 cpar, double cutoff, u8* d_flag, u64* d_key_all, u32* d_idx_a, u32* d_idx_b, u64* d_key_a, u64* d_key_b,
                      u32* d_count, u32 n_kept_known, void* temp, size_t temp_bytes, size_t* need_bytes, int phase) {
    const u32 n = b->n;
    size_t need1 = 0, need2 = 0;
    rocprim::counting_iterator<u32> ids(0);
    if (rocprim::select((void*)nullptr, need1, ids, d_flag, d_idx_a, d_count, (size_t)n, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "select (size query) failed");
    if (rocprim::radix_sort_pairs((void*)nullptr, need2, d_key_a, d_key_b, d_idx_a, d_idx_b, (size_t)n, 0u, 64u, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "radix_sort_pairs (size query) failed");
    if (need_bytes) *need_bytes = std::max(need1, need2);
    if (!temp) return SVT_OK;
    if (temp_bytes < std::max(need1, need2)) return svt_fail(c, SVT_ERR_STATE, "launch_twin_order: temporary storage too small");
    if (n == 0) return SVT_OK;
    const SeedsDev& s = b->seeds;
    if (phase == 0) {                                                              // flags + ordered selection of the passing reads (ascending read index)
        ProfScope ps(c, "k_twin_order", (double)n * 40.0, (double)n);
        hipLaunchKernelGGL(k_twin_flags, dim3((n + 255) / 256), dim3(256), 0, c->stream, b->d_off, s.status, s.n_solid, s.est_id, s.est_valid, n, min_len, max_len, cpar, cutoff, d_flag, d_key_all);
        if (rocprim::select(temp, need1, ids, d_flag, d_idx_a, d_count, (size_t)n, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "select failed");
    } else if (n_kept_known) {                                                     // keys of the kept reads, stable sort
        ProfScope ps(c, "k_twin_order", (double)n_kept_known * (12.0 + 8.0 * 24.0), (double)n_kept_known);
        hipLaunchKernelGGL(k_twin_keys, dim3((n_kept_known + 255) / 256), dim3(256), 0, c->stream, d_idx_a, n_kept_known, d_key_all, d_key_a);
        if (rocprim::radix_sort_pairs(temp, need2, d_key_a, d_key_b, d_idx_a, d_idx_b, (size_t)n_kept_known, 0u, 64u, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "radix_sort_pairs failed");
    }
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
int launch_twin_gather(svt_ctx* c, const svt_batch* b, const u32* d_order, u32 n, u32* o_len, u32* o_nmini, u32* o_nuniq, u32* o_nsnpf, double* o_est, u8* o_ev, u8* o_lv, u64* o_lsh) {
    if (n == 0) return SVT_OK;
    ProfScope ps(c, "k_twin_gather", (double)n * (29.0 + 2.0 * 160.0), (double)n);
    hipLaunchKernelGGL(k_twin_gather, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_order, n, b->d_off, b->seeds, o_len, o_nmini, o_nuniq, o_nsnpf, o_est, o_ev, o_lv);
    if (o_lsh) { const u64 tot = (u64)n * SVT_LSH_TABLES; hipLaunchKernelGGL(k_twin_gather_lsh, dim3((u32)((tot + 255) / 256)), dim3(256), 0, c->stream, d_order, n, b->seeds.lsh, o_lsh); }
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// sorts the `n` compacted entries (km, rv, fw) into (okm, orv, ofw); key / idx buffers are caller scratch of n entries each (x2 for the sort's ping-pong)
int launch_table_sort(svt_ctx* c, u32 k, u64 n, const u64* km, const u32* rv, const u32* fw, u64* okm, u32* orv, u32* ofw,
                      u64* key_a, u64* key_b, u32* idx_a, u32* idx_b, void* temp, size_t temp_bytes, size_t* need_bytes) {
    const unsigned bits = 2 * k + 2;
    size_t need = 0;
    if (rocprim::radix_sort_pairs((void*)nullptr, need, key_a, key_b, idx_a, idx_b, (size_t)n, 0u, bits, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "radix_sort_pairs (size query) failed");
    if (need_bytes) *need_bytes = need;
    if (!temp) return SVT_OK;
    if (temp_bytes < need) return svt_fail(c, SVT_ERR_STATE, "launch_table_sort: temporary storage too small");
    if (n == 0) return SVT_OK;
    ProfScope ps(c, "k_table_sort", (double)n * (16.0 + 5.0 * 24.0 + 32.0), (double)n);
    const u32 blocks = (u32)((n + 255) / 256);
    hipLaunchKernelGGL(k_table_keys, dim3(blocks), dim3(256), 0, c->stream, km, n, k, key_a, idx_a);
    if (rocprim::radix_sort_pairs(temp, need, key_a, key_b, idx_a, idx_b, (size_t)n, 0u, bits, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "radix_sort_pairs failed");
    hipLaunchKernelGGL(k_table_gather, dim3(blocks), dim3(256), 0, c->stream, idx_b, n, km, rv, fw, okm, orv, ofw);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ordered index lists of the two selections over the sorted table; d_counts[0] = #candidates, d_counts[1] = #heavy
int launch_table_select(svt_ctx* c, u32 k, u64 n, const u64* km, const u32* rv, const u32* fw, u8* fl_grp, u8* fl_heavy, u32* out_grp, u32* out_heavy, u32* d_counts,
                        void* temp, size_t temp_bytes, size_t* need_bytes) {
    size_t need = 0;
    rocprim::counting_iterator<u32> ids(0);
    if (rocprim::select((void*)nullptr, need, ids, fl_grp, out_grp, d_counts, (size_t)n, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "select (size query) failed");
    if (need_bytes) *need_bytes = need;
    if (!temp) return SVT_OK;
    if (temp_bytes < need) return svt_fail(c, SVT_ERR_STATE, "launch_table_select: temporary storage too small");
    if (n == 0) { HIPCHK(c, hipMemsetAsync(d_counts, 0, 8, c->stream)); return SVT_OK; }
    ProfScope ps(c, "k_table_select", (double)n * (16.0 + 2.0 + 2.0 * 5.0), (double)n);
    hipLaunchKernelGGL(k_table_flags, dim3((u32)((n + 255) / 256)), dim3(256), 0, c->stream, km, rv, fw, n, k, fl_grp, fl_heavy);
    if (rocprim::select(temp, need, ids, fl_grp, out_grp, d_counts, (size_t)n, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "select failed");
    if (rocprim::select(temp, need, ids, fl_heavy, out_heavy, d_counts + 1, (size_t)n, c->stream) != hipSuccess) return svt_fail(c, SVT_ERR_HIP, "select failed");
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

int launch_table_gather(svt_ctx* c, const u32* idx, u64 n, const u64* km, const u32* rv, const u32* fw, u64* okm, u32* orv, u32* ofw) {
    if (n == 0) return SVT_OK;
    hipLaunchKernelGGL(k_table_gather, dim3((u32)((n + 255) / 256)), dim3(256), 0, c->stream, idx, n, km, rv, fw, okm, orv, ofw);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
