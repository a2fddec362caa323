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
}  // namespace

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
