// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_copy.hip -o build_tmp/hbm_copy   (run on a GPU box: gpurun -- ./build_tmp/hbm_copy)
// HBM streaming-copy rate on this box for a few launch shapes (the denominator next to the 8 TB/s datasheet figure; MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int U> __global__ void __launch_bounds__(256) kc(const v4u* __restrict__ s, v4u* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4u v[U];
        #pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(&s[i + u * stride]);
        #pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], &d[i + u * stride]);
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int U> void run(const v4u* a, v4u* b, size_t n, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; it++) { hipEventRecord(e0); hipLaunchKernelGGL(kc<U>, dim3(blocks), dim3(256), 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
    printf("unroll %d blocks %6d: %.3f ms  %.0f GB/s\n", U, blocks, best, 2.0 * n * 16 / 1e9 / (best * 1e-3));
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    v4u *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) { run<1>(a, b, n, blocks); run<2>(a, b, n, blocks); run<4>(a, b, n, blocks); run<8>(a, b, n, blocks); }
    return 0;
}
