// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/lone_wave.hip -o build_tmp/lone_wave   (run on a GPU box: gpurun -- ./build_tmp/lone_wave)
// How fast does ONE wave per CU run?  Dependent VALU chain, dependent LDS round trips and a DPP scan chain, 105 single-wave blocks
// (the shape of K11: one wave per partial-order graph).  Prints cycles per operation from s_memtime and wall-clock nanoseconds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_valu(int n, int* out, long long* cyc) {
    int v = threadIdx.x; long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) { v = v * 3 + i; v ^= v >> 3; v += 7; v = max(v, i); }   // 5 dependent ops (mul-add counts 1-2)
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = v; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_lds(int n, int* out, long long* cyc) {
    __shared__ int a[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) a[i] = (i * 7 + 1) & 4095;
    __syncthreads();
    int v = threadIdx.x; long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) v = a[v];                                                 // dependent LDS round trip
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = v; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dpp(int n, int* out, long long* cyc) {
    int v = threadIdx.x; long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) { v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false)) + 1; }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = v; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 105, n = 200000;
    int* out; long long* cyc; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int which = 0; which < 3; which++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (which == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64), 0, 0, n, out, cyc);
        if (which == 1) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), 0, 0, n, out, cyc);
        if (which == 2) hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(64), 0, 0, n, out, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        if (rep) printf("%s: %d blocks x 1 wave, %d iterations: %.3f ms = %.1f ns/iter, counter ticks/iter %.1f\n", which == 0 ? "valu(5 dep ops)" : which == 1 ? "lds round trip" : "dpp+max+add", blocks, n, ms, ms * 1e6 / n, (double)c0 / n);
    }
    return 0;
}
