// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o build_tmp/valu_rates   (run on a GPU box: gpurun -- ./build_tmp/valu_rates)
// Issue rate of the integer VALU instructions the bit-parallel aligners are made of, gfx950: 8 independent chains per lane, 8 waves per
// SIMD, every CU busy.  Prints wave-instructions per SIMD-cycle-pair relative to v_and_b32 (a full-rate op: 2 cycles per wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
template <int OP> __global__ void __launch_bounds__(256) k(int n, unsigned* out) {
    unsigned v[CHAINS], a = threadIdx.x * 2654435761u, b = blockIdx.x + 12345u;
    #pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = a + c * 977u;
    for (int i = 0; i < n; i++) {
        #pragma unroll
        for (int u = 0; u < 8; u++)                        // 64 VALU instructions per loop trip: the loop overhead (2 SALU + branch) stays below 5 %
        #pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            // every variant is two instructions that the compiler cannot fold across trips (an op followed by an add of a trip-dependent value)
            const unsigned y = b + c + u;
            if (OP == 0) v[c] = (v[c] & a) + y;                                        // v_and_b32 + v_add_u32
            if (OP == 1) v[c] = (v[c] ^ a) + y;                                        // v_xor_b32 + v_add_u32
            if (OP == 2) v[c] = __builtin_amdgcn_alignbit(v[c], a, 1) + y;            // v_alignbit_b32 + v_add_u32
            if (OP == 3) v[c] = ((v[c] & a) | (~v[c] & y)) + y;                        // 3-input logic (v_bfi / v_bitop3) + v_add_u32
            if (OP == 4) v[c] = __popc(v[c]) + y;                                      // v_bcnt_u32_b32 (adds its second operand): ONE instruction
            if (OP == 5) v[c] = ((v[c] << 1) | a) + y;                                 // v_lshl_or_b32 + v_add_u32
            if (OP == 6) v[c] = (v[c] + a) + y;                                        // v_add3_u32: ONE instruction
            if (OP == 7) v[c] = v[c] * 3u + y;                                         // v_mad_u32_u24 / v_mul_lo + add
        }
    }
    unsigned s = 0;
    #pragma unroll
    for (int c = 0; c < CHAINS; c++) s ^= v[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> double run(int n, unsigned* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, 16, out);
    hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, n, out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winstr = (double)n * 8 * CHAINS * (256.0 * 8 * 4);   // wave-instructions
    return winstr / (ms * 1e-3) / 1e12;                           // T wave-instr/s
}
int main() {
    unsigned* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int n = 4000;
    const char* names[] = {"v_and + v_add (2)", "v_xor + v_add (2)", "v_alignbit + v_add (2)", "bitop3/bfi + v_add (2)", "v_bcnt (1)", "v_lshl_or + v_add (2)", "v_add3 (1)", "mul*3+add (1-2)"};
    const int per[] = {2, 2, 2, 2, 1, 2, 1, 1};
    double r[8] = {run<0>(n, out), run<1>(n, out), run<2>(n, out), run<3>(n, out), run<4>(n, out), run<5>(n, out), run<6>(n, out), run<7>(n, out)};
    for (int i = 0; i < 8; i++) printf("%-28s %.3f T wave-instr/s = %.1f T lane-ops/s = %.2f SIMD cycles per wave64 instruction at 2.4 GHz\n", names[i], r[i] * per[i], r[i] * per[i] * 64, 1024 * 2.4e9 / (r[i] * per[i] * 1e12));
    return 0;
}
