// Semantics of the packed 16-bit instructions K8a's packed cell relies on (gfx950): signed saturation of v_pk_add_i16 / v_pk_mad_i16 with `clamp`, an SGPR as the
// second operand of a VOP3P instruction, v_pk_max_i16 / v_pk_min_i16 on both halves.  Prints PASS / FAIL per check.
// build + run (GPU box): /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/pk16_check.hip -o /tmp/pk16_check && /tmp/pk16_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t pk_add_c(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_add_cs(uint32_t a, uint32_t s) { uint32_t r; asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(s)); return r; }
__device__ __forceinline__ uint32_t pk_mad_c(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t pk_mad_cs(uint32_t a, uint32_t s, uint32_t c) { uint32_t r; asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "s"(s), "v"(c)); return r; }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_max_s(uint32_t a, uint32_t s) { uint32_t r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__global__ void k(const uint32_t* in, uint32_t* out, uint32_t sconst) {
    const uint32_t a = in[0], b = in[1], c = in[2], w = in[3];
    uint32_t s = __builtin_amdgcn_readfirstlane(sconst);
    out[0] = pk_add_c(a, b); out[1] = pk_add_cs(a, s); out[2] = pk_mad_c(w, b, a); out[3] = pk_mad_cs(w, s, a);
    out[4] = pk_max(a, c); out[5] = pk_min(a, c); out[6] = pk_max_s(a, s);
}
static uint32_t P(int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }
static int sat(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }
int main() {
    uint32_t h_in[4] = {P(-32700, -100), P(-769, -769), P(-5, -32768), P(1, 0)}, h_out[7];
    uint32_t *d_in, *d_out; hipMalloc(&d_in, 16); hipMalloc(&d_out, 28);
    hipMemcpy(d_in, h_in, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, P(-769, -769));
    hipMemcpy(h_out, d_out, 28, hipMemcpyDeviceToHost);
    const uint32_t want[7] = {P(sat(-32700 - 769), sat(-100 - 769)), P(sat(-32700 - 769), sat(-100 - 769)), P(sat(-32700 + 1 * -769), sat(-100 + 0)), P(sat(-32700 - 769), -100),
                              P(-5, -100), P(-32700, -32768), P(-769, -100)};
    const char* name[7] = {"v_pk_add_i16 clamp", "v_pk_add_i16 clamp (sgpr)", "v_pk_mad_i16 clamp", "v_pk_mad_i16 clamp (sgpr multiplier)", "v_pk_max_i16", "v_pk_min_i16", "v_pk_max_i16 (sgpr)"};
    int bad = 0;
    for (int i = 0; i < 7; i++) { const bool ok = h_out[i] == want[i]; bad += !ok; printf("%-40s %s  got %08x want %08x\n", name[i], ok ? "PASS" : "FAIL", h_out[i], want[i]); }
    return bad;
}
