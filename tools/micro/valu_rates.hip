// Issue rate of the VALU instructions the aligners are made of, gfx950, one instruction kind per kernel, written in inline assembly so that the
// loop body IS what runs: 64 instructions per trip (8 independent register chains x 8 rounds, all operands VGPRs) + 3 scalar instructions,
// 8 waves per SIMD, every CU busy.  Prints wave-instructions per second per instruction kind and the SIMD cycles one wave64 instruction takes
// (shader clock read from the device).  `make check` below disassembles the kernels: every k<OP> must show 64 of its instruction per trip.
// build + run (GPU box): /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// check the ISA (here):   /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only tools/micro/valu_rates.hip -o - | grep -c v_alignbit_b32
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BODY8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I)
// one instruction per chain register %c; %8 and %9 are two more VGPRs
#define I_AND(c)      "v_and_b32 %" #c ", %" #c ", %8\n"
#define I_XOR(c)      "v_xor_b32 %" #c ", %" #c ", %8\n"
#define I_ADD(c)      "v_add_u32 %" #c ", %" #c ", %8\n"
#define I_MAX(c)      "v_max_i32 %" #c ", %" #c ", %8\n"
#define I_LSHL(c)     "v_lshlrev_b32 %" #c ", 1, %" #c "\n"
#define I_ALIGNBIT(c) "v_alignbit_b32 %" #c ", %" #c ", %8, %9\n"
#define I_BFI(c)      "v_bfi_b32 %" #c ", %" #c ", %8, %9\n"
#define I_ANDOR(c)    "v_and_or_b32 %" #c ", %" #c ", %8, %9\n"
#define I_LSHLOR(c)   "v_lshl_or_b32 %" #c ", %" #c ", 1, %9\n"
#define I_ADD3(c)     "v_add3_u32 %" #c ", %" #c ", %8, %9\n"
#define I_MAX3(c)     "v_max3_i32 %" #c ", %" #c ", %8, %9\n"
#define I_BCNT(c)     "v_bcnt_u32_b32 %" #c ", %" #c ", %9\n"
#define I_FMA(c)      "v_fma_f32 %" #c ", %" #c ", %8, %9\n"
#define I_MAD24(c)    "v_mad_u32_u24 %" #c ", %" #c ", %8, %9\n"
#define I_CNDMASK(c)  "v_cndmask_b32 %" #c ", %" #c ", %8, vcc\n"
#define I_ADDCO(c)    "v_add_co_u32 %" #c ", vcc, %" #c ", %8\n"
#define I_PKADD(c)    "v_pk_add_i16 %" #c ", %" #c ", %8\n"
#define I_CNDMASK_S(c) "v_cndmask_b32 %" #c ", %" #c ", %8, %10\n"       /* lane mask in an SGPR pair (what v_cmp ..., s[a:b] leaves) */
#define I_CMP_CND(c)  "v_cmp_lt_u32 vcc, %" #c ", %9\n v_cndmask_b32 %" #c ", %" #c ", %8, vcc\n"   /* the compare + select pair of a per-lane branch */

#define I_MAXF(c)     "v_max_f32 %" #c ", %" #c ", %8\n"
#define I_MINF(c)     "v_min_f32 %" #c ", %" #c ", %8\n"
#define I_ADDF(c)     "v_add_f32 %" #c ", %" #c ", %8\n"
#define I_MAX3F(c)    "v_max3_f32 %" #c ", %" #c ", %8, %9\n"
#define I_MED3F(c)    "v_med3_f32 %" #c ", %" #c ", %8, %9\n"
#define I_MED3I(c)    "v_med3_i32 %" #c ", %" #c ", %8, %9\n"
#define I_CVTUB(c)    "v_cvt_f32_ubyte1 %" #c ", %" #c "\n"
#define I_BFEI(c)     "v_bfe_i32 %" #c ", %" #c ", 3, 1\n"
#define I_MADI24(c)   "v_mad_i32_i24 %" #c ", %" #c ", %8, %9\n"
#define I_MAXI_DPP(c) "v_max_i32_dpp %" #c ", %" #c ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MAXF_DPP(c) "v_max_f32_dpp %" #c ", %" #c ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDF_DPP(c) "v_add_f32_dpp %" #c ", %" #c ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOV_DPP(c)  "v_mov_b32_dpp %" #c ", %" #c " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MAXU16(c)   "v_pk_max_i16 %" #c ", %" #c ", %8\n"
#define I_SUBF(c)     "v_sub_f32 %" #c ", %" #c ", %8\n"
#define I_MAXU(c)     "v_max_u32 %" #c ", %" #c ", %8\n"
#define I_PKADDF(c)   "v_pk_add_f32 %" #c ", %" #c ", %8\n"
#define I_PKFMAF(c)   "v_pk_fma_f32 %" #c ", %" #c ", %8, %9\n"

#define KERNEL(NAME, INS)                                                                                   \
__global__ void __launch_bounds__(256) NAME(int n, unsigned* out) {                                         \
    unsigned v0 = threadIdx.x * 2654435761u, v1 = v0 + 977u, v2 = v0 + 2 * 977u, v3 = v0 + 3 * 977u,        \
             v4 = v0 + 4 * 977u, v5 = v0 + 5 * 977u, v6 = v0 + 6 * 977u, v7 = v0 + 7 * 977u;                \
    unsigned a = blockIdx.x + 12345u, b = (threadIdx.x & 15) + 1;                                           \
    const unsigned long long m = 0x5555AAAA3333CCCCull ^ (unsigned long long)n;                             \
    for (int i = 0; i < n; i++)                                                                             \
        asm volatile(BODY8(INS) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b), "s"(m) : "vcc"); \
    out[blockIdx.x * 256 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;                            \
}
KERNEL(k_and, I_AND) KERNEL(k_xor, I_XOR) KERNEL(k_add, I_ADD) KERNEL(k_max, I_MAX) KERNEL(k_lshl, I_LSHL) KERNEL(k_alignbit, I_ALIGNBIT) KERNEL(k_bfi, I_BFI)
KERNEL(k_andor, I_ANDOR) KERNEL(k_lshlor, I_LSHLOR) KERNEL(k_add3, I_ADD3) KERNEL(k_max3, I_MAX3) KERNEL(k_bcnt, I_BCNT) KERNEL(k_fma, I_FMA) KERNEL(k_mad24, I_MAD24)
KERNEL(k_cndmask, I_CNDMASK) KERNEL(k_addco, I_ADDCO) KERNEL(k_pkadd, I_PKADD) KERNEL(k_cndmask_s, I_CNDMASK_S) KERNEL(k_cmp_cnd, I_CMP_CND)

KERNEL(k_maxf, I_MAXF) KERNEL(k_minf, I_MINF) KERNEL(k_addf, I_ADDF) KERNEL(k_max3f, I_MAX3F) KERNEL(k_med3f, I_MED3F) KERNEL(k_med3i, I_MED3I) KERNEL(k_cvtub, I_CVTUB)
KERNEL(k_bfei, I_BFEI) KERNEL(k_madi24, I_MADI24) KERNEL(k_maxi_dpp, I_MAXI_DPP) KERNEL(k_maxf_dpp, I_MAXF_DPP) KERNEL(k_addf_dpp, I_ADDF_DPP) KERNEL(k_mov_dpp, I_MOV_DPP)
KERNEL(k_pkmaxi16, I_MAXU16) KERNEL(k_subf, I_SUBF) KERNEL(k_maxu, I_MAXU)
// packed fp32: every chain register is a VGPR pair
#define KERNEL64(NAME, INS)                                                                                 \
__global__ void __launch_bounds__(256) NAME(int n, unsigned* out) {                                         \
    unsigned long long v0 = threadIdx.x * 2654435761ull, v1 = v0 + 977u, v2 = v0 + 2 * 977u, v3 = v0 + 3 * 977u, \
             v4 = v0 + 4 * 977u, v5 = v0 + 5 * 977u, v6 = v0 + 6 * 977u, v7 = v0 + 7 * 977u;                \
    unsigned long long a = blockIdx.x + 12345u, b = (threadIdx.x & 15) + 1;                                 \
    const unsigned long long m = 0x5555AAAA3333CCCCull ^ (unsigned long long)n;                             \
    for (int i = 0; i < n; i++)                                                                             \
        asm volatile(BODY8(INS) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b), "s"(m) : "vcc"); \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7);                \
}
KERNEL64(k_pkaddf, I_PKADDF) KERNEL64(k_pkfmaf, I_PKFMAF)

typedef void (*kern_t)(int, unsigned*);
static double run(kern_t k, int n, unsigned* out, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, 16, out);
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, n, out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)n * 64 * (blocks * 4.0) / (ms * 1e-3);        // wave-instructions per second
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    const int blocks = cus * 8;                                   // 8 blocks of 4 waves per CU: 8 waves per SIMD
    unsigned* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int n = 4000;
    struct { const char* name; const char* enc; kern_t k; } t[] = {
        {"v_and_b32", "VOP2", k_and}, {"v_xor_b32", "VOP2", k_xor}, {"v_add_u32", "VOP2", k_add}, {"v_max_i32", "VOP2", k_max}, {"v_lshlrev_b32", "VOP2", k_lshl},
        {"v_cndmask_b32 (vcc)", "VOP2", k_cndmask}, {"v_add_co_u32 (vcc)", "VOP3b", k_addco},
        {"v_alignbit_b32", "VOP3", k_alignbit}, {"v_bfi_b32", "VOP3", k_bfi}, {"v_and_or_b32", "VOP3", k_andor}, {"v_lshl_or_b32", "VOP3", k_lshlor},
        {"v_add3_u32", "VOP3", k_add3}, {"v_max3_i32", "VOP3", k_max3}, {"v_bcnt_u32_b32", "VOP3", k_bcnt}, {"v_mad_u32_u24", "VOP3", k_mad24},
        {"v_fma_f32", "VOP3", k_fma}, {"v_pk_add_i16", "VOP3P", k_pkadd},
        {"v_cndmask_b32 (sgpr pair)", "VOP3", k_cndmask_s}, {"v_cmp + v_cndmask (x2)", "VOPC+VOP2", k_cmp_cnd},
        {"v_max_f32", "VOP2", k_maxf}, {"v_min_f32", "VOP2", k_minf}, {"v_add_f32", "VOP2", k_addf}, {"v_sub_f32", "VOP2", k_subf}, {"v_max_u32", "VOP2", k_maxu},
        {"v_max3_f32", "VOP3", k_max3f}, {"v_med3_f32", "VOP3", k_med3f}, {"v_med3_i32", "VOP3", k_med3i}, {"v_cvt_f32_ubyte1", "VOP1", k_cvtub}, {"v_bfe_i32", "VOP3", k_bfei},
        {"v_mad_i32_i24", "VOP3", k_madi24}, {"v_max_i32 dpp row_shr", "DPP", k_maxi_dpp}, {"v_max_f32 dpp row_shr", "DPP", k_maxf_dpp}, {"v_add_f32 dpp row_shr", "DPP", k_addf_dpp},
        {"v_mov_b32 dpp row_shr", "DPP", k_mov_dpp}, {"v_pk_max_i16", "VOP3P", k_pkmaxi16}, {"v_pk_add_f32 (2 per lane)", "VOP3P", k_pkaddf}, {"v_pk_fma_f32 (2 per lane)", "VOP3P", k_pkfmaf}};
    printf("%d CUs, %.2f GHz shader clock, %d SIMDs\n", cus, ghz, cus * 4);
    for (auto& e : t) {
        const double w = run(e.k, n, out, blocks);
        printf("%-26s %-9s %7.3f T wave-instr/s  %7.1f T lane-ops/s  %.2f SIMD cycles per wave64 instruction\n", e.name, e.enc, w / 1e12, w * 64 / 1e12, cus * 4 * ghz * 1e9 / w);
    }
    return 0;
}
