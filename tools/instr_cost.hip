// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD at 8 waves/SIMD) of the instruction kinds the
// blend kernels are made of, on gfx950.  Eight independent instances per loop iteration, inline asm.
// Build: hipcc --offload-arch=gfx950 -O3 tools/instr_cost.hip -o tools/instr_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)

template <int MODE>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#define S(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 1) {
#define S(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 2) {
#define S(n) "v_exp_f32 %" #n ", %" #n "\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 3) {
#define S(n) "v_rcp_f32 %" #n ", %" #n "\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 4) {  // DPP add inside a row (each instance reads a register written 8 instructions earlier)
#define S(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 5) {
            asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n" OPERANDS);
        } else if (MODE == 6) {
            asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                         "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n" OPERANDS);
        } else if (MODE == 7) {  // compare into an SGPR pair + select on it
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n"
                         "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %5, vcc\n"
                         OPERANDS : "vcc");
        } else if (MODE == 8) {  // scalar ALU only
            asm volatile("s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20\n s_add_u32 s22, s22, 1\n s_and_b32 s23, s23, s22\n"
                         "s_add_u32 s24, s24, 1\n s_and_b32 s25, s25, s24\n s_add_u32 s26, s26, 1\n s_and_b32 s27, s27, s26\n"
                         ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        } else if (MODE == 9) {  // 4 VALU + 4 SALU interleaved
            asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_and_b32 s21, s21, s20\n"
                         "v_fma_f32 %2, %2, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %3, %3, %8, %9\n s_and_b32 s23, s23, s22\n"
                         OPERANDS : "s20", "s21", "s22", "s23", "scc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char *name, float *out) {
    const int iters = 20000, wps = 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wps), block(256);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 8.0 * iters * wps;
    printf("%-28s %.3f ms -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", out); run<1>("v_mul_f32", out); run<2>("v_exp_f32", out); run<3>("v_rcp_f32", out);
    run<4>("v_add_f32_dpp row_shr", out); run<5>("v_permlane32_swap", out); run<6>("v_permlane16_swap", out);
    run<7>("v_cmp + v_cndmask (vcc)", out); run<8>("s_add / s_and", out); run<9>("4 v_fma + 4 salu", out);
    return 0;
}
