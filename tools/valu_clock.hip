// Microbenchmark: what does a wave64 VALU instruction cost in SHADER cycles, and at what clock does the chip run a
// VALU-dense kernel?  Reconciles tools/instr_cost.hip (which converts wall time to cycles at the nominal 2.4 GHz and
// reads 3.1 "cycles" per v_fma_f32) with MI355X_MICROARCH.md (2 cycles per wave64 VALU instruction on a SIMD-32).
//
// Every wave stamps s_memtime (shader clock ticks) and s_memrealtime (constant 100 MHz) around its loop:
//   in-kernel clock       = d(memtime) / d(memrealtime) * 100 MHz
//   cycles per instruction = d(memtime) / (instructions per wave * waves per SIMD)      [issue slots of one SIMD]
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_clock.hip -o tools/valu_clock ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *stamps, int iters) {
    float a0 = threadIdx.x + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    unsigned s0 = blockIdx.x, s1 = 3, s2 = 1, s3 = 0xff;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#define S(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 1) {
#define S(n) "v_exp_f32 %" #n ", %" #n "\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 2) {
#define S(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE == 3) {  // the blend loops' mix: 6 fma/mul + 1 exp + 1 cmp/cndmask pair per 9
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_mul_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_mul_f32 %5, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n" OPERANDS : "vcc");
        } else if (MODE == 4) {  // wave-level rotate of a register (DPP wave_ror:1), the systolic hand-over
#define S(n) "v_mov_b32_dpp %" #n ", %" #n " wave_ror:1 row_mask:0xf bank_mask:0xf\n"
            asm volatile(REP8(S) OPERANDS);
#undef S
        } else if (MODE >= 5 && MODE <= 7) {  // scalar instructions beside (5: four, 6: eight) or instead of (7) the eight v_fma_f32
#define S(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
            if (MODE != 7) asm volatile(REP8(S) OPERANDS);
#undef S
            asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 5\n s_lshl_b32 %2, %2, 1\n s_and_b32 %3, %3, 7\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            if (MODE != 5) asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 5\n s_lshl_b32 %2, %2, 1\n s_and_b32 %3, %3, 7\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 8) {  // two taken branches beside the eight v_fma_f32
#define S(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n s_branch 1f\n s_nop 0\n 1:\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n s_branch 2f\n s_nop 0\n 2:\n" OPERANDS);
#undef S
        } else if (MODE == 9) {  // a compare, its mask combined on the scalar unit, and a select, four times (the blend loops' hit tests)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %1, %1, %0, vcc\n v_cmp_lt_f32 vcc, %2, %8\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %3, %3, %2, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %5, %5, %4, vcc\n v_cmp_lt_f32 vcc, %6, %8\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %7, %7, %6, vcc\n" OPERANDS : "vcc", "scc");
        }
    }
    out[0] = (float)(s0 + s1 + s2 + s3);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char *name, float *out, unsigned long long *stamps, int wps, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wps), block(256);
    const size_t waves = (size_t)grid.x * 4;
    for (int warm = 0; warm < 3; warm++) hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, stamps, iters);  // let the clock settle
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, stamps, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> clk(waves), cyc(waves);
    for (size_t w = 0; w < waves; w++) { clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 0.1; cyc[w] = (double)h[2 * w]; }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double instr_per_simd = 8.0 * iters * wps;
    printf("%-22s waves/SIMD %d: %8.3f ms wall | clock %.3f GHz (median over waves) | %.2f shader cycles per wave-instr per SIMD "
           "(in-kernel) | %.2f 'cycles' if wall time is priced at 2.4 GHz\n",
           name, wps, ms, clk[waves / 2], cyc[waves / 2] / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400000;  // ~40 ms per launch at 8 waves/SIMD: long enough for DVFS to settle
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    unsigned long long *stamps; hipMalloc(&stamps, 256 * 8 * 4 * 2 * sizeof(unsigned long long));
    for (int wps : {1, 2, 4, 8}) run<0>("v_fma_f32", out, stamps, wps, iters);
    run<1>("v_exp_f32", out, stamps, 8, iters);
    run<2>("v_add_f32_dpp row_shr", out, stamps, 8, iters);
    run<3>("blend-like mix", out, stamps, 8, iters);
    run<4>("v_mov_dpp wave_ror:1", out, stamps, 8, iters);
    for (int wps : {4, 8}) {
        run<5>("8 v_fma + 4 scalar", out, stamps, wps, iters);
        run<6>("8 v_fma + 8 scalar", out, stamps, wps, iters);
        run<7>("8 scalar only", out, stamps, wps, iters);
        run<8>("8 v_fma + 2 taken branches", out, stamps, wps, iters);
        run<9>("4 x (cmp, s_and, cndmask)", out, stamps, wps, iters);
    }
    run<0>("v_fma_f32 (short)", out, stamps, 8, 20000);  // the 1.7 ms launch tools/instr_cost.hip times
    return 0;
}
