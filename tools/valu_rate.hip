// Microbenchmark: wave64 VALU issue rate on gfx950 for plain and packed f32 FMA, by waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0001f, c = 0.5f;
    const float2v m2 = {m, m}, c2 = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 independent plain FMAs
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 1) {  // 4 independent packed FMAs (8 FMAs per lane)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m2), "v"(c2));
        } else {  // dependent chain of plain FMAs (8 per iteration)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char *name, float *out) {
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD: block = 256 threads = 4 waves = 1 per SIMD; wps blocks per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        dim3 grid(256 * wps), block(256);
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (MODE == 1 ? 4.0 : 8.0) * iters;
        const double wave_instr_per_simd = instr_per_wave * wps;  // waves on one SIMD
        printf("%-14s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz), %.1f TFLOP/s\n", name, wps, ms,
               ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4,
               (MODE == 1 ? 2.0 : 1.0) * 2 * 64 * instr_per_wave * 4 * 256 * wps / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("fma x8 indep", out); run<1>("pk_fma x4", out); run<2>("fma chain", out);
    return 0;
}
