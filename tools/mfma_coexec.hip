// Microbenchmark: do v_mfma_f32_16x16x4_f32 and plain VALU FMAs overlap on one SIMD of gfx950?
// Even blocks run VALU only, odd blocks MFMA only (mode 2) -- or every wave interleaves both (mode 3) -- and the
// time is compared with the VALU-only (mode 0) and MFMA-only (mode 1) runs at the same occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_coexec.hip -o tools/mfma_coexec ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void valu8(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5, float &a6, float &a7, float m, float c) {
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
}

// per iteration: 32 VALU FMAs and / or 4 MFMAs (2 independent accumulators) = 128 VALU cycles and / or 128 MFMA cycles
template <int MODE>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    const float m = 1.0001f, c = 0.5f, x = a0 * 1e-3f, y = 1e-3f;
    // MODE 4 (512-thread workgroups): waves 0-3 of a workgroup -- one per SIMD -- run VALU only, waves 4-7 -- one per SIMD -- MFMA
    // only: a vector wave and a matrix wave side by side on EVERY SIMD (mode 2 leaves the pairing to the dispatcher)
    const bool do_valu = MODE == 0 || MODE == 3 || (MODE == 2 && (blockIdx.x & 1) == 0) || (MODE == 4 && threadIdx.x < 256);
    const bool do_mfma = MODE == 1 || MODE == 3 || (MODE == 2 && (blockIdx.x & 1) == 1) || (MODE == 4 && threadIdx.x >= 256);
    for (int i = 0; i < iters; i++) {
        if (do_mfma) { d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, d1, 0, 0, 0); }
        if (do_valu) { valu8(a0, a1, a2, a3, a4, a5, a6, a7, m, c); valu8(a0, a1, a2, a3, a4, a5, a6, a7, m, c); }
        if (do_mfma) { d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, d1, 0, 0, 0); }
        if (do_valu) { valu8(a0, a1, a2, a3, a4, a5, a6, a7, m, c); valu8(a0, a1, a2, a3, a4, a5, a6, a7, m, c); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
}

template <int MODE>
float run(float *out, int wps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(MODE == 4 ? 256 * wps / 2 : 256 * wps), block(MODE == 4 ? 512 : 256);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int wps = 2; wps <= 8; wps *= 2) {
        const float v = run<0>(out, wps), m = run<1>(out, wps), split = run<2>(out, wps), mixed = run<3>(out, wps), paired = run<4>(out, wps);
        printf("waves/SIMD %d: VALU only %.3f ms, MFMA only %.3f ms | half the blocks each (mode 2) %.3f ms, a vector and a matrix wave paired on every SIMD (mode 4) %.3f ms "
               "(no overlap would be %.3f, full overlap %.3f) | every wave both (mode 3) %.3f ms (no overlap %.3f, full overlap %.3f)\n",
               wps, v, m, split, paired, (v + m) / 2, (v > m ? v : m) / 2, mixed, v + m, v > m ? v : m);
    }
    return 0;
}
