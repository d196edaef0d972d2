// Checks the operand layout of v_mfma_f32_16x16x4_f32 assumed by blend_bwd's moment reduction:
//   A[i][k]: lane = i + 16 k;  B[k][j]: lane = j + 16 k;  D[i][j]: lane = j + 16 (i / 4), register i % 4.
// Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/mfma_layout.hip -o tools/mfma_layout && tools/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void k(const float *A, const float *B, float *D) {  // A 16x4, B 4x16 row-major, D 16x16
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16];
    const float b = B[(l / 16) * 16 + l % 16];
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * (l / 16) + r) * 16 + l % 16] = c[r];
}

int main() {
    float hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; i++) { hA[i] = (float)((i * 37) % 11) - 5.f; hB[i] = (float)((i * 53) % 13) * 0.25f - 1.f; }
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            float s = 0.f;
            for (int kk = 0; kk < 4; kk++) s += hA[i * 4 + kk] * hB[kk * 16 + j];
            ref[i * 16 + j] = s;
        }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 256; i++) err = fmax(err, fabs(hD[i] - ref[i]));
    printf("mfma_f32_16x16x4 layout check: max abs error %.3g -> %s\n", err, err < 1e-5 ? "OK" : "MISMATCH");
    return err < 1e-5 ? 0 : 1;
}
