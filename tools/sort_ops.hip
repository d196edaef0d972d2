// Microbenchmark: issue cost of the instructions the per-tile bitonic sort is made of (csrc/tilesort.hip), in shader cycles per
// wave-instruction per SIMD -- 64-bit against 32-bit compares, selects, the DPP moves of the lane exchanges, ds_bpermute.
// Same method as tools/valu_clock.hip (s_memtime around the loop, 8 instructions per iteration and wave).
// Build: hipcc --offload-arch=gfx950 -O3 tools/sort_ops.hip -o tools/sort_ops ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned *out, unsigned long long *stamps, int iters) {
    unsigned a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned addr = ((threadIdx.x & 63) ^ 32) << 2;
    unsigned long long m0 = 0, m1 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // eight 64-bit compares (pairs of registers as the operands)
            asm volatile("v_cmp_lt_u64 vcc, %[a], %[b]\n v_cmp_lt_u64 %[m0], %[b], %[c]\n v_cmp_lt_u64 vcc, %[c], %[d]\n v_cmp_lt_u64 %[m1], %[d], %[a]\n"
                         "v_cmp_gt_u64 vcc, %[a], %[b]\n v_cmp_gt_u64 %[m0], %[b], %[c]\n v_cmp_gt_u64 vcc, %[c], %[d]\n v_cmp_gt_u64 %[m1], %[d], %[a]\n"
                         : [m0] "+s"(m0), [m1] "+s"(m1)
                         : [a] "v"(((unsigned long long)a0 << 32) | a1), [b] "v"(((unsigned long long)a2 << 32) | a3), [c] "v"(((unsigned long long)a4 << 32) | a5), [d] "v"(((unsigned long long)a6 << 32) | a7) : "vcc");
        } else if (MODE == 1) {   // eight 32-bit compares
            asm volatile("v_cmp_lt_u32 vcc, %[a], %[b]\n v_cmp_lt_u32 %[m0], %[b], %[c]\n v_cmp_lt_u32 vcc, %[c], %[d]\n v_cmp_lt_u32 %[m1], %[d], %[a]\n"
                         "v_cmp_gt_u32 vcc, %[a], %[b]\n v_cmp_gt_u32 %[m0], %[b], %[c]\n v_cmp_gt_u32 vcc, %[c], %[d]\n v_cmp_gt_u32 %[m1], %[d], %[a]\n"
                         : [m0] "+s"(m0), [m1] "+s"(m1) : [a] "v"(a0), [b] "v"(a2), [c] "v"(a4), [d] "v"(a6) : "vcc");
        } else if (MODE == 2) {   // eight selects on a mask in an SGPR pair
#define S(n) "v_cndmask_b32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr), "s"(0x5555aaaa5555aaaaull));
#undef S
        } else if (MODE == 3) {   // DPP move, quad_perm [1,0,3,2]
#define S(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
            asm volatile(REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#undef S
        } else if (MODE == 4) {   // DPP move, row_ror:8
#define S(n) "v_mov_b32_dpp %" #n ", %" #n " row_ror:8 row_mask:0xf bank_mask:0xf\n"
            asm volatile(REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#undef S
        } else if (MODE == 5) {   // ds_bpermute: eight out, then waited for
#define S(n) "ds_bpermute_b32 %" #n ", %8, %" #n "\n"
            asm volatile(REP8(S) "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));
#undef S
        } else if (MODE == 6) {   // one key-stage of the sort as compiled today, twice: 2 DPP moves, 64-bit compare, mask xor, 2 selects (= 12 instructions; 8 counted)
            asm volatile("v_mov_b32_dpp %[o0], %[k0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %[o1], %[k1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "s_nop 0\n v_cmp_lt_u64 vcc, %[o], %[k]\n s_xor_b64 vcc, vcc, %[dir]\n v_cndmask_b32 %[k0], %[k0], %[o0], vcc\n v_cndmask_b32 %[k1], %[k1], %[o1], vcc\n"
                         : [k0] "+v"(a0), [k1] "+v"(a1), [o0] "+v"(a2), [o1] "+v"(a3), [k] "+v"(m0), [o] "+v"(m1) : [dir] "s"(0x5555aaaa5555aaaaull) : "vcc", "scc");
        } else if (MODE == 7) {   // 8 x v_xor_b32 (plain full-rate integer reference)
#define S(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
            asm volatile(REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));
#undef S
        } else if (MODE == 8) {   // selects on VCC written by a 32-bit compare just before (cmp, sel, sel) x 2 + 2 xor
            asm volatile("v_cmp_lt_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_xor_b32 %6, %6, %8\n"
                         "v_cmp_lt_u32 vcc, %4, %6\n v_cndmask_b32 %4, %4, %6, vcc\n v_cndmask_b32 %5, %5, %7, vcc\n v_xor_b32 %2, %2, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr) : "vcc");
        } else if (MODE == 9) {   // the same with 64-bit compares
            asm volatile("v_cmp_lt_u64 vcc, %[p], %[q]\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_xor_b32 %6, %6, %8\n"
                         "v_cmp_lt_u64 vcc, %[q], %[p]\n v_cndmask_b32 %4, %4, %6, vcc\n v_cndmask_b32 %5, %5, %7, vcc\n v_xor_b32 %2, %2, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr), [p] "v"(m0 + i), [q] "v"(m1 + 7) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)m0 + (unsigned)m1;
    if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, unsigned *out, unsigned long long *stamps, int wps, int iters, double per_iter = 8.0) {
    dim3 grid(256 * wps), block(256);
    const size_t waves = (size_t)grid.x * 4;
    for (int warm = 0; warm < 3; warm++) hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, stamps, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s waves/SIMD %d: %6.2f shader cycles per wave-instruction per SIMD\n", name, wps, (double)h[waves / 2] / (per_iter * iters * wps));
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    unsigned *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned));
    unsigned long long *stamps; hipMalloc(&stamps, 256 * 8 * 4 * sizeof(unsigned long long));
    for (int wps : {2, 4, 8}) {
        run<7>("v_xor_b32", out, stamps, wps, iters);
        run<0>("v_cmp_{lt,gt}_u64", out, stamps, wps, iters);
        run<1>("v_cmp_{lt,gt}_u32", out, stamps, wps, iters);
        run<2>("v_cndmask_b32 (SGPR mask)", out, stamps, wps, iters);
        run<3>("v_mov_b32_dpp quad_perm", out, stamps, wps, iters);
        run<4>("v_mov_b32_dpp row_ror:8", out, stamps, wps, iters);
        run<5>("ds_bpermute_b32 (8 in flight)", out, stamps, wps, iters);
        run<6>("key-stage: 2 dpp, cmp64, s_xor, 2 sel (per 6)", out, stamps, wps, iters, 6.0);
        run<8>("cmp32 + 2 sel + xor", out, stamps, wps, iters);
        run<9>("cmp64 + 2 sel + xor", out, stamps, wps, iters);
    }
    return 0;
}
