// Microbenchmark: what does it cost a wave to receive a 64-byte per-Gaussian record that every lane needs?
//   mode 0  LDS broadcast: 3 x ds_read_b128 from a wave-uniform address (what the blend kernels do today)
//   mode 1  scalar path, no prefetch: v_readlane of the record's byte offset, s_load_dwordx16, wait, use
//   mode 2  scalar path, record k+1 requested before record k is used (two SGPR sets, wait placed before the request)
// followed in every mode by WORK dependent-ish vector instructions that take the record's fields as scalar / vector
// operands (the blend loops issue ~30 per entry).  Every workgroup reads the 64 records of a 4 KiB window that moves on every 64 entries (a round of its
// tile list); the windows wrap inside `span` bytes (4 KiB: always the same lines; 1 MiB: L2-resident; 24 MiB: the size of
// a 360k-Gaussian record array).
// Prints shader cycles per entry per wave (s_memtime) and wall time.
// Build: hipcc --offload-arch=gfx950 -O3 tools/scalar_rec.hip -o tools/scalar_rec ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define USE16(acc, r)                                                                                              \
    asm volatile("v_fma_f32 %0, %4, %1, %0\n v_fma_f32 %1, %5, %2, %1\n v_fma_f32 %2, %6, %3, %2\n v_fma_f32 %3, %7, %0, %3\n"  \
                 "v_fma_f32 %0, %8, %1, %0\n v_fma_f32 %1, %9, %2, %1\n v_fma_f32 %2, %10, %3, %2\n v_fma_f32 %3, %11, %0, %3\n" \
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                            \
                 : "s"(r[0]), "s"(r[1]), "s"(r[2]), "s"(r[3]), "s"(r[4]), "s"(r[5]), "s"(r[6]), "s"(r[7]))

template <int N>
__device__ inline void filler(float (&acc)[4]) {
#pragma unroll
    for (int i = 0; i < N; i++)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}

template <int MODE, int WORK4>
__global__ void __launch_bounds__(256) k(const char *recs, unsigned span, int entries, float *out, unsigned long long *stamps) {
    __shared__ float4 s_rec[64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every lane holds the byte offset of "its" staged record, as in the culling layout (lane -> Gaussian)
    unsigned h = (blockIdx.x * 64u + lane) * 2654435761u;
    h ^= h >> 15;
    // the workgroup's window of 64 records moves on every 64 entries (a new round of its tile list)
    auto window_offset = [&](int round) {
        return ((blockIdx.x * 131u + (unsigned)round) % (span >> 12)) * 4096u + (h & 63u) * 64u;
    };
    unsigned v_off = window_offset(0);
    for (int i = threadIdx.x; i < 256; i += 256) s_rec[i] = make_float4(1.f + i, 0.5f, 0.25f, 2.f);
    __syncthreads();
    float acc[4] = {1.f + lane, 2.f, 3.f, 4.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int e = 0; e < entries; e++) {
            const int j = (e * 7 + wave) & 63;
            v4f a, b, c;
            asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b128 %2, %3 offset:32\n s_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c) : "v"((unsigned)(size_t)&s_rec[4 * j]) : "memory");
            acc[0] = fmaf(a.x, acc[1], acc[0]); acc[1] = fmaf(a.y, acc[2], acc[1]); acc[2] = fmaf(a.z, acc[3], acc[2]); acc[3] = fmaf(a.w, acc[0], acc[3]);
            acc[0] = fmaf(b.x, acc[1], acc[0]); acc[1] = fmaf(c.y, acc[2], acc[1]); acc[2] = fmaf(b.z, acc[3], acc[2]); acc[3] = fmaf(c.w, acc[0], acc[3]);
            filler<WORK4>(acc);
        }
    } else if (MODE == 1) {
        for (int e = 0; e < entries; e++) {
            const int j = (e * 7 + wave) & 63;
            if ((e & 63) == 0) v_off = window_offset(e >> 6);
            const unsigned off = __builtin_amdgcn_readlane(v_off, j);
            v16i r;
            asm volatile("s_load_dwordx16 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(recs), "s"(off) : "memory");
            USE16(acc, r);
            filler<WORK4>(acc);
        }
    } else {
        v16i r0, r1;
        unsigned off = __builtin_amdgcn_readlane(v_off, wave & 63);
        asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r0) : "s"(recs), "s"(off) : "memory");
        for (int e = 0; e < entries; e += 2) {
            const int j1 = ((e + 1) * 7 + wave) & 63, j2 = ((e + 2) * 7 + wave) & 63;
            if ((e & 63) == 0) v_off = window_offset(e >> 6);
            off = __builtin_amdgcn_readlane(v_off, j1);
            asm volatile("s_waitcnt lgkmcnt(0)\n s_load_dwordx16 %0, %2, %3" : "=s"(r1), "+s"(r0) : "s"(recs), "s"(off) : "memory");
            USE16(acc, r0);
            filler<WORK4>(acc);
            off = __builtin_amdgcn_readlane(v_off, j2);
            asm volatile("s_waitcnt lgkmcnt(0)\n s_load_dwordx16 %0, %2, %3" : "=s"(r0), "+s"(r1) : "s"(recs), "s"(off) : "memory");
            USE16(acc, r1);
            filler<WORK4>(acc);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (lane == 0) stamps[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE, int WORK4>
void run(const char *name, const char *recs, unsigned span, int wgs_per_cu, int entries, float *out, unsigned long long *stamps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<MODE, WORK4>), dim3(grid), dim3(256), 0, 0, recs, span, entries, out, stamps);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WORK4>), dim3(grid), dim3(256), 0, 0, recs, span, entries, out, stamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-34s span %9u B  %d waves/SIMD  %2d vector instr/entry: %7.1f cycles per entry per wave (median), %7.1f per entry per SIMD, wall %.3f ms\n",
           name, span, wgs_per_cu, 8 + 4 * WORK4, (double)h[h.size() / 2] / entries, (double)h[h.size() / 2] / entries / wgs_per_cu, ms);
}

template <int WORK4>
void sweep(const char *recs, int entries, float *out, unsigned long long *stamps) {
    for (int wps : {4, 5, 8}) {
        run<0, WORK4>("LDS broadcast 3 x ds_read_b128", recs, 4096, wps, entries, out, stamps);
        for (unsigned span : {4096u, 1u << 20, 24u << 20}) {
            run<1, WORK4>("s_load_dwordx16, no prefetch", recs, span, wps, entries, out, stamps);
            run<2, WORK4>("s_load_dwordx16, one ahead", recs, span, wps, entries, out, stamps);
        }
    }
}

int main() {
    const size_t bytes = 64ull << 20;
    char *recs; hipMalloc(&recs, bytes);
    std::vector<float> init(bytes / 4, 0.001f);
    hipMemcpy(recs, init.data(), bytes, hipMemcpyHostToDevice);
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long *stamps; hipMalloc(&stamps, 256 * 8 * 4 * 8);
    const int entries = 4096;
    sweep<6>(recs, entries, out, stamps);    // 32 vector instructions per entry
    sweep<12>(recs, entries, out, stamps);   // 56
    return 0;
}
