// What rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the blend kernels -- the guide
// (MI355X_MICROARCH.md, HBM section) calibrates 16-byte-per-lane streams only ("other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern before trusting an absolute").  Every kernel here moves a
// KNOWN number of bytes, printed by main(); tools/traffic_calib.sh runs the binary under the two --pmc passes and divides.
//
//   read16   16 bytes per lane, streaming                      (the guide's case: FETCH_SIZE = 1/2 of the bytes)
//   read4    4 bytes per lane, streaming (a wave: 256 contiguous bytes)
//   tile4    a 16 x 16 pixel tile per workgroup out of seven image planes: a wave reads four 64-byte row segments per
//            plane (blend_bwd's prologue; the tile to the right holds the other half of each 128-byte line)
//   tile4xcd the same with a tile's right-hand neighbour on the SAME XCD (the blend kernels' workgroup -> tile mapping)
//   gather44 one 44-byte record per lane at a random place of a table, every record once (blend's staging of a list entry:
//            no re-use, so the count is what ONE miss on such a record moves)
//   gather44x4  the same table a quarter the size, every record four times (D / V = 4.2 at config 3)
//   store16  16 bytes per lane, streaming
//   store40  one 40-byte record (five 8-byte stores) per lane at a random slot, every slot once (blend_bwd's pair records)
//   store1   one byte per lane at a random place of a byte array (the pair_valid flags as they are)
//   or1      one bit per lane, atomic OR without return at agent scope into a bit array (the flags as bits)
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/traffic_calib tools/traffic_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_read16_kernel(const uint4 *__restrict__ src, size_t n, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void calib_read4_kernel(const uint32_t *__restrict__ src, size_t n, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// one workgroup of 256 per tile, tiles in row-major order (neighbours close in time, as in the blend kernels' runs of tiles)
template <bool XCD_RUNS>
__global__ void calib_tile4_kernel(const float *__restrict__ planes, int W, int H, int nplanes, uint32_t *__restrict__ sink) {
    const int gx = (W + 15) / 16;
    // XCD_RUNS: workgroup b runs on XCD b % 8; XCD x takes the x-th eighth of the tiles in row-major order, so that a tile's right-hand
    // neighbour (the other half of its 128-byte lines) is the same L2's next workgroup -- the blend kernels' mapping
    int t = blockIdx.x;
    if (XCD_RUNS) { const int per = (gridDim.x + 7) / 8; t = (blockIdx.x & 7) * per + (blockIdx.x >> 3); if (t >= (int)gridDim.x) return; }
    const int tx = t % gx, ty = t / gx;
    const int x = tx * 16 + (threadIdx.x & 15), y = ty * 16 + (threadIdx.x >> 4);
    float acc = 0.f;
    if (x < W && y < H)
        for (int k = 0; k < nplanes; k++) acc += planes[(size_t)k * W * H + (size_t)y * W + x];
    if (acc == 1.2345f) sink[0] = 1;
}
__global__ void calib_gather44_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ perm, size_t n, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t *r = table + 11 * (size_t)perm[i];
#pragma unroll
        for (int k = 0; k < 11; k++) acc ^= r[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void calib_store16_kernel(uint4 *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void calib_store40_kernel(float2 *__restrict__ dst, const uint32_t *__restrict__ perm, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float2 *r = dst + 5 * (size_t)perm[i];
#pragma unroll
        for (int k = 0; k < 5; k++) r[k] = make_float2((float)i, (float)k);
    }
}
__global__ void calib_store1_kernel(uint8_t *__restrict__ dst, const uint32_t *__restrict__ perm, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[perm[i]] = 1;
}
__global__ void calib_or1_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ perm, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t s = perm[i];
        __hip_atomic_fetch_or(dst + (s >> 5), 1u << (s & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// (the L2s and the Infinity Cache hold what the previous launch left: 512 MiB streamed through between two measured launches)
__global__ void calib_flush_kernel(uint4 *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4(0u, 0u, 0u, 0u);
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const size_t R = 1500000;                  // records / pairs (config 3: D = 1.50 M)
    const int W = 1920, H = 1080, NPL = 7;
    const size_t stream_bytes = (size_t)256 << 20;
    const size_t flush_bytes = (size_t)512 << 20;
    uint4 *stream, *flush; uint32_t *table, *perm, *perm4, *sink, *bits; float *planes; float2 *rec40; uint8_t *flags;
    CHECK(hipMalloc(&stream, stream_bytes)); CHECK(hipMalloc(&flush, flush_bytes));
    CHECK(hipMalloc(&table, R * 44)); CHECK(hipMalloc(&perm, R * 4)); CHECK(hipMalloc(&perm4, R * 4)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&planes, (size_t)NPL * W * H * 4)); CHECK(hipMalloc(&rec40, R * 40)); CHECK(hipMalloc(&flags, R)); CHECK(hipMalloc(&bits, R / 8 + 64));
    CHECK(hipMemset(stream, 1, stream_bytes)); CHECK(hipMemset(table, 1, R * 44)); CHECK(hipMemset(planes, 0, (size_t)NPL * W * H * 4));
    CHECK(hipMemset(flags, 0, R)); CHECK(hipMemset(bits, 0, R / 8 + 64)); CHECK(hipMemset(rec40, 0, R * 40));
    {
        std::vector<uint32_t> h(R), h4(R);
        for (size_t i = 0; i < R; i++) h[i] = (uint32_t)i;
        std::mt19937 rng(7);
        std::shuffle(h.begin(), h.end(), rng);
        for (size_t i = 0; i < R; i++) h4[i] = h[i] % (uint32_t)(R / 4);
        CHECK(hipMemcpy(perm, h.data(), R * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(perm4, h4.data(), R * 4, hipMemcpyHostToDevice));
    }
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    auto flush_caches = [&]() { calib_flush_kernel<<<2048, 256>>>(flush, flush_bytes / 16); };
    for (int r = 0; r < reps; r++) {
        flush_caches(); calib_read16_kernel<<<2048, 256>>>(stream, stream_bytes / 16, sink);
        flush_caches(); calib_read4_kernel<<<2048, 256>>>((const uint32_t *)stream, stream_bytes / 4, sink);
        flush_caches(); calib_tile4_kernel<false><<<gx * gy, 256>>>(planes, W, H, NPL, sink);
        flush_caches(); calib_tile4_kernel<true><<<(gx * gy + 7) / 8 * 8, 256>>>(planes, W, H, NPL, sink);
        flush_caches(); calib_gather44_kernel<<<2048, 256>>>(table, perm, R, sink);
        flush_caches(); calib_gather44_kernel<<<2048, 256>>>(table, perm4, R, sink);
        flush_caches(); calib_store16_kernel<<<2048, 256>>>(stream, stream_bytes / 16);
        flush_caches(); calib_store40_kernel<<<2048, 256>>>(rec40, perm, R);
        flush_caches(); calib_store1_kernel<<<2048, 256>>>(flags, perm, R);
        flush_caches(); calib_or1_kernel<<<2048, 256>>>(bits, perm, R);
    }
    CHECK(hipDeviceSynchronize());
    // bytes moved per launch, in launch order (gather44 twice: the second is the x4 form); perm reads (4 bytes per lane) listed apart
    printf("{\"launch_order\": [\"read16\", \"read4\", \"tile4\", \"tile4xcd\", \"gather44\", \"gather44x4\", \"store16\", \"store40\", \"store1\", \"or1\"],\n");
    printf(" \"read_bytes\": {\"read16\": %zu, \"read4\": %zu, \"tile4\": %zu, \"tile4xcd\": %zu, \"gather44\": %zu, \"gather44x4\": %zu, \"store40\": %zu, \"store1\": %zu, \"or1\": %zu},\n",
           stream_bytes, stream_bytes, (size_t)NPL * W * H * 4, (size_t)NPL * W * H * 4, R * 44 + R * 4, R / 4 * 44 + R * 4, R * 4, R * 4, R * 4);
    printf(" \"gather44x4_requested_bytes\": %zu,\n", R * 44 + R * 4);
    printf(" \"write_bytes\": {\"store16\": %zu, \"store40\": %zu, \"store1\": %zu, \"or1_bits_as_bytes\": %zu, \"or1_as_dwords\": %zu}}\n", stream_bytes, R * 40, R, R / 8, R * 4);
    return 0;
}
