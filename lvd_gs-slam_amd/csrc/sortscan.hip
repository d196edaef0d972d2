// Stable LSD radix sort of (u32 key, u32 value) pairs and the tiles-touched prefix sum.
//
// The sort groups the D (Gaussian, tile) pairs by tile id (<= 16 bits) instead of sorting D 64-bit
// (tile | depth) keys; being stable, it leaves every tile's segment in emission (Gaussian id) order, and
// tilesort.hip then orders each segment by (depth, id) in LDS.  Together that is the (tile, depth, id) order a
// stable sort of the 64-bit keys gives, at a fraction of the HBM traffic and launches.
//
// One radix pass = three launches, no inter-workgroup communication inside a launch:
//   radix_hist    per-workgroup digit histogram          -> hist[digit][workgroup]
//   radix_rowscan exclusive scan of each digit's row over workgroups + digit totals
//   radix_scatter stable rank inside the workgroup (wave ballot matching) + scattered store
// Digits are up to 8 bits wide (one digit per thread in the scatter); the scatter re-orders its chunk in LDS
// so that the stores of each digit's run are consecutive addresses (4-byte scattered stores cost ~4.5x the
// bytes in HBM write traffic, measured with WRITE_SIZE).
#include "common.hpp"
#include "device_utils.hpp"

namespace lvdgs {

namespace {

__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint32_t *__restrict__ keys, int64_t n_cap,
                                                                  const uint32_t *__restrict__ n_dev, int shift, int nbits,
                                                                  uint32_t *__restrict__ hist, int nblk) {
    __shared__ uint32_t s_hist[1 << SORT_MAX_BITS];
    // the element count may live on the device (pair count of this frame): clamp it to the capacity
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    const int nbins = 1 << nbits;
    const uint32_t mask = (uint32_t)nbins - 1u;
    for (int d = threadIdx.x; d < nbins; d += SORT_THREADS) s_hist[d] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
#pragma unroll 4
    for (int i = 0; i < SORT_IPT; i++) {
        const int64_t idx = base + (int64_t)i * SORT_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&s_hist[(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nbins; d += SORT_THREADS) hist[(size_t)d * nblk + blockIdx.x] = s_hist[d];
}

// one wave per digit row: in-place exclusive scan over the workgroups, row total -> totals[d]
__global__ void __launch_bounds__(256) radix_rowscan_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ totals,
                                                            int nbins, int nblk) {
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= nbins) return;
    const int lane = threadIdx.x & 63;
    uint32_t *row = hist + (size_t)d * nblk;
    uint32_t carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const uint32_t v = b < nblk ? row[b] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (b < nblk) row[b] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) totals[d] = carry;
}

__global__ void __launch_bounds__(SORT_THREADS) radix_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                                     const uint32_t *__restrict__ vals_in,
                                                                     uint32_t *__restrict__ keys_out,
                                                                     uint32_t *__restrict__ vals_out,
                                                                     const uint32_t *__restrict__ rowprefix,
                                                                     const uint32_t *__restrict__ totals, int64_t n_cap,
                                                                     const uint32_t *__restrict__ n_dev, int shift, int nbits,
                                                                     int nblk) {
    constexpr int NB = 1 << SORT_MAX_BITS;  // 256 >= nbins: one digit per thread
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_cap) : n_cap;
    if ((int64_t)blockIdx.x * SORT_CHUNK >= n) return;  // workgroup-uniform: capacity beyond this frame's count
    __shared__ uint32_t s_cnt[4][NB];       // per-wave digit counts, then per-wave local bases
    __shared__ uint32_t s_scan[SORT_THREADS];
    __shared__ uint32_t s_gofs[NB];         // global position of local index 0 of each digit's run
    __shared__ uint32_t s_key[SORT_CHUNK], s_val[SORT_CHUNK];  // the chunk in digit order
    const int nbins = 1 << nbits;
    const uint32_t mask = (uint32_t)nbins - 1u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int d = tid; d < 4 * NB; d += SORT_THREADS) (&s_cnt[0][0])[d] = 0;

    // ---- load this wave's contiguous sub-chunk ----
    const int64_t wbase = (int64_t)blockIdx.x * SORT_CHUNK + (int64_t)wave * (64 * SORT_IPT);
    uint32_t key[SORT_IPT], val[SORT_IPT], rank[SORT_IPT];
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < SORT_IPT; i++) {
        const int64_t idx = wbase + (int64_t)i * 64 + lane;
        const bool valid = idx < n;
        key[i] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[i] = valid ? vals_in[idx] : 0u;
    }
    // global base of this thread's digit: exclusive scan of the digit totals + this workgroup's row prefix
    const uint32_t my_total = tid < nbins ? totals[tid] : 0u;
    const uint32_t my_rowprefix = tid < nbins ? rowprefix[(size_t)tid * nblk + blockIdx.x] : 0u;
    __syncthreads();  // s_cnt zeroed

    // ---- stable rank inside the wave: same-digit lanes found by ballots, running per-wave counters in LDS ----
#pragma unroll
    for (int i = 0; i < SORT_IPT; i++) {
        const int64_t idx = wbase + (int64_t)i * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (key[i] >> shift) & mask;
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < nbits; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        uint32_t prior = 0;
        if (valid) prior = s_cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) s_cnt[wave][d] = prior + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[i] = prior + before;
    }
    __syncthreads();

    // ---- one digit per thread: local (in-chunk) base of the digit, per-wave bases, global offset of the run ----
    const uint32_t c0 = s_cnt[0][tid], c1 = s_cnt[1][tid], c2 = s_cnt[2][tid], c3 = s_cnt[3][tid];
    const uint32_t cnt = c0 + c1 + c2 + c3;
    // two independent workgroup scans (digit totals -> global digit base, chunk counts -> local base), interleaved
    s_scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < SORT_THREADS; off <<= 1) {
        const uint32_t t = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += t;
        __syncthreads();
    }
    const uint32_t lbase = s_scan[tid] - cnt;
    __syncthreads();
    s_scan[tid] = my_total;
    __syncthreads();
    for (int off = 1; off < SORT_THREADS; off <<= 1) {
        const uint32_t t = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += t;
        __syncthreads();
    }
    const uint32_t dbase = s_scan[tid] - my_total;
    s_cnt[0][tid] = lbase;
    s_cnt[1][tid] = lbase + c0;
    s_cnt[2][tid] = lbase + c0 + c1;
    s_cnt[3][tid] = lbase + c0 + c1 + c2;
    s_gofs[tid] = dbase + my_rowprefix - lbase;
    __syncthreads();

    // ---- reorder through LDS so that each digit's run leaves as consecutive addresses ----
#pragma unroll
    for (int i = 0; i < SORT_IPT; i++) {
        const int64_t idx = wbase + (int64_t)i * 64 + lane;
        if (idx < n) {
            const uint32_t d = (key[i] >> shift) & mask;
            const uint32_t lpos = s_cnt[wave][d] + rank[i];
            s_key[lpos] = key[i];
            s_val[lpos] = val[i];
        }
    }
    __syncthreads();
    const int count = (int)min((int64_t)SORT_CHUNK, n - (int64_t)blockIdx.x * SORT_CHUNK);
#pragma unroll
    for (int i = 0; i < SORT_IPT; i++) {
        const int li = i * SORT_THREADS + tid;
        if (li < count) {
            const uint32_t k = s_key[li];
            const uint32_t pos = s_gofs[(k >> shift) & mask] + (uint32_t)li;
            keys_out[pos] = k;
            vals_out[pos] = s_val[li];
        }
    }
}

// ---- prefix sum of tiles_touched in id order ----
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_IPT = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_IPT;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s, uint32_t *total) {
    const int tid = threadIdx.x;
    s[tid] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        const uint32_t t = tid >= off ? s[tid - off] : 0u;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    const uint32_t incl = s[tid];
    *total = s[SCAN_THREADS - 1];
    __syncthreads();
    return incl - v;
}

// slot_base[i] = exclusive scan over i of tiles_touched[i]: where Gaussian i's pairs (and, in the backward pass, its
// per-pair partial gradients) start in the unsorted pair list.
// One launch: preprocess_fwd left the pair count of every 256 Gaussians in `blocksums`; every workgroup here first
// adds up the sums in front of its chunk (at most a few thousand values: cheaper than a launch for a scan of them),
// then scans its own chunk of SCAN_CHUNK = 8 x 256 Gaussians.
__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const uint32_t *__restrict__ tt, int N,
                                                                  const uint32_t *__restrict__ blocksums,
                                                                  uint32_t *__restrict__ slot_base, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t s[SCAN_THREADS];
    uint32_t before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x * (SCAN_CHUNK / 256); b += SCAN_THREADS) before += blocksums[b];
    uint32_t prefix;
    block_exclusive_scan(before, s, &prefix);  // only the total is of interest
    uint32_t v[SCAN_IPT];
    uint32_t sum = 0;
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_IPT;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) {
        v[k] = base + k < N ? tt[base + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, s, &total) + prefix;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) {
        if (base + k < N) slot_base[base + k] = run;
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = prefix + total;
}

}  // namespace

int radix_num_passes(int total_bits) { return total_bits <= 0 ? 0 : (total_bits + SORT_MAX_BITS - 1) / SORT_MAX_BITS; }

size_t radix_hist_entries(int64_t n) { return (size_t)(1 << SORT_MAX_BITS) * (size_t)(cdiv(n > 0 ? n : 1, SORT_CHUNK)); }

int radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, int64_t n, int total_bits,
                     uint32_t *hist, uint32_t *totals, bool *result_in_a, int dbg, hipStream_t s, const uint32_t *n_dev) {
    *result_in_a = true;
    const int passes = radix_num_passes(total_bits);
    if (n <= 0 || passes == 0) {
        *result_in_a = (passes % 2) == 0;
        return LVDGS_OK;
    }
    const int nblk = cdiv(n, SORT_CHUNK);
    uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
    int shift = 0;
    for (int p = 0; p < passes; p++) {
        // spread the bits evenly over the passes (13 bits -> 7 + 6)
        const int nbits = (total_bits - shift + (passes - p) - 1) / (passes - p);
        const int nbins = 1 << nbits;
        {
            ProfScope ps("radix_hist", s);
            hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(SORT_THREADS), 0, s, kin, n, n_dev, shift, nbits, hist, nblk);
            LVDGS_LAUNCH_CHECK("radix_hist", dbg, s);
        }
        {
            ProfScope ps("radix_rowscan", s);
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(cdiv(nbins, 4)), dim3(256), 0, s, hist, totals, nbins, nblk);
            LVDGS_LAUNCH_CHECK("radix_rowscan", dbg, s);
        }
        {
            ProfScope ps("radix_scatter", s);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(SORT_THREADS), 0, s, kin, vin, kout, vout, hist, totals,
                               n, n_dev, shift, nbits, nblk);
            LVDGS_LAUNCH_CHECK("radix_scatter", dbg, s);
        }
        shift += nbits;
        uint32_t *t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    *result_in_a = (kin == keys_a);
    return LVDGS_OK;
}

int launch_slot_scan(const uint32_t *tiles_touched, uint32_t *slot_base, uint32_t *blocksums, uint32_t *total_dev, int N, int dbg,
                     hipStream_t s) {
    if (N == 0) {
        return check_hip(hipMemsetAsync(total_dev, 0, sizeof(uint32_t), s), "memset total");
    }
    const int nblk = cdiv(N, SCAN_CHUNK);
    {
        ProfScope ps("scan_apply", s);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk), dim3(SCAN_THREADS), 0, s, tiles_touched, N, (const uint32_t *)blocksums, slot_base,
                           total_dev);
        LVDGS_LAUNCH_CHECK("scan_apply", dbg, s);
    }
    return LVDGS_OK;
}

}  // namespace lvdgs
