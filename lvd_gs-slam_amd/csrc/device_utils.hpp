// Wave-level device helpers for gfx950 (wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace lvdgs {

// v_mov_b32 with a DPP modifier; lanes whose source is out of range (or masked) read 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_or_zero(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, BANK_MASK, false));
}

// Sum over the 64 lanes of the wave; the total is valid in lane 63 only.
// row_shr:1,2,4,8 build inclusive prefixes inside each 16-lane row, row_bcast:15 / :31 chain the rows.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_or_zero<0x111>(v);
    v += dpp_or_zero<0x112>(v);
    v += dpp_or_zero<0x114>(v);
    v += dpp_or_zero<0x118>(v);
    v += dpp_or_zero<0x142, 0xa>(v);
    v += dpp_or_zero<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace lvdgs
