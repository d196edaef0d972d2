// Wave-level device helpers for gfx950 (wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace lvdgs {

// v_mov_b32 with a DPP modifier; lanes whose source is out of range (or masked) read 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_or_zero(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, BANK_MASK, false));
}

// Rotate a register by N lanes inside every 16-lane row (row_ror:N): lane i receives the value of lane (i - N) mod 16 of
// its row.  Every lane has a source, so the "old" operand is never used; passing the input itself spares the
// zero-initialisation the compiler emits for a constant.
template <int N>
__device__ __forceinline__ float row_rotate(float x) {
    const int v = __float_as_int(x);
    return __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false));
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

// Pairwise folds for multi-value wave reductions (gfx950 v_permlane32_swap / v_permlane16_swap).
// fold32(a, b): lanes 0-31 hold a[l] + a[l+32], lanes 32-63 hold b[l-32] + b[l].
__device__ __forceinline__ float fold32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// fold16(a, b): 16-lane rows 0 and 2 hold a's row pairs (0+1, 2+3) summed, rows 1 and 3 hold b's.
__device__ __forceinline__ float fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// In-row (16 lanes) inclusive sums of three registers at once; lane 15 of every row gets the row
// total.  Interleaving the three keeps two instructions between a register's write and its DPP read.
__device__ __forceinline__ void row_sums3(float &x, float &y, float &z) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y), "+v"(z));
}

// Clear / set one bit of a wave-uniform 64-bit mask in one scalar instruction (the compiler's shift + and-not, or
// subtract-with-carry + and for m &= m - 1, are two to three on a scalar unit four SIMDs share).
__device__ __forceinline__ uint64_t mask_clear_bit(uint64_t m, int bit) {
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit));
    return m;
}
__device__ __forceinline__ uint64_t mask_set_bit(uint64_t m, int bit) {
    asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(bit));
    return m;
}

// v with lane `lane` (wave-uniform) replaced by the wave-uniform value x (v_writelane_b32; clang has no builtin for
// it, the intrinsic is reached by name; the compiler puts the lane select into M0 itself, as gfx9's one-SGPR-per-
// instruction rule demands).
extern "C" __device__ int lvdgs_writelane_i32(int x, int lane, int v) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int write_lane(int v, int x, int lane) { return lvdgs_writelane_i32(x, lane, v); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace lvdgs
