// What the back end derives from one view's render package (reference utils/slam_backend.py:311-315, :350-357): shared by
// lvdgs_view_stats (adam.hip) and lvdgs_map_view_tail (pose.hip), which does it in the launch that finishes the view's loss.
#pragma once
#include "common.hpp"

namespace lvdgs {

struct ViewStats {
    int N;
    const int32_t *radii, *n_touched;
    const float *viewspace_grad;   // N x 3 or null
    int32_t *radii_max;
    float *norm_sum, *vis_count;   // vis_count may be null (the view is counted by another band's rank)
    uint8_t *touched_row;          // null for views outside the window
    float *split_xy;               // N x 2 or null: the band's share of the screen-space gradient instead of its norm
};

// element i: radii_max = max(radii_max, radii); visible: norm_sum += |grad xy| (or split_xy = grad xy), vis_count += 1;
// touched_row = n_touched > 0
__device__ __forceinline__ void view_stats_one(const ViewStats &v, int i) {
    const int r = v.radii[i];
    const bool vis = r > 0;
    if (r > v.radii_max[i]) v.radii_max[i] = r;
    float gx = 0.f, gy = 0.f;
    if (vis && v.viewspace_grad) { gx = v.viewspace_grad[3 * (size_t)i]; gy = v.viewspace_grad[3 * (size_t)i + 1]; }
    if (v.split_xy) *reinterpret_cast<float2 *>(v.split_xy + 2 * (size_t)i) = make_float2(gx, gy);
    else if (vis && v.viewspace_grad) v.norm_sum[i] += sqrtf(gx * gx + gy * gy);
    if (vis && v.vis_count) v.vis_count[i] += 1.f;
    if (v.touched_row) v.touched_row[i] = v.n_touched[i] > 0 ? 1 : 0;
}

}  // namespace lvdgs
