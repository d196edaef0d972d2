// simple-knn and RoPE entry points (implemented in a later milestone of this round).
#include "common.hpp"

using namespace lvdgs;

extern "C" {

size_t lvdgs_knn_scratch_bytes(int32_t num_points) { (void)num_points; return 256; }

int lvdgs_dist2_knn3(int32_t, const float *, float *, void *, size_t, void *) {
    set_error("lvdgs_dist2_knn3: not implemented in this build");
    return LVDGS_E_INVALID;
}

int lvdgs_rope2d(float *, const int64_t *, int32_t, int32_t, int32_t, int32_t, float, float, void *) {
    set_error("lvdgs_rope2d: not implemented in this build");
    return LVDGS_E_INVALID;
}

}  // extern "C"
