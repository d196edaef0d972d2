// C ABI entry points (include/lvdgs.h): argument checks, state-buffer layouts, launch sequencing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include <cstdlib>

#include <chrono>
#include <thread>

#include "common.hpp"
#include "photometric.hpp"

namespace lvdgs {

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return LVDGS_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return LVDGS_E_HIP;
}

// ---------------------------------------------------------------- profiling
namespace {
struct ProfSlot {
    std::string name;
    int64_t launches = 0;
    double total_ms = 0.0;
};
struct ProfPending {
    int slot;
    hipEvent_t start, stop;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfSlot> g_slots;
std::vector<ProfPending> g_pending;
std::vector<hipEvent_t> g_free_events;

hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
int slot_of(const char *name) {
    for (size_t i = 0; i < g_slots.size(); i++)
        if (g_slots[i].name == name) return (int)i;
    g_slots.push_back(ProfSlot{name});
    return (int)g_slots.size() - 1;
}
void drain_locked() {
    for (auto &p : g_pending) {
        (void)hipEventSynchronize(p.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            g_slots[p.slot].launches++;
            g_slots[p.slot].total_ms += ms;
        }
        g_free_events.push_back(p.start);
        g_free_events.push_back(p.stop);
    }
    g_pending.clear();
}
}  // namespace

ProfScope::ProfScope(const char *name, hipStream_t s) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    slot = slot_of(name);
    ProfPending p{slot, get_event(), get_event()};
    (void)hipEventRecord(p.start, s);
    g_pending.push_back(p);
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    // the matching entry is the last one pushed for this scope
    for (size_t i = g_pending.size(); i-- > 0;)
        if (g_pending[i].slot == slot) {
            (void)hipEventRecord(g_pending[i].stop, stream);
            break;
        }
}

int allow_dynamic_lds(const void *kernel, int bytes, unsigned char (&done)[16]) {
    int dev = 0;
    if (int e = check_hip(hipGetDevice(&dev), "hipGetDevice")) return e;
    // Two host threads may get here together (one stream each): the flag is read and written atomically, and setting the
    // attribute twice is harmless, so the worst case is one redundant call.
    unsigned char &flag = done[dev & 15];
    if (!__atomic_load_n(&flag, __ATOMIC_ACQUIRE)) {
        if (int e = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "reserve dynamic LDS")) return e;
        __atomic_store_n(&flag, (unsigned char)1, __ATOMIC_RELEASE);
    }
    return LVDGS_OK;
}

// Images of up to group_max_tiles() tiles group their pairs by counting (binning.hip); larger ones, or every image
// when LVDGS_FORCE_RADIX_GROUPING is set in the environment (tests), take the radix sort on the tile id.
static bool use_counting_path(int num_tiles) {
    static const bool forced = getenv("LVDGS_FORCE_RADIX_GROUPING") != nullptr;
    return !forced && num_tiles <= group_max_tiles();
}

// Small grids (a KITTI frame: 1848 tiles for 2048 places) are resident all at once, so nothing rebalances the CUs while
// they run: the blend kernels then take their tiles longest list first, dealt round-robin (binning.hip writes the order
// into the second half of long_tiles).
#ifndef LVDGS_TILE_ORDER_MAX_TILES
#define LVDGS_TILE_ORDER_MAX_TILES 4096   // A/B builds: 0 = never
#endif
bool tile_order_in_use(int num_tiles) {
    return use_counting_path(num_tiles) && num_tiles <= LVDGS_TILE_ORDER_MAX_TILES && num_tiles > 0;
}

// ---------------------------------------------------------------- layouts
template <typename T>
static void carve(T *&ptr, size_t count, char *base, size_t &off) {
    ptr = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += align256(count * sizeof(T));
}

size_t geom_layout(int N, GeomView *v, void *base) {
    GeomView tmp;
    if (!v) v = &tmp;
    size_t off = 0;
    char *b = (char *)base;
    const size_t n = (size_t)(N > 0 ? N : 1);
    carve(v->rec, n * REC_FLOATS, b, off);
    carve(v->tiles_touched, n, b, off);
    carve(v->depth_bits, n, b, off);
    carve(v->rect, n, b, off);
    carve(v->slot_base, n, b, off);
    carve(v->total, 64, b, off);
    return off;
}

static size_t scan_blocks(int N) { return (size_t)cdiv(N > 0 ? N : 1, 256) + 1; }  // one sum per preprocess_fwd workgroup

size_t prep_scratch_layout(int N, PrepScratch *v, void *base) {
    PrepScratch tmp;
    if (!v) v = &tmp;
    size_t off = 0;
    char *b = (char *)base;
    carve(v->blocksums, scan_blocks(N), b, off);
    return off;
}

size_t bin_layout(int64_t D, BinView *v, void *base) {
    BinView tmp;
    if (!v) v = &tmp;
    size_t off = 0;
    char *b = (char *)base;
    const size_t d = (size_t)(D > 0 ? D : 1);
    carve(v->point_list, d, b, off);
    carve(v->tile_keys, d, b, off);
    carve(v->pair_valid, d + 16, b, off);
    return off;
}

// Where the arrays lie is a function of the BUFFER's size, not of the pair count of the call: the forward pass lays the
// buffer out for its pair capacity, the backward pass is told the frame's actual pair count -- and both must find
// pair_valid (cleared by one, read by the other) in the same place.  The layout is the one of the largest pair count the
// buffer holds; a buffer of exactly lvdgs_binning_bytes(D) bytes has the layout bin_layout(D) describes.
int64_t bin_view(const lvdgs_args *a, BinView *v) {
    if (!a->binning_state || a->binning_bytes < bin_layout(1, nullptr, nullptr)) { *v = BinView{}; return 0; }
    int64_t p = a->binning_bytes > 2048 ? (int64_t)((a->binning_bytes - 3 * 256 - 16) / 9) : 1;   // 4 + 4 + 1 bytes per pair, three 256-byte roundings
    while (bin_layout(p + 1, nullptr, nullptr) <= a->binning_bytes) p++;
    while (p > 1 && bin_layout(p, nullptr, nullptr) > a->binning_bytes) p--;
    bin_layout(p, v, a->binning_state);
    return p;
}

size_t image_layout(int W, int H, ImageView *v, void *base) {
    ImageView tmp;
    if (!v) v = &tmp;
    size_t off = 0;
    char *b = (char *)base;
    const size_t P = (size_t)W * H, T = (size_t)cdiv(W, TILE) * cdiv(H, TILE);
    carve(v->ranges, T ? T : 1, b, off);
    carve(v->long_count, 64, b, off);  // directly behind ranges: one memset clears both
    carve(v->long_tiles, 2 * (T ? T : 1), b, off);  // queue of up to T tile ids (the second half is unused)
    carve(v->final_T, P ? P : 1, b, off);
    carve(v->n_contrib, P ? P : 1, b, off);
    return off;
}

size_t render_scratch_layout(int N, int64_t D, int W, int H, RenderScratch *v, void *base) {
    RenderScratch tmp;
    if (!v) v = &tmp;
    size_t off = 0;
    char *b = (char *)base;
    const size_t d = (size_t)(D > 0 ? D : 1);
    const int T = cdiv(W, TILE) * cdiv(H, TILE);
    carve(v->blocksums, scan_blocks(N), b, off);  // as in prep_scratch_layout: left by preprocess_fwd, read by count_pairs
    carve(v->keys, d, b, off);
    carve(v->vals, d, b, off);
    v->hist = v->totals = v->group_hist = v->group_totals = v->chunk_sums = nullptr;
    v->super = SuperView{};
    if (use_counting_path(T)) {
        carve(v->group_hist, group_hist_entries(N, T), b, off);
        carve(v->group_totals, (size_t)(T > 0 ? T : 1) + 4, b, off);   // (+4: the tile-range scan reads them 16 bytes at a time)
        carve(v->chunk_sums, group_chunks(N), b, off);
        // two-level grouping (LVDGS_FLAG_SUPER_TILES): the super-tile grid's own counting state
        const size_t Ts = (size_t)super_tiles_of(W, H);
        carve(v->super.rect, (size_t)(N > 0 ? N : 1), b, off);
        carve(v->super.hist, group_chunks(N) * (Ts ? Ts : 1), b, off);
        carve(v->super.totals, (Ts ? Ts : 1) + 4, b, off);
        carve(v->super.ranges, Ts ? Ts : 1, b, off);
        carve(v->super.long_count, 64, b, off);
        carve(v->super.long_tiles, 2 * (Ts ? Ts : 1), b, off);
        carve(v->super.total, 64, b, off);
    } else {
        carve(v->hist, radix_hist_entries(D), b, off);
        carve(v->totals, (size_t)1 << SORT_MAX_BITS, b, off);
    }
    return off;
}

size_t bwd_scratch_layout(int N, int64_t D, BwdScratch *v, void *base) {
    BwdScratch tmp;
    if (!v) v = &tmp;
    size_t off = 0;
    char *b = (char *)base;
    // (the pose-gradient partials first: their place must not depend on the pair count -- lvdgs_forward_backward_fused_loss
    // enqueues the backward before the host knows it, and lvdgs_tracking_tail looks for them with the count the caller then has)
    carve(v->tau_part, (size_t)(cdiv(N > 0 ? N : 1, 256)) * 6, b, off);
    carve(v->pair_grads, (size_t)(D > 0 ? D : 1) * PAIR_FLOATS + 8, b, off);   // (+8: the staging loads of preprocess_bwd are 16 bytes wide)
    return off;
}

int tile_sort_bits(int W, int H) {
    const int T = cdiv(W, TILE) * cdiv(H, TILE);
    int bits = 0;
    while ((1 << bits) < T) bits++;
    return bits < 1 ? 1 : bits;
}

// ---------------------------------------------------------------- argument checks
static int check_common(const lvdgs_args *a) {
    if (!a) { set_error("args is NULL"); return LVDGS_E_INVALID; }
    if (a->image_width <= 0 || a->image_height <= 0) { set_error("bad image size %dx%d", a->image_width, a->image_height); return LVDGS_E_INVALID; }
    if (a->num_gaussians < 0) { set_error("negative Gaussian count"); return LVDGS_E_INVALID; }
    if (!(a->tanfovx > 0.f) || !(a->tanfovy > 0.f)) { set_error("tanfov must be positive"); return LVDGS_E_INVALID; }
    if (a->sh_degree < 0 || a->sh_degree > 3) { set_error("sh_degree %d outside 0..3", a->sh_degree); return LVDGS_E_INVALID; }
    if (!a->bg || !a->viewmatrix || !a->projmatrix) { set_error("bg / viewmatrix / projmatrix is NULL"); return LVDGS_E_INVALID; }
    if ((int64_t)cdiv(a->image_width, TILE) * cdiv(a->image_height, TILE) > (1 << 22)) { set_error("image too large"); return LVDGS_E_RANGE; }
    return LVDGS_OK;
}

static int check_gaussians(const lvdgs_args *a) {
    if (a->num_gaussians == 0) return LVDGS_OK;
    if (!a->means3D || !a->opacities) { set_error("means3D / opacities is NULL"); return LVDGS_E_INVALID; }
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) { set_error("provide exactly one of shs / colors_precomp"); return LVDGS_E_INVALID; }
    const bool sr = a->scales && a->rotations;
    if (sr == (a->cov3D_precomp != nullptr) || (!!a->scales != !!a->rotations)) { set_error("provide exactly one of scales+rotations / cov3D_precomp"); return LVDGS_E_INVALID; }
    if (a->shs) {
        if (!a->campos) { set_error("campos is NULL"); return LVDGS_E_INVALID; }
        if (a->sh_coeffs < (a->sh_degree + 1) * (a->sh_degree + 1)) { set_error("sh_coeffs %d too small for degree %d", a->sh_coeffs, a->sh_degree); return LVDGS_E_INVALID; }
    }
    return LVDGS_OK;
}

}  // namespace lvdgs

using namespace lvdgs;

extern "C" {

const char *lvdgs_last_error(void) { return g_err; }
const char *lvdgs_version(void) { return "lvdgs 0.1.0 (gfx950)"; }

size_t lvdgs_geom_bytes(int32_t N) { return geom_layout(N, nullptr, nullptr); }
size_t lvdgs_prepare_scratch_bytes(int32_t N) { return prep_scratch_layout(N, nullptr, nullptr); }
size_t lvdgs_binning_bytes(int64_t D) { return bin_layout(D, nullptr, nullptr); }
size_t lvdgs_image_bytes(int32_t W, int32_t H) { return image_layout(W, H, nullptr, nullptr); }
size_t lvdgs_render_scratch_bytes(int32_t N, int64_t D, int32_t W, int32_t H) { return render_scratch_layout(N, D, W, H, nullptr, nullptr); }
size_t lvdgs_backward_scratch_bytes(int32_t N, int64_t D) { return bwd_scratch_layout(N, D, nullptr, nullptr); }

int lvdgs_state_layout_query(int32_t N, int64_t D, int32_t W, int32_t H, lvdgs_state_layout *out) {
    if (!out) { set_error("out is NULL"); return LVDGS_E_INVALID; }
    char *base = (char *)4096;  // offsets are computed by carving from a fake base
    GeomView g; BinView b; ImageView im;
    geom_layout(N, &g, base); bin_layout(D, &b, base); image_layout(W, H, &im, base);
    out->geom_rec = (char *)g.rec - base; out->geom_tiles_touched = (char *)g.tiles_touched - base;
    out->geom_slot_base = (char *)g.slot_base - base;
    out->bin_point_list = (char *)b.point_list - base; out->bin_tile_keys = (char *)b.tile_keys - base;
    out->img_ranges = (char *)im.ranges - base; out->img_final_T = (char *)im.final_T - base;
    out->img_n_contrib = (char *)im.n_contrib - base;
    out->geom_rec_floats = REC_FLOATS;
    return LVDGS_OK;
}

// Per-thread, per-device host resources of the single-call forward: 4 pinned bytes that receive the
// pair count and the event that says they have arrived.  Nothing else in the library is stateful.
namespace {
struct PairProbe {
    int device = -1;
    uint32_t *pinned = nullptr;   // [0] pair count, [1] longest queued tile segment of the frame, [2] length of that queue, [3] sequence number of the call that wrote them;
                                  // [16 + 4 k ...]: the same four words of view k of lvdgs_forward_batch
    uint32_t *pinned_dev = nullptr;   // the same words as the device addresses them
    uint32_t seq = 0;             // of the last single-call forward on this thread and device
    hipEvent_t ready = nullptr;
    int longest_super = -1, queued_super = 0;   // the same hints for the super-tile lists of the two-level grouping (LVDGS_FLAG_SUPER_TILES)
    int longest = 0, queued = 0;  // of the previous frame on this device: which kernels for long segments the next frame
    int keep = 0;                 // launches behind its tile sort (a hint, never a result); kept for a few frames
};
constexpr int PROBE_WORDS = 16 + 4 * FWD_BATCH_VIEWS + 8;
thread_local PairProbe g_probe[16];

int get_probe(PairProbe **out) {
    int dev = 0;
    if (int e = check_hip(hipGetDevice(&dev), "hipGetDevice")) return e;
    PairProbe &p = g_probe[dev & 15];
    if (p.device != dev || !p.pinned) {
        p.device = dev;
        if (int e = check_hip(hipHostMalloc((void **)&p.pinned, PROBE_WORDS * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent), "pinned pair count")) return e;
        for (int k = 0; k < PROBE_WORDS; k++) p.pinned[k] = 0u;
        if (int e = check_hip(hipHostGetDevicePointer((void **)&p.pinned_dev, p.pinned, 0), "device address of the pinned pair count")) return e;
        if (int e = check_hip(hipEventCreateWithFlags(&p.ready, hipEventDisableTiming), "pair count event")) return e;
    }
    *out = &p;
    return LVDGS_OK;
}

// preprocess -> prefix sum of tiles touched; the pair count ends up in g.total (device)
int enqueue_prepare(const lvdgs_args *a, const GeomView &g, hipStream_t s) {
    const int N = a->num_gaussians;
    PrepScratch w;
    prep_scratch_layout(N, &w, a->scratch);
    if (int e = launch_preprocess_fwd(*a, g, w.blocksums, s)) return e;
    return launch_slot_scan(g.tiles_touched, g.slot_base, w.blocksums, g.total, N, a->debug, s);
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}

// The tile scan of this call has written the pinned words when their fourth holds the call's sequence number.
// (words: the four words of the call -- or of one view of lvdgs_forward_batch)
int wait_for_sequence(PairProbe *probe, const uint32_t *words = nullptr) {
    volatile const uint32_t *flag = (words ? words : probe->pinned) + 3;
    const uint32_t want = probe->seq;
    const auto t0 = std::chrono::steady_clock::now();
    // The count arrives some tens of microseconds after the call got here (the projection and the two scans); the wait is a
    // spin on a pinned word, polite to the core's sibling thread (pause) and, past ~20 us, to the scheduler (yield) -- the
    // reference's front end and back end are two processes that each sit in this wait once per render.
    for (uint64_t spins = 0;; spins++) {
        if (__atomic_load_n(const_cast<uint32_t *>(flag), __ATOMIC_ACQUIRE) == want) return LVDGS_OK;
        if (spins < 2000) cpu_relax();
        else std::this_thread::yield();
        if ((spins & 0xffffu) == 0xffffu) {
            if (int e = check_hip(hipGetLastError(), "while waiting for the pair count")) return e;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
                set_error("timed out waiting for the pair count (the kernels before it never finished)");
                return LVDGS_E_HIP;
            }
        }
    }
}

// (radix path: the slot scan leaves the pair count alone -- ONE word is copied; the words behind it are the counting path's
// hints and the sequence number lvdgs_forward's wait spins on, which a copy of uninitialised device words must not touch)
int enqueue_count_probe(PairProbe *probe, const uint32_t *total, hipStream_t s) {
    if (int e = check_hip(hipMemcpyAsync(probe->pinned, total, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read pair count")) return e;
    return check_hip(hipEventRecord(probe->ready, s), "record pair count event");
}

// (pair emission ->) grouping by tile -> ranges -> depth order inside each tile -> blend.  `cap` sizes grids and buffers; when `count_on_device`
// the kernels take the actual pair count from g.total (clamped to cap), otherwise cap IS the count.
// counted (lvdgs_forward on the counting path): the projection kernel has left the per-chunk tile counts already; the
// tile-range scan then makes the pair count, which is copied to the host (probe) as soon as it exists, and the scatter
// makes slot_base.
int enqueue_render(const lvdgs_args *a, int64_t cap, bool count_on_device, hipStream_t s, bool counted = false, PairProbe *probe = nullptr) {
    const int N = a->num_gaussians, W = a->image_width, H = a->image_height;
    GeomView g{}; BinView b{}; ImageView im; RenderScratch w{};
    image_layout(W, H, &im, a->image_state);
    const int gx = cdiv(W, TILE), num_tiles = gx * cdiv(H, TILE);
    int row0, row1;
    tile_row_band(*a, &row0, &row1);
    const uint32_t *count = nullptr;
    if (N > 0) {
        geom_layout(N, &g, a->geom_state);
        // blend_fwd accumulates into n_touched: the counting path clears it in its first kernel, the radix path here
        if (!(cap > 0 && use_counting_path(num_tiles)))
            if (int e = check_hip(hipMemsetAsync(a->n_touched, 0, sizeof(int32_t) * (size_t)N, s), "memset n_touched")) return e;
        if (count_on_device) count = g.total;
    }
    bool grouped = false, super = false;
    if (cap > 0) {
        if (bin_view(a, &b) < cap) { set_error("internal: binning_state smaller than the pair capacity"); return LVDGS_E_INVALID; }
        render_scratch_layout(N, cap, W, H, &w, a->scratch);
        if (use_counting_path(num_tiles)) {
            // counting path: no pair list is materialised, the tile ranges fall out of the counts
            if (!counted)
                if (int e = launch_group_count(*a, g, im, w, s)) return e;
            super = super_tiles_in_use(*a);
            if (super && !counted)   // two-level grouping: the super-tile grid's counts first (single-call forward: the projection kernel has made them), the scan below then takes both grids in its two launches
                if (int e = launch_super_count(*a, g, w.super, s)) return e;
            // (single-call forward: the tile scan writes the pair count and the hints into the caller thread's pinned words itself)
            if (int e = launch_group_scan(*a, im, w, cap, counted ? g.total : nullptr, s, counted && probe ? probe->pinned_dev : nullptr,
                                          counted && probe ? probe->seq : 0u)) return e;
            if (super) {
                // two-level grouping: keys scattered and sorted per 64 x 64-pixel super-tile, the tiles' lists read off the sorted super lists
                // (binning.hip); b.tile_keys -- the radix path's -- takes the sorted super lists
                if (int e = launch_super_scatter(*a, g, w.super, w, (unsigned long long *)w.keys, cap, counted, b.pair_valid, s)) return e;
                ImageView ims{};
                ims.ranges = w.super.ranges; ims.long_count = w.super.long_count; ims.long_tiles = w.super.long_tiles;
                const int Ts = super_tiles_of(W, H);
                if (int e = launch_tile_depth_sort(ims, Ts, 0, Ts, g.rec, b.tile_keys, w.keys, true, probe ? probe->longest_super : -1,
                                                   probe ? probe->queued_super : 0, a->debug, s)) return e;
                if (int e = launch_super_expand(*a, g, w.super, im, b.tile_keys, b.point_list, s)) return e;
            } else if (int e = launch_group_scatter(*a, g, im, w, (unsigned long long *)w.keys, cap, counted, b.pair_valid, s, counted ? g.total : nullptr,
                                                    counted && probe ? probe->pinned_dev : nullptr, counted && probe ? probe->seq : 0u)) return e;
            grouped = true;
        } else {
            if (!w.hist) { set_error("internal: scratch was not laid out for the radix grouping"); return LVDGS_E_INVALID; }
            const int bits = tile_sort_bits(W, H);
            // start in the buffer that makes the sorted result land in binning_state
            const bool start_in_state = (radix_num_passes(bits) % 2) == 0;
            uint32_t *k0 = start_in_state ? b.tile_keys : w.keys, *v0 = start_in_state ? b.point_list : w.vals;
            uint32_t *k1 = start_in_state ? w.keys : b.tile_keys, *v1 = start_in_state ? w.vals : b.point_list;
            if (int e = check_hip(hipMemsetAsync(b.pair_valid, 0, (size_t)cap, s), "memset pair_valid")) return e;
            if (int e = launch_emit_pairs(*a, g, k0, v0, cap, s)) return e;
            bool in_first = true;
            if (int e = radix_sort_pairs(k0, v0, k1, v1, cap, bits, w.hist, w.totals, &in_first, a->debug, s, count)) return e;
            if ((in_first ? k0 : k1) != b.tile_keys) { set_error("internal: sorted list not in binning_state"); return LVDGS_E_INVALID; }
        }
    }
    if (!grouped)
        if (int e = launch_tile_ranges(b.tile_keys, cap, count, im, num_tiles, a->debug, s)) return e;
    // w.keys + w.vals: the (depth, id) keys the counting path scattered, or scratch for over-long segments after the radix path
    if (cap > 0 && !super) {
        // long segments: expected as the recent frames on this device had them (single-call forward), unknown otherwise
        if (int e = launch_tile_depth_sort(im, num_tiles, row0 * gx, row1 * gx, g.rec, b.point_list, w.keys, grouped, probe ? probe->longest : -1,
                                           probe ? probe->queued : 0, a->debug, s)) return e;
    }
    // (tile lists beyond what one wave sorts on recent frames: the blend kernel built for deep lists)
    if (a->flags & LVDGS_FLAG_NO_BLEND) return LVDGS_OK;   // the caller blends several views in one launch: lvdgs_blend_forward_batch
    return launch_blend_fwd(*a, g, b, im, probe && probe->longest > 0, s);
}

int check_render_buffers(const lvdgs_args *a, int64_t cap) {
    const int N = a->num_gaussians, W = a->image_width, H = a->image_height;
    if (!a->out_color || !a->out_depth || !a->out_opacity || !a->image_state) { set_error("output image or image_state is NULL"); return LVDGS_E_INVALID; }
    if (a->image_bytes < lvdgs_image_bytes(W, H)) { set_error("image_state too small"); return LVDGS_E_INVALID; }
    if (N > 0 && (!a->n_touched || !a->geom_state)) { set_error("n_touched / geom_state is NULL"); return LVDGS_E_INVALID; }
    if (cap > 0) {
        if (!a->binning_state || !a->scratch) { set_error("binning_state / scratch is NULL"); return LVDGS_E_INVALID; }
        if (a->binning_bytes < lvdgs_binning_bytes(cap) || a->scratch_bytes < lvdgs_render_scratch_bytes(N, cap, W, H)) {
            set_error("binning_state or scratch too small for %lld pairs", (long long)cap); return LVDGS_E_INVALID;
        }
    }
    return LVDGS_OK;
}
}  // namespace

int lvdgs_forward_prepare(const lvdgs_args *a, int64_t *num_rendered, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_common(a)) return e;
    if (int e = check_gaussians(a)) return e;
    if (!num_rendered) { set_error("num_rendered is NULL"); return LVDGS_E_INVALID; }
    const int N = a->num_gaussians;
    *num_rendered = 0;
    if (N == 0) return LVDGS_OK;
    if (!a->radii || !a->geom_state || !a->scratch) { set_error("radii / geom_state / scratch is NULL"); return LVDGS_E_INVALID; }
    if (a->geom_bytes < lvdgs_geom_bytes(N) || a->scratch_bytes < lvdgs_prepare_scratch_bytes(N)) {
        set_error("geom_state or scratch too small"); return LVDGS_E_INVALID;
    }
    GeomView g;
    geom_layout(N, &g, a->geom_state);
    if (int e = enqueue_prepare(a, g, s)) return e;
    uint32_t total = 0;
    if (int e = check_hip(hipMemcpyAsync(&total, g.total, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read pair count")) return e;
    if (int e = check_hip(hipStreamSynchronize(s), "synchronize after prepare")) return e;
    if (total > 0x7FFFFFFFu) { set_error("%u (Gaussian, tile) pairs exceed the 2^31 limit", total); return LVDGS_E_RANGE; }
    *num_rendered = (int64_t)total;
    return LVDGS_OK;
}

int lvdgs_forward_render(const lvdgs_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_common(a)) return e;
    const int64_t D = a->num_gaussians == 0 ? 0 : a->num_rendered;
    if (D < 0) { set_error("negative num_rendered"); return LVDGS_E_INVALID; }
    if (int e = check_render_buffers(a, D)) return e;
    return enqueue_render(a, D, false, s);
}

int lvdgs_forward(const lvdgs_args *a, int64_t *num_rendered, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_common(a)) return e;
    if (int e = check_gaussians(a)) return e;
    if (!num_rendered) { set_error("num_rendered is NULL"); return LVDGS_E_INVALID; }
    const int N = a->num_gaussians;
    *num_rendered = 0;
    if (N == 0) {
        if (int e = check_render_buffers(a, 0)) return e;
        return enqueue_render(a, 0, false, s);
    }
    const int64_t cap = a->pair_capacity;
    if (cap <= 0 || cap > 0x7FFFFFFFll) { set_error("pair_capacity must be in 1..2^31-1"); return LVDGS_E_INVALID; }
    if (!a->radii || !a->geom_state || !a->scratch) { set_error("radii / geom_state / scratch is NULL"); return LVDGS_E_INVALID; }
    if (a->geom_bytes < lvdgs_geom_bytes(N) || a->scratch_bytes < lvdgs_prepare_scratch_bytes(N)) {
        set_error("geom_state or scratch too small"); return LVDGS_E_INVALID;
    }
    if (int e = check_render_buffers(a, cap)) return e;
    PairProbe *probe = nullptr;
    if (int e = get_probe(&probe)) return e;
    GeomView g;
    geom_layout(N, &g, a->geom_state);
    // On the counting path the projection kernel counts the pairs per (chunk, tile) as it goes, the tile-range scan makes
    // the pair count and the scatter the slot prefix sum: three launches fewer than project / scan / count in turn.
    const bool counted = use_counting_path(cdiv(a->image_width, TILE) * cdiv(a->image_height, TILE));
    if (counted) {
        ImageView im; RenderScratch w{};
        image_layout(a->image_width, a->image_height, &im, a->image_state);
        render_scratch_layout(N, cap, a->image_width, a->image_height, &w, a->scratch);
        if (int e = launch_preprocess_count(*a, g, im, w, s)) return e;
    } else {
        if (int e = enqueue_prepare(a, g, s)) return e;
        if (int e = enqueue_count_probe(probe, g.total, s)) return e;
    }
    // Everything after the count is enqueued BEFORE the host waits for it: the GPU keeps working on
    // the tile sort and the blend while the host learns whether the capacity was enough.
    if (counted) { probe->seq++; if (probe->seq == 0u) probe->seq = 1u; }
    if (int e = enqueue_render(a, cap, true, s, counted, probe)) return e;
    if (counted) {
        if (int e = wait_for_sequence(probe)) return e;
    } else if (int e = check_hip(hipEventSynchronize(probe->ready), "wait for pair count")) return e;
    const uint32_t total = probe->pinned[0];
    if (counted) {   // what the next frames expect: this frame's long segments, or a recent frame's for a while (views alternate)
        const int longest = (int)probe->pinned[1], queued = (int)probe->pinned[2];
        if (longest >= probe->longest || probe->keep == 0) { probe->longest = longest; probe->queued = queued; probe->keep = longest ? 32 : 0; }
        else probe->keep--;
        if (super_tiles_in_use(*a)) {
            // (the super scan runs behind the tile scan whose sequence number was waited for: these two words may still be the previous
            // frame's -- they are hints for the next frame's sort launch either way)
            probe->longest_super = (int)probe->pinned[8 + 1]; probe->queued_super = (int)probe->pinned[8 + 2];
        }
    }
    if (total > 0x7FFFFFFFu) { set_error("%u (Gaussian, tile) pairs exceed the 2^31 limit", total); return LVDGS_E_RANGE; }
    *num_rendered = (int64_t)total;
    if ((int64_t)total > cap) {
        set_error("%u pairs exceed pair_capacity %lld: grow binning_state / scratch and call lvdgs_forward_render", total, (long long)cap);
        return LVDGS_E_CAPACITY;
    }
    return LVDGS_OK;
}

// views_out (lvdgs_blend_backward_fused_loss_batch): the call checks its arguments, lays its buffers out and stops there
struct BackwardViews { GeomView g; BinView b; ImageView im; BwdScratch w; };
static int backward_impl(const lvdgs_args *a, const LossParams *fused, int propagate_opacity, hipStream_t s, BackwardViews *views_out = nullptr,
                         const MaskedLossView *masked = nullptr) {
    if (int e = check_common(a)) return e;
    if (int e = check_gaussians(a)) return e;
    const int N = a->num_gaussians, W = a->image_width, H = a->image_height;
    const int64_t D = N == 0 ? 0 : a->num_rendered;
    if (D < 0) { set_error("negative num_rendered"); return LVDGS_E_INVALID; }
    if (!a->image_state || a->image_bytes < lvdgs_image_bytes(W, H)) { set_error("image_state is NULL or too small"); return LVDGS_E_INVALID; }
    GeomView g{}; BinView b{}; ImageView im; BwdScratch w{};
    image_layout(W, H, &im, a->image_state);
    const bool pose_only = (a->flags & LVDGS_FLAG_POSE_ONLY) != 0;
    if (N > 0) {
        if ((!fused && !masked && !a->dL_dout_color) || !a->projmatrix_raw || !a->radii) { set_error("a required backward pointer is NULL"); return LVDGS_E_INVALID; }
        if (masked && pose_only) { set_error("LVDGS_FLAG_POSE_ONLY: the static-mask mapping loss is a mapping loss, its backward makes every gradient"); return LVDGS_E_INVALID; }
        if (pose_only) {
            // a view-dependent colour moves with the camera centre: its gradient feeds dL/dtau (preprocess.hip), and the
            // pose-only passes do not make it
            if (a->shs && a->sh_degree > 0) { set_error("LVDGS_FLAG_POSE_ONLY needs sh_degree 0 or colors_precomp"); return LVDGS_E_INVALID; }
            if (a->flags & LVDGS_FLAG_ACCUMULATE_PARAM_GRADS) { set_error("LVDGS_FLAG_POSE_ONLY writes no parameter gradients: nothing to accumulate"); return LVDGS_E_INVALID; }
        } else {
            if (!a->dL_dmeans3D || !a->dL_dmeans2D || !a->dL_dopacities) { set_error("a required backward pointer is NULL"); return LVDGS_E_INVALID; }
            if (a->cov3D_precomp ? !a->dL_dcov3D : (!a->dL_dscales || !a->dL_drotations)) { set_error("covariance gradient output is NULL"); return LVDGS_E_INVALID; }
            if (a->shs ? !a->dL_dshs : !a->dL_dcolors) { set_error("colour gradient output is NULL"); return LVDGS_E_INVALID; }
        }
        if (!a->geom_state || !a->scratch || (D > 0 && !a->binning_state)) { set_error("a state / scratch buffer is NULL"); return LVDGS_E_INVALID; }
        if (a->geom_bytes < lvdgs_geom_bytes(N) || (D > 0 && a->binning_bytes < lvdgs_binning_bytes(D)) ||
            a->scratch_bytes < lvdgs_backward_scratch_bytes(N, D)) {
            set_error("a state / scratch buffer is too small"); return LVDGS_E_INVALID;
        }
        geom_layout(N, &g, a->geom_state);
        bwd_scratch_layout(N, D, &w, a->scratch);
        if (D > 0) bin_view(a, &b);
    }
    // With the loss inside, the blend pass runs even over empty lists (a view that sees nothing, an empty map): it is what
    // evaluates the loss of the background image -- value and exposure gradients -- and no pair record is written.
    if (views_out) { *views_out = BackwardViews{g, b, im, w}; return LVDGS_OK; }
    if (fused) {
        if (!(a->flags & LVDGS_FLAG_NO_BLEND))
            if (int e = launch_blend_bwd_fused_loss(*a, g, b, im, w, *fused, propagate_opacity, s)) return e;
    } else if (masked) {
        if (D > 0 && !(a->flags & LVDGS_FLAG_NO_BLEND))
            if (int e = launch_blend_bwd_masked_loss(*a, g, b, im, w, *masked, s)) return e;
    } else if (D > 0) {
        if (int e = launch_blend_bwd(*a, g, b, im, w, s)) return e;
    }
    if (N == 0) return a->dL_dtau ? check_hip(hipMemsetAsync(a->dL_dtau, 0, 6 * sizeof(float), s), "memset tau") : LVDGS_OK;
    return launch_preprocess_bwd(*a, g, w, b.pair_valid, s);
}

int lvdgs_backward(const lvdgs_args *a, void *stream) {
    if (a && (a->flags & LVDGS_FLAG_NO_BLEND)) { set_error("LVDGS_FLAG_NO_BLEND: lvdgs_backward runs its own blend pass; the calls that leave it to a batched one (lvdgs_blend_backward_fused_loss_batch / lvdgs_blend_backward_window_batch) are lvdgs_backward_fused_loss, lvdgs_backward_masked_loss and lvdgs_gaussian_backward_batch"); return LVDGS_E_INVALID; }
    return backward_impl(a, nullptr, 0, (hipStream_t)stream);
}

int lvdgs_backward_fused_loss(const lvdgs_args *a, const lvdgs_loss_args *loss, int32_t propagate_opacity_grad, void *stream) {
    if (!a) { set_error("backward: args is NULL"); return LVDGS_E_INVALID; }
    LossParams lp;
    if (int e = loss_fused_params(loss, &lp)) return e;
    if (loss->width != a->image_width || loss->height != a->image_height) { set_error("fused loss: image size differs from the rasterizer's"); return LVDGS_E_INVALID; }
    return backward_impl(a, &lp, propagate_opacity_grad != 0, (hipStream_t)stream);
}

// The forward passes of `count` views of one map and one image size, every stage ONE launch (include/lvdgs.h).
int lvdgs_forward_batch(const lvdgs_args *const *views, int32_t count, int64_t *num_rendered, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || (count > 0 && (!views || !num_rendered))) { set_error("forward batch: bad view list"); return LVDGS_E_INVALID; }
    if (count == 0) return LVDGS_OK;
    for (int k = 0; k < count; k++) num_rendered[k] = 0;
    const lvdgs_args *a0 = views[0];
    if (!a0) { set_error("forward batch: view 0 is NULL"); return LVDGS_E_INVALID; }
    const int N = a0->num_gaussians, W = a0->image_width, H = a0->image_height;
    if (int e = check_common(a0)) return e;
    const int num_tiles = cdiv(W, TILE) * cdiv(H, TILE);
    if (N <= 0 || !use_counting_path(num_tiles)) { set_error("forward batch: needs a map (N > 0) and an image of at most %d tiles (the views go through lvdgs_forward one by one otherwise)", group_max_tiles()); return LVDGS_E_INVALID; }
    std::vector<GeomView> g(count); std::vector<BinView> b(count); std::vector<ImageView> im(count); std::vector<RenderScratch> w(count);
    std::vector<int64_t> caps(count);
    for (int k = 0; k < count; k++) {
        const lvdgs_args *a = views[k];
        if (!a) { set_error("forward batch: view %d is NULL", k); return LVDGS_E_INVALID; }
        if (int e = check_common(a)) return e;
        if (int e = check_gaussians(a)) return e;
        if (a->num_gaussians != N || a->image_width != W || a->image_height != H || a->tile_row_begin != a0->tile_row_begin || a->tile_row_end != a0->tile_row_end ||
            a->means3D != a0->means3D || a->opacities != a0->opacities || a->scales != a0->scales || a->rotations != a0->rotations ||
            a->cov3D_precomp != a0->cov3D_precomp || a->shs != a0->shs || a->colors_precomp != a0->colors_precomp || a->sh_coeffs != a0->sh_coeffs ||
            a->activations != a0->activations || ((a->flags ^ a0->flags) & (LVDGS_FLAG_LIST_ALL_TILES | LVDGS_FLAG_NO_BLEND | LVDGS_FLAG_SUPER_TILES))) {
            set_error("forward batch: the views differ in map (means3D / opacities / scales / rotations / cov3D_precomp / shs / colors_precomp / activations), image size, band or flags"); return LVDGS_E_INVALID;
        }
        const int64_t cap = a->pair_capacity;
        if (cap <= 0 || cap > 0x7FFFFFFFll) { set_error("pair_capacity must be in 1..2^31-1"); return LVDGS_E_INVALID; }
        if (!a->radii || !a->geom_state || !a->scratch) { set_error("radii / geom_state / scratch is NULL"); return LVDGS_E_INVALID; }
        if (a->geom_bytes < lvdgs_geom_bytes(N) || a->scratch_bytes < lvdgs_prepare_scratch_bytes(N)) { set_error("geom_state or scratch too small"); return LVDGS_E_INVALID; }
        if (int e = check_render_buffers(a, cap)) return e;
        caps[k] = cap;
        geom_layout(N, &g[k], a->geom_state);
        image_layout(W, H, &im[k], a->image_state);
        render_scratch_layout(N, cap, W, H, &w[k], a->scratch);
        if (bin_view(a, &b[k]) < cap) { set_error("internal: binning_state smaller than the pair capacity"); return LVDGS_E_INVALID; }
    }
    PairProbe *probe = nullptr;
    if (int e = get_probe(&probe)) return e;
    probe->seq++; if (probe->seq == 0u) probe->seq = 1u;
    for (int first = 0; first < count; first += FWD_BATCH_VIEWS) {
        const int m = count - first < FWD_BATCH_VIEWS ? count - first : FWD_BATCH_VIEWS;
        const lvdgs_args *const *av = views + first;
        uint32_t *words = probe->pinned + 16, *words_dev = probe->pinned_dev + 16;
        if (int e = launch_preprocess_count_batch(av, &g[first], &im[first], &w[first], m, s)) return e;
        if (int e = launch_group_scan_batch(av, &g[first], &im[first], &w[first], &caps[first], m, words_dev, probe->seq, s)) return e;
        if (int e = launch_group_scatter_batch(av, &g[first], &im[first], &w[first], &b[first], &caps[first], m, s)) return e;
        const bool super = super_tiles_in_use(*a0);   // two-level grouping: the launches above worked on the super-tile grid; the tiles' lists are read off the sorted super lists
        if (int e = launch_tile_depth_sort_batch(av, &g[first], &im[first], &w[first], &b[first], m, super ? probe->longest_super : probe->longest,
                                                 super ? probe->queued_super : probe->queued, s)) return e;
        if (super)
            if (int e = launch_super_expand_batch(av, &g[first], &im[first], &w[first], &b[first], m, s)) return e;
        if (!(a0->flags & LVDGS_FLAG_NO_BLEND))
            if (int e = launch_blend_fwd_batch(av, &g[first], &b[first], &im[first], m, probe->longest > 0, s)) return e;
        // everything is enqueued: now the counts (the GPU is busy with the scatter, the sorts and the blend meanwhile)
        int longest = 0, queued = 0;
        for (int k = 0; k < m; k++) {
            if (int e = wait_for_sequence(probe, words + 4 * k)) return e;
            num_rendered[first + k] = (int64_t)words[4 * k];
            longest = std::max(longest, (int)words[4 * k + 1]); queued = std::max(queued, (int)words[4 * k + 2]);
        }
        if (super) { probe->longest_super = (int)probe->pinned[8 + 1]; probe->queued_super = (int)probe->pinned[8 + 2]; }   // (hints; possibly still the previous call's)
        if (m < count) { probe->seq++; if (probe->seq == 0u) probe->seq = 1u; }   // (the next group of views re-uses the words)
        if (longest >= probe->longest || probe->keep == 0) { probe->longest = longest; probe->queued = queued; probe->keep = longest ? 32 : 0; }
        else probe->keep--;
    }
    int status = LVDGS_OK;
    for (int k = 0; k < count; k++) {
        if (num_rendered[k] > 0x7FFFFFFFll) { set_error("%lld (Gaussian, tile) pairs exceed the 2^31 limit", (long long)num_rendered[k]); return LVDGS_E_RANGE; }
        if (num_rendered[k] > caps[k]) {
            set_error("view %d: %lld pairs exceed pair_capacity %lld: grow binning_state / scratch and call lvdgs_forward_render for it", k, (long long)num_rendered[k], (long long)caps[k]);
            status = LVDGS_E_CAPACITY;
        }
    }
    return status;
}

// lvdgs_forward + lvdgs_backward_fused_loss as one call; on small grids the two blend passes are ONE launch (blend.hip:
// blend_fwd_bwd_kernel).  Frames beyond this many tiles fill the chip in either blend kernel by themselves: the calls in turn.
#ifndef LVDGS_FUSED_BLEND_MAX_TILES
#define LVDGS_FUSED_BLEND_MAX_TILES 4096   // A/B builds: 0 = never
#endif
int lvdgs_forward_backward_fused_loss(const lvdgs_args *a, const lvdgs_loss_args *loss, int32_t propagate_opacity_grad, int64_t *num_rendered,
                                      void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!a || !num_rendered) { set_error("forward + backward: args / num_rendered is NULL"); return LVDGS_E_INVALID; }
    if (int e = check_common(a)) return e;
    const int N = a->num_gaussians, W = a->image_width, H = a->image_height;
    const int num_tiles = cdiv(W, TILE) * cdiv(H, TILE);
    const int64_t cap = a->pair_capacity;
    static const bool off = getenv("LVDGS_NO_FUSED_BLEND") != nullptr;   // (test / A-B hook: the two calls in turn whatever the grid)
    const bool fuse = !off && N > 0 && use_counting_path(num_tiles) && num_tiles <= LVDGS_FUSED_BLEND_MAX_TILES && !(a->flags & LVDGS_FLAG_NO_BLEND) &&
                      cap > 0 && cap <= 0x7FFFFFFFll;
    if (!fuse) {
        if (int e = lvdgs_forward(a, num_rendered, stream)) return e;
        lvdgs_args b = *a;
        b.num_rendered = *num_rendered;
        return lvdgs_backward_fused_loss(&b, loss, propagate_opacity_grad, stream);
    }
    *num_rendered = 0;
    if (int e = check_gaussians(a)) return e;
    LossParams lp;
    if (int e = loss_fused_params(loss, &lp)) return e;
    if (loss->width != W || loss->height != H) { set_error("fused loss: image size differs from the rasterizer's"); return LVDGS_E_INVALID; }
    if (!a->radii || !a->geom_state || !a->scratch) { set_error("radii / geom_state / scratch is NULL"); return LVDGS_E_INVALID; }
    if (a->geom_bytes < lvdgs_geom_bytes(N) || a->scratch_bytes < lvdgs_prepare_scratch_bytes(N)) { set_error("geom_state or scratch too small"); return LVDGS_E_INVALID; }
    if (int e = check_render_buffers(a, cap)) return e;
    // the backward's arguments, checked with the pair CAPACITY standing in for the count (the scratch must hold that many records)
    lvdgs_args bw = *a;
    bw.num_rendered = cap;
    BackwardViews v;
    if (int e = backward_impl(&bw, &lp, propagate_opacity_grad != 0, s, &v)) return e;
    PairProbe *probe = nullptr;
    if (int e = get_probe(&probe)) return e;
    GeomView g;
    geom_layout(N, &g, a->geom_state);
    {
        ImageView im; RenderScratch w{};
        image_layout(W, H, &im, a->image_state);
        render_scratch_layout(N, cap, W, H, &w, a->scratch);
        if (int e = launch_preprocess_count(*a, g, im, w, s)) return e;
    }
    probe->seq++; if (probe->seq == 0u) probe->seq = 1u;
    lvdgs_args fw = *a;
    fw.flags |= LVDGS_FLAG_NO_BLEND;   // (grouping and tile sort; the blend follows below, together with the backward's)
    if (int e = enqueue_render(&fw, cap, true, s, true, probe)) return e;
    // (both halves of the backward look at the frame's pair count on the device and stand down when it exceeds the capacity)
    if (int e = launch_blend_fwd_bwd_fused_loss(*a, v.g, v.b, v.im, v.w, lp, propagate_opacity_grad != 0, probe->longest > 0, g.total, (uint32_t)cap, s)) return e;
    if (int e = launch_preprocess_bwd(*a, v.g, v.w, v.b.pair_valid, s, g.total, (uint32_t)cap)) return e;
    if (int e = wait_for_sequence(probe)) return e;
    const uint32_t total = probe->pinned[0];
    {
        const int longest = (int)probe->pinned[1], queued = (int)probe->pinned[2];
        if (longest >= probe->longest || probe->keep == 0) { probe->longest = longest; probe->queued = queued; probe->keep = longest ? 32 : 0; }
        else probe->keep--;
        if (super_tiles_in_use(*a)) {
            // (the super scan runs behind the tile scan whose sequence number was waited for: these two words may still be the previous
            // frame's -- they are hints for the next frame's sort launch either way)
            probe->longest_super = (int)probe->pinned[8 + 1]; probe->queued_super = (int)probe->pinned[8 + 2];
        }
    }
    if (total > 0x7FFFFFFFu) { set_error("%u (Gaussian, tile) pairs exceed the 2^31 limit", total); return LVDGS_E_RANGE; }
    *num_rendered = (int64_t)total;
    if ((int64_t)total > cap) {
        set_error("%u pairs exceed pair_capacity %lld: grow binning_state / scratch, then lvdgs_forward_render and lvdgs_backward_fused_loss", total, (long long)cap);
        return LVDGS_E_CAPACITY;
    }
    return LVDGS_OK;
}

int lvdgs_blend_forward_batch(const lvdgs_args *const *views, int32_t count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || (count > 0 && !views)) { set_error("blend batch: bad view list"); return LVDGS_E_INVALID; }
    if (count == 0) return LVDGS_OK;
    std::vector<GeomView> g(count); std::vector<BinView> b(count); std::vector<ImageView> im(count);
    for (int k = 0; k < count; k++) {
        const lvdgs_args *a = views[k];
        if (!a) { set_error("blend batch: view %d is NULL", k); return LVDGS_E_INVALID; }
        if (int e = check_common(a)) return e;
        const int64_t D = a->num_gaussians == 0 ? 0 : a->num_rendered;
        if (D < 0) { set_error("negative num_rendered"); return LVDGS_E_INVALID; }
        if (int e = check_render_buffers(a, D)) return e;
        if (a->image_width != views[0]->image_width || a->image_height != views[0]->image_height || a->tile_row_begin != views[0]->tile_row_begin ||
            a->tile_row_end != views[0]->tile_row_end) { set_error("blend batch: the views differ in image size or band"); return LVDGS_E_INVALID; }
        g[k] = GeomView{}; b[k] = BinView{};
        if (a->num_gaussians > 0) geom_layout(a->num_gaussians, &g[k], a->geom_state);
        bin_view(a, &b[k]);
        image_layout(a->image_width, a->image_height, &im[k], a->image_state);
    }
    PairProbe *probe = nullptr;
    if (int e = get_probe(&probe)) return e;
    return launch_blend_fwd_batch(views, g.data(), b.data(), im.data(), count, probe->longest > 0, s);
}

// lvdgs_masked_loss_args as the backward reads it
static int masked_loss_view(const lvdgs_args *a, const lvdgs_masked_loss_args *m, MaskedLossView *out) {
    if (!m || !m->d_image || !m->out) { set_error("masked loss backward: loss / d_image / out is NULL"); return LVDGS_E_INVALID; }
    if (m->width != a->image_width || m->height != a->image_height) { set_error("masked loss backward: image size differs from the rasterizer's"); return LVDGS_E_INVALID; }
    if (m->gt_depth && !m->depth) { set_error("masked loss backward: gt_depth without the rendered depth"); return LVDGS_E_INVALID; }
    if (a->tile_row_begin != 0 || a->tile_row_end != 0) { set_error("masked loss backward: the loss is not a sum over pixels, a view scored by it cannot be rendered in bands"); return LVDGS_E_INVALID; }
    *out = MaskedLossView{m->d_image, m->depth, m->gt_depth, m->static_mask, m->depth_lambda, m->out};
    return LVDGS_OK;
}

int lvdgs_backward_masked_loss(const lvdgs_args *a, const lvdgs_masked_loss_args *loss, void *stream) {
    if (!a) { set_error("backward: args is NULL"); return LVDGS_E_INVALID; }
    MaskedLossView mv;
    if (int e = masked_loss_view(a, loss, &mv)) return e;
    return backward_impl(a, nullptr, 0, (hipStream_t)stream, nullptr, &mv);
}

int lvdgs_blend_backward_window_batch(const lvdgs_args *const *views, const lvdgs_loss_args *const *losses,
                                      const lvdgs_masked_loss_args *const *masked, int32_t count, int32_t propagate_opacity_grad, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || (count > 0 && (!views || (!losses && !masked)))) { set_error("blend batch: bad view list"); return LVDGS_E_INVALID; }
    if (count == 0) return LVDGS_OK;
    std::vector<GeomView> g(count); std::vector<BinView> b(count); std::vector<ImageView> im(count); std::vector<BwdScratch> w(count);
    std::vector<LossParams> lp(count);
    std::vector<MaskedLossView> mv(count); std::vector<const MaskedLossView *> mp(count, nullptr);
    for (int k = 0; k < count; k++) {
        const lvdgs_args *a = views[k];
        const lvdgs_masked_loss_args *m = masked ? masked[k] : nullptr;
        if (!a || (!m && !(losses && losses[k]))) { set_error("blend batch: view %d is NULL or has no loss", k); return LVDGS_E_INVALID; }
        if (a->image_width != views[0]->image_width || a->image_height != views[0]->image_height || a->tile_row_begin != views[0]->tile_row_begin ||
            a->tile_row_end != views[0]->tile_row_end || ((a->flags ^ views[0]->flags) & LVDGS_FLAG_POSE_ONLY)) {
            set_error("blend batch: the views differ in image size, band or LVDGS_FLAG_POSE_ONLY"); return LVDGS_E_INVALID;
        }
        BackwardViews v;
        if (m) {
            if (int e = masked_loss_view(a, m, &mv[k])) return e;
            mp[k] = &mv[k];
            if (int e = backward_impl(a, nullptr, 0, s, &v, &mv[k])) return e;
        } else {
            if (int e = loss_fused_params(losses[k], &lp[k])) return e;
            if (losses[k]->width != a->image_width || losses[k]->height != a->image_height) { set_error("fused loss: image size differs from the rasterizer's"); return LVDGS_E_INVALID; }
            if (int e = backward_impl(a, &lp[k], propagate_opacity_grad != 0, s, &v)) return e;
        }
        g[k] = v.g; b[k] = v.b; im[k] = v.im; w[k] = v.w;
    }
    return launch_blend_bwd_fused_loss_batch(views, g.data(), b.data(), im.data(), w.data(), lp.data(), masked ? mp.data() : nullptr, count,
                                             propagate_opacity_grad != 0, s);
}

int lvdgs_blend_backward_fused_loss_batch(const lvdgs_args *const *views, const lvdgs_loss_args *const *losses, int32_t count,
                                          int32_t propagate_opacity_grad, void *stream) {
    if (count > 0 && !losses) { set_error("blend batch: bad view list"); return LVDGS_E_INVALID; }
    return lvdgs_blend_backward_window_batch(views, losses, nullptr, count, propagate_opacity_grad, stream);
}

// The per-Gaussian passes of `count` views behind their (batched) backward blend pass, in one launch (include/lvdgs.h).
int lvdgs_gaussian_backward_batch(const lvdgs_args *const *views, int32_t count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || (count > 0 && !views)) { set_error("gaussian backward batch: bad view list"); return LVDGS_E_INVALID; }
    if (count == 0) return LVDGS_OK;
    std::vector<GeomView> g(count); std::vector<BinView> b(count); std::vector<BwdScratch> w(count);
    const lvdgs_args *a0 = views[0];
    if (!a0) { set_error("gaussian backward batch: view 0 is NULL"); return LVDGS_E_INVALID; }
    for (int k = 0; k < count; k++) {
        const lvdgs_args *a = views[k];
        if (!a) { set_error("gaussian backward batch: view %d is NULL", k); return LVDGS_E_INVALID; }
        if (a->flags & LVDGS_FLAG_POSE_ONLY) { set_error("gaussian backward batch: LVDGS_FLAG_POSE_ONLY makes no parameter gradients"); return LVDGS_E_INVALID; }
        if (a->num_gaussians != a0->num_gaussians || a->means3D != a0->means3D || a->opacities != a0->opacities || a->scales != a0->scales ||
            a->rotations != a0->rotations || a->shs != a0->shs || a->activations != a0->activations || a->dL_dmeans3D != a0->dL_dmeans3D ||
            a->dL_dopacities != a0->dL_dopacities || a->dL_dscales != a0->dL_dscales || a->dL_drotations != a0->dL_drotations || a->dL_dshs != a0->dL_dshs) {
            set_error("gaussian backward batch: the views differ in map or gradient buffers"); return LVDGS_E_INVALID;
        }
        if (!a->shs || a->colors_precomp || a->cov3D_precomp || a->sh_coeffs != 1 || a->sh_degree != 0) {
            set_error("gaussian backward batch: needs SH colours of one coefficient (degree 0) and scales + rotations (the views go through lvdgs_backward_fused_loss / _masked_loss one by one otherwise)");
            return LVDGS_E_INVALID;
        }
        if (k > 0 && !(a->flags & LVDGS_FLAG_ACCUMULATE_PARAM_GRADS)) { set_error("gaussian backward batch: view %d does not carry LVDGS_FLAG_ACCUMULATE_PARAM_GRADS (the views' gradients are summed)", k); return LVDGS_E_INVALID; }
    }
    for (int k = 0; k < count; k++) {
        const lvdgs_args *a = views[k];
        BackwardViews v;
        LossParams unused{};   // (the checks of a backward call whose pixel gradients came from a loss: the blend pass has run)
        if (int e = backward_impl(a, &unused, 0, s, &v)) return e;
        g[k] = v.g; b[k] = v.b; w[k] = v.w;
    }
    return launch_preprocess_bwd_views(views, g.data(), w.data(), b.data(), count, s);
}

int lvdgs_mark_visible(int32_t N, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                       void *stream) {
    (void)projmatrix;
    if (N < 0 || (N > 0 && (!means3D || !viewmatrix || !present))) { set_error("bad mark_visible arguments"); return LVDGS_E_INVALID; }
    return launch_mark_visible(N, means3D, viewmatrix, present, (hipStream_t)stream);
}

void lvdgs_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}
void lvdgs_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    drain_locked();
    g_slots.clear();
}
int lvdgs_profile_read(lvdgs_kernel_time *out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    drain_locked();
    int n = 0;
    for (auto &sl : g_slots) {
        if (n >= cap) break;
        memset(&out[n], 0, sizeof(out[n]));
        strncpy(out[n].name, sl.name.c_str(), sizeof(out[n].name) - 1);
        out[n].launches = sl.launches;
        out[n].total_ms = sl.total_ms;
        n++;
    }
    return n;
}

}  // extern "C"
