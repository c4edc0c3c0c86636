// gsr_api.hip -- the C ABI of libgsr_hip.so (include/gsr.h): argument validation, workspace
// carving, stage sequencing on the caller's stream, error reporting.  Host code only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "../../include/gsr.h"
#include "../../include/gsr_knn.h"
#include "../../include/gsr_loss.h"
#include "../../include/gsr_optim.h"
#include "gsr_internal.h"

namespace gsr {
hipError_t launch_adam(int n_groups, const gsr_adam_group_t *groups, double beta1, double beta2, double eps, hipStream_t s);   // adam.hip

static thread_local char g_err[512] = "";
// process-wide (not thread-local): PyTorch's autograd engine calls gsr_backward from its own thread
static std::atomic<int> g_profiling{0};
static std::atomic<int> g_exact_cull{1};       // output-invariant exact splat-vs-tile culling
static std::atomic<int> g_bwd_npx{2};          // 8x8 pixel blocks per wave in the reverse compositing kernel (1, 2 or 4)
static std::atomic<int> g_fwd_npx{2};          // same for the forward compositing kernel
static std::atomic<int> g_wpb{1};              // waves per workgroup of the compositing kernels (waves are independent)
static std::atomic<int> g_two_level_sort{1};   // 1: depth order first, then per-tile lists; 0: one global sort on tile<<32|depth
static std::atomic<int> g_tile_lists{2};       // 2: supertile_sort.hip (per-super-tile LDS order, no global sort); 1: depth_order.hip + tile_lists.hip
                                               // (round 1's path, also the fall-back of 2); 0: key emission + rocPRIM sort + range detection
static std::atomic<int> g_depth_buckets{1};    // 0: rocPRIM radix sort + scan; 1: depth_order.hip when P is large enough; 2: always (tests)
int g_composite_lds_pad = 0;                    // debug: extra dynamic LDS bytes per compositing workgroup (occupancy experiments)
static std::atomic<int> g_count_lanes{0};      // 1: instrumented compositing kernels (lane-slot accounting, slower)
static std::atomic<int> g_deterministic_bwd{0};   // 1: fixed-order reduction of the reverse pass's partial gradients
static std::atomic<int> g_seg_len{256};           // entries per segment of the reverse pass's work units (multiple of 64); 0: whole half tiles
static std::atomic<int> g_dense_pergauss{2};      // per-Gaussian backward on the Gaussians with a gradient only, zero rows filled on a second stream: 0 off, 1 on, 2 = from GSR_DENSE_MIN_P Gaussians
static std::atomic<int> g_prefill_at{1};           // announced gradient outputs (gsr_backward_prefill): zero-filled 1 = beside the forward compositing kernel, 2 = beside the list-ordering kernel already, 0 = announcements ignored
static std::atomic<int> g_dense_fork{2};           // dense per-Gaussian stage: 1 = the second stream is forked after the accumulator rows are cleared, 0 = before, 2 = after below GSR_DENSE_FORK_EARLY_P Gaussians
static std::atomic<int> g_fwd_pair_long{-1};       // forward pass on small images (seg_plan: persistent reverse kernel in use): half tiles whose list exceeds this many entries are walked by two waves, one per block; 0 = off, -1 = GSR_PAIR_LONG_DEFAULT
static std::atomic<int> g_bwd_lpt{1};             // large images: the reverse pass's half tiles in order of decreasing length (composite_bwd_lpt_kernel); 0 = in tile order
static std::atomic<int> g_asm_walk{1};            // 1: compositing walks written in gfx950 assembly where they exist (same results, bit for bit), 0: the C++ walks
static std::atomic<int> g_fill_in_tail{0};        // 1: with the persistent reverse kernel, the zero rows of Gaussians without a gradient are written by its idle waves
                                                  // (measured at config 3: pergauss_bwd 84 -> 62 us, but the compositing kernel + 40..66 us: off)
static std::atomic<int> g_persistent_bwd{2};      // persistent reverse compositing kernel drawing length-ordered work units (2 blocks per wave only): 0 never, 1 always,
                                                  // 2 (default) when the image has at most GSR_PERSISTENT_MAX_TILES tiles, i.e. when its half tiles fill the chip less
                                                  // than 1.5 times over and the longest chain, not the throughput, sets the kernel's time (measured: -25 % at 800 x 800,
                                                  // -10 % at 1600 x 900, +-0 at 1080p and 4K where the classic kernel's second generation of waves hides the long chains)
#define GSR_PERSISTENT_MAX_TILES 6144
// what gsr_forward and gsr_backward decide from the image size and the options alone: is the segmented machinery on, with what
// segment length (the forward pass then takes checkpoints and leaves the half tiles' lengths), is the reverse kernel the persistent one
struct SegPlan { int seg_len; bool persistent_bwd, small_image; };
static SegPlan seg_plan(int W, int H);
#define GSR_DEPTH_BUCKETS_MIN_P 1024           // measured at P = 10 k: 25 us against 48 us for rocPRIM sort + scan + copy-back

// State that adapts to what a device has rendered lives per DEVICE, not per process: a frame with depth outliers on one
// GPU must not change the path of another GPU driven by the same process (SURVEY 8b "several devices in one process").
// The knobs above are deliberate process-wide settings (gsr_set_option); everything below is keyed by hipGetDevice().
#define GSR_MAX_DEVICES 32
struct DeviceState {
    std::atomic<int> bucket_fail_p{0x7fffffff};   // smallest P whose buckets overflowed even under the log map: not tried again
    std::atomic<int> depth_log_map{0};            // set once a frame overflowed a depth bucket under the linear map: log map from then on
    std::atomic<int> poll_timeouts{0};            // N read-backs whose pinned-word poll timed out (diagnostic)
    std::atomic<uint32_t> frame_seq{0};           // forward passes so far: the mark composite_fwd leaves in GeomView::touched cycles with it
    std::mutex mu;                                // guards stage_ms and counters
    float stage_ms[GSR_NUM_STAGES] = {0};         // last profiled forward / backward on this device
    CompositeCounters *counters = nullptr;        // [2] device memory: forward, reverse (allocated on first use of count_lanes)
    // gsr_backward's second stream (lowest priority): the zero-fill of the gradient outputs runs there, beside the compositing kernel
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_prefill = nullptr;
    bool side_failed = false;
    // gsr_backward_prefill: the outputs announced for the backward call that follows the next forward pass (pending), and the ones
    // that forward pass zero-filled (done; single use, dropped by the next gsr_forward or gsr_backward on the device)
    struct Prefill { bool pending = false, done = false; int P = 0, M = 0; float *p[9] = {nullptr}; } prefill;
};
static DeviceState g_dev[GSR_MAX_DEVICES];
static DeviceState &dev_state();
// device buffer of the instrumented compositing kernels (debug facility: the only allocation the library makes besides
// its pinned read-back words); NULL when counting is off or the allocation failed
static CompositeCounters *lane_counters(int which) {
    if (!g_count_lanes.load()) return nullptr;
    DeviceState &ds = dev_state();
    std::lock_guard<std::mutex> lk(ds.mu);
    if (!ds.counters) {
        // two counter blocks, then two per-unit trace arrays (wave timeline of the instrumented kernels)
        const size_t tr_bytes = (size_t)GSR_TRACE_UNITS * sizeof(uint4);
        if (hipMalloc((void **)&ds.counters, 2 * sizeof(CompositeCounters) + 2 * tr_bytes) != hipSuccess) { ds.counters = nullptr; return nullptr; }
        (void)hipMemset(ds.counters, 0, 2 * sizeof(CompositeCounters) + 2 * tr_bytes);
        CompositeCounters h[2];
        memset(h, 0, sizeof(h));
        for (int w = 0; w < 2; w++) {
            h[w].trace = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(ds.counters + 2) + (size_t)w * tr_bytes);
            h[w].trace_cap = GSR_TRACE_UNITS;
        }
        (void)hipMemcpy(ds.counters, h, sizeof(h), hipMemcpyHostToDevice);
    }
    return ds.counters + which;
}
static DeviceState &dev_state() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return g_dev[d % GSR_MAX_DEVICES];
}
// the device's second stream and its two events, made on first use (false: could not be made -- the caller keeps everything on one stream)
static bool side_stream(DeviceState &ds) {
    if (ds.side) return true;
    if (ds.side_failed) return false;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&ds.side, hipStreamNonBlocking, least) != hipSuccess ||
        hipEventCreateWithFlags(&ds.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ds.ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ds.ev_prefill, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        ds.side = nullptr; ds.side_failed = true;
        return false;
    }
    return true;
}
#define GSR_PAIR_LONG_DEFAULT 64
#define GSR_LPT_SPAN 512              // length classes of the reverse pass's order on large images: 16 of 32 entries (SegView, plan_units)
#define GSR_DENSE_MIN_P 500000
#define GSR_DENSE_FORK_EARLY_P 2000000
static const char *const k_stage_names[GSR_NUM_STAGES] = {
    // lists.bin = entries binned per super-tile (count + scan + scatter; round 1's path: depth order + scan); lists.order = per-super-tile order +
    // expansion into the tile lists (sort path: the radix sort); emit_keys / ranges only run on the sort path
    "fwd.preprocess", "fwd.lists.bin", "fwd.readback_N", "fwd.lists.emit_keys", "fwd.lists.order", "fwd.lists.ranges", "(unused)",
    "fwd.composite", "bwd.clear+plan", "bwd.composite", "bwd.pergauss", "fwd.total", "bwd.total"};

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr, what)                                                                              \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) return fail(GSR_ERR_HIP, "%s: %s (%d)", what, hipGetErrorString(_e), (int)_e); \
    } while (0)

GeomView carve_geom(void *base, int P, size_t scan_tb, size_t dsort_tb) {
    GeomView g;
    const size_t n = (size_t)(P > 0 ? P : 1);
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    g.rec = (float *)take(n * GSR_REC_FLOATS * sizeof(float));
    g.depth = (float *)take(n * sizeof(float));
    g.opac = (float *)take(n * sizeof(float));
    g.rect = (uint4 *)take(n * sizeof(uint4));
    g.tiles = (uint32_t *)take(n * sizeof(uint32_t));
    g.offsets = (uint32_t *)take(n * sizeof(uint32_t));
    g.clamped = (uint8_t *)take(n);
    g.perm = (uint32_t *)take(n * sizeof(uint32_t));
    g.depth_sorted = (uint32_t *)take(n * sizeof(uint32_t));
    g.orect = (uint4 *)take(n * sizeof(uint4));
    g.ss_rec = (uint4 *)take(n * sizeof(uint4));
    g.hot = (uint32_t *)take(n * sizeof(uint32_t));
    g.ss_entries = (uint4 *)take((size_t)GSR_SS_ENT_PER_G * n * sizeof(uint4));
    // the bin count is an image property the workspace size cannot depend on (gsr_workspace_sizes is asked per (P, W, H) but
    // carve_geom only sees P): room for GSR_SS_WGCNT_WORDS words; supertile_sort.hip is skipped when nblk * S exceeds it
    g.ss_wg_cnt = (uint32_t *)take((size_t)GSR_SS_WGCNT_WORDS * sizeof(uint32_t));
    g.tl_mat1 = (uint32_t *)take((size_t)GSR_TL_MAX_S * ((n + GSR_TL_L1 - 1) / GSR_TL_L1) * sizeof(uint32_t));
    g.tl_bin_total = (uint32_t *)take(GSR_TL_MAX_S * sizeof(uint32_t));
    g.scan_temp = take(scan_tb);
    g.scan_temp_bytes = scan_tb;
    g.dsort_temp = take(dsort_tb);
    g.dsort_temp_bytes = dsort_tb;
    const DepthOrderPlan pl = depth_order_plan(P, 0);          // npre; the bucket tables are sized for the maximum
    g.dord.hdr = (uint32_t *)take(GSR_DO_ZERO_WORDS * sizeof(uint32_t));
    g.dord.gpair = reinterpret_cast<unsigned long long *>(g.dord.hdr + DO_HDR_WORDS);     // DO_HDR_WORDS is even: 8-byte aligned
    g.dord.gcur = g.dord.hdr + DO_HDR_WORDS + 2 * GSR_DO_MAXB;
    g.dord.bstart = (uint32_t *)take((GSR_DO_MAXB + 1) * sizeof(uint32_t));
    g.dord.tbase = (uint32_t *)take((GSR_DO_MAXB + 1) * sizeof(uint32_t));
    g.dord.blkmin = (uint32_t *)take(pl.npre * sizeof(uint32_t));
    g.dord.blkmax = (uint32_t *)take(pl.npre * sizeof(uint32_t));
    g.dord.blkent = (uint32_t *)take(pl.npre * sizeof(uint32_t));
    g.dord.comp = (uint64_t *)take(n * sizeof(uint64_t));
    g.touched = (uint8_t *)take(n);
    g.touch_mark = (uint32_t *)take(sizeof(uint32_t));
    g.total_bytes = off;
    return g;
}

ImageView carve_image(void *base, int W, int H) {
    ImageView v;
    const size_t T = (size_t)((W + GSR_TILE_HOST - 1) / GSR_TILE_HOST) * ((H + GSR_TILE_HOST - 1) / GSR_TILE_HOST);
    const size_t HW = (size_t)W * H;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    v.ranges = (uint2 *)take((T > 0 ? T : 1) * sizeof(uint2));
    v.final_T = (float *)take((HW > 0 ? HW : 1) * sizeof(float));
    v.n_contrib = (uint32_t *)take((HW > 0 ? HW : 1) * sizeof(uint32_t));
    const size_t units = 2 * (T > 0 ? T : 1);
    v.seg.units = (uint32_t)units;
    v.seg.band_units = (uint32_t)((units + GSR_SEG_BANDS - 1) / GSR_SEG_BANDS);
    v.seg.pool_cap = (uint32_t)((units * GSR_SEG_POOL_PER_UNIT + GSR_SEG_BANDS - 1) / GSR_SEG_BANDS * GSR_SEG_BANDS);
    v.seg.hdr = (uint32_t *)take(GSR_SEG_HDR_WORDS * sizeof(uint32_t));
    v.seg.info = (uint2 *)take(units * sizeof(uint2));
    v.seg.ck_slot = (uint32_t *)take(units * 8 * sizeof(uint32_t));
    v.seg.list_cap = (uint32_t)((size_t)v.seg.band_units * (1 + GSR_SEG_MAXCK) + GSR_SEG_FILL_CAP);
    v.seg.bq = (uint4 *)take((size_t)GSR_SEG_BANDS * v.seg.list_cap * sizeof(uint4));
    v.seg.pool = (float4 *)take((size_t)v.seg.pool_cap * 128 * sizeof(float4));
    v.total_bytes = off;
    return v;
}

BinningView carve_binning(void *base, int64_t N, size_t sort_tb) {
    BinningView b;
    const size_t n = (size_t)(N > 0 ? N : 1);
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    b.point_list = (uint32_t *)take(n * sizeof(uint32_t));
    b.contrib = (uint8_t *)take(4 * n);
    b.list_bytes = off;
    b.keys_sorted = (uint64_t *)take(n * sizeof(uint64_t));
    b.keys_unsorted = (uint64_t *)take(n * sizeof(uint64_t));
    b.point_list_unsorted = (uint32_t *)take(n * sizeof(uint32_t));
    b.tkeys_unsorted = (uint32_t *)b.keys_sorted; b.ids_sorted = b.point_list; b.ids_unsorted = (uint32_t *)b.keys_unsorted;
    b.tkeys_sorted = b.point_list_unsorted;
    b.sort_temp = take(sort_tb);
    b.sort_temp_bytes = sort_tb;
    b.total_bytes = off;
    return b;
}

TileListView carve_tile_lists(void *base, const TileListPlan &pl, int64_t E) {
    TileListView v;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    const size_t S = (size_t)pl.S, e = (size_t)(E > 0 ? E : 1), nseg = (size_t)(pl.nseg_max > 0 ? pl.nseg_max : 1);
    v.binstart = (uint32_t *)take((S + 1) * sizeof(uint32_t));
    v.segbase = (uint32_t *)take((S + 1) * sizeof(uint32_t));
    v.seg_super = (uint32_t *)take(nseg * sizeof(uint32_t));
    v.entries = (uint4 *)take(e * sizeof(uint4));
    v.segcnt = (uint32_t *)take(nseg * 64 * sizeof(uint32_t));
    v.tile_off = (uint32_t *)take(S * 64 * sizeof(uint32_t));
    v.tile_tot = (uint32_t *)take(S * 64 * sizeof(uint32_t));
    v.st_pairs = (uint32_t *)take(S * sizeof(uint32_t));
    v.total_bytes = off;
    return v;
}

static inline int grid_dim(int px) { return (px + GSR_TILE_HOST - 1) / GSR_TILE_HOST; }
static SegPlan seg_plan(int W, int H) {
    SegPlan p = {0, false, false};
    const long long T = (long long)grid_dim(W) * grid_dim(H);
    const int pk = g_persistent_bwd.load();
    const bool small = T <= GSR_PERSISTENT_MAX_TILES;
    p.persistent_bwd = (pk == 1 || (pk == 2 && small)) && g_bwd_npx.load() == 2 && T <= (1 << 28);
    const bool fwd_seg = (pk == 1 || (pk == 2 && small)) && g_fwd_npx.load() == 2 && T <= (1 << 28);
    // not under deterministic_bwd: which half tiles get checkpoints once the pool runs out is a race between the forward waves, and a
    // segment that starts from a stored transmittance differs in the last bits from the same entries reached by dividing back
    p.seg_len = fwd_seg && !g_deterministic_bwd.load() ? g_seg_len.load() : 0;
    p.small_image = small;
    return p;
}
static inline int tile_bits(int W, int H) { return ceil_log2_u32((uint32_t)(grid_dim(W) * grid_dim(H))); }
static inline int key_bits(int W, int H) { return 32 + tile_bits(W, H); }
// temp storage that serves either sort flavour
static hipError_t any_sort_temp_bytes(int64_t N, int W, int H, size_t *bytes) {
    size_t a = 0, b = 0;
    hipError_t e = sort_temp_bytes(N, key_bits(W, H), &a);
    if (e != hipSuccess) return e;
    e = sort2_temp_bytes(N, tile_bits(W, H) > 0 ? tile_bits(W, H) : 1, &b);
    *bytes = a > b ? a : b;
    return e;
}

// Four pinned host words per in-flight forward call: the bucket-scan kernel stores {overflow, Pv, N, seq}
// into them and the host polls `seq`, so nothing (copy engine, barrier packet) sits between the kernels that
// produce the totals and the kernels queued behind them while the host catches up.
struct ReadbackSlot {
    std::atomic<int> busy{0};
    uint32_t *host = nullptr;     // 8 pinned, device-visible words: overflow, Pv, N, E, seq
    int device = -1;
};
static ReadbackSlot g_slots[8];
static std::atomic<uint32_t> g_seq{1};
static ReadbackSlot *acquire_slot() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (auto &sl : g_slots) {
        int expect = 0;
        if (!sl.busy.compare_exchange_strong(expect, 1)) continue;
        if (sl.host && sl.device != dev) { sl.busy.store(0); continue; }
        if (!sl.host) {
            // coherent + mapped: the kernel's system-scope store must become visible to the polling host while the
            // stream is still running, whatever HIP_HOST_COHERENT says
            if (hipHostMalloc((void **)&sl.host, 8 * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { sl.host = nullptr; sl.busy.store(0); return nullptr; }
            sl.device = dev;
        }
        return &sl;
    }
    return nullptr;
}
static void release_slot(ReadbackSlot *sl) { if (sl) sl->busy.store(0); }
// spin on the sequence word; gives up after ~2 s (the caller then synchronises the stream instead, counts the
// time-out in the device's "poll_timeouts" and reuses the slot: after the synchronisation nothing writes to it)
static bool wait_seq(const ReadbackSlot *sl, uint32_t seq) {
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t spin = 1;; spin++) {
        if (__atomic_load_n(&sl->host[4], __ATOMIC_ACQUIRE) == seq) return true;
        __builtin_ia32_pause();
        if ((spin & 0xfffffu) == 0) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (t1.tv_sec - t0.tv_sec > 2) return false;
        }
    }
}

struct StageTimer {   // hipEvent pairs on the caller's stream; active only under gsr_set_profiling(1)
    hipStream_t s;
    bool on;
    hipEvent_t ev[GSR_NUM_STAGES + 1];
    int idx[GSR_NUM_STAGES + 1];
    float ms_out[GSR_NUM_STAGES];
    int n = 0;
    StageTimer(hipStream_t s_, bool on_) : s(s_), on(on_) { for (float &m : ms_out) m = -1.f; }
    void zero(int stage) { ms_out[stage] = 0.f; }
    void mark(int stage_about_to_start) {
        if (!on || n > GSR_NUM_STAGES) return;
        if (hipEventCreate(&ev[n]) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(ev[n], s);
        idx[n] = stage_about_to_start;
        n++;
    }
    void finish(int total_slot) {   // call after a final mark(-1)
        if (!on || n < 2) return;
        (void)hipEventSynchronize(ev[n - 1]);
        for (int i = 0; i + 1 < n; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            if (idx[i] >= 0) ms_out[idx[i]] = ms;
        }
        float tot = 0.f;
        (void)hipEventElapsedTime(&tot, ev[0], ev[n - 1]);
        ms_out[total_slot] = tot;
        {
            DeviceState &ds = dev_state();
            std::lock_guard<std::mutex> lk(ds.mu);
            for (int i = 0; i < GSR_NUM_STAGES; i++) if (ms_out[i] >= 0.f) ds.stage_ms[i] = ms_out[i];
        }
        for (int i = 0; i < n; i++) (void)hipEventDestroy(ev[i]);
        n = 0;
    }
};

}  // namespace gsr

using namespace gsr;

extern "C" {

int32_t gsr_abi_version(void) { return GSR_ABI_VERSION; }
const char *gsr_last_error(void) { return g_err; }

int32_t gsr_set_option(const char *name, int32_t value) {
    if (name && !strcmp(name, "exact_tile_cull")) { g_exact_cull.store(value ? 1 : 0); return GSR_OK; }
    if (name && !strcmp(name, "two_level_sort")) { g_two_level_sort.store(value ? 1 : 0); return GSR_OK; }
    if (name && !strcmp(name, "tile_lists")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARGUMENT, "tile_lists must be 0, 1 or 2");
        g_tile_lists.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "depth_log_map")) { DeviceState &ds = dev_state(); ds.depth_log_map.store(value ? 1 : 0); ds.bucket_fail_p.store(0x7fffffff); return GSR_OK; }
    if (name && !strcmp(name, "count_lanes")) { g_count_lanes.store(value == 2 ? 2 : (value ? 1 : 0)); return GSR_OK; }
    if (name && !strcmp(name, "composite_lds_pad")) { g_composite_lds_pad = value < 0 ? 0 : value; return GSR_OK; }
    if (name && !strcmp(name, "deterministic_bwd")) { g_deterministic_bwd.store(value ? 1 : 0); return GSR_OK; }
    if (name && !strcmp(name, "persistent_bwd")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARGUMENT, "persistent_bwd must be 0, 1 or 2");
        g_persistent_bwd.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "fill_in_tail")) { g_fill_in_tail.store(value ? 1 : 0); return GSR_OK; }
    if (name && !strcmp(name, "asm_walk")) { g_asm_walk.store(value ? 1 : 0); return GSR_OK; }
    if (name && !strcmp(name, "bwd_lpt")) { g_bwd_lpt.store(value ? 1 : 0); return GSR_OK; }
    if (name && !strcmp(name, "fwd_pair_long")) { g_fwd_pair_long.store(value < -1 ? -1 : value); return GSR_OK; }
    if (name && !strcmp(name, "prefill_at")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARGUMENT, "prefill_at must be 0, 1 or 2");
        g_prefill_at.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "dense_fork")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARGUMENT, "dense_fork must be 0, 1 or 2");
        g_dense_fork.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "dense_pergauss")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARGUMENT, "dense_pergauss must be 0, 1 or 2");
        g_dense_pergauss.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "segment_entries")) {
        if (value < 0 || value > 65536 || (value & 63)) return fail(GSR_ERR_INVALID_ARGUMENT, "segment_entries must be 0 or a multiple of 64 up to 65536");
        g_seg_len.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "depth_buckets")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARGUMENT, "depth_buckets must be 0, 1 or 2");
        g_depth_buckets.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "composite_waves_per_block")) {
        if (value != 1 && value != 2 && value != 4) return fail(GSR_ERR_INVALID_ARGUMENT, "composite_waves_per_block must be 1, 2 or 4");
        g_wpb.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "fwd_blocks_per_wave")) {
        if (value != 1 && value != 2 && value != 4) return fail(GSR_ERR_INVALID_ARGUMENT, "fwd_blocks_per_wave must be 1, 2 or 4");
        g_fwd_npx.store(value); return GSR_OK;
    }
    if (name && !strcmp(name, "bwd_blocks_per_wave")) {
        if (value != 1 && value != 2 && value != 4) return fail(GSR_ERR_INVALID_ARGUMENT, "bwd_blocks_per_wave must be 1, 2 or 4");
        g_bwd_npx.store(value); return GSR_OK;
    }
    return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_set_option: unknown option '%s'", name ? name : "(null)");
}
int32_t gsr_get_option(const char *name, int32_t *value) {
    if (name && value && !strcmp(name, "exact_tile_cull")) { *value = g_exact_cull.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "bwd_blocks_per_wave")) { *value = g_bwd_npx.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "fwd_blocks_per_wave")) { *value = g_fwd_npx.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "two_level_sort")) { *value = g_two_level_sort.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "tile_lists")) { *value = g_tile_lists.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "depth_log_map")) { *value = dev_state().depth_log_map.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "count_lanes")) { *value = g_count_lanes.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "deterministic_bwd")) { *value = g_deterministic_bwd.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "persistent_bwd")) { *value = g_persistent_bwd.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "segment_entries")) { *value = g_seg_len.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "fill_in_tail")) { *value = g_fill_in_tail.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "asm_walk")) { *value = g_asm_walk.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "bwd_lpt")) { *value = g_bwd_lpt.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "fwd_pair_long")) { *value = g_fwd_pair_long.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "prefill_at")) { *value = g_prefill_at.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "dense_fork")) { *value = g_dense_fork.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "dense_pergauss")) { *value = g_dense_pergauss.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "poll_timeouts")) { *value = dev_state().poll_timeouts.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "depth_buckets")) { *value = g_depth_buckets.load(); return GSR_OK; }
    if (name && value && !strcmp(name, "composite_waves_per_block")) { *value = g_wpb.load(); return GSR_OK; }
    return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_get_option: unknown option '%s'", name ? name : "(null)");
}

int32_t gsr_set_profiling(int32_t enable) { g_profiling.store(enable ? 1 : 0); return GSR_OK; }
int32_t gsr_get_stage_times(const char **names, float *ms) {
    DeviceState &ds = dev_state();
    std::lock_guard<std::mutex> lk(ds.mu);
    for (int i = 0; i < GSR_NUM_STAGES; i++) {
        if (names) names[i] = k_stage_names[i];
        if (ms) ms[i] = ds.stage_ms[i];
    }
    return GSR_OK;
}

int32_t gsr_workspace_sizes(int32_t P, int32_t W, int32_t H, size_t *geom_bytes, size_t *img_bytes, size_t *bwd_bytes) {
    if (P < 0 || W <= 0 || H <= 0) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_workspace_sizes: P=%d W=%d H=%d", P, W, H);
    if (W > 65535 * GSR_TILE_HOST || H > 65535 * GSR_TILE_HOST) return fail(GSR_ERR_INVALID_ARGUMENT, "image too large");
    size_t stb = 0, dtb = 0;
    HIP_TRY(scan_temp_bytes(P, &stb), "scan temp query");
    HIP_TRY(depth_sort_temp_bytes(P, &dtb), "depth sort temp query");
    if (geom_bytes) *geom_bytes = carve_geom(nullptr, P, stb, dtb).total_bytes;
    if (img_bytes) *img_bytes = carve_image(nullptr, W, H).total_bytes;
    if (bwd_bytes) *bwd_bytes = align_up(acc_rows(P) * GSR_ACC_FLOATS * sizeof(float));
    return GSR_OK;
}

int32_t gsr_backward_workspace_bytes(int32_t P, int64_t R, size_t *bytes) {
    if (P < 0 || R < 0 || !bytes) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_workspace_bytes: bad argument");
    const size_t acc_bytes = align_up(acc_rows(P) * GSR_ACC_FLOATS * sizeof(float));
    const size_t det_bytes = g_deterministic_bwd.load() ? (size_t)R * (size_t)(4 / g_bwd_npx.load()) * GSR_ACC_FLOATS * sizeof(float) : 0;
    // + the dense per-Gaussian stage's list and record buffer (pergauss_bwd.hip), behind the accumulators
    *bytes = acc_bytes + align_up(det_bytes) + pergauss_vis_bytes(P);
    return GSR_OK;
}

int32_t gsr_binning_bytes(int64_t N, int32_t W, int32_t H, size_t *bytes) {
    if (N < 0 || W <= 0 || H <= 0 || !bytes) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_binning_bytes: bad argument");
    size_t stb = 0;
    HIP_TRY(any_sort_temp_bytes(N, W, H, &stb), "sort temp query");
    *bytes = carve_binning(nullptr, N, stb).total_bytes;
    return GSR_OK;
}

int32_t gsr_forward(gsr_stream_t stream, int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float *bg,
                    const float *means3D, const float *shs, const float *colors_precomp, const float *opacities,
                    const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                    const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
                    float tanfovy, int32_t prefiltered, int32_t debug, float *out_color, int32_t *radii, void *geom_ws,
                    size_t geom_bytes, gsr_alloc_fn binning_alloc, void *binning_user, void *img_ws, size_t img_bytes,
                    int64_t *num_rendered, const float *shs_rest, int32_t raw_params) {
    (void)prefiltered;   // culled Gaussians are always skipped, as with prefiltered=False (the only value the reference passes)
    hipStream_t s = (hipStream_t)stream;
    if (num_rendered) *num_rendered = 0;
    // Gradient outputs announced for this render's backward call (gsr_backward_prefill): their zeros are written on the device's second
    // stream beside this pass's last kernels (half of the CUs idle under the list-ordering kernel, most of them in the compositing
    // kernel's tail) instead of beside the reverse compositing kernel, whose waves leave no slot free until it drains.  Only where
    // gsr_backward will take the dense per-Gaussian stage (same rule as there), which is what needs the zeros.
    DeviceState::Prefill pre;
    bool prefill = false, prefilled = false;
    {
        DeviceState &d0 = dev_state();
        std::lock_guard<std::mutex> lk(d0.mu);
        if (d0.prefill.done) {       // an earlier render's fill that no gsr_backward took over: ordered before everything this call writes
            d0.prefill.done = false;
            if (hipStreamWaitEvent(s, d0.ev_prefill, 0) != hipSuccess) return fail(GSR_ERR_HIP, "prefill join");
        }
        if (d0.prefill.pending) {
            d0.prefill.pending = false;
            const int dopt = g_dense_pergauss.load();
            pre = d0.prefill;
            prefill = g_prefill_at.load() != 0 && P > 0 && pre.P == P && pre.M == M && !debug && (dopt == 1 || (dopt == 2 && P >= GSR_DENSE_MIN_P)) &&
                      shs && !shs_rest && M == 16 && scales && rotations && pre.p[5] && !pre.p[6] && side_stream(d0);
        }
    }
    if (P < 0 || W <= 0 || H <= 0 || !out_color || !bg || !viewmatrix || !projmatrix)
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: bad sizes or missing bg/matrices/out_color");
    if (W > 65535 * GSR_TILE_HOST || H > 65535 * GSR_TILE_HOST) return fail(GSR_ERR_INVALID_ARGUMENT, "image too large");
    const size_t HW = (size_t)W * H;
    if (P == 0) {   // upstream returns a zero image, not the background, when there is nothing to draw
        HIP_TRY(hipMemsetAsync(out_color, 0, 3 * HW * sizeof(float), s), "memset out_color");
        return GSR_OK;
    }
    if (!means3D || !opacities || !radii || !geom_ws || !img_ws || !binning_alloc)
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: missing means3D/opacities/radii/workspaces/allocator");
    if ((shs != nullptr) == (colors_precomp != nullptr))
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: exactly one of shs / colors_precomp must be given");
    if (((scales != nullptr) && (rotations != nullptr)) == (cov3D_precomp != nullptr) || ((scales != nullptr) != (rotations != nullptr)))
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: exactly one of (scales, rotations) / cov3D_precomp must be given");
    if (shs_rest && (!shs || M < 2)) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: shs_rest needs shs (= features_dc) and M >= 2");
    if (raw_params && cov3D_precomp) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: raw_params needs scales/rotations, not cov3D_precomp");
    if (shs) {
        if (D < 0 || D > 3) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: SH degree %d not in 0..3", D);
        if (M < (D + 1) * (D + 1)) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: M=%d < (D+1)^2=%d", M, (D + 1) * (D + 1));
        if (!campos) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_forward: campos required with shs");
    }
    size_t stb = 0, dtb = 0;
    HIP_TRY(scan_temp_bytes(P, &stb), "scan temp query");
    HIP_TRY(depth_sort_temp_bytes(P, &dtb), "depth sort temp query");
    GeomView g = carve_geom(geom_ws, P, stb, dtb);
    ImageView im = carve_image(img_ws, W, H);
    if (geom_bytes < g.total_bytes) return fail(GSR_ERR_WORKSPACE, "geom workspace %zu < %zu", geom_bytes, g.total_bytes);
    if (img_bytes < im.total_bytes) return fail(GSR_ERR_WORKSPACE, "image workspace %zu < %zu", img_bytes, im.total_bytes);
    const int gridx = grid_dim(W), gridy = grid_dim(H), T = gridx * gridy;

    auto prefill_now = [&]() -> int32_t {
        if (!prefill) return GSR_OK;
        prefill = false;
        DeviceState &d0 = dev_state();
        std::lock_guard<std::mutex> lk(d0.mu);
        PergaussBwdArgs fa{};
        fa.P = P; fa.M = M; fa.shs = shs;
        fa.dL_dmeans2D = pre.p[0]; fa.dL_dopacity = pre.p[1]; fa.dL_dcolors = pre.p[2]; fa.dL_dmeans3D = pre.p[3]; fa.dL_dcov3D = pre.p[4];
        fa.dL_dsh = pre.p[5]; fa.dL_dscales = pre.p[7]; fa.dL_drots = pre.p[8];
        HIP_TRY(hipEventRecord(d0.ev_fork, s), "prefill fork event");
        HIP_TRY(hipStreamWaitEvent(d0.side, d0.ev_fork, 0), "prefill fork wait");
        HIP_TRY(launch_fill_zero(fa, d0.side), "gradient zero-fill launch");
        HIP_TRY(hipEventRecord(d0.ev_prefill, d0.side), "prefill event");
        prefilled = true;
        return GSR_OK;
    };

    StageTimer tm(s, g_profiling.load() != 0);
    tm.mark(0);
    PreprocessArgs pa;
    pa.P = P; pa.D = D; pa.M = M; pa.W = W; pa.H = H; pa.gridx = gridx; pa.gridy = gridy;
    pa.raw_params = raw_params ? 1 : 0; pa.shs_rest = shs_rest;
    pa.means3D = means3D; pa.shs = shs; pa.colors_precomp = colors_precomp; pa.opacities = opacities;
    pa.scales = scales; pa.rotations = rotations; pa.cov3D_precomp = cov3D_precomp;
    pa.viewmatrix = viewmatrix; pa.projmatrix = projmatrix; pa.campos = campos;
    pa.scale_modifier = scale_modifier; pa.tanfovx = tanfovx; pa.tanfovy = tanfovy; pa.radii = radii; pa.exact_cull = g_exact_cull.load(); pa.g = g;
    pa.touch_mark = 1u + dev_state().frame_seq.fetch_add(1u) % 255u;
    pa.seg_hdr = im.seg.hdr;
    HIP_TRY(launch_preprocess_fwd(pa, s), "preprocess launch");
    if (debug) HIP_TRY(hipStreamSynchronize(s), "preprocess");
    tm.mark(1);
    uint32_t n32 = 0, e32 = 0;
    const int two_level = g_two_level_sort.load();
    const int tl_mode = g_tile_lists.load();
    DeviceState &ds = dev_state();
    BinningView b;
    bool lists_done = false;
    // ---- default: entries binned per super-tile, every bin ordered in LDS (supertile_sort.hip) ----
    const SuperSortPlan ssp = super_sort_plan(P, W, H);
    if (tl_mode == 2 && two_level && !debug && ssp.S <= GSR_SS_MAXS && ssp.chunk <= GSR_SS_MAX_CHUNK &&
        (size_t)ssp.S * ssp.nblk <= (size_t)GSR_SS_WGCNT_WORDS) {
        uint32_t h[4] = {1u, 0u, 0u, 0u};
        ReadbackSlot *sl = acquire_slot();
        const uint32_t seq = sl ? (g_seq.fetch_add(1) | 0x80000000u) : 0u;
        if (sl) sl->host[4] = 0u;
        hipError_t e = launch_super_sort_count(g, P, W, H, pa.exact_cull, sl ? sl->host : nullptr, seq, s);
        if (e == hipSuccess) e = launch_super_sort_scatter(g, P, W, H, pa.exact_cull, s);     // runs while the host waits for the totals
        if (e != hipSuccess) { release_slot(sl); return fail(GSR_ERR_HIP, "super-tile lists: %s (%d)", hipGetErrorString(e), (int)e); }
        tm.mark(2);
        if (sl && wait_seq(sl, seq)) {
            h[0] = sl->host[0]; h[1] = sl->host[1]; h[2] = sl->host[2]; h[3] = sl->host[3];
            release_slot(sl);
        } else {
            uint32_t w4[4] = {0u, 0u, 0u, 0u};
            hipError_t ce = hipMemcpyAsync(w4, g.dord.hdr + DO_OVERFLOW, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
            if (ce == hipSuccess) ce = hipMemcpyAsync(w4 + 1, g.dord.hdr + SS_HDR_MAXBIN, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
            if (ce == hipSuccess) ce = hipStreamSynchronize(s);
            if (sl) { ds.poll_timeouts.fetch_add(1); release_slot(sl); }
            if (ce != hipSuccess) return fail(GSR_ERR_HIP, "read N: %s (%d)", hipGetErrorString(ce), (int)ce);
            h[0] = w4[0]; h[1] = w4[1]; h[2] = w4[2]; h[3] = w4[3];
        }
        if (!h[0]) {
            n32 = h[2]; e32 = h[3];
            const int64_t N = (int64_t)n32;
            if (num_rendered) *num_rendered = N;
            const size_t pl_bytes = carve_binning(nullptr, N, 0).list_bytes;   // point_list + contrib
            char *bin_ptr = (char *)binning_alloc(binning_user, pl_bytes);
            if (!bin_ptr) return fail(GSR_ERR_ALLOC, "binning allocator returned NULL for %zu bytes (N=%lld)", pl_bytes, (long long)N);
            b = carve_binning(bin_ptr, N, 0);
            tm.zero(3); tm.zero(5);
            tm.mark(4);
            if (g_prefill_at.load() == 2) { const int32_t rc = prefill_now(); if (rc != GSR_OK) return rc; }
            HIP_TRY(launch_super_sort_expand(g, im, b.point_list, P, W, H, h[1], s), "super-tile lists: order + expand");
            tm.mark(7);
            lists_done = true;
        } else {
            // a bin beyond the LDS capacity or more entries than the workspace holds: this frame takes round 1's path, which
            // shares the (now dirty) counter region
            HIP_TRY(hipMemsetAsync(g.dord.hdr, 0, GSR_DO_ZERO_WORDS * sizeof(uint32_t), s), "reset counters");
            tm.mark(1);
        }
    }
    if (!lists_done) {
    // whether tile_lists.hip builds the per-tile lists is known up front (image size, options): its level-1 counting
    // is queued before the host waits for N, and the depth order then skips the scan only key emission needs
    const bool want_tile_lists = g_tile_lists.load() != 0 && two_level && tile_list_plan(P, 0, W, H).S <= GSR_TL_MAX_S;
    int P_list = P;                                   // entries of the depth-ordered list (perm / offsets)
    const int dbopt = g_depth_buckets.load();
    bool bucketed = dbopt == 2 || (dbopt == 1 && P >= GSR_DEPTH_BUCKETS_MIN_P && P < ds.bucket_fail_p.load());
    if (bucketed) {
        uint32_t h[4] = {1u, 0u, 0u, 0u};
        ReadbackSlot *sl = debug ? nullptr : acquire_slot();
        const uint32_t seq = sl ? (g_seq.fetch_add(1) | 0x80000000u) : 0u;
        if (sl) sl->host[4] = 0u;
        const int log_map = ds.depth_log_map.load();
        hipError_t e = launch_depth_order_count(g, P, log_map, sl ? sl->host : nullptr, seq, s);
        if (e == hipSuccess) e = launch_depth_order_place(g, P, log_map, want_tile_lists ? 0 : 1, s);   // runs while the host waits for the totals
        if (e == hipSuccess && want_tile_lists) e = launch_tile_lists_count(g, P, g.dord.hdr, W, H, s);
        if (e != hipSuccess) { release_slot(sl); return fail(GSR_ERR_HIP, "depth order: %s (%d)", hipGetErrorString(e), (int)e); }
        if (debug) HIP_TRY(hipStreamSynchronize(s), "depth order");
        tm.mark(2);
        if (sl && wait_seq(sl, seq)) {
            h[0] = sl->host[0]; h[1] = sl->host[1]; h[2] = sl->host[2]; h[3] = sl->host[3];
            release_slot(sl);
        } else {                                      // no slot, debug mode, or the poll timed out
            hipError_t ce = hipMemcpyAsync(h, g.dord.hdr + DO_OVERFLOW, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
            if (ce == hipSuccess) ce = hipStreamSynchronize(s);
            // the stream has drained: the kernel that writes the slot has finished, so the slot is free again
            if (sl) { ds.poll_timeouts.fetch_add(1); release_slot(sl); }
            if (ce != hipSuccess) return fail(GSR_ERR_HIP, "read N: %s (%d)", hipGetErrorString(ce), (int)ce);
        }
        e32 = h[3];
        if (h[0]) {               // a bucket exceeds the LDS capacity: general sort below, log map next time
            bucketed = false;
            if (log_map) {        // already the robust map: stop paying for the attempt at this size
                int cur = ds.bucket_fail_p.load();
                while (P < cur && !ds.bucket_fail_p.compare_exchange_weak(cur, P)) {}
            }
            ds.depth_log_map.store(1);
        }
        else { P_list = (int)h[1]; n32 = h[2]; }
    }
    if (!bucketed) {
        HIP_TRY(launch_depth_sort(g, P, s), "depth sort");
        HIP_TRY(launch_ordered_scan(g, P, s), "ordered scan");
        HIP_TRY(launch_entry_total(g, P, s), "entry total");
        if (want_tile_lists) HIP_TRY(launch_tile_lists_count(g, P, nullptr, W, H, s), "tile lists: count");
        if (debug) HIP_TRY(hipStreamSynchronize(s), "depth order + scan");
        tm.mark(2);
        HIP_TRY(hipMemcpyAsync(&n32, g.offsets + (P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read N");
        HIP_TRY(hipMemcpyAsync(&e32, g.dord.hdr + DO_ETOT, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read E");
        HIP_TRY(hipStreamSynchronize(s), "read N sync");
    }
    const int64_t N = (int64_t)n32;
    if (num_rendered) *num_rendered = N;

    size_t sort_tb = 0;
    const int bits = key_bits(W, H);
    const int64_t E = (int64_t)e32;
    const TileListPlan tlp = tile_list_plan(P, E, W, H);
    const bool tile_lists = want_tile_lists && N > 0;
    TileListView tv;
    if (tile_lists) {             // point_list first (what backward and the debug reader expect), then the entry workspace
        const size_t pl_bytes = carve_binning(nullptr, N, 0).list_bytes;   // point_list + contrib
        tv = carve_tile_lists(nullptr, tlp, E);
        const size_t total = pl_bytes + tv.total_bytes;
        char *bin_ptr = (char *)binning_alloc(binning_user, total);
        if (!bin_ptr) return fail(GSR_ERR_ALLOC, "binning allocator returned NULL for %zu bytes (N=%lld)", total, (long long)N);
        b = carve_binning(bin_ptr, N, 0);
        tv = carve_tile_lists(bin_ptr + pl_bytes, tlp, E);
        tm.zero(3); tm.zero(5);                         // no key emission / range detection on this path
        tm.mark(4);
        HIP_TRY(launch_tile_lists(g, tv, im, b.point_list, P, P_list, E, W, H, pa.exact_cull, s), "tile lists");
        if (debug) HIP_TRY(hipStreamSynchronize(s), "tile lists");
        tm.mark(7);
    } else {
        HIP_TRY(any_sort_temp_bytes(N, W, H, &sort_tb), "sort temp query");
        b = carve_binning(nullptr, N, sort_tb);
        void *bin_ptr = binning_alloc(binning_user, b.total_bytes);
        if (!bin_ptr) return fail(GSR_ERR_ALLOC, "binning allocator returned NULL for %zu bytes (N=%lld)", b.total_bytes, (long long)N);
        b = carve_binning(bin_ptr, N, sort_tb);
        tm.mark(3);
        if (N > 0) {
            HIP_TRY(launch_emit_keys(g, b, P_list, W, H, pa.exact_cull, two_level, s), "emit keys launch");
            if (debug) HIP_TRY(hipStreamSynchronize(s), "emit keys");
            tm.mark(4);
            if (two_level) HIP_TRY(launch_sort2_by_tile(b, N, tile_bits(W, H) > 0 ? tile_bits(W, H) : 1, s), "radix sort by tile");
            else HIP_TRY(launch_sort(b, N, bits, s), "radix sort");
            if (debug) HIP_TRY(hipStreamSynchronize(s), "radix sort");
        }
        tm.mark(5);
        HIP_TRY(launch_ranges(b, im, N, T, two_level, s), "tile ranges");
        if (debug) HIP_TRY(hipStreamSynchronize(s), "tile ranges");
        tm.mark(7);
    }
    }
    CompositeArgs ca;
    ca.W = W; ca.H = H; ca.gridx = gridx; ca.gridy = gridy; ca.ranges = im.ranges; ca.point_list = b.point_list;
    ca.contrib = b.contrib; ca.contrib_stride = (size_t)(n32 > 0 ? n32 : 1);
    ca.rec = g.rec; ca.bg = bg; ca.final_T = im.final_T; ca.n_contrib = im.n_contrib; ca.out_color = out_color; ca.touched = g.touched; ca.touch_mark = pa.touch_mark;
    ca.counters = lane_counters(0); ca.count_mode = g_count_lanes.load();
    ca.seg = im.seg;
    // checkpoints + per-half-tile lengths for the segmented reverse pass: only where gsr_backward will use them (same rule as there)
    const SegPlan sp = seg_plan(W, H);
    ca.seg_len = sp.seg_len;
    ca.asm_walk = g_asm_walk.load();
    {   // long lists by pairs of block waves: only where the longest list sets the kernel's time (the images the persistent reverse kernel serves)
        const int pl = g_fwd_pair_long.load();
        ca.pair_long_n = sp.small_image && g_fwd_npx.load() == 2 ? (pl < 0 ? GSR_PAIR_LONG_DEFAULT : pl) : 0;
    }
    ca.lpt_span = (!sp.small_image && sp.seg_len == 0 && g_bwd_lpt.load() && g_fwd_npx.load() == 2 && g_bwd_npx.load() == 2 && T <= (1 << 28)) ? GSR_LPT_SPAN : 0;
    { const int32_t rc = prefill_now(); if (rc != GSR_OK) return rc; }
    HIP_TRY(launch_composite_fwd(ca, g_fwd_npx.load(), pa.exact_cull, g_wpb.load(), s), "composite launch");
    if (debug) HIP_TRY(hipStreamSynchronize(s), "composite");
    if (prefilled) {        // joined by the gsr_backward that takes the buffers over, or by the next call on the device (no wait here: the
                            // fill's tail runs on beside the kernels between the two passes)
        DeviceState &d0 = dev_state();
        std::lock_guard<std::mutex> lk(d0.mu);
        d0.prefill = pre; d0.prefill.pending = false; d0.prefill.done = true;
    }
    tm.mark(-1);
    tm.finish(11);
    return GSR_OK;
}

int32_t gsr_backward_prefill(int32_t P, int32_t M, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors, float *dL_dmeans3D,
                             float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drots, float *dL_dsh_rest) {
    DeviceState &ds = dev_state();
    std::lock_guard<std::mutex> lk(ds.mu);
    ds.prefill.pending = false;
    if (P <= 0) return GSR_OK;               // withdraws an announcement
    if (M < 0) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward_prefill: M < 0");
    float *const outs[9] = {dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscales, dL_drots};
    ds.prefill.P = P; ds.prefill.M = M;
    memcpy(ds.prefill.p, outs, sizeof outs);
    ds.prefill.pending = true;
    return GSR_OK;
}

int32_t gsr_backward(gsr_stream_t stream, int32_t P, int32_t D, int32_t M, int64_t R, int32_t W, int32_t H,
                     const float *bg, const float *means3D, const int32_t *radii, const float *shs,
                     const float *colors_precomp, const float *scales, float scale_modifier, const float *rotations,
                     const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
                     float tanfovx, float tanfovy, const float *dL_dpix, const void *geom_ws, size_t geom_bytes,
                     const void *binning_ws, size_t binning_bytes, const void *img_ws, size_t img_bytes, void *bwd_ws,
                     size_t bwd_bytes, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors, float *dL_dmeans3D,
                     float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drots, int32_t debug,
                     const float *shs_rest, int32_t raw_params, float *dL_dsh_rest) {
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || W <= 0 || H <= 0 || R < 0) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: bad sizes");
    if (P == 0) return GSR_OK;
    if (!bg || !means3D || !radii || !viewmatrix || !projmatrix || !dL_dpix || !geom_ws || !img_ws || !bwd_ws ||
        !dL_dmeans2D || !dL_dopacity || !dL_dmeans3D)
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: missing input, workspace or gradient buffer");
    if ((colors_precomp && !dL_dcolors) || (cov3D_precomp && !dL_dcov3D))
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: dL_dcolors / dL_dcov3D required with colors_precomp / cov3D_precomp");
    if ((shs != nullptr) == (colors_precomp != nullptr))
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: exactly one of shs / colors_precomp must be given");
    if (((scales != nullptr) && (rotations != nullptr)) == (cov3D_precomp != nullptr) || ((scales != nullptr) != (rotations != nullptr)))
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: exactly one of (scales, rotations) / cov3D_precomp must be given");
    if (shs && (!dL_dsh || !campos || D < 0 || D > 3 || M < (D + 1) * (D + 1)))
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: SH inputs inconsistent");
    if (scales && (!dL_dscales || !dL_drots)) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: dL_dscales/dL_drots required");
    if (shs_rest && (!shs || !dL_dsh_rest || M < 2)) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: shs_rest needs shs, dL_dsh_rest and M >= 2");
    if (raw_params && cov3D_precomp) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: raw_params needs scales/rotations");
    if (R > 0 && !binning_ws) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: binning workspace missing");
    size_t stb = 0, dtb = 0;
    HIP_TRY(scan_temp_bytes(P, &stb), "scan temp query");
    HIP_TRY(depth_sort_temp_bytes(P, &dtb), "depth sort temp query");
    GeomView g = carve_geom(const_cast<void *>(geom_ws), P, stb, dtb);
    ImageView im = carve_image(const_cast<void *>(img_ws), W, H);
    if (geom_bytes < g.total_bytes) return fail(GSR_ERR_WORKSPACE, "geom workspace %zu < %zu", geom_bytes, g.total_bytes);
    if (img_bytes < im.total_bytes) return fail(GSR_ERR_WORKSPACE, "image workspace %zu < %zu", img_bytes, im.total_bytes);
    const SegPlan sp = seg_plan(W, H);
    BinningView b = carve_binning(const_cast<void *>(binning_ws), R, 0);
    if (R > 0 && binning_bytes < b.list_bytes)
        return fail(GSR_ERR_WORKSPACE, "binning workspace too small for R=%lld (%zu < %zu)", (long long)R, binning_bytes, b.list_bytes);
    if (acc_rows(P) >= (1u << 28)) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_backward: P too large for the 28-bit accumulator row index");
    const size_t acc_bytes = acc_rows(P) * GSR_ACC_FLOATS * sizeof(float);
    if (bwd_bytes < acc_bytes) return fail(GSR_ERR_WORKSPACE, "backward workspace %zu < %zu", bwd_bytes, acc_bytes);
    const bool det = g_deterministic_bwd.load() != 0 && R > 0;
    const int bwd_npx = g_bwd_npx.load();
    const size_t det_bytes = det ? (size_t)R * (size_t)(4 / bwd_npx) * GSR_ACC_FLOATS * sizeof(float) : 0;
    if (det && bwd_bytes < align_up(acc_bytes) + det_bytes)
        return fail(GSR_ERR_WORKSPACE, "deterministic_bwd: backward workspace %zu < %zu (size it with gsr_backward_workspace_bytes)", bwd_bytes,
                    align_up(acc_bytes) + det_bytes);
    const int gridx = grid_dim(W), gridy = grid_dim(H);
    const bool persistent = sp.persistent_bwd && R > 0;
    const SegView &segv = im.seg;
    // the written-out reverse walk addresses the accumulator rows with a 32-bit byte offset: rows < 2^26
    const int bwd_asm = g_asm_walk.load() && !det && (!lane_counters(1) || g_count_lanes.load() == 2) && acc_rows(P) < (1u << 26) ? 1 : 0;
    const int pk_grid = persistent ? composite_bwd_persistent_grid(gridx * gridy, det ? 1 : 0, lane_counters(1) ? g_count_lanes.load() : 0, bwd_asm) : 0;
    // large images: half tiles in order of decreasing length (the forward pass filed the lengths when it ran with the same options)
    const bool lpt = !persistent && bwd_asm && !lane_counters(1) && g_bwd_lpt.load() && !sp.small_image && bwd_npx == 2 && g_fwd_npx.load() == 2 && R > 0 && g_wpb.load() == 1;
    const int fill_chunk = persistent && g_fill_in_tail.load() ? seg_fill_chunk(P) : 0;       // zero-fill units in the persistent kernel's lists

    PergaussBwdArgs pa;
    pa.raw_params = raw_params ? 1 : 0; pa.shs_rest = shs_rest; pa.rec = g.rec; pa.dL_dsh_rest = dL_dsh_rest;
    pa.P = P; pa.D = D; pa.M = M; pa.W = W; pa.H = H; pa.means3D = means3D; pa.shs = shs; pa.colors_precomp = colors_precomp;
    pa.scales = scales; pa.rotations = rotations; pa.cov3D_precomp = cov3D_precomp; pa.viewmatrix = viewmatrix;
    pa.projmatrix = projmatrix; pa.campos = campos; pa.scale_modifier = scale_modifier; pa.tanfovx = tanfovx;
    pa.tanfovy = tanfovy; pa.radii = radii; pa.clamped = g.clamped; pa.opac = g.opac; pa.acc = (const float *)bwd_ws; pa.hot = g.hot; pa.touched = g.touched; pa.touch_mark = g.touch_mark;
    pa.skip_unmarked = fill_chunk > 0 && R > 0 ? 1 : 0;
    pa.dense = 0; pa.vis_count = nullptr; pa.vis_list = nullptr; pa.vis_rec = nullptr; pa.vis_cap = 0;
    pa.dL_dmeans2D = dL_dmeans2D; pa.dL_dopacity = dL_dopacity; pa.dL_dcolors = dL_dcolors; pa.dL_dmeans3D = dL_dmeans3D;
    pa.dL_dcov3D = dL_dcov3D; pa.dL_dsh = dL_dsh; pa.dL_dscales = dL_dscales; pa.dL_drots = dL_drots;

    StageTimer tm(s, g_profiling.load() != 0);
    tm.mark(8);
    // Dense per-Gaussian stage: the zeros of every gradient output are written by a kernel of their own on the device's second stream,
    // forked here and joined in front of pergauss_bwd: it runs beside the compositing kernel (FP32-issue-bound, the memory system idle).
    const int dense_opt = g_dense_pergauss.load();
    const size_t vis_off = align_up(acc_bytes) + align_up(det_bytes);
    bool dense = R > 0 && fill_chunk == 0 && (dense_opt == 1 || (dense_opt == 2 && P >= GSR_DENSE_MIN_P)) && pergauss_dense_eligible(pa) &&
                 bwd_bytes >= vis_off + pergauss_vis_bytes(P);            // (a workspace sized before this stage existed: the streaming kernel)
    if (dense) {
        char *vis = (char *)bwd_ws + vis_off;
        pa.vis_count = (uint32_t *)vis;
        pa.vis_list = (uint32_t *)(vis + 256);
        pa.vis_rec = (float4 *)(vis + 256 + (((size_t)P * 4 + 255) / 256 * 256));
        pa.vis_cap = pergauss_vis_cap(P);
        DeviceState &ds = dev_state();
        std::lock_guard<std::mutex> lk(ds.mu);
        if (!side_stream(ds)) dense = false;
    }
    // outputs the matching forward pass has zero-filled already (gsr_backward_prefill): no fill here, and the second stream's order
    // (that fill, then this call's gathering kernel, then the join event) covers it.  Any other outstanding fill is joined first.
    bool prefilled = false;
    {
        DeviceState &ds = dev_state();
        std::lock_guard<std::mutex> lk(ds.mu);
        if (ds.prefill.done) {
            ds.prefill.done = false;
            float *const outs[9] = {dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscales, dL_drots};
            prefilled = dense && ds.prefill.P == P && ds.prefill.M == M && !memcmp(outs, ds.prefill.p, sizeof outs);
            if (!prefilled) HIP_TRY(hipStreamWaitEvent(s, ds.ev_prefill, 0), "prefill join");
        }
    }
    // the fork: after the accumulator rows are cleared, the clearing kernel has the chip to itself (7 us; 18 with the gathering kernel
    // starting beside it) and the second stream's work starts with the compositing kernel -- config 3: 4 us better; at 5 M Gaussians
    // the second stream's work (1.2 GB of zeros, 450 k records) outlasts the compositing kernel's shadow and every microsecond of
    // head start counts: fork first (config 5: 2.37 against 2.42 ms)
    auto fork = [&]() -> int32_t {
        DeviceState &ds = dev_state();
        std::lock_guard<std::mutex> lk(ds.mu);
        HIP_TRY(hipEventRecord(ds.ev_fork, s), "fork event");
        HIP_TRY(hipStreamWaitEvent(ds.side, ds.ev_fork, 0), "fork wait");
        HIP_TRY(hipMemsetAsync(pa.vis_count, 0, 256, ds.side), "visible counter");
        HIP_TRY(launch_gather_visible(pa, ds.side), "gather launch");
        if (!prefilled) HIP_TRY(launch_fill_zero(pa, ds.side), "gradient zero-fill launch");
        HIP_TRY(hipEventRecord(ds.ev_join, ds.side), "join event");
        return GSR_OK;
    };
    const int fork_late = g_dense_fork.load() == 2 ? (P < GSR_DENSE_FORK_EARLY_P ? 1 : 0) : g_dense_fork.load();
    if (dense && !fork_late) { const int32_t rc = fork(); if (rc != GSR_OK) return rc; }
    if (det) {
        HIP_TRY(hipMemsetAsync(bwd_ws, 0, align_up(acc_bytes) + det_bytes, s), "zero accumulators");
        if (persistent) HIP_TRY(launch_zero_marked_rows(P, g.touched, g.touch_mark, (float *)bwd_ws, 0, segv, pk_grid, fill_chunk, s), "unit lists");
    } else HIP_TRY(launch_zero_marked_rows(P, g.touched, g.touch_mark, (float *)bwd_ws, acc_rows(P), segv, persistent ? pk_grid : (lpt ? 1 : 0), fill_chunk, s), "zero accumulators");
    if (dense && fork_late) { const int32_t rc = fork(); if (rc != GSR_OK) return rc; }
    tm.mark(9);
    if (R > 0) {
        CompositeBwdArgs ca;
        ca.W = W; ca.H = H; ca.gridx = gridx; ca.gridy = gridy; ca.ranges = im.ranges; ca.point_list = b.point_list;
        ca.contrib = b.contrib; ca.contrib_stride = (size_t)R;
        ca.rec = g.rec; ca.bg = bg; ca.final_T = im.final_T; ca.n_contrib = im.n_contrib; ca.dL_dpix = dL_dpix;
        ca.acc = (float *)bwd_ws;
        ca.counters = lane_counters(1); ca.count_mode = g_count_lanes.load();
        ca.det = det ? (float *)((char *)bwd_ws + align_up(acc_bytes)) : nullptr;
        ca.P = P; ca.rect = g.rect; ca.tiles = g.tiles; ca.depth_bits = reinterpret_cast<const uint32_t *>(g.depth);
        ca.seg = segv;
        ca.asm_walk = bwd_asm;
        ca.fill.P = P; ca.fill.M = M; ca.fill.chunk = fill_chunk; ca.fill.radii = radii; ca.fill.touched = g.touched; ca.fill.mark = g.touch_mark;
        ca.fill.means2D = dL_dmeans2D; ca.fill.opacity = dL_dopacity; ca.fill.colors = dL_dcolors; ca.fill.means3D = dL_dmeans3D;
        ca.fill.cov3D = dL_dcov3D; ca.fill.sh = shs ? dL_dsh : nullptr; ca.fill.sh_rest = shs_rest ? dL_dsh_rest : nullptr;
        ca.fill.scales = scales ? dL_dscales : nullptr; ca.fill.rots = scales ? dL_drots : nullptr;
        if (persistent) HIP_TRY(launch_composite_bwd_persistent(ca, pk_grid, s), "composite backward launch");
        else if (lpt) HIP_TRY(launch_composite_bwd_lpt(ca, s), "composite backward launch");
        else HIP_TRY(launch_composite_bwd(ca, bwd_npx, g_exact_cull.load(), g_wpb.load(), s), "composite backward launch");
        if (debug) HIP_TRY(hipStreamSynchronize(s), "composite backward");
    }
    tm.mark(10);
    if (dense) {
        DeviceState &ds = dev_state();
        std::lock_guard<std::mutex> lk(ds.mu);
        HIP_TRY(hipStreamWaitEvent(s, ds.ev_join, 0), "join wait");
        pa.skip_unmarked = 1; pa.dense = 1;
    }
    HIP_TRY(launch_pergauss_bwd(pa, s), "per-Gaussian backward launch");
    if (debug) HIP_TRY(hipStreamSynchronize(s), "per-Gaussian backward");
    tm.mark(-1);
    tm.finish(12);
    return GSR_OK;
}

// ---- fused training loss (include/gsr_loss.h) ----
static inline size_t loss_blocks(int C, int H, int W) { return (size_t)((W + 15) / 16) * ((H + 15) / 16) * C; }

int32_t gsr_l1_ssim_workspace(int32_t C, int32_t H, int32_t W, size_t *bytes) {
    if (C <= 0 || H <= 0 || W <= 0 || !bytes) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_l1_ssim_workspace: bad argument");
    *bytes = align_up((size_t)3 * C * H * W * sizeof(float)) + align_up(loss_blocks(C, H, W) * 2 * sizeof(float));
    return GSR_OK;
}

int32_t gsr_l1_ssim_forward(gsr_stream_t stream, int32_t C, int32_t H, int32_t W, const float *img, const float *gt,
                            float lambda_dssim, float *out3, void *ws, size_t ws_bytes) {
    size_t need = 0;
    if (gsr_l1_ssim_workspace(C, H, W, &need) != GSR_OK) return GSR_ERR_INVALID_ARGUMENT;
    if (!img || !gt || !out3 || !ws) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_l1_ssim_forward: null pointer");
    if (ws_bytes < need) return fail(GSR_ERR_WORKSPACE, "loss workspace %zu < %zu", ws_bytes, need);
    float *dmaps = (float *)ws;
    float *partial = (float *)((char *)ws + align_up((size_t)3 * C * H * W * sizeof(float)));
    HIP_TRY(launch_l1_ssim_forward(C, H, W, img, gt, lambda_dssim, dmaps, partial, out3, (hipStream_t)stream), "l1+ssim forward launch");
    return GSR_OK;
}

int32_t gsr_l1_ssim_backward(gsr_stream_t stream, int32_t C, int32_t H, int32_t W, const float *img, const float *gt,
                             float lambda_dssim, const float *grad_loss, const void *ws, size_t ws_bytes, float *grad_img) {
    size_t need = 0;
    if (gsr_l1_ssim_workspace(C, H, W, &need) != GSR_OK) return GSR_ERR_INVALID_ARGUMENT;
    if (!img || !gt || !ws || !grad_img) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_l1_ssim_backward: null pointer");
    if (ws_bytes < need) return fail(GSR_ERR_WORKSPACE, "loss workspace %zu < %zu", ws_bytes, need);
    HIP_TRY(launch_l1_ssim_backward(C, H, W, img, gt, lambda_dssim, (const float *)ws, grad_loss, grad_img, (hipStream_t)stream),
            "l1+ssim backward launch");
    return GSR_OK;
}

// ---- Adam step over all parameter groups (include/gsr_optim.h) ----
int32_t gsr_adam_step(gsr_stream_t stream, int32_t n_groups, const gsr_adam_group_t *groups, double beta1, double beta2, double eps) {
    if (n_groups < 0 || n_groups > GSR_ADAM_MAX_GROUPS || (n_groups > 0 && !groups))
        return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: %d groups (at most %d)", n_groups, GSR_ADAM_MAX_GROUPS);
    for (int k = 0; k < n_groups; k++) {
        const gsr_adam_group_t &g = groups[k];
        if (g.n < 0 || (g.n > 0 && (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq)) || g.step < 1)
            return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: group %d: n=%lld step=%d or a NULL buffer", k, (long long)g.n, g.step);
    }
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_adam_step: betas");
    HIP_TRY(launch_adam(n_groups, groups, beta1, beta2, eps, (hipStream_t)stream), "adam launch");
    return GSR_OK;
}

// ---- simple_knn.distCUDA2 equivalent (include/gsr_knn.h) ----
int32_t gsr_knn_workspace(int32_t N, size_t *bytes) {
    if (N < 0 || !bytes) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_knn_workspace: bad argument");
    HIP_TRY(knn_workspace_bytes(N, bytes), "knn temp query");
    return GSR_OK;
}

int32_t gsr_knn_mean_dist2(gsr_stream_t stream, int32_t N, const float *points, float *mean_dist2, void *ws, size_t ws_bytes) {
    if (N < 0 || (N > 0 && (!points || !mean_dist2 || !ws))) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_knn_mean_dist2: bad argument");
    size_t need = 0;
    HIP_TRY(knn_workspace_bytes(N, &need), "knn temp query");
    if (N > 0 && ws_bytes < need) return fail(GSR_ERR_WORKSPACE, "knn workspace %zu < %zu", ws_bytes, need);
    HIP_TRY(launch_knn(N, points, mean_dist2, ws, (hipStream_t)stream), "knn launch");
    return GSR_OK;
}

int32_t gsr_mark_visible(gsr_stream_t stream, int32_t P, const float *means3D, const float *viewmatrix,
                         const float *projmatrix, uint8_t *present) {
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_mark_visible: bad argument");
    HIP_TRY(launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream), "mark_visible launch");
    return GSR_OK;
}

int32_t gsr_composited_mask(gsr_stream_t stream, int32_t P, const void *geom_ws, size_t geom_bytes, uint8_t *out) {
    if (P < 0 || (P > 0 && (!geom_ws || !out))) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_composited_mask: bad argument");
    if (P == 0) return GSR_OK;
    size_t stb = 0, dtb = 0;
    HIP_TRY(scan_temp_bytes(P, &stb), "scan temp query");
    HIP_TRY(depth_sort_temp_bytes(P, &dtb), "depth sort temp query");
    const GeomView g = carve_geom(const_cast<void *>(geom_ws), P, stb, dtb);
    if (geom_bytes < g.total_bytes) return fail(GSR_ERR_WORKSPACE, "geom workspace %zu < %zu", geom_bytes, g.total_bytes);
    HIP_TRY(launch_composited_mask(P, g.touched, g.touch_mark, out, (hipStream_t)stream), "composited mask launch");
    return GSR_OK;
}

int32_t gsr_debug_read_geom(gsr_stream_t stream, int32_t P, const void *geom_ws, float *depth, float *xy,
                            float *conic_opacity, float *rgb, uint32_t *tiles_touched, uint8_t *clamped) {
    hipStream_t s = (hipStream_t)stream;
    if (P <= 0) return GSR_OK;
    size_t stb = 0, dtb = 0;
    HIP_TRY(scan_temp_bytes(P, &stb), "scan temp query");
    HIP_TRY(depth_sort_temp_bytes(P, &dtb), "depth sort temp query");
    GeomView g = carve_geom(const_cast<void *>(geom_ws), P, stb, dtb);
    HIP_TRY(hipStreamSynchronize(s), "sync");
    float *rec = (float *)malloc((size_t)P * GSR_REC_FLOATS * sizeof(float));
    uint8_t *cl = (uint8_t *)malloc((size_t)P);
    if (!rec || !cl) { free(rec); free(cl); return fail(GSR_ERR_ALLOC, "host malloc"); }
    hipError_t e = hipMemcpy(rec, g.rec, (size_t)P * GSR_REC_FLOATS * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(cl, g.clamped, (size_t)P, hipMemcpyDeviceToHost);
    if (e == hipSuccess && tiles_touched) e = hipMemcpy(tiles_touched, g.tiles, (size_t)P * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && depth) e = hipMemcpy(depth, g.depth, (size_t)P * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(rec); free(cl); return fail(GSR_ERR_HIP, "debug copy: %s", hipGetErrorString(e)); }
    for (int i = 0; i < P; i++) {
        const float *r = rec + (size_t)i * GSR_REC_FLOATS;
        if (xy) { xy[2 * i] = r[0]; xy[2 * i + 1] = r[1]; }
        if (conic_opacity) { conic_opacity[4 * i] = r[2]; conic_opacity[4 * i + 1] = r[3]; conic_opacity[4 * i + 2] = r[4]; conic_opacity[4 * i + 3] = r[5]; }
        if (rgb) { rgb[3 * i] = r[6]; rgb[3 * i + 1] = r[7]; rgb[3 * i + 2] = r[8]; }
        if (clamped) { clamped[3 * i] = cl[i] & 1; clamped[3 * i + 1] = (cl[i] >> 1) & 1; clamped[3 * i + 2] = (cl[i] >> 2) & 1; }
    }
    free(rec); free(cl);
    return GSR_OK;
}

int32_t gsr_debug_read_binning(gsr_stream_t stream, int64_t N, int32_t W, int32_t H, const void *binning_ws,
                               const void *img_ws, uint64_t *keys_sorted, uint32_t *point_list, uint32_t *ranges) {
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(s), "sync");
    if (N > 0 && binning_ws) {
        BinningView b = carve_binning(const_cast<void *>(binning_ws), N, 0);
        // the 64-bit tile<<32|depth keys only exist in global-sort mode (the other layouts may not even span that region)
        if (keys_sorted) {
            if (g_two_level_sort.load() == 0) HIP_TRY(hipMemcpy(keys_sorted, b.keys_sorted, (size_t)N * 8, hipMemcpyDeviceToHost), "copy keys");
            else memset(keys_sorted, 0, (size_t)N * 8);
        }
        if (point_list) HIP_TRY(hipMemcpy(point_list, b.point_list, (size_t)N * 4, hipMemcpyDeviceToHost), "copy point list");
    }
    if (ranges && img_ws) {
        ImageView im = carve_image(const_cast<void *>(img_ws), W, H);
        const size_t T = (size_t)grid_dim(W) * grid_dim(H);
        HIP_TRY(hipMemcpy(ranges, im.ranges, T * sizeof(uint2), hipMemcpyDeviceToHost), "copy ranges");
        if (point_list && N > 0) {
            // canonical (tile-major) order: tile_lists.hip lays the per-tile slices out super-tile-major; the
            // slices themselves are what the parity tests compare.  Empty tiles read (0, 0) as upstream's do.
            uint32_t *tmp = (uint32_t *)malloc((size_t)N * sizeof(uint32_t));
            if (!tmp) return fail(GSR_ERR_ALLOC, "host malloc");
            size_t pos = 0;
            for (size_t t = 0; t < T; t++) {
                const uint32_t a = ranges[2 * t], e = ranges[2 * t + 1];
                if (e <= a || (size_t)e > (size_t)N || pos + (e - a) > (size_t)N) { ranges[2 * t] = 0; ranges[2 * t + 1] = 0; if (e > a) { free(tmp); return fail(GSR_ERR_INVALID_ARGUMENT, "tile %zu: bad range [%u, %u)", t, a, e); } continue; }
                memcpy(tmp + pos, point_list + a, (size_t)(e - a) * sizeof(uint32_t));
                ranges[2 * t] = (uint32_t)pos; ranges[2 * t + 1] = (uint32_t)(pos + (e - a));
                pos += e - a;
            }
            if (pos != (size_t)N) { free(tmp); return fail(GSR_ERR_INVALID_ARGUMENT, "tile ranges cover %zu of %lld pairs", pos, (long long)N); }
            memcpy(point_list, tmp, (size_t)N * sizeof(uint32_t));
            free(tmp);
        }
    }
    return GSR_OK;
}

int32_t gsr_debug_read_lane_counters(uint64_t *fwd, uint64_t *bwd) {
    DeviceState &ds = dev_state();
    HIP_TRY(hipDeviceSynchronize(), "sync");
    std::lock_guard<std::mutex> lk(ds.mu);
    static_assert(sizeof(CompositeCounters) == 16 * sizeof(uint64_t), "counter block is 16 words");
    if (!ds.counters) {
        if (fwd) memset(fwd, 0, sizeof(CompositeCounters));
        if (bwd) memset(bwd, 0, sizeof(CompositeCounters));
        return GSR_OK;
    }
    if (fwd) HIP_TRY(hipMemcpy(fwd, ds.counters, sizeof(CompositeCounters), hipMemcpyDeviceToHost), "copy counters");
    if (bwd) HIP_TRY(hipMemcpy(bwd, ds.counters + 1, sizeof(CompositeCounters), hipMemcpyDeviceToHost), "copy counters");
    if (fwd) { fwd[9] = 0; fwd[10] = 0; }      // device pointer / capacity of the trace: not counters
    if (bwd) { bwd[9] = 0; bwd[10] = 0; }
    for (int w = 0; w < 2; w++)                 // reset the nine counters, keep the trace pointers
        HIP_TRY(hipMemset(ds.counters + w, 0, 9 * sizeof(unsigned long long)), "reset counters");
    return GSR_OK;
}

int32_t gsr_debug_read_wave_trace(int32_t which, uint32_t *out, int64_t max_units) {
    DeviceState &ds = dev_state();
    HIP_TRY(hipDeviceSynchronize(), "sync");
    std::lock_guard<std::mutex> lk(ds.mu);
    if (which < 0 || which > 1 || !out || max_units < 0) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_debug_read_wave_trace: bad argument");
    if (!ds.counters) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_debug_read_wave_trace: count_lanes has not run on this device");
    const size_t n = (size_t)(max_units < (int64_t)GSR_TRACE_UNITS ? max_units : (int64_t)GSR_TRACE_UNITS);
    const char *base = reinterpret_cast<const char *>(ds.counters + 2) + (size_t)which * GSR_TRACE_UNITS * sizeof(uint4);
    HIP_TRY(hipMemcpy(out, base, n * sizeof(uint4), hipMemcpyDeviceToHost), "copy trace");
    HIP_TRY(hipMemset(const_cast<char *>(base), 0, (size_t)GSR_TRACE_UNITS * sizeof(uint4)), "reset trace");
    return GSR_OK;
}

int32_t gsr_debug_read_bound_errors(gsr_stream_t stream, int32_t P, const void *geom_ws, int32_t W, int32_t H, const void *img_ws, uint32_t *out) {
    hipStream_t s = (hipStream_t)stream;
    if (!out) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_debug_read_bound_errors: bad argument");
    memset(out, 0, 12 * sizeof(uint32_t));
#ifdef GSR_DEBUG_BOUNDS
    out[8] = 1u;
#endif
    HIP_TRY(hipStreamSynchronize(s), "sync");
    if (geom_ws && P > 0) {
        size_t stb = 0, dtb = 0;
        HIP_TRY(scan_temp_bytes(P, &stb), "scan temp query");
        HIP_TRY(depth_sort_temp_bytes(P, &dtb), "depth sort temp query");
        const GeomView g = carve_geom(const_cast<void *>(geom_ws), P, stb, dtb);
        HIP_TRY(hipMemcpy(out, g.dord.hdr + GSR_DBG_GEOM_WORD, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy bound words");
    }
    if (img_ws && W > 0 && H > 0) {
        const ImageView im = carve_image(const_cast<void *>(img_ws), W, H);
        HIP_TRY(hipMemcpy(out + 4, im.seg.hdr + GSR_DBG_SEG_WORD, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy bound words");
        // self test of the mechanism, on words of its own (seg.hdr[8..11])
        HIP_TRY(hipMemsetAsync(im.seg.hdr + 8, 0, 4 * sizeof(uint32_t), s), "clear self-test words");
        HIP_TRY(launch_bound_selftest(im.seg.hdr + 8, s), "bound self test");
        HIP_TRY(hipStreamSynchronize(s), "sync");
        uint32_t st[4];
        HIP_TRY(hipMemcpy(st, im.seg.hdr + 8, sizeof(st), hipMemcpyDeviceToHost), "copy self-test words");
        out[9] = st[0]; out[10] = st[1]; out[11] = st[2];
    }
    return GSR_OK;
}

int32_t gsr_debug_read_segments(gsr_stream_t stream, int32_t W, int32_t H, const void *img_ws, uint32_t *summary) {
    hipStream_t s = (hipStream_t)stream;
    if (!img_ws || !summary || W <= 0 || H <= 0) return fail(GSR_ERR_INVALID_ARGUMENT, "gsr_debug_read_segments: bad argument");
    HIP_TRY(hipStreamSynchronize(s), "sync");
    ImageView im = carve_image(const_cast<void *>(img_ws), W, H);
    uint32_t *h = (uint32_t *)malloc(GSR_SEG_HDR_WORDS * sizeof(uint32_t));
    if (!h) return fail(GSR_ERR_ALLOC, "host malloc");
    const hipError_t e = hipMemcpy(h, im.seg.hdr, GSR_SEG_HDR_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(h); return fail(GSR_ERR_HIP, "copy segment header: %s", hipGetErrorString(e)); }
    uint32_t units = 0, slots = 0, drawn = 0;
    for (int b = 0; b < GSR_SEG_BANDS; b++) {
        units += h[SEG_BCOUNT + b];
        const uint32_t share = im.seg.pool_cap / GSR_SEG_BANDS, got = h[SEG_POOL + GSR_SEG_CTR_STRIDE * b];
        slots += got < share ? got : share;
        drawn += h[SEG_BTICKET + GSR_SEG_CTR_STRIDE * b];
    }
    summary[0] = h[SEG_SEG]; summary[1] = units; summary[2] = slots; summary[3] = im.seg.pool_cap; summary[4] = im.seg.units; summary[5] = drawn;
    summary[6] = 0; summary[7] = 0;
    free(h);
    return GSR_OK;
}

int32_t gsr_debug_read_image_state(gsr_stream_t stream, int32_t W, int32_t H, const void *img_ws, float *final_T,
                                   uint32_t *n_contrib) {
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(s), "sync");
    ImageView im = carve_image(const_cast<void *>(img_ws), W, H);
    const size_t HW = (size_t)W * H;
    if (final_T) HIP_TRY(hipMemcpy(final_T, im.final_T, HW * 4, hipMemcpyDeviceToHost), "copy final_T");
    if (n_contrib) HIP_TRY(hipMemcpy(n_contrib, im.n_contrib, HW * 4, hipMemcpyDeviceToHost), "copy n_contrib");
    return GSR_OK;
}

}  // extern "C"
