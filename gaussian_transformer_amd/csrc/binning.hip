// binning.hip -- the rocPRIM-based binning path: depth sort of the Gaussians + scan of tiles-touched, (tile, depth)
// key emission (S7), radix sort of the pairs, per-tile range detection (S8).  The default path is depth_order.hip +
// tile_lists.hip; this one runs when those do not apply (a depth bucket beyond LDS capacity, more than 512
// super-tiles) and under gsr_set_option("depth_buckets" / "tile_lists" / "two_level_sort", 0) as the reference
// the parity tests compare the fast paths with.
//
// Key emission is work-balanced in two levels (rows of the splat rectangles, then output pairs), each a
// flattened walk with a 6-step ds_bpermute search over a wave-wide scan -- see emit_keys_kernel.
#include <cstring>  // ROCm 7.2 rocprim/texture_cache_iterator.hpp uses memset without including it
#include <mutex>

#include <rocprim/rocprim.hpp>

#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

// rocPRIM's temp-size queries look the device up on every call (microseconds each, three per gsr_forward and per gsr_backward: a
// tenth of a small scene's host time); the answers depend on P and the device only
struct TempSizeCache {
    std::mutex mu;
    struct E { int dev, P; size_t scan, dsort; bool has_scan, has_dsort; } e[8] = {};
    int next = 0;
    E *find(int dev, int P) {
        for (auto &x : e) if ((x.has_scan || x.has_dsort) && x.dev == dev && x.P == P) return &x;
        return nullptr;
    }
    E *slot(int dev, int P) {
        E *x = find(dev, P);
        if (x) return x;
        x = &e[next]; next = (next + 1) % 8;
        *x = E{dev, P, 0, 0, false, false};
        return x;
    }
};
static TempSizeCache g_tsc;
static int current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : 0; }

hipError_t scan_temp_bytes(int P, size_t *bytes) {
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_tsc.mu);
        TempSizeCache::E *x = g_tsc.find(dev, P);
        if (x && x->has_scan) { *bytes = x->scan; return hipSuccess; }
    }
    size_t tb = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (size_t)(P > 0 ? P : 1), rocprim::plus<uint32_t>(), (hipStream_t)0, false);
    *bytes = tb;
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_tsc.mu);
        TempSizeCache::E *x = g_tsc.slot(dev, P);
        x->scan = tb; x->has_scan = true;
    }
    return e;
}

hipError_t sort_temp_bytes(int64_t N, int bits, size_t *bytes) {
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (size_t)(N > 0 ? N : 1), 0u, (unsigned)bits, (hipStream_t)0, false);
    *bytes = tb;
    return e;
}

hipError_t launch_sort(const BinningView &b, int64_t N, int bits, hipStream_t s) {
    size_t tb = b.sort_temp_bytes;
    return rocprim::radix_sort_pairs(b.sort_temp, tb, (const uint64_t *)b.keys_unsorted, b.keys_sorted,
                                     (const uint32_t *)b.point_list_unsorted, b.point_list, (size_t)N, 0u,
                                     (unsigned)bits, s, false);
}

// ---- two-level sort.  Level 1 sorts the P GAUSSIANS by depth (stable radix sort of 32-bit depth bits,
// value = Gaussian id): 8 bytes x P instead of 12 bytes x N.  Key emission then walks the Gaussians in
// that order, so the pairs leave in global depth order, and level 2 -- a STABLE rocPRIM radix sort of
// (tile id, Gaussian id) pairs on the ceil(log2 T) tile bits only (2 onesweep passes moving 8 bytes per
// pair, instead of 5 passes over 32+log2 T bits moving 12) -- keeps every tile's slice in depth order.
// Ties in depth keep ascending Gaussian id (both sorts are stable), i.e. exactly the order of one global
// stable sort on tile<<32|depth with emission in id order.
// Onesweep for every size: rocPRIM's default switches to a merge sort below 1 M items, which costs
// ~20 small launches (~140 us) for 1 M Gaussians against ~5 launches for the radix passes.
using DepthSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

// below this size rocPRIM's default choice (single-block / merge sort) wins: 10 k items 15 us vs 118 us
#define GSR_DEPTH_SORT_ONESWEEP_MIN 262144

hipError_t depth_sort_temp_bytes(int P, size_t *bytes) {
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_tsc.mu);
        TempSizeCache::E *x = g_tsc.find(dev, P);
        if (x && x->has_dsort) { *bytes = x->dsort; return hipSuccess; }
    }
    size_t tb = 0, tb2 = 0;
    hipError_t e2 = rocprim::radix_sort_pairs(nullptr, tb2, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                              rocprim::counting_iterator<uint32_t>(0), (uint32_t *)nullptr,
                                              (size_t)(P > 0 ? P : 1), 0u, 32u, (hipStream_t)0, false);
    if (e2 != hipSuccess) return e2;
    hipError_t e = rocprim::radix_sort_pairs<DepthSortConfig>(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             rocprim::counting_iterator<uint32_t>(0), (uint32_t *)nullptr,
                                             (size_t)(P > 0 ? P : 1), 0u, 32u, (hipStream_t)0, false);
    *bytes = tb > tb2 ? tb : tb2;
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_tsc.mu);
        TempSizeCache::E *x = g_tsc.slot(dev, P);
        x->dsort = *bytes; x->has_dsort = true;
    }
    return e;
}

// Level 1: Gaussian ids in (depth, id) order.  Culled Gaussians carry depth 0 and emit no pairs, so
// their position is irrelevant.  (Running this on a helper stream beside preprocess was tried: the
// radix passes just queue behind preprocess's resident workgroups, no overlap -- kept on one stream.)
hipError_t launch_depth_sort(const GeomView &g, int P, hipStream_t s) {
    size_t tb = g.dsort_temp_bytes;
    if (P < GSR_DEPTH_SORT_ONESWEEP_MIN)
        return rocprim::radix_sort_pairs(g.dsort_temp, tb, reinterpret_cast<const uint32_t *>(g.depth), g.depth_sorted,
                                         rocprim::counting_iterator<uint32_t>(0), g.perm, (size_t)P, 0u, 32u, s, false);
    return rocprim::radix_sort_pairs<DepthSortConfig>(g.dsort_temp, tb, reinterpret_cast<const uint32_t *>(g.depth), g.depth_sorted,
                                                      rocprim::counting_iterator<uint32_t>(0), g.perm, (size_t)P, 0u, 32u, s, false);
}

struct TilesOf {
    const uint32_t *tiles;
    __host__ __device__ uint32_t operator()(uint32_t g) const { return tiles[g]; }
};

// perm = Gaussian ids in (depth, id) order (launch_depth_sort); offsets[i] = inclusive scan of tiles[perm[i]]
hipError_t launch_ordered_scan(const GeomView &g, int P, hipStream_t s) {
    size_t sb = g.scan_temp_bytes;
    auto in = rocprim::make_transform_iterator(static_cast<const uint32_t *>(g.perm), TilesOf{g.tiles});
    return rocprim::inclusive_scan(g.scan_temp, sb, in, g.offsets, (size_t)P, rocprim::plus<uint32_t>(), s, false);
}

hipError_t sort2_temp_bytes(int64_t N, int tile_bits, size_t *bytes) {
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (size_t)(N > 0 ? N : 1), 0u, (unsigned)tile_bits, (hipStream_t)0, false);
    *bytes = tb;
    return e;
}

hipError_t launch_sort2_by_tile(const BinningView &b, int64_t N, int tile_bits, hipStream_t s) {
    size_t tb = b.sort_temp_bytes;
    return rocprim::radix_sort_pairs(b.sort_temp, tb, (const uint32_t *)b.tkeys_unsorted, b.tkeys_sorted,
                                     (const uint32_t *)b.ids_unsorted, b.point_list, (size_t)N, 0u,
                                     (unsigned)tile_bits, s, false);
}

// wave-wide inclusive scan (all 64 lanes active)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
// number of lanes whose inclusive-scan value is <= j  (== index of the lane that owns slot j)
__device__ __forceinline__ int owner_lane(uint32_t incl, uint32_t j) {
    int lo = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        const uint32_t v = __shfl(incl, (lo + step - 1) & 63);
        if (v <= j) lo += step;
    }
    return lo;
}

// Key emission, two balanced levels.  A wave owns 64 consecutive Gaussians of the depth-ordered list.
// Level 1 walks the wave's flattened list of (Gaussian, tile ROW) items, one item per lane per step: the
// ellipse-vs-row span (two sqrt) is evaluated once per row, not once per candidate tile.  Level 2 walks
// the flattened list of the step's output pairs, one pair per lane: a splat covering a thousand tiles
// costs the same per pair as one covering four, and every store is a contiguous 64-lane row.
__global__ __launch_bounds__(256) void emit_keys_kernel(int P, int W, int H, int gridx, int exact_cull, int two_level,
                                                        const uint32_t *__restrict__ perm,
                                                        const uint32_t *__restrict__ tiles,
                                                        const uint32_t *__restrict__ offsets,
                                                        const uint4 *__restrict__ rect, const float *__restrict__ rec,
                                                        uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        uint32_t *__restrict__ tkeys, uint32_t *__restrict__ ids) {
    const int lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (g0 >= P) return;                                  // wave-uniform
    // the wave owns positions [g0, g0+64) of the depth-ordered Gaussian list; `offsets` is in that order
    const int g = g0 + lane;
    const uint32_t out_start = g0 > 0 ? offsets[g0 - 1] : 0u;   // first output slot of this wave
    const uint32_t out_total = offsets[min(g0 + 63, P - 1)] - out_start;
    if (out_total == 0) return;                           // wave-uniform
    const int gc = (int)perm[g < P ? g : P - 1];          // Gaussian id of this lane
    const uint4 rc4 = rect[gc];
    const uint2 rc = make_uint2(rc4.x, rc4.y);
    const uint32_t rh = (rc.y >> 16) - (rc.y & 0xffffu);
    // level-1 items = tile rows of the 3-sigma rectangle; splats that emit nothing are not walked at all
    const uint32_t nrows = (g < P && tiles[gc] > 0u) ? rh : 0u;
    const uint32_t rincl = wave_inclusive_scan(nrows, lane);
    const uint32_t rexcl = rincl - nrows;
    const uint32_t rtotal = __shfl(rincl, 63);
    const float4 r0 = reinterpret_cast<const float4 *>(rec)[3 * (size_t)gc];
    const float4 r1 = reinterpret_cast<const float4 *>(rec)[3 * (size_t)gc + 1];
    const float4 r2 = reinterpret_cast<const float4 *>(rec)[3 * (size_t)gc + 2];
    const uint32_t dbits = __float_as_uint(r2.y);
    const CullParams cp = make_cull(r0.z, r0.w, r1.x, r2.z);
    uint32_t running = 0;
    for (uint32_t rbase = 0; rbase < rtotal; rbase += 64) {      // wave-uniform trip count
        // ---- level 1: one (Gaussian, row) item per lane ----
        const uint32_t j = rbase + lane;
        const int src = owner_lane(rincl, j) & 63;               // == 64 only for j >= rtotal (masked below)
        const uint32_t rx = __shfl(rc.x, src), ry = __shfl(rc.y, src);
        const uint32_t x0 = rx & 0xffffu, x1 = rx >> 16, y0 = ry & 0xffffu;
        const uint32_t row = y0 + (j - __shfl(rexcl, src));
        const uint32_t gid = __shfl((uint32_t)gc, src);
        const uint32_t db = __shfl(dbits, src);
        int c0 = (int)x0, c1 = (int)x1;
        if (exact_cull) {                                        // wave-uniform
            CullParams c;
            c.tau = __shfl(cp.tau, src); c.xmax = __shfl(cp.xmax, src); c.ymax = __shfl(cp.ymax, src);
            c.dy_at_xmax = __shfl(cp.dy_at_xmax, src); c.det = __shfl(cp.det, src);
            const float px = __shfl(r0.x, src), py = __shfl(r0.y, src), A = __shfl(r0.z, src), B = __shfl(r0.w, src);
            tile_row_span(c, px, py, A, B, (int)row, W, H, (int)x0, (int)x1, c0, c1);
        }
        const uint32_t cnt = j < rtotal ? (uint32_t)(c1 - c0) : 0u;
        const uint32_t cincl = wave_inclusive_scan(cnt, lane);
        const uint32_t cexcl = cincl - cnt;
        const uint32_t ctotal = __shfl(cincl, 63);
        const uint32_t tile0 = row * (uint32_t)gridx + (uint32_t)c0;      // first tile of this item's span
        // ---- level 2: one output pair per lane ----
        for (uint32_t obase = 0; obase < ctotal; obase += 64) {           // wave-uniform trip count
            const uint32_t t = obase + lane;
            const int it = owner_lane(cincl, t) & 63;
            const uint32_t tile = __shfl(tile0, it) + (t - __shfl(cexcl, it));
            const uint32_t id = __shfl(gid, it);
            const uint32_t d = __shfl(db, it);
            const uint32_t slot = running + t;
            if (t < ctotal && slot < out_total) {                 // slot < out_total always holds (same span function as the count)
                const size_t o = (size_t)out_start + slot;
                if (two_level) {                                  // wave-uniform
                    tkeys[o] = tile;
                    ids[o] = id;
                } else {
                    keys[o] = ((uint64_t)tile << 32) | d;
                    vals[o] = id;
                }
            }
        }
        running += ctotal;
    }
}

// P = entries of the depth-ordered list (all Gaussians after the rocPRIM sort, the emitting ones after depth_order.hip)
hipError_t launch_emit_keys(const GeomView &g, const BinningView &b, int P, int W, int H, int exact_cull, int two_level,
                            hipStream_t s) {
    if (P <= 0) return hipSuccess;
    const int gridx = (W + GSR_TILE - 1) / GSR_TILE;
    // two-level: level 2 writes the sorted ids straight into point_list
    hipLaunchKernelGGL(emit_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, W, H, gridx, exact_cull, two_level,
                       g.perm, g.tiles, g.offsets, g.rect, g.rec, b.keys_unsorted, b.point_list_unsorted, b.tkeys_unsorted,
                       b.ids_unsorted);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(int64_t N, const uint64_t *__restrict__ keys,
                                                          uint2 *__restrict__ ranges) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const uint32_t t = (uint32_t)(keys[j] >> 32);
    if (j == 0) {
        ranges[t].x = 0u;
    } else {
        const uint32_t tp = (uint32_t)(keys[j - 1] >> 32);
        if (tp != t) { ranges[tp].y = (uint32_t)j; ranges[t].x = (uint32_t)j; }
    }
    if (j == N - 1) ranges[t].y = (uint32_t)N;
}

__global__ __launch_bounds__(256) void tile_ranges32_kernel(int64_t N, const uint32_t *__restrict__ tkeys,
                                                            uint2 *__restrict__ ranges) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const uint32_t t = tkeys[j];
    if (j == 0) {
        ranges[t].x = 0u;
    } else {
        const uint32_t tp = tkeys[j - 1];
        if (tp != t) { ranges[tp].y = (uint32_t)j; ranges[t].x = (uint32_t)j; }
    }
    if (j == N - 1) ranges[t].y = (uint32_t)N;
}

hipError_t launch_ranges(const BinningView &b, const ImageView &im, int64_t N, int T, int two_level, hipStream_t s) {
    hipError_t e = hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)T, s);
    if (e != hipSuccess || N <= 0) return e;
    const dim3 grid((unsigned)((N + 255) / 256));
    if (two_level) hipLaunchKernelGGL(tile_ranges32_kernel, grid, dim3(256), 0, s, N, b.tkeys_sorted, im.ranges);
    else hipLaunchKernelGGL(tile_ranges_kernel, grid, dim3(256), 0, s, N, b.keys_sorted, im.ranges);
    return hipGetLastError();
}

}  // namespace gsr
