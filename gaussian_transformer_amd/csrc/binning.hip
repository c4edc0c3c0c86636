// binning.hip -- scan of tiles-touched, (tile, depth) key emission (S7), global radix sort,
// per-tile range detection (S8).  All HBM-streaming stages.
//
// Key emission is work-balanced: a wave owns 64 consecutive Gaussians and its lanes walk the
// wave's flattened output range (one output pair per lane per step, found by a 6-step search
// over the wave's 64 scan values with ds_bpermute), so a splat covering a thousand tiles costs
// the same per pair as one covering four, and every store is a contiguous 64-lane row.
#include <cstring>  // ROCm 7.2 rocprim/texture_cache_iterator.hpp uses memset without including it
#include <rocprim/rocprim.hpp>

#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

hipError_t scan_temp_bytes(int P, size_t *bytes) {
    size_t tb = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (size_t)(P > 0 ? P : 1), rocprim::plus<uint32_t>(), (hipStream_t)0, false);
    *bytes = tb;
    return e;
}

hipError_t launch_scan(const GeomView &g, int P, hipStream_t s) {
    size_t tb = g.scan_temp_bytes;
    return rocprim::inclusive_scan(g.scan_temp, tb, (const uint32_t *)g.tiles, g.offsets, (size_t)P,
                                   rocprim::plus<uint32_t>(), s, false);
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

// ---- two-level sort: rocPRIM orders the pairs by TILE only (stable, 2 onesweep passes over
// ceil(log2 T) bits instead of 5 over 32+log2 T), then one workgroup per tile orders its
// contiguous slice by (depth bits, Gaussian id) in LDS.  Within a tile the stable tile sort keeps
// emission order = ascending Gaussian id, so (depth, id) reproduces exactly the order of a global
// stable sort on tile<<32|depth.
hipError_t sort2_temp_bytes(int64_t N, int tile_bits, size_t *bytes) {
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (size_t)(N > 0 ? N : 1), 0u, (unsigned)tile_bits, (hipStream_t)0, false);
    *bytes = tb;
    return e;
}

hipError_t launch_sort2_by_tile(const BinningView &b, int64_t N, int tile_bits, hipStream_t s) {
    size_t tb = b.sort_temp_bytes;
    return rocprim::radix_sort_pairs(b.sort_temp, tb, (const uint32_t *)b.tkeys_unsorted, b.tkeys_sorted,
                                     (const uint64_t *)b.dvals_unsorted, b.dvals_sorted, (size_t)N, 0u,
                                     (unsigned)tile_bits, s, false);
}

// One workgroup per tile: bitonic sort of the tile's (depth<<32 | id) words in LDS, ids out.
// CAP = LDS capacity in elements; the kernel instance handles tiles with LO < n <= CAP.  The last
// instance (CAP = 16384) also takes larger tiles through a (slow) in-place global-memory path.
template <int CAP, int LO, bool LAST>
__global__ __launch_bounds__(256) void tile_depth_sort_kernel(const uint2 *__restrict__ ranges,
                                                              uint64_t *__restrict__ dvals,
                                                              uint32_t *__restrict__ point_list,
                                                              uint32_t *__restrict__ scratch) {
    extern __shared__ __align__(16) uint64_t lds_keys[];
    const uint2 r = ranges[blockIdx.x];
    const uint32_t n = r.y - r.x;
    if (n <= (uint32_t)LO || (!LAST && n > (uint32_t)CAP)) return;     // workgroup-uniform
    const uint32_t tid = threadIdx.x;
    const uint64_t *src = dvals + r.x;
    if (n == 1) { if (tid == 0) point_list[r.x] = (uint32_t)src[0]; return; }
    uint32_t m = 4;
    while (m < n) m <<= 1;
    if (LAST && n > (uint32_t)CAP) {
        // rare (> 16384 pairs in one tile): rank sort, keys streamed through LDS in CAP-sized chunks.
        // All keys of a tile are distinct (the id is part of the key), so rank = #smaller keys.
        for (uint32_t i = tid; i < n; i += 256) scratch[r.x + i] = 0u;
        for (uint32_t c0 = 0; c0 < n; c0 += CAP) {
            const uint32_t cn = min((uint32_t)CAP, n - c0);
            __syncthreads();
            for (uint32_t i = tid; i < cn; i += 256) lds_keys[i] = src[c0 + i];
            __syncthreads();
            for (uint32_t i = tid; i < n; i += 256) {
                const uint64_t mine = src[i];
                uint32_t less = 0;
                for (uint32_t t = 0; t < cn; t++) less += lds_keys[t] < mine ? 1u : 0u;
                scratch[r.x + i] += less;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) point_list[r.x + scratch[r.x + i]] = (uint32_t)src[i];
        return;
    }
    if (n == 2) {
        if (tid == 0) {
            const uint64_t a = src[0], bb = src[1];
            point_list[r.x] = (uint32_t)(a < bb ? a : bb);
            point_list[r.x + 1] = (uint32_t)(a < bb ? bb : a);
        }
        return;
    }
    // ---- bitonic network, m = next power of two >= n (>= 4), padded with +inf ----
    // Each thread owns groups of 4 consecutive elements: the j = 2 and j = 1 steps of every level run in
    // registers (one 32-byte LDS read + write per group instead of two passes).  Each wave owns whole
    // 256-element blocks, so steps with j <= 128 need no workgroup barrier (LDS operations of one wave
    // execute in order); only the cross-block steps (j >= 256: 1 for m = 512, 3 for m = 1024, ...) do.
    for (uint32_t i = tid; i < m; i += 256) lds_keys[i] = i < n ? src[i] : ~0ull;
    __syncthreads();
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t bsize = m < 256u ? m : 256u;            // elements per block
    const uint32_t nblk = m / bsize;
    auto cex = [](uint64_t &a, uint64_t &bb, bool asc) __attribute__((always_inline)) {
        const bool sw = (a > bb) == asc;
        const uint64_t lo = sw ? bb : a, hi = sw ? a : bb;
        a = lo; bb = hi;
    };
    using u64x2 = uint64_t __attribute__((ext_vector_type(2)));
    // phase 0: levels k = 2 and k = 4 == sort every 4-group, ascending iff bit 2 of its base index is 0
    for (uint32_t b = wave; b < nblk; b += 4)
        for (uint32_t q = lane; q < (bsize >> 2); q += 64) {
            const uint32_t i0 = b * 256 + q * 4;
            u64x2 *p = reinterpret_cast<u64x2 *>(lds_keys + i0);
            u64x2 lo2 = p[0], hi2 = p[1];
            uint64_t e0 = lo2.x, e1 = lo2.y, e2 = hi2.x, e3 = hi2.y;
            const bool asc = (m == 4u) || ((i0 & 4u) == 0u);
            cex(e0, e1, asc); cex(e2, e3, asc); cex(e0, e2, asc); cex(e1, e3, asc); cex(e1, e2, asc);
            lo2.x = e0; lo2.y = e1; hi2.x = e2; hi2.y = e3;
            p[0] = lo2; p[1] = hi2;
        }
    for (uint32_t k = 8; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j >= 4; j >>= 1) {
            const uint32_t sh = 31u - (uint32_t)__builtin_clz(j);          // log2 j
            if (j >= 256u) {                                               // cross-block step
                __syncthreads();
                for (uint32_t p = tid; p < (m >> 1); p += 256) {
                    const uint32_t i = ((p >> sh) << (sh + 1)) + (p & (j - 1)), l = i + j;
                    uint64_t a = lds_keys[i], bb = lds_keys[l];
                    cex(a, bb, (i & k) == 0u);
                    lds_keys[i] = a; lds_keys[l] = bb;
                }
                __syncthreads();
            } else {                                                       // inside this wave's blocks
                __builtin_amdgcn_wave_barrier();
                for (uint32_t b = wave; b < nblk; b += 4)
                    for (uint32_t lp = lane; lp < (bsize >> 1); lp += 64) {
                        const uint32_t i = b * 256 + ((lp >> sh) << (sh + 1)) + (lp & (j - 1)), l = i + j;
                        uint64_t a = lds_keys[i], bb = lds_keys[l];
                        cex(a, bb, (i & k) == 0u);
                        lds_keys[i] = a; lds_keys[l] = bb;
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t b = wave; b < nblk; b += 4)                          // j = 2 and j = 1 in registers
            for (uint32_t q = lane; q < (bsize >> 2); q += 64) {
                const uint32_t i0 = b * 256 + q * 4;
                u64x2 *p = reinterpret_cast<u64x2 *>(lds_keys + i0);
                u64x2 lo2 = p[0], hi2 = p[1];
                uint64_t e0 = lo2.x, e1 = lo2.y, e2 = hi2.x, e3 = hi2.y;
                const bool asc = (i0 & k) == 0u;
                cex(e0, e2, asc); cex(e1, e3, asc); cex(e0, e1, asc); cex(e2, e3, asc);
                lo2.x = e0; lo2.y = e1; hi2.x = e2; hi2.y = e3;
                p[0] = lo2; p[1] = hi2;
            }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) point_list[r.x + i] = (uint32_t)lds_keys[i];
}

hipError_t launch_tile_depth_sort(const BinningView &b, const ImageView &im, int T, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)tile_depth_sort_kernel<16384, 4096, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
        attr_set = true;
    }
    hipLaunchKernelGGL((tile_depth_sort_kernel<1024, 0, false>), dim3(T), dim3(256), 1024 * 8, s, im.ranges, b.dvals_sorted, b.point_list, (uint32_t *)b.dvals_unsorted);
    hipLaunchKernelGGL((tile_depth_sort_kernel<4096, 1024, false>), dim3(T), dim3(256), 4096 * 8, s, im.ranges, b.dvals_sorted, b.point_list, (uint32_t *)b.dvals_unsorted);
    hipLaunchKernelGGL((tile_depth_sort_kernel<16384, 4096, true>), dim3(T), dim3(256), 16384 * 8, s, im.ranges, b.dvals_sorted, b.point_list, (uint32_t *)b.dvals_unsorted);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void emit_keys_kernel(int P, int W, int H, int gridx, int exact_cull, int two_level,
                                                        const uint32_t *__restrict__ tiles,
                                                        const uint32_t *__restrict__ offsets,
                                                        const uint2 *__restrict__ rect, const float *__restrict__ rec,
                                                        uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        uint32_t *__restrict__ tkeys, uint64_t *__restrict__ dvals) {
    const int lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (g0 >= P) return;                                  // wave-uniform
    const int g = g0 + lane;
    const int gc = g < P ? g : P - 1;
    const uint32_t out_start = g0 > 0 ? offsets[g0 - 1] : 0u;   // first output slot of this wave
    const uint32_t out_total = offsets[min(g0 + 63, P - 1)] - out_start;
    if (out_total == 0) return;                           // wave-uniform
    const uint2 rc = rect[gc];
    const uint32_t rw = (rc.x >> 16) - (rc.x & 0xffffu), rh = (rc.y >> 16) - (rc.y & 0xffffu);
    // candidates = tiles of the 3-sigma rectangle; splats that emit nothing are not walked at all
    const uint32_t cand = (g < P && tiles[gc] > 0u) ? rw * rh : 0u;
    uint32_t incl = cand;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    const uint32_t excl = incl - cand;
    const uint32_t total = __shfl(incl, 63);
    const float4 r0 = reinterpret_cast<const float4 *>(rec)[3 * (size_t)gc];
    const float4 r1 = reinterpret_cast<const float4 *>(rec)[3 * (size_t)gc + 1];
    const float4 r2 = reinterpret_cast<const float4 *>(rec)[3 * (size_t)gc + 2];
    const uint32_t dbits = __float_as_uint(r2.y);
    const CullParams cp = make_cull(r0.z, r0.w, r1.x, r2.z);
    uint32_t running = 0;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // wave-uniform trip count: every lane stays active for the cross-lane reads
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t j = base + lane;
        int lo = 0;                                       // number of lanes whose incl <= j
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const uint32_t v = __shfl(incl, (lo + step - 1) & 63);
            if (v <= j) lo += step;
        }
        const int src = lo & 63;                          // lo == 64 only for j >= total (not stored)
        const uint32_t e = __shfl(excl, src);
        const uint32_t rx = __shfl(rc.x, src), ry = __shfl(rc.y, src);
        const uint32_t db = __shfl(dbits, src);
        const uint32_t x0 = rx & 0xffffu, x1 = rx >> 16, y0 = ry & 0xffffu;
        const uint32_t w = max(x1 - x0, 1u);
        const uint32_t k = j - e;
        const uint32_t ty = k / w, tx = k - ty * w;
        bool pass = j < total;
        if (exact_cull) {                                 // wave-uniform
            CullParams c;
            c.tau = __shfl(cp.tau, src); c.xmax = __shfl(cp.xmax, src); c.ymax = __shfl(cp.ymax, src);
            c.dy_at_xmax = __shfl(cp.dy_at_xmax, src); c.det = __shfl(cp.det, src);
            const float px = __shfl(r0.x, src), py = __shfl(r0.y, src), A = __shfl(r0.z, src), B = __shfl(r0.w, src);
            int c0, c1;
            tile_row_span(c, px, py, A, B, (int)(y0 + ty), W, H, (int)x0, (int)x1, c0, c1);
            pass = pass && ((int)(x0 + tx) >= c0) && ((int)(x0 + tx) < c1);
        }
        const uint64_t ballot = __ballot(pass);
        const uint32_t slot = running + (uint32_t)__popcll(ballot & lt_mask);
        if (pass && slot < out_total) {                   // slot < out_total always holds (same span function as the count)
            const uint32_t tile = (y0 + ty) * (uint32_t)gridx + (x0 + tx);
            const size_t o = (size_t)out_start + slot;
            if (two_level) {                              // wave-uniform
                tkeys[o] = tile;
                dvals[o] = ((uint64_t)db << 32) | (uint32_t)(g0 + src);
            } else {
                keys[o] = ((uint64_t)tile << 32) | db;
                vals[o] = (uint32_t)(g0 + src);
            }
        }
        running += (uint32_t)__popcll(ballot);
    }
}

hipError_t launch_emit_keys(const GeomView &g, const BinningView &b, int P, int W, int H, int exact_cull, int two_level,
                            hipStream_t s) {
    if (P <= 0) return hipSuccess;
    const int gridx = (W + GSR_TILE - 1) / GSR_TILE;
    hipLaunchKernelGGL(emit_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, W, H, gridx, exact_cull, two_level,
                       g.tiles, g.offsets, g.rect, g.rec, b.keys_unsorted, b.point_list_unsorted, b.tkeys_unsorted,
                       b.dvals_unsorted);
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
