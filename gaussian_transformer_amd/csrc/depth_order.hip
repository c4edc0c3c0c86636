// depth_order.hip -- the depth order of the emitting Gaussians and the scan of their pair counts, without
// a general radix sort.
//
// Output (same contract as launch_depth_sort + launch_ordered_scan of binning.hip):
//   perm[0..Pv)     ids of the Pv Gaussians that emit at least one (tile, Gaussian) pair, ordered by
//                   (depth bits, id) -- the order a stable sort on depth of id-ordered input gives
//   offsets[0..Pv)  inclusive scan of tiles[perm[.]]        (only when key emission will run: NEED_OFFSETS)
//   orect[0..Pv)    rect[perm[.]], the 16-byte rectangle + span records in depth order   (for tile_lists.hip)
//   hdr[DO_PV], hdr[DO_NTOT], hdr[DO_ETOT]  Pv, N (total pairs) and E (super-tile entries); hdr[DO_OVERFLOW] != 0 ->
//                   nothing usable, the caller falls back to the rocPRIM path.
//
// rocPRIM's onesweep needs 4 digit passes + histogram + scan + 4 fills (~165 us for 1 M Gaussians, almost all
// of it launch latency and decoupled-lookback chains over 8 MB of data).  Here: the depth range is cut into
// nb level-1 buckets (linear in depth, monotone), one counting pass + one scatter pass (workgroup-local LDS
// histograms; one global atomic per workgroup and touched bucket) place every Gaussian in its bucket in
// arbitrary order, and one workgroup per bucket finishes the job in LDS:
// a second, finer counting split (GSR_DO_NSUB sub-buckets) and a rank-by-counting inside each sub-bucket on the
// full 64-bit (depth bits, id) key, so the result does not depend on arrival order.  A bucket whose keys pile
// up in one sub-bucket (coplanar splats) is sorted by an in-LDS bitonic network instead: bounded time for any
// input.  The same workgroup then scans the pair counts of its sorted slice (or just carries the records along).
// Every kernel here is latency-bound (8 MB of keys): the design minimises dependent memory round trips and
// launches -- preprocess leaves per-workgroup depth extrema, so the whole stage is four launches, and the
// totals the host needs reach it through pinned memory while the last two kernels run.
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_internal.h"

namespace gsr {

#define DO_SORT_THREADS 1024
#define DO_ITEMS (GSR_DO_CAP / DO_SORT_THREADS)   // 8 register-resident items per thread
#define DO_RANK_MAX 96                            // largest sub-bucket handled by rank-by-counting
#define DO_SCAN_PER (GSR_DO_MAXB / 1024)          // buckets per thread of the one-workgroup bucket scan
#define DO_CNT_THREADS 1024                       // counting / scatter workgroups: one uint4 (4 Gaussians) per thread per step

DepthOrderPlan depth_order_plan(int P, int log_map) {
    DepthOrderPlan p;
    const long n = P > 0 ? P : 1;
    p.nb = 64;
    while (p.nb < GSR_DO_MAXB / 4 && (long)p.nb * 2048 < n) p.nb *= 2;      // more buckets cost more global atomics
    if (log_map) p.nb = p.nb * 4 < GSR_DO_MAXB ? p.nb * 4 : GSR_DO_MAXB;   // outliers leave most buckets empty: more of them
    p.chunk = 4 * DO_CNT_THREADS;
    while ((n + p.chunk - 1) / p.chunk > GSR_DO_MAXBLK) p.chunk *= 2;
    p.nblk = (int)((n + p.chunk - 1) / p.chunk);
    p.npre = (int)((n + 255) / 256);
    return p;
}

// The bucket map: a monotone non-decreasing function of the key (depth bits of a positive float) onto [0, nfine).
// Linear in DEPTH by default; linear in the depth BITS (logarithmic in depth, exact integer difference first) once a
// frame has overflowed a bucket -- a few far outliers stretch a linear map until the bulk of the scene shares one
// bucket.  Subtraction, int->float conversion, multiplication by a non-negative constant, truncation and the clamp
// all preserve order.
struct DoMap {
    uint32_t kmin, nfine;
    float dmin, scale;
    int log_map;
    __device__ __forceinline__ uint32_t fine(uint32_t key) const {
        const float x = log_map ? (float)(key - kmin) : (__uint_as_float(key) - dmin);
        const float v = x * scale;
        const uint32_t f = v > 0.f ? (uint32_t)v : 0u;
        return f < nfine ? f : nfine - 1u;
    }
};
__device__ __forceinline__ DoMap do_map(uint32_t kmin, uint32_t kmax, uint32_t nfine, int log_map) {
    DoMap m;
    m.kmin = kmin; m.nfine = nfine; m.log_map = log_map;
    m.dmin = __uint_as_float(kmin);
    const float span = log_map ? (float)(kmax - kmin) : (__uint_as_float(kmax) - m.dmin);
    m.scale = (kmax > kmin && span > 0.f) ? (float)nfine / span : 0.f;
    return m;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m));
    return v;
}

// depth range of the emitting Gaussians from the per-workgroup extrema preprocess left (every workgroup
// reduces the whole list: npre words x 2 out of L2, one round trip)
__device__ __forceinline__ void do_key_range(int npre, const uint32_t *__restrict__ blkmin, const uint32_t *__restrict__ blkmax,
                                             uint32_t *s_red /*[2 * 16]*/, uint32_t &kmin, uint32_t &kmax) {
    uint32_t mn = 0xffffffffu, mx = 0u;
    for (int j = threadIdx.x; j < npre; j += blockDim.x) { mn = min(mn, blkmin[j]); mx = max(mx, blkmax[j]); }
    mn = wave_min_u32(mn); mx = wave_max_u32(mx);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[w] = mn; s_red[16 + w] = mx; }
    __syncthreads();
    mn = 0xffffffffu; mx = 0u;
    for (int k = 0; k < nw; k++) { mn = min(mn, s_red[k]); mx = max(mx, s_red[16 + k]); }
    kmin = mn; kmax = mx;
}

// total of the per-workgroup entry counts preprocess left (one workgroup)
__global__ __launch_bounds__(1024) void do_entry_total_kernel(int npre, const uint32_t *__restrict__ blkent, uint32_t *__restrict__ hdr) {
    __shared__ uint32_t s_sum[16];
    uint32_t e = 0;
    for (int j = threadIdx.x; j < npre; j += 1024) e += blkent[j];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) e += (uint32_t)__shfl_xor((int)e, m);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < 16; k++) t += s_sum[k];
        hdr[DO_ETOT] = t;
    }
}
// rocPRIM depth order (all P Gaussians listed): rectangles in list order, empty for the ones that emit nothing
__global__ __launch_bounds__(256) void do_gather_rect_kernel(int P, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ tiles,
                                                             const uint4 *__restrict__ rect, uint4 *__restrict__ orect) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P) return;
    const uint32_t id = perm[r];
    orect[r] = tiles[id] > 0u ? rect[id] : make_uint4(0u, 0u, 0u, 0u);
}
hipError_t launch_entry_total(const GeomView &g, int P, hipStream_t s) {
    const DepthOrderPlan pl = depth_order_plan(P, 0);
    hipLaunchKernelGGL(do_entry_total_kernel, dim3(1), dim3(1024), 0, s, pl.npre, g.dord.blkent, g.dord.hdr);
    hipLaunchKernelGGL(do_gather_rect_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.perm, g.tiles, g.rect, g.orect);
    return hipGetLastError();
}

// Level-1 histogram (count, pair-count sum per bucket): LDS per workgroup, then one global atomic per touched
// bucket.  (A "last workgroup scans the totals" tail needs an agent-scope fence in every workgroup -- an L2
// write-back per workgroup on this multi-XCD part, 55 us measured -- so the scan is its own one-workgroup launch.)
__global__ __launch_bounds__(DO_CNT_THREADS) void do_hist_kernel(int P, int chunk, int nb, int npre, int log_map, const uint32_t *__restrict__ depth,
                                                                 const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ blkmin,
                                                                 const uint32_t *__restrict__ blkmax, const uint32_t *__restrict__ blkent,
                                                                 uint32_t *__restrict__ hdr, unsigned long long *__restrict__ gpair) {
    extern __shared__ uint32_t sm[];
    __shared__ uint32_t s_red[32];
    __shared__ uint32_t s_ent[16];
    if (blockIdx.x == 0) {                                             // workgroup 0 also totals the super-tile entries
        uint32_t e = 0;
        for (int j = threadIdx.x; j < npre; j += DO_CNT_THREADS) e += blkent[j];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) e += (uint32_t)__shfl_xor((int)e, m);
        if ((threadIdx.x & 63) == 0) s_ent[threadIdx.x >> 6] = e;
    }
    uint32_t *h = sm, *ts = sm + nb;
    for (int b = threadIdx.x; b < 2 * nb; b += DO_CNT_THREADS) sm[b] = 0u;
    uint32_t kmin, kmax;
    do_key_range(npre, blkmin, blkmax, s_red, kmin, kmax);          // contains a barrier: sm is zeroed for everyone after it
    const DoMap map = do_map(kmin, kmax, (uint32_t)nb * GSR_DO_NSUB, log_map);
    if (blockIdx.x == 0 && threadIdx.x == 0) {                         // s_ent is complete: do_key_range has a barrier
        uint32_t e = 0;
        for (int k = 0; k < DO_CNT_THREADS / 64; k++) e += s_ent[k];
        hdr[DO_KMIN] = kmin; hdr[DO_KMAX] = kmax; hdr[DO_ETOT] = e;
    }
    const int i0 = blockIdx.x * chunk, i1 = min(P, i0 + chunk);
    for (int i = i0 + 4 * (int)threadIdx.x; i < i1; i += 4 * DO_CNT_THREADS) {
        const uint4 t4 = *reinterpret_cast<const uint4 *>(tiles + i);   // padded allocations: the tail read stays inside
        const uint4 d4 = *reinterpret_cast<const uint4 *>(depth + i);
        const uint32_t tt[4] = {t4.x, t4.y, t4.z, t4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (i + q < i1 && tt[q] > 0u) {
                const uint32_t b = map.fine(dd[q]) / GSR_DO_NSUB;
                atomicAdd(&h[b], 1u);
                atomicAdd(&ts[b], tt[q]);
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += DO_CNT_THREADS) {
        const uint32_t c = h[b];
        if (c) atomicAdd(&gpair[b], (unsigned long long)c | ((unsigned long long)ts[b] << 32));   // count | pair-count sum: one request
    }
}

// one workgroup: bucket starts, pair-count bases, Pv, N, overflow flag; the totals also go straight to the
// caller's pinned host words (host_out may be NULL), sequence number last
__global__ __launch_bounds__(DO_CNT_THREADS) void do_bucket_scan_kernel(int nb, const unsigned long long *__restrict__ gpair,
                                                                        uint32_t *__restrict__ bstart,
                                                                        uint32_t *__restrict__ tbase, uint32_t *__restrict__ hdr,
                                                                        uint32_t *host_out, uint32_t seq) {
    __shared__ uint32_t wtot[2][16];
    __shared__ uint32_t s_over;
    if (threadIdx.x == 0) s_over = 0u;
    __syncthreads();
    // thread t owns the DO_SCAN_PER consecutive buckets from t * DO_SCAN_PER (nb <= GSR_DO_MAXB = DO_SCAN_PER * DO_CNT_THREADS)
    const int b0 = DO_SCAN_PER * (int)threadIdx.x;
    uint32_t c[DO_SCAN_PER], t[DO_SCAN_PER];
    uint32_t sumc = 0, sumt = 0;
    bool over = false;
#pragma unroll
    for (int q = 0; q < DO_SCAN_PER; q++) {
        c[q] = 0u; t[q] = 0u;
        if (b0 + q < nb) { const unsigned long long pr = gpair[b0 + q]; c[q] = (uint32_t)pr; t[q] = (uint32_t)(pr >> 32); }
        sumc += c[q]; sumt += t[q];
        over = over || c[q] > GSR_DO_CAP;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t ic = wave_incl_scan_u32(sumc, lane), it = wave_incl_scan_u32(sumt, lane);
    if (lane == 63) { wtot[0][w] = ic; wtot[1][w] = it; }
    if (over) atomicOr(&s_over, 1u);
    __syncthreads();
    uint32_t ec = ic - sumc, et = it - sumt;
    for (int k = 0; k < w; k++) { ec += wtot[0][k]; et += wtot[1][k]; }
    {
        uint32_t rc = ec, rt = et;
#pragma unroll
        for (int q = 0; q < DO_SCAN_PER; q++) {
            if (b0 + q < nb) { bstart[b0 + q] = rc; tbase[b0 + q] = rt; }
            rc += c[q]; rt += t[q];
        }
    }
    if (threadIdx.x == DO_CNT_THREADS - 1) {                       // owns nothing or the last buckets: ec + sumc is the total
        const uint32_t pv = ec + sumc, ntot = et + sumt, over = s_over;
        bstart[nb] = pv; tbase[nb] = ntot;
        hdr[DO_PV] = pv; hdr[DO_NTOT] = ntot; hdr[DO_OVERFLOW] = over;
        if (host_out) {
            host_out[0] = over; host_out[1] = pv; host_out[2] = ntot; host_out[3] = hdr[DO_ETOT];
            __threadfence_system();
            __hip_atomic_store(&host_out[4], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// place every emitting Gaussian in its level-1 bucket: the workgroup reserves one run per touched bucket
// (returning global atomic), arrival order inside the bucket is arbitrary
__global__ __launch_bounds__(DO_CNT_THREADS) void do_scatter_kernel(int P, int chunk, int nb, int log_map, const uint32_t *__restrict__ depth,
                                                                    const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ hdr,
                                                                    const uint32_t *__restrict__ bstart, uint32_t *__restrict__ gcur,
                                                                    uint64_t *__restrict__ comp) {
    extern __shared__ uint32_t h[];                                // counts, then run bases
    if (hdr[DO_OVERFLOW]) return;                                  // grid-uniform: the host takes the rocPRIM path
    for (int b = threadIdx.x; b < nb; b += DO_CNT_THREADS) h[b] = 0u;
    __syncthreads();
    const DoMap map = do_map(hdr[DO_KMIN], hdr[DO_KMAX], (uint32_t)nb * GSR_DO_NSUB, log_map);
    const int i0 = blockIdx.x * chunk, i1 = min(P, i0 + chunk);
    const int steps = chunk / (4 * DO_CNT_THREADS);                // 1 unless P > 2 M
    for (int st = 0; st < steps; st++) {
        // one pass per step keeps the per-thread state at 4 items; runs reserved by different steps are independent
        const int i = i0 + st * 4 * DO_CNT_THREADS + 4 * (int)threadIdx.x;
        uint32_t key[4] = {0u, 0u, 0u, 0u}, br[4] = {~0u, ~0u, ~0u, ~0u};
        if (i < i1) {
            const uint4 t4 = *reinterpret_cast<const uint4 *>(tiles + i);
            const uint4 d4 = *reinterpret_cast<const uint4 *>(depth + i);
            const uint32_t tt[4] = {t4.x, t4.y, t4.z, t4.w};
            key[0] = d4.x; key[1] = d4.y; key[2] = d4.z; key[3] = d4.w;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (i + q < i1 && tt[q] > 0u) {
                    const uint32_t b = map.fine(key[q]) / GSR_DO_NSUB;
                    br[q] = b | (atomicAdd(&h[b], 1u) << 13);      // bucket (< 8192) | arrival rank in this step (<= 4096)
                }
            }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < nb; b += DO_CNT_THREADS) {
            const uint32_t c = h[b];
            h[b] = c ? bstart[b] + atomicAdd(&gcur[b], c) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (br[q] != ~0u) comp[h[br[q] & 0x1fffu] + (br[q] >> 13)] = ((uint64_t)key[q] << 32) | (uint32_t)(i + q);
        __syncthreads();
        if (st + 1 < steps) {
            for (int b = threadIdx.x; b < nb; b += DO_CNT_THREADS) h[b] = 0u;
            __syncthreads();
        }
    }
}

// one workgroup per level-1 bucket: order its keys in LDS, then scan the pair counts of the ordered slice
template <bool NEED_OFFSETS>
__global__ __launch_bounds__(DO_SORT_THREADS) void do_local_sort_kernel(int nb, int log_map, const uint32_t *__restrict__ hdr,
                                                                        const uint32_t *__restrict__ bstart,
                                                                        const uint32_t *__restrict__ tbase,
                                                                        const uint64_t *__restrict__ comp,
                                                                        const uint32_t *__restrict__ tiles,
                                                                        const uint4 *__restrict__ rect, uint32_t *__restrict__ perm,
                                                                        uint32_t *__restrict__ offsets, uint4 *__restrict__ orect) {
    extern __shared__ uint64_t buf[];                              // [GSR_DO_CAP]
    __shared__ uint32_t start[GSR_DO_NSUB + 1], cur[GSR_DO_NSUB];
    __shared__ uint32_t wsum[DO_SORT_THREADS / 64];
    __shared__ uint32_t s_max;
    if (hdr[DO_OVERFLOW]) return;                                  // grid-uniform
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t s0 = bstart[b];
    const int n = (int)(bstart[b + 1] - s0);
    if (n == 0) return;                                            // workgroup-uniform
    const DoMap map = do_map(hdr[DO_KMIN], hdr[DO_KMAX], (uint32_t)nb * GSR_DO_NSUB, log_map);
    if (tid <= GSR_DO_NSUB) start[tid] = 0u;
    if (tid == 0) s_max = 0u;
    __syncthreads();
    // ---- sub-bucket histogram, then placement with a second pass over the keys (L2-resident): nothing is held in
    //      registers across the barriers, which keeps two workgroups per CU resident ----
#pragma unroll
    for (int q = 0; q < DO_ITEMS; q++) {
        const int j = tid + q * DO_SORT_THREADS;
        if (j < n) {
            const uint32_t key = (uint32_t)(comp[s0 + j] >> 32);
            atomicAdd(&start[map.fine(key) & (GSR_DO_NSUB - 1)], 1u);
        }
    }
    __syncthreads();
    {   // exclusive scan of the sub-bucket counts (thread t < 512 owns sub-bucket t) + their maximum
        const uint32_t v = tid < GSR_DO_NSUB ? start[tid] : 0u;
        const uint32_t mx = wave_max_u32(v);
        const uint32_t incl = wave_incl_scan_u32(v, lane);
        if (lane == 63) { wsum[w] = incl; atomicMax(&s_max, mx); }
        __syncthreads();
        uint32_t ex = incl - v;
        for (int k = 0; k < w; k++) ex += wsum[k];
        if (tid < GSR_DO_NSUB) { start[tid] = ex; cur[tid] = ex; }
        if (tid == GSR_DO_NSUB) start[GSR_DO_NSUB] = (uint32_t)n;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DO_ITEMS; q++) {
        const int j = tid + q * DO_SORT_THREADS;
        if (j < n) {
            const uint64_t c = comp[s0 + j];
            const uint32_t sub = map.fine((uint32_t)(c >> 32)) & (GSR_DO_NSUB - 1);
            buf[atomicAdd(&cur[sub], 1u)] = c;                     // arrival order inside the sub-bucket is irrelevant
        }
    }
    __syncthreads();
    uint32_t *sid = reinterpret_cast<uint32_t *>(buf);             // sorted ids, aliasing buf once it has been consumed
    if (s_max <= DO_RANK_MAX) {                                    // workgroup-uniform
        // the items are grouped by sub-bucket now: thread takes the item at position j, counts the smaller
        // keys of its sub-bucket
        uint32_t rk[DO_ITEMS], id[DO_ITEMS];
#pragma unroll
        for (int q = 0; q < DO_ITEMS; q++) {
            const int j = tid + q * DO_SORT_THREADS;
            rk[q] = 0u; id[q] = 0u;
            if (j < n) {
                const uint64_t me = buf[j];
                const uint32_t sub = map.fine((uint32_t)(me >> 32)) & (GSR_DO_NSUB - 1);
                const uint32_t a0 = start[sub], a1 = start[sub + 1];
                uint32_t r = a0;
                for (uint32_t k = a0; k < a1; k++) r += buf[k] < me ? 1u : 0u;
                rk[q] = r; id[q] = (uint32_t)me;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DO_ITEMS; q++) {
            const int j = tid + q * DO_SORT_THREADS;
            if (j < n) sid[rk[q]] = id[q];
        }
    } else {                                                       // degenerate depth distribution: bitonic network
        int m = 64;
        while (m < n) m <<= 1;
        for (int j = n + tid; j < m; j += DO_SORT_THREADS) buf[j] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (m >> 1); t += DO_SORT_THREADS) {
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int hi = lo + j;
                    const uint64_t x = buf[lo], y = buf[hi];
                    const bool up = (lo & k) == 0;
                    if ((x > y) == up) { buf[lo] = y; buf[hi] = x; }
                }
                __syncthreads();
            }
        }
        uint32_t id[DO_ITEMS];
#pragma unroll
        for (int q = 0; q < DO_ITEMS; q++) {
            const int j = tid + q * DO_SORT_THREADS;
            id[q] = j < n ? (uint32_t)buf[j] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DO_ITEMS; q++) {
            const int j = tid + q * DO_SORT_THREADS;
            if (j < n) sid[j] = id[q];
        }
    }
    __syncthreads();
    if (!NEED_OFFSETS) {                                           // tile_lists.hip only wants the ordered records
        for (int j = tid; j < n; j += DO_SORT_THREADS) {
            const uint32_t id = sid[j];
            perm[s0 + j] = id;
            orect[s0 + j] = rect[id];
        }
        return;
    }
    // ---- scan of the pair counts in sorted order: thread tid owns positions [tid*k, tid*k + k) ----
    const int k = (n + DO_SORT_THREADS - 1) / DO_SORT_THREADS;     // <= DO_ITEMS
    uint32_t ids[DO_ITEMS], tt[DO_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < DO_ITEMS; q++) {
        const int j = tid * k + q;
        ids[q] = 0u; tt[q] = 0u;
        if (q < k && j < n) { ids[q] = sid[j]; tt[q] = tiles[ids[q]]; }
    }
#pragma unroll
    for (int q = 0; q < DO_ITEMS; q++) sum += tt[q];
    const uint32_t incl = wave_incl_scan_u32(sum, lane);
    __syncthreads();                                               // wsum is reused
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = tbase[b] + incl - sum;
    for (int kk = 0; kk < w; kk++) run += wsum[kk];
#pragma unroll
    for (int q = 0; q < DO_ITEMS; q++) {
        const int j = tid * k + q;
        if (q < k && j < n) {
            run += tt[q];
            perm[s0 + j] = ids[q];
            offsets[s0 + j] = run;
        }
    }
}

hipError_t launch_depth_order_count(const GeomView &g, int P, int log_map, uint32_t *host_out, uint32_t seq, hipStream_t s) {
    const DepthOrderPlan pl = depth_order_plan(P, log_map);
    const DepthOrderView &d = g.dord;
    static std::atomic<uint64_t> attr_set{0};   // one bit per device (the attribute is per device and idempotent)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t dev_bit = 1ull << (dev & 63);
    if (!(attr_set.load() & dev_bit) && 2 * pl.nb * sizeof(uint32_t) > 48 * 1024) {
        // only the largest bucket counts need it (2 x 8192 counters = 64 KB next to a few static words); raising the
        // limit when it is not needed costs the kernel ~4 us (measured), so it is set on first use
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(do_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * GSR_DO_MAXB * (int)sizeof(uint32_t));
        if (e != hipSuccess) return e;
        attr_set.fetch_or(dev_bit);
    }
    hipLaunchKernelGGL(do_hist_kernel, dim3(pl.nblk), dim3(DO_CNT_THREADS), 2 * pl.nb * sizeof(uint32_t), s, P, pl.chunk, pl.nb, pl.npre, log_map,
                       reinterpret_cast<const uint32_t *>(g.depth), g.tiles, d.blkmin, d.blkmax, d.blkent, d.hdr, d.gpair);
    hipLaunchKernelGGL(do_bucket_scan_kernel, dim3(1), dim3(DO_CNT_THREADS), 0, s, pl.nb, d.gpair, d.bstart, d.tbase, d.hdr, host_out, seq);
    return hipGetLastError();
}

hipError_t launch_depth_order_place(const GeomView &g, int P, int log_map, int need_offsets, hipStream_t s) {
    const DepthOrderPlan pl = depth_order_plan(P, log_map);
    const DepthOrderView &d = g.dord;
    static std::atomic<uint64_t> attr_set{0};   // one bit per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t dev_bit = 1ull << (dev & 63);
    if (!(attr_set.load() & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(do_local_sort_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           GSR_DO_CAP * (int)sizeof(uint64_t));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(do_local_sort_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    GSR_DO_CAP * (int)sizeof(uint64_t));
        if (e != hipSuccess) return e;
        attr_set.fetch_or(dev_bit);
    }
    hipLaunchKernelGGL(do_scatter_kernel, dim3(pl.nblk), dim3(DO_CNT_THREADS), pl.nb * sizeof(uint32_t), s, P, pl.chunk, pl.nb, log_map,
                       reinterpret_cast<const uint32_t *>(g.depth), g.tiles, d.hdr, d.bstart, d.gcur, d.comp);
    if (need_offsets)
        hipLaunchKernelGGL(do_local_sort_kernel<true>, dim3(pl.nb), dim3(DO_SORT_THREADS), GSR_DO_CAP * sizeof(uint64_t), s, pl.nb, log_map, d.hdr, d.bstart,
                           d.tbase, d.comp, g.tiles, g.rect, g.perm, g.offsets, g.orect);
    else
        hipLaunchKernelGGL(do_local_sort_kernel<false>, dim3(pl.nb), dim3(DO_SORT_THREADS), GSR_DO_CAP * sizeof(uint64_t), s, pl.nb, log_map, d.hdr, d.bstart,
                           d.tbase, d.comp, g.tiles, g.rect, g.perm, g.offsets, g.orect);
    return hipGetLastError();
}

}  // namespace gsr
