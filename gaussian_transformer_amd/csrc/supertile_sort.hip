// supertile_sort.hip -- per-tile, depth-ordered Gaussian lists (S7 + S8) in four launches, with no global sort at all.
//
// Upstream sorts N (tile, depth) keyed pairs; round 1 of this library ordered the P Gaussians by depth first (four
// launches) and then multi-split them into per-tile lists (six more).  But depth order is only ever needed INSIDE a
// tile, so the order of operations can be turned around:
//   1. ss_count    bin the emitting Gaussians by SUPER-TILE (4 x 4 tiles = 64 x 64 pixels): entries per bin, counted in LDS
//                  per workgroup of 4096 Gaussians, one global atomic per (workgroup, touched bin);
//   2. ss_scan     one workgroup: bin starts, totals (pairs N, entries E, largest bin) -> pinned host words;
//   3. ss_scatter  every (Gaussian, super-tile) ENTRY = (depth bits, id, 16-bit tile mask) goes to its bin, in arbitrary
//                  order (workgroup-aggregated reservations); preprocess leaves, per Gaussian, the masks of the 2 x 2
//                  super-tiles under its rectangle (GeomView::ss_rec), larger rectangles re-evaluate their row spans;
//   4. ss_sort_expand   one workgroup per super-tile: its <= 7168 entries are ordered by (depth bits, id) in LDS
//                  (sub-buckets by log-depth + rank-by-counting inside a sub-bucket) -- the order depends on the keys
//                  only, not on the arrival order of step 3 -- and then
//                  wave t of the workgroup writes tile t's list: it sweeps the sorted entries 64 at a time, ballots
//                  the tile's mask bit and appends the selected ids as contiguous runs.  ranges[] fall out of it.
// Same per-tile lists as a stable sort of the pairs on tile << 32 | depth (ties: ascending id), laid out super-tile-major.
// A bin larger than the LDS capacity (or more super-tiles than the LDS histograms hold) makes the host take the
// round-1 path (depth_order.hip + tile_lists.hip) or rocPRIM for that frame.
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

#define SS_THREADS 1024
#ifndef SS_BIN_THREADS
#define SS_BIN_THREADS 1024          // workgroup size of the counting / scatter kernels
#endif
#define SS_NSUB 512               // depth sub-buckets of one bin
#define SS_NSPLIT 256             // sub-buckets under the splitter map (ss_sort_expand_kernel)
#define SS_RANK_MAX 96            // largest sub-bucket the depth-linear map may produce before the bit-linear map takes over
#define SS_RANK_MAX_LOG 1024       // ... and the bit-linear map before the splitter map does: depths that coincide by the thousand only (the
                                  // splitter map's binary searches cost more than ranking a sub-bucket of a few hundred: config 4, 107 vs 62 us)

SuperSortPlan super_sort_plan(int P, int W, int H) {
    SuperSortPlan p;
    const int gridx = (W + GSR_TILE - 1) / GSR_TILE, gridy = (H + GSR_TILE - 1) / GSR_TILE;
    p.SX = (gridx + GSR_SS_TILES - 1) / GSR_SS_TILES;
    p.SY = (gridy + GSR_SS_TILES - 1) / GSR_SS_TILES;
    p.S = p.SX * p.SY;
    const long n = P > 0 ? P : 1;
    p.chunk = 4 * SS_BIN_THREADS;
    while ((n + p.chunk - 1) / p.chunk > 1024) p.chunk *= 2;
    p.nblk = (int)((n + p.chunk - 1) / p.chunk);
    p.ecap = (int64_t)GSR_SS_ENT_PER_G * n;
    return p;
}

__device__ __forceinline__ uint32_t ss_wave_sum(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}
__device__ __forceinline__ uint32_t ss_wave_max(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m));
    return v;
}
__device__ __forceinline__ uint32_t ss_wave_min(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m));
    return v;
}
__device__ __forceinline__ uint32_t ss_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

struct SsRect { int x0, x1, y0, y1, sx0, sx1, sy0, sy1; };
__device__ __forceinline__ SsRect ss_rect(uint4 rc, int SX, int SY) {
    SsRect r;
    r.x0 = (int)(rc.x & 0xffffu); r.x1 = (int)(rc.x >> 16); r.y0 = (int)(rc.y & 0xffffu); r.y1 = (int)(rc.y >> 16);
    r.sx0 = r.x0 / GSR_SS_TILES; r.sx1 = min((r.x1 + GSR_SS_TILES - 1) / GSR_SS_TILES, SX);
    r.sy0 = r.y0 / GSR_SS_TILES; r.sy1 = min((r.y1 + GSR_SS_TILES - 1) / GSR_SS_TILES, SY);
    return r;
}

struct SsBinArgs {
    int P, chunk, SX, SY, W, H, exact_cull;
    const uint4 *ss_rec;             // per Gaussian: what preprocess prepared for this path (GeomView::ss_rec)
    const uint4 *rect;               // large rectangles only
    const uint32_t *depth_bits;
    float *rec;
    const uint32_t *tiles;           // pairs per Gaussian (large rectangles: who gets replica accumulator rows)
    uint32_t *hot;                   // [P] replica codes, written by the counting kernel for the largest splats (only theirs)
    uint8_t *clamped;                // [P] bit 7 := "has a replica code"
    uint32_t hot_cap;                // replica rows available (acc_extra_rows)
    uint32_t *hdr;
    uint32_t *bin_cnt;               // [S] global entry counts (count kernel: atomics)
    uint32_t *wg_cnt;                // [nblk][S] entries of every counting workgroup per bin (count kernel writes, scatter reads)
    const uint32_t *bin_start;
    uint32_t *bin_cur, *bin_pairs;
    uint4 *entries;
};

// Entries of the workgroup's chunk of Gaussians: f(bin, mask, gaussian, depth bits) for every (super-tile, non-empty mask).
// preprocess left one 16-byte record per Gaussian (GeomView::ss_rec):
//   kind 1  rectangle inside 2 x 2 super-tiles: the four masks are in the record, the lane emits them;
//   kind 3  up to 8 x 15 tiles: the row spans are in the record.  Assembling up to 15 masks from them in place would leave the
//           other lanes of the wave waiting, so the record goes to a list in LDS and the list is worked off one item per lane;
//   kind 2  larger (about 1 % of the Gaussians, up to a thousand tiles each): deferred to a second list and taken one per WAVE:
//           the lanes first evaluate the ellipse-vs-tile-row span of 64 tile rows in parallel (exactly as preprocess counted
//           them: same function, same rounding, -ffp-contract=off), then switch to one lane per super-tile of the band and
//           assemble the masks from those spans.  Their rect / rec loads are issued before the kind-3 list is worked off.
struct SsLds {
    uint4 *midrec;        // [GSR_SS_MIDCAP] kind-3 records
    uint32_t *midid;      // [GSR_SS_MIDCAP] their Gaussian ids
    uint32_t *big;        // [chunk] ids of kind-2 Gaussians
    uint32_t *spans;      // [waves][64]
    uint32_t *nbig, *nmid;   // zeroed
};
#define SS_LDS_LIST_WORDS(chunk) (5 * GSR_SS_MIDCAP + (chunk) + (SS_BIN_THREADS / 64) * 64)
__device__ __forceinline__ SsLds ss_carve_lds(uint32_t *sm, int chunk, uint32_t *nbig, uint32_t *nmid) {
    SsLds l;
    l.midrec = reinterpret_cast<uint4 *>(sm);
    l.midid = sm + 4 * GSR_SS_MIDCAP;
    l.big = l.midid + GSR_SS_MIDCAP;
    l.spans = l.big + chunk;
    l.nbig = nbig; l.nmid = nmid;
    return l;
}

template <class F>
__device__ __forceinline__ void ss_mid_item(const uint4 sr, uint32_t id, int SX, F f) {
    const uint32_t y = sr.y;
    const int bin0 = (int)(y & 0x3ffffu), lx0 = (int)((y >> 18) & 3u), ly0 = (int)((y >> 20) & 3u);
    const int rows = (int)((y >> 22) & 7u) + 1, cols = (int)((y >> 25) & 15u) + 1;
    const uint64_t sp = (uint64_t)sr.z | ((uint64_t)sr.w << 32);
    const int ndx = (lx0 + cols + GSR_SS_TILES - 1) / GSR_SS_TILES, ndy = (ly0 + rows + GSR_SS_TILES - 1) / GSR_SS_TILES;
    for (int dy = 0; dy < ndy; dy++)
        for (int dx = 0; dx < ndx; dx++) {
            uint32_t m = 0u;
            const int bx = dx * GSR_SS_TILES;                         // tile columns relative to the first super-tile
#pragma unroll
            for (int k = 0; k < GSR_SS_TILES; k++) {
                const int r = dy * GSR_SS_TILES + k - ly0;
                if (r < 0 || r >= rows) continue;
                const uint32_t bb = (uint32_t)(sp >> (8 * r)) & 0xffu;
                const int c0 = lx0 + (int)(bb & 15u), c1 = lx0 + (int)(bb >> 4);
                const int lo = max(c0, bx) - bx, hi = min(c1, bx + GSR_SS_TILES) - bx;
                if (hi > lo) m |= (((1u << hi) - 1u) & ~((1u << lo) - 1u)) << (4 * k);
            }
            if (m) f(bin0 + dy * SX + dx, m, id, sr.x);
        }
}

// HOT (the counting kernel only): the lane that fetches a large rectangle also decides whether the splat gets replica rows
template <bool HOT, class F>
__device__ __forceinline__ void ss_for_chunk_entries(const SsBinArgs &a, const SsLds &l, F f) {
    const int i0 = blockIdx.x * a.chunk, i1 = min(a.P, i0 + a.chunk);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = i0; base < i1; base += 4 * SS_BIN_THREADS) {
        // four Gaussians per thread, every load issued before the first use
        uint4 sr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + k * SS_BIN_THREADS + (int)threadIdx.x;
            sr[k] = i < i1 ? a.ss_rec[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t kind = sr[k].y >> 29;
            if (!kind) continue;
            const int i = base + k * SS_BIN_THREADS + (int)threadIdx.x;
            if (kind == 2u) {                                                               // at most `chunk` of them
                const uint32_t at = atomicAdd(l.nbig, 1u);
                if (GSR_IDX_OK(at, a.chunk + (SS_BIN_THREADS / 64) * 64, a.hdr + GSR_DBG_GEOM_WORD, GSR_BOUND_SS_BIG_LIST)) l.big[at] = (uint32_t)i;
                continue;
            }
            if (kind == 3u) {
                const uint32_t slot = atomicAdd(l.nmid, 1u);
                if (slot < GSR_SS_MIDCAP) { l.midrec[slot] = sr[k]; l.midid[slot] = (uint32_t)i; }
                else ss_mid_item(sr[k], (uint32_t)i, a.SX, f);                           // list full: in place
                continue;
            }
            const int bin0 = (int)(sr[k].y & 0x3ffffu);
            // the four masks preprocess assembled: super-tiles (dx, dy) of the 2 x 2 block from bin0 (an empty mask where the
            // block leaves the grid)
            if (sr[k].z & 0xffffu) f(bin0, sr[k].z & 0xffffu, (uint32_t)i, sr[k].x);
            if (sr[k].z >> 16) f(bin0 + 1, sr[k].z >> 16, (uint32_t)i, sr[k].x);
            if (sr[k].w & 0xffffu) f(bin0 + a.SX, sr[k].w & 0xffffu, (uint32_t)i, sr[k].x);
            if (sr[k].w >> 16) f(bin0 + a.SX + 1, sr[k].w >> 16, (uint32_t)i, sr[k].x);
        }
    }
    __syncthreads();
    const int nb = (int)*l.nbig, nm = min((int)*l.nmid, GSR_SS_MIDCAP);
    uint32_t *my_spans = l.spans + w * 64;
    // one wave per large rectangle; wave w takes items w, w + 16, ...  Their records are fetched 64 at a time, one item per
    // lane, and broadcast from that lane when the item's turn comes: one memory latency per 64 items instead of one each --
    // and the first batch is in flight while the medium rectangles are worked off
    constexpr int NW = SS_BIN_THREADS / 64;
    for (int k0 = w, first = 1; first || k0 < nb; k0 += NW * 64, first = 0) {
        const int kl = k0 + NW * lane;
        const bool have = kl < nb;
        const int iv = have ? (int)l.big[kl] : 0;
        uint4 rcv = make_uint4(0u, 0u, 0u, 0u);
        uint32_t dv = 0u;
        float4 r0v = make_float4(0.f, 0.f, 0.f, 0.f);
        float conCv = 0.f, tauv = 0.f;
        if (have) {
            rcv = a.rect[iv]; dv = a.depth_bits[iv];
            const float4 *rp = reinterpret_cast<const float4 *>(a.rec) + 3 * (size_t)iv;
            r0v = rp[0]; conCv = rp[1].x; tauv = rp[2].z;
            if (HOT) {
                const uint32_t t = a.tiles[iv];
                if (t >= GSR_HOT_MIN_TILES) {                            // K = 4 .. 16 replicas, 64 tiles each or more
                    const uint32_t lg = min(4u, 31u - (uint32_t)__clz((int)(t / 64u)));
                    const uint32_t base = atomicAdd(&a.hdr[SS_HDR_HOT], 1u << lg);
                    if (base + (1u << lg) <= a.hot_cap) {
                        const uint32_t code = (base << 4) | lg;
                        a.hot[iv] = code;
                        a.clamped[iv] |= 0x80u;                          // tells pergauss_bwd.hip to look at hot[] (single writer)
                        reinterpret_cast<uint32_t *>(a.rec)[(size_t)GSR_REC_FLOATS * iv + 11] = code;
                    }
                }
            }
        }
        if (first)                                                      // medium rectangles: one per lane, densely
            for (int k = (int)threadIdx.x; k < nm; k += SS_BIN_THREADS) ss_mid_item(l.midrec[k], l.midid[k], a.SX, f);
        const int cnt = k0 < nb ? min(64, (nb - k0 + NW - 1) / NW) : 0;    // wave-uniform
        for (int j = 0; j < cnt; j++) {
#define SS_BC(x) __builtin_amdgcn_readlane((int)(x), j)
#define SS_BCF(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j))
            const int i = SS_BC(iv);
            const SsRect q = ss_rect(make_uint4((uint32_t)SS_BC(rcv.x), (uint32_t)SS_BC(rcv.y), 0u, 0u), a.SX, a.SY);
            const uint32_t d = (uint32_t)SS_BC(dv);
            const float4 r0 = make_float4(SS_BCF(r0v.x), SS_BCF(r0v.y), SS_BCF(r0v.z), SS_BCF(r0v.w));
            const CullParams cp = make_cull(r0.z, r0.w, SS_BCF(conCv), SS_BCF(tauv));
#undef SS_BC
#undef SS_BCF
            const int nsx = q.sx1 - q.sx0;
            for (int sy0 = q.sy0; sy0 < q.sy1; sy0 += 16) {              // bands of 16 super rows = 64 tile rows
                const int ty = sy0 * GSR_SS_TILES + lane;
                int c0 = 0, c1 = 0;
                if (ty >= q.y0 && ty < q.y1) {
                    c0 = q.x0; c1 = q.x1;
                    if (a.exact_cull) tile_row_span(cp, r0.x, r0.y, r0.z, r0.w, ty, a.W, a.H, q.x0, q.x1, c0, c1);
                }
                __builtin_amdgcn_wave_barrier();
                my_spans[lane] = (uint32_t)c0 | ((uint32_t)c1 << 16);
                __builtin_amdgcn_wave_barrier();
                const int nsy = min(16, q.sy1 - sy0);
                for (int e = lane; e < nsy * nsx; e += 64) {
                    const int syl = e / nsx, sx = q.sx0 + (e - syl * nsx);
                    const int bx = sx * GSR_SS_TILES;
                    uint32_t m = 0u;
#pragma unroll
                    for (int r = 0; r < GSR_SS_TILES; r++) {
                        const uint32_t spn = my_spans[syl * GSR_SS_TILES + r];
                        const int lo = max((int)(spn & 0xffffu), bx) - bx, hi = min((int)(spn >> 16), bx + GSR_SS_TILES) - bx;
                        if (hi > lo) m |= (((1u << hi) - 1u) & ~((1u << lo) - 1u)) << (4 * r);
                    }
                    if (m) f((sy0 + syl) * a.SX + sx, m, (uint32_t)i, d);
                }
            }
        }
    }
}

// ---- 1: exact entry counts: per (workgroup, super-tile) for the scatter pass, per super-tile (global) for the scan ----
__global__ __launch_bounds__(SS_BIN_THREADS) void ss_count_kernel(SsBinArgs a) {
    extern __shared__ __align__(16) uint32_t sm[];         // lists (ss_carve_lds) | h[S]
    __shared__ uint32_t s_sum[2][SS_BIN_THREADS / 64];
    __shared__ uint32_t s_nbig, s_nmid;
    const int S = a.SX * a.SY;
    const SsLds l = ss_carve_lds(sm, a.chunk, &s_nbig, &s_nmid);
    uint32_t *h = sm + SS_LDS_LIST_WORDS(a.chunk);
    for (int b = threadIdx.x; b < S; b += SS_BIN_THREADS) h[b] = 0u;
    if (threadIdx.x == 0) { s_nbig = 0u; s_nmid = 0u; }
    __syncthreads();
    uint32_t pairs = 0, ents = 0;
    ss_for_chunk_entries<true>(a, l, [&](int bin, uint32_t m, uint32_t, uint32_t) {
        atomicAdd(&h[bin], 1u); pairs += (uint32_t)__popc(m); ents++;
    });
    pairs = ss_wave_sum(pairs); ents = ss_wave_sum(ents);
    if ((threadIdx.x & 63) == 0) { s_sum[0][threadIdx.x >> 6] = pairs; s_sum[1][threadIdx.x >> 6] = ents; }
    __syncthreads();
    uint32_t *row = a.wg_cnt + (size_t)blockIdx.x * S;
    for (int b = threadIdx.x; b < S; b += SS_BIN_THREADS) {
        const uint32_t c = h[b];
        row[b] = c;
        if (c) atomicAdd(&a.bin_cnt[b], c);
    }
    if (threadIdx.x == 0) {
        uint32_t p = 0, e = 0;
        for (int k = 0; k < SS_BIN_THREADS / 64; k++) { p += s_sum[0][k]; e += s_sum[1][k]; }
        if (p) atomicAdd(&a.hdr[SS_HDR_N], p);
        if (e) atomicAdd(&a.hdr[SS_HDR_E], e);
    }
}

// ---- 2: one workgroup: exclusive scan of the bin counts, the totals, the overflow verdict; totals to the pinned words ----
__global__ __launch_bounds__(SS_THREADS) void ss_scan_kernel(int S, uint32_t ecap, const uint32_t *__restrict__ bin_cnt,
                                                             uint32_t *__restrict__ bin_start, uint32_t *__restrict__ hdr,
                                                             uint32_t *host_out, uint32_t seq) {
    __shared__ uint32_t wtot[SS_THREADS / 64], wmax[SS_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = (S + SS_THREADS - 1) / SS_THREADS;     // consecutive bins per thread (<= 8 for S <= 8192)
    const int b0 = per * (int)threadIdx.x;
    uint32_t c[GSR_SS_MAXS / SS_THREADS], sum = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < GSR_SS_MAXS / SS_THREADS; q++) {
        c[q] = (q < per && b0 + q < S) ? bin_cnt[b0 + q] : 0u;
        sum += c[q]; mx = max(mx, c[q]);
    }
    const uint32_t incl = ss_wave_incl_scan(sum, lane);
    mx = ss_wave_max(mx);
    if (lane == 63) { wtot[w] = incl; wmax[w] = mx; }
    __syncthreads();
    uint32_t ex = incl - sum;
    for (int k = 0; k < w; k++) ex += wtot[k];
    {
        uint32_t run = ex;
#pragma unroll
        for (int q = 0; q < GSR_SS_MAXS / SS_THREADS; q++) {
            if (q < per && b0 + q < S) bin_start[b0 + q] = run;
            run += c[q];
        }
    }
    if (threadIdx.x == 0) {
        uint32_t tot = 0, m = 0;
        for (int k = 0; k < SS_THREADS / 64; k++) { tot += wtot[k]; m = max(m, wmax[k]); }
        bin_start[S] = tot;
        const uint32_t over = (m > GSR_SS_CAP_BIG || tot > ecap) ? 1u : 0u;
        hdr[DO_OVERFLOW] = over; hdr[SS_HDR_MAXBIN] = m;
        if (host_out) {
            host_out[0] = over; host_out[1] = m; host_out[2] = hdr[SS_HDR_N]; host_out[3] = tot;
            __threadfence_system();
            __hip_atomic_store(&host_out[4], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- 3: entries to their bins (arbitrary order inside a bin; the sort of step 4 only looks at the keys).  The workgroup
//      knows its exact count per bin from step 1 (same chunk, same code), reserves one run per touched bin and fills it ----
__global__ __launch_bounds__(SS_BIN_THREADS) void ss_scatter_kernel(SsBinArgs a) {
    extern __shared__ __align__(16) uint32_t sm[];         // lists (ss_carve_lds) | run base[S] | rank[S] | prs[S]
    __shared__ uint32_t s_nbig, s_nmid;
    const int S = a.SX * a.SY;
    if (a.hdr[DO_OVERFLOW]) return;                        // grid-uniform: the host takes another path for this frame
    const SsLds l = ss_carve_lds(sm, a.chunk, &s_nbig, &s_nmid);
    uint32_t *basep = sm + SS_LDS_LIST_WORDS(a.chunk), *rank = basep + S, *prs = basep + 2 * S;
    const uint32_t *row = a.wg_cnt + (size_t)blockIdx.x * S;
    for (int b = threadIdx.x; b < S; b += SS_BIN_THREADS) {
        const uint32_t c = row[b];
        basep[b] = c ? a.bin_start[b] + atomicAdd(&a.bin_cur[b], c) : 0u;      // this workgroup's run inside the bin
        rank[b] = 0u; prs[b] = 0u;
    }
    if (threadIdx.x == 0) { s_nbig = 0u; s_nmid = 0u; }
    __syncthreads();
    ss_for_chunk_entries<false>(a, l, [&](int bin, uint32_t m, uint32_t id, uint32_t d) {
        const uint32_t at = basep[bin] + atomicAdd(&rank[bin], 1u);
        if (GSR_IDX_OK(at, (unsigned long long)GSR_SS_ENT_PER_G * a.P, a.hdr + GSR_DBG_GEOM_WORD, GSR_BOUND_SS_ENTRIES)) a.entries[at] = make_uint4(d, id, m, 0u);
        atomicAdd(&prs[bin], (uint32_t)__popc(m));
    });
    __syncthreads();
    for (int b = threadIdx.x; b < S; b += SS_BIN_THREADS)
        if (prs[b]) atomicAdd(&a.bin_pairs[b], prs[b]);
}

// ---- 4: one workgroup per super-tile: order the bin by (depth bits, id) in LDS, then one wave per tile writes its list ----
struct SsSortArgs {
    int S, SX, gridx, gridy;
    const uint32_t *hdr, *bin_start, *bin_cur, *bin_pairs;
    const uint4 *entries;
    uint32_t *point_list;
    uint2 *ranges;
    int n_lo, n_hi;          // this launch takes the bins with n_lo <= entries <= n_hi (a frame with a few crowded super-tiles runs the
                             // small-buffer instantiation over the rest and the big one over those only)
    uint32_t *unsplit;       // [S] crowded bins the split kernel could not cut into parts (cleared by the first launch, set by the split
                             // kernel, read by the big-buffer launch, which then takes only those); lives in the counting scratch (wg_cnt)
    int flagged_only;        // big-buffer launch: 1 = only bins with unsplit[s] != 0
};
template <int CAP>
__global__ __launch_bounds__(SS_THREADS, CAP <= GSR_SS_CAP ? 8 : 4) void ss_sort_expand_kernel(SsSortArgs a) {
    constexpr int ITEMS = CAP / SS_THREADS;
    extern __shared__ uint64_t buf[];                              // [CAP] keys; later: sorted ids (u32) | sorted masks (u16)
    uint16_t *mbuf = reinterpret_cast<uint16_t *>(buf + CAP);      // [CAP] masks travelling with the keys
    __shared__ uint32_t start[SS_NSUB + 1], cur[SS_NSUB];
    __shared__ uint32_t wred[3][SS_THREADS / 64];
    __shared__ uint32_t s_kmin, s_kmax, s_before;
    __shared__ uint32_t tile_start[16];
    __shared__ uint64_t split[SS_NSPLIT];                          // third map: splitters drawn from the bin itself (split[0] unused); 256 of them:
                                                                   // with 512 the static LDS would push two workgroups past a CU's 160 KB
    if (a.hdr[DO_OVERFLOW]) return;                                // grid-uniform
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t e0 = a.bin_start[s];
    int n = (int)a.bin_cur[s];                                     // entries actually written (empty masks were dropped)
    if (a.n_lo == 0 && tid == 0 && a.unsplit) a.unsplit[s] = 0u;   // first launch of the frame: clear the flag of every bin
    if (n < a.n_lo || n > a.n_hi) return;                          // workgroup-uniform: another launch's bin
    if (a.flagged_only && a.unsplit[s] == 0u) return;              // the split kernel took it

    if (!GSR_IDX_OK(n, CAP + 1, a.hdr + GSR_DBG_GEOM_WORD, GSR_BOUND_SS_BIN_SIZE)) n = CAP;      // debug build: a bin beyond the LDS buffer
    // pairs of the super-tiles before this one = where its region of point_list starts
    {
        uint32_t before = 0;
        for (int k = tid; k < s; k += SS_THREADS) before += a.bin_pairs[k];
        before = ss_wave_sum(before);
        if (lane == 0) wred[0][w] = before;
    }
    if (tid <= SS_NSUB) start[tid] = 0u;
    // ---- keys into registers, their depth range ----
    uint64_t key[ITEMS];
    uint32_t msk[ITEMS];
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int j = tid + q * SS_THREADS;
        key[q] = ~0ull; msk[q] = 0u;
        if (j < n) {
            const uint4 e = a.entries[e0 + j];
            key[q] = ((uint64_t)e.x << 32) | e.y; msk[q] = e.z;
            kmin = min(kmin, e.x); kmax = max(kmax, e.x);
        }
    }
    kmin = ss_wave_min(kmin); kmax = ss_wave_max(kmax);
    if (lane == 0) { wred[1][w] = kmin; wred[2][w] = kmax; }
    __syncthreads();
    if (tid == 0) {
        uint32_t b = 0, mn = 0xffffffffu, mx = 0u;
        for (int k = 0; k < SS_THREADS / 64; k++) { b += wred[0][k]; mn = min(mn, wred[1][k]); mx = max(mx, wred[2][k]); }
        s_before = b; s_kmin = mn; s_kmax = mx;
    }
    __syncthreads();
    // Monotone map of the depth bits onto the sub-buckets, over the bin's own range.  First choice: linear in DEPTH, which
    // spreads a typical bin evenly (rank-by-counting then compares a key with ~10 others).  A few splats right in front of
    // the camera stretch such a map until the bulk of the bin shares a handful of sub-buckets; when the largest sub-bucket
    // exceeds SS_RANK_MAX the histogram is redone with a map linear in the depth BITS (logarithmic in depth, exact integer
    // arithmetic), which no outlier can stretch that way.  Either map is monotone, so the order is the same.
    const uint32_t kmin0 = s_kmin;
    const uint64_t kspan = (uint64_t)(s_kmax >= s_kmin ? s_kmax - s_kmin : 0u) + 1ull;
    const float dmin = __uint_as_float(s_kmin);
    const float fspan = __uint_as_float(s_kmax) - dmin;
    const float fscale = (s_kmax > s_kmin && fspan > 0.f) ? (float)SS_NSUB / fspan : 0.f;
    bool log_map = false, split_map = false;                           // workgroup-uniform
    auto sub_of = [&](uint64_t k) -> uint32_t {
        if (split_map) {                                               // number of splitters <= k: binary search over split[1..SS_NSPLIT)
            uint32_t lo = 0u, hi = SS_NSPLIT - 1u;                     // answer in [lo, hi]
#pragma unroll
            for (int it = 0; it < 8; it++) {                           // SS_NSPLIT = 256
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (split[mid] <= k) lo = mid; else hi = mid - 1u;
            }
            return lo;
        }
        const uint32_t kb = (uint32_t)(k >> 32);
        if (log_map) return (uint32_t)(((uint64_t)(kb - kmin0) * (uint64_t)SS_NSUB) / kspan);     // < SS_NSUB
        const float v = (__uint_as_float(kb) - dmin) * fscale;
        const uint32_t f = v > 0.f ? (uint32_t)v : 0u;
        return f < SS_NSUB ? f : SS_NSUB - 1u;
    };
    if (n > 0) {
        // ---- sub-bucket histogram (again if the map turns out lopsided: linear in depth, then linear in the depth bits, then splitters
        //      drawn from the bin's own keys), exclusive scan, placement ----
        for (int attempt = 0; attempt < 3; attempt++) {
#pragma unroll
            for (int q = 0; q < ITEMS; q++)
                if (tid + q * SS_THREADS < n) atomicAdd(&start[sub_of(key[q])], 1u);
            __syncthreads();
            if (attempt == 2) break;
            const uint32_t mx = ss_wave_max(tid < SS_NSUB ? start[tid] : 0u);
            if (lane == 0) wred[1][w] = mx;
            __syncthreads();
            uint32_t m = 0;
            for (int k = 0; k < SS_THREADS / 64; k++) m = max(m, wred[1][k]);
            if (m <= (attempt == 0 ? SS_RANK_MAX : SS_RANK_MAX_LOG)) break;      // workgroup-uniform
            if (attempt == 0) log_map = true;
            else {
                // Both analytic maps leave a sub-bucket of hundreds (a scene seen from inside: a few splats at the lens stretch the range and
                // the object sits in a sliver of it; config 4: 590 of a bin's 1781 entries in one sub-bucket, the ranking below is quadratic in
                // that).  Third map: the first min(n, 1024) entries -- arrival order is arbitrary, so they are a random sample -- are ranked
                // among themselves and every (ns / 256)-th becomes a splitter; keys are unique (depth bits | id), so the sub-buckets come
                // out even whatever the depths are, coinciding ones included.
                const int ns = min(n, SS_THREADS);
                uint64_t *samp = buf, *sorted = buf + SS_THREADS;      // buf is not in use yet (CAP >= 2 * SS_THREADS)
                samp[tid] = tid < ns ? key[0] : ~0ull;
                __syncthreads();
                if (tid < ns) {
                    const uint64_t me = samp[tid];
                    uint32_t r = 0u;
#pragma unroll 8
                    for (int u = 0; u < ns; u++) r += samp[u] < me ? 1u : 0u;
                    sorted[r] = me;
                }
                __syncthreads();
                if (tid >= 1 && tid < SS_NSPLIT) split[tid] = sorted[min(ns - 1, (tid * ns) / SS_NSPLIT)];
                split_map = true; log_map = false;
            }
            __syncthreads();
            if (tid <= SS_NSUB) start[tid] = 0u;
            __syncthreads();
        }
        {
            const uint32_t v = tid < SS_NSUB ? start[tid] : 0u;
            const uint32_t incl = ss_wave_incl_scan(v, lane);
            if (lane == 63) wred[0][w] = incl;
            __syncthreads();
            uint32_t ex = incl - v;
            for (int k = 0; k < w; k++) ex += wred[0][k];
            if (tid < SS_NSUB) { start[tid] = ex; cur[tid] = ex; }
            if (tid == SS_NSUB) start[SS_NSUB] = (uint32_t)n;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ITEMS; q++)
            if (tid + q * SS_THREADS < n) {
                const uint32_t pos = atomicAdd(&cur[sub_of(key[q])], 1u);      // arrival order inside the sub-bucket is irrelevant
                if (GSR_IDX_OK(pos, CAP, a.hdr + GSR_DBG_GEOM_WORD, GSR_BOUND_SS_LDS_POS)) { buf[pos] = key[q]; mbuf[pos] = (uint16_t)msk[q]; }
            }
        __syncthreads();
    }
    uint32_t *sid = reinterpret_cast<uint32_t *>(buf);                 // sorted ids, aliasing buf once it has been consumed
    uint16_t *smask = reinterpret_cast<uint16_t *>(sid + CAP);         // sorted masks behind them (6 * CAP <= 8 * CAP bytes)
    if (n > 0) {
        // The items are grouped by sub-bucket now: the thread that holds position j counts the smaller keys of j's
        // sub-bucket (keys are unique: depth bits | id).  Correct for any distribution; a bin whose depths all coincide
        // costs n comparisons per item (slow, never wrong) -- there is no power-of-two network to overflow the LDS.
        uint32_t rk[ITEMS], id[ITEMS], mm[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            const int j = tid + q * SS_THREADS;
            rk[q] = 0u; id[q] = 0u; mm[q] = 0u;
            if (j < n) {
                const uint64_t me = buf[j];
                const uint32_t sb = sub_of(me);
                const uint32_t a0 = start[sb], a1 = start[sb + 1];
                // (eight reads in flight: one LDS round trip per comparison made this loop the whole kernel on scenes whose depths
                //  crowd a few sub-buckets -- config 4: 170 of 216 us)
                uint32_t r = a0, k = a0;
                for (; k + 8 <= a1; k += 8) {
                    uint64_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = buf[k + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) r += v[u] < me ? 1u : 0u;
                }
                for (; k < a1; k++) r += buf[k] < me ? 1u : 0u;
                rk[q] = r; id[q] = (uint32_t)me; mm[q] = mbuf[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ITEMS; q++)
            if (tid + q * SS_THREADS < n) { sid[rk[q]] = id[q]; smask[rk[q]] = (uint16_t)mm[q]; }
    }
    __syncthreads();
    // ---- expansion: the sorted entries -> the 16 per-tile lists.  Every thread takes ITEMS consecutive sorted entries and needs,
    // for each of the 16 tiles, how many earlier entries carry that tile's bit: a 16-component prefix sum.  The components are
    // packed two to a word (16 bits each, n <= 14 336), so that one workgroup scan of 8 words does all tiles at once; a wave
    // sweeping all n entries for one tile (the first version) spent 26 of this kernel's 42 us on ballots.
    uint32_t m[ITEMS], idv[ITEMS];
    const int jb = tid * ITEMS;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int j = jb + q;
        m[q] = j < n ? (uint32_t)smask[j] : 0u;
        idv[q] = j < n ? sid[j] : 0u;
    }
    // this thread's entries per tile, two tiles to a word (word k: tile 2k | tile 2k + 1 << 16)
    auto local_counts = [&](uint32_t c[8]) {
        // bit k of a byte -> nibble k of a word; the per-thread counts (<= ITEMS <= 14 < 16) add up nibble-wise
        auto spread8 = [](uint32_t x) {
            x = (x | (x << 12)) & 0x000f000fu; x = (x | (x << 6)) & 0x03030303u; x = (x | (x << 3)) & 0x11111111u;
            return x;
        };
        uint32_t nlo = 0u, nhi = 0u;
#pragma unroll
        for (int q = 0; q < ITEMS; q++) { nlo += spread8(m[q] & 0xffu); nhi += spread8(m[q] >> 8); }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t src = k < 4 ? nlo : nhi;
            const int sh = 8 * (k & 3);
            c[k] = ((src >> sh) & 0xfu) | (((src >> (sh + 4)) & 0xfu) << 16);
        }
    };
    uint32_t c[8];
    local_counts(c);
    __shared__ uint32_t wtab[SS_THREADS / 64][8];
    uint32_t ex[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t incl = ss_wave_incl_scan(c[k], lane);
        ex[k] = incl - c[k];
        if (lane == 63) wtab[w][k] = incl;
    }
    __syncthreads();
    for (int k = 0; k < w; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++) ex[u] += wtab[k][u];
    }
    if (tid < 16) {                                                    // tile totals, their exclusive scan, the ranges
        uint32_t tot = 0u;
        for (int k = 0; k < SS_THREADS / 64; k++) tot += (wtab[k][tid >> 1] >> (16 * (tid & 1))) & 0xffffu;
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (tid >= d) incl += t;
        }
        const uint32_t st = s_before + incl - tot;
        tile_start[tid] = st;

        const int tx = (s % a.SX) * GSR_SS_TILES + (tid & 3), ty = (s / a.SX) * GSR_SS_TILES + (tid >> 2);
        if (tx < a.gridx && ty < a.gridy) a.ranges[ty * a.gridx + tx] = make_uint2(st, st + tot);
    }
    __syncthreads();
    // 16 * ITEMS predicated 4-byte stores per thread; consecutive lanes hold consecutive entries, so the lanes that do store for
    // a tile hit neighbouring addresses.  (Staging the super-tile's lists in LDS and copying them out densely was measured: slower.)
    uint32_t pos[16];
#pragma unroll
    for (int t = 0; t < 16; t++) pos[t] = tile_start[t] + ((ex[t >> 1] >> (16 * (t & 1))) & 0xffffu);
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const uint32_t bitv = (m[q] >> t) & 1u;
            if (bitv && GSR_IDX_OK(pos[t], a.hdr[SS_HDR_N], a.hdr + GSR_DBG_GEOM_WORD, GSR_BOUND_POINT_LIST)) a.point_list[pos[t]] = idv[q];
            pos[t] += bitv;
        }
    }
}

// ---- 4b: a crowded super-tile (GSR_SS_CAP < n <= GSR_SS_CAP_BIG entries: a dense cloud centre) cut across SS_SPLIT_PARTS workgroups.
// One workgroup ordering such a bin alone takes ~60 us (config 4: the whole stage was that bin).  The parts share nothing at run
// time: every part reads ALL the bin's entries (224 KB), builds the same sub-bucket histogram, cuts the sub-buckets into PARTS
// contiguous groups of about n / PARTS entries each -- the same cut in every part, it is a function of the histogram -- and then
// orders and expands only its own group; where its entries start in each tile's list follows from what it saw of the groups
// before it (per-tile counts of their mask bits).  A bin whose histogram does not cut that way (one sub-bucket beyond the LDS
// buffer: thousands of equal depths) is flagged and left to the single-workgroup kernel.
#define SS_SPLIT_PARTS 4
template <int PARTS>
__global__ __launch_bounds__(SS_THREADS, 4) void ss_sort_expand_split_kernel(SsSortArgs a) {
    constexpr int CAP = GSR_SS_CAP, ITEMS = GSR_SS_CAP_BIG / SS_THREADS, ITEMS_P = CAP / SS_THREADS;
    extern __shared__ uint64_t buf[];                              // [CAP] keys of this part; later: sorted ids (u32) | sorted masks (u16)
    uint16_t *mbuf = reinterpret_cast<uint16_t *>(buf + CAP);
    __shared__ uint32_t start[SS_NSUB + 1], cur[SS_NSUB];
    __shared__ uint32_t wred[3][SS_THREADS / 64];
    __shared__ uint32_t s_kmin, s_kmax, s_before;
    __shared__ uint32_t tile_start[16], tile_before[16];
    __shared__ uint32_t s_g[PARTS + 1];
    __shared__ uint32_t wsum[2][SS_THREADS / 64][8];
    if (a.hdr[DO_OVERFLOW]) return;                                // grid-uniform
    const int s = blockIdx.x / PARTS, part = blockIdx.x % PARTS, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t e0 = a.bin_start[s];
    const int n = (int)a.bin_cur[s];
    if (n <= GSR_SS_CAP || n > GSR_SS_CAP_BIG) return;             // workgroup-uniform: not a crowded bin
    {
        uint32_t before = 0;
        for (int k = tid; k < s; k += SS_THREADS) before += a.bin_pairs[k];
        before = ss_wave_sum(before);
        if (lane == 0) wred[0][w] = before;
    }
    if (tid <= SS_NSUB) start[tid] = 0u;
    if (tid <= PARTS) s_g[tid] = tid == 0 ? 0u : (uint32_t)SS_NSUB;
    uint64_t key[ITEMS];
    uint32_t msk[ITEMS];
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int j = tid + q * SS_THREADS;
        key[q] = ~0ull; msk[q] = 0u;
        if (j < n) {
            const uint4 e = a.entries[e0 + j];
            key[q] = ((uint64_t)e.x << 32) | e.y; msk[q] = e.z;
            kmin = min(kmin, e.x); kmax = max(kmax, e.x);
        }
    }
    kmin = ss_wave_min(kmin); kmax = ss_wave_max(kmax);
    if (lane == 0) { wred[1][w] = kmin; wred[2][w] = kmax; }
    __syncthreads();
    if (tid == 0) {
        uint32_t b = 0, mn = 0xffffffffu, mx = 0u;
        for (int k = 0; k < SS_THREADS / 64; k++) { b += wred[0][k]; mn = min(mn, wred[1][k]); mx = max(mx, wred[2][k]); }
        s_before = b; s_kmin = mn; s_kmax = mx;
    }
    __syncthreads();
    const uint32_t kmin0 = s_kmin;
    const uint64_t kspan = (uint64_t)(s_kmax >= s_kmin ? s_kmax - s_kmin : 0u) + 1ull;
    const float dmin = __uint_as_float(s_kmin);
    const float fspan = __uint_as_float(s_kmax) - dmin;
    const float fscale = (s_kmax > s_kmin && fspan > 0.f) ? (float)SS_NSUB / fspan : 0.f;
    bool log_map = false;                                              // workgroup-uniform
    auto sub_of = [&](uint64_t k) -> uint32_t {                        // the maps of ss_sort_expand_kernel
        const uint32_t kb = (uint32_t)(k >> 32);
        if (log_map) return (uint32_t)(((uint64_t)(kb - kmin0) * (uint64_t)SS_NSUB) / kspan);
        const float v = (__uint_as_float(kb) - dmin) * fscale;
        const uint32_t f = v > 0.f ? (uint32_t)v : 0u;
        return f < SS_NSUB ? f : SS_NSUB - 1u;
    };
    for (int attempt = 0; attempt < 2; attempt++) {
#pragma unroll
        for (int q = 0; q < ITEMS; q++)
            if (tid + q * SS_THREADS < n) atomicAdd(&start[sub_of(key[q])], 1u);
        __syncthreads();
        if (attempt == 1) break;
        const uint32_t mx = ss_wave_max(tid < SS_NSUB ? start[tid] : 0u);
        if (lane == 0) wred[1][w] = mx;
        __syncthreads();
        uint32_t m = 0;
        for (int k = 0; k < SS_THREADS / 64; k++) m = max(m, wred[1][k]);
        if (m <= SS_RANK_MAX) break;                                   // workgroup-uniform
        log_map = true;
        __syncthreads();
        if (tid <= SS_NSUB) start[tid] = 0u;
        __syncthreads();
    }
    {
        const uint32_t v = tid < SS_NSUB ? start[tid] : 0u;
        const uint32_t incl = ss_wave_incl_scan(v, lane);
        if (lane == 63) wred[0][w] = incl;
        __syncthreads();
        uint32_t ex = incl - v;
        for (int k = 0; k < w; k++) ex += wred[0][k];
        if (tid < SS_NSUB) start[tid] = ex;
        if (tid == SS_NSUB) start[SS_NSUB] = (uint32_t)n;
    }
    __syncthreads();
    // the cut: group p starts at the first sub-bucket whose first position is at or beyond p n / PARTS
    if (tid < SS_NSUB) {
#pragma unroll
        for (int p = 1; p < PARTS; p++) {
            const uint32_t target = (uint32_t)(((uint64_t)n * p) / PARTS);
            if (start[tid] >= target && (tid == 0 || start[tid - 1] < target)) s_g[p] = (uint32_t)tid;
        }
    }
    __syncthreads();
    bool fits = true;
#pragma unroll
    for (int p = 0; p < PARTS; p++) fits = fits && (start[s_g[p + 1]] - start[s_g[p]] <= (uint32_t)CAP) && s_g[p] <= s_g[p + 1];
    if (!fits) {                                                       // workgroup-uniform, and the same in every part
        if (part == 0 && tid == 0) a.unsplit[s] = 1u;
        return;
    }
    const uint32_t g0 = s_g[part], g1 = s_g[part + 1];
    const uint32_t base0 = start[g0];
    const int np = (int)(start[g1] - base0);                           // entries of this part
    if (tid < SS_NSUB) cur[tid] = start[tid] - base0;                  // only [g0, g1) is used
    __syncthreads();
    // own entries into LDS; what the bin's entries of the groups before this one, and all of them, put into each tile
    auto spread8 = [](uint32_t x) {
        x = (x | (x << 12)) & 0x000f000fu; x = (x | (x << 6)) & 0x03030303u; x = (x | (x << 3)) & 0x11111111u;
        return x;
    };
    uint32_t tlo = 0u, thi = 0u, blo = 0u, bhi = 0u;                   // nibble counters (<= ITEMS < 16 per thread)
#pragma unroll
    for (int q = 0; q < ITEMS; q++)
        if (tid + q * SS_THREADS < n) {
            const uint32_t sb = sub_of(key[q]);
            const uint32_t lo = spread8(msk[q] & 0xffu), hi = spread8(msk[q] >> 8);
            tlo += lo; thi += hi;
            if (sb < g0) { blo += lo; bhi += hi; }
            else if (sb < g1) {
                const uint32_t pos = atomicAdd(&cur[sb], 1u);
                if (GSR_IDX_OK(pos, CAP, a.hdr + GSR_DBG_GEOM_WORD, GSR_BOUND_SS_LDS_POS)) { buf[pos] = key[q]; mbuf[pos] = (uint16_t)msk[q]; }
            }
        }
    {   // workgroup sums of the per-tile counts, two tiles to a word (<= 14 336 < 2^16 each)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int sh = 8 * (k & 3);
            const uint32_t ts = k < 4 ? tlo : thi, bs = k < 4 ? blo : bhi;
            const uint32_t tw = ((ts >> sh) & 0xfu) | (((ts >> (sh + 4)) & 0xfu) << 16);
            const uint32_t bw = ((bs >> sh) & 0xfu) | (((bs >> (sh + 4)) & 0xfu) << 16);
            const uint32_t tsum = ss_wave_sum(tw), bsum = ss_wave_sum(bw);
            if (lane == 0) { wsum[0][w][k] = tsum; wsum[1][w][k] = bsum; }
        }
    }
    __syncthreads();
    if (tid < 16) {
        uint32_t tot = 0u, bef = 0u;
        for (int k = 0; k < SS_THREADS / 64; k++) {
            tot += (wsum[0][k][tid >> 1] >> (16 * (tid & 1))) & 0xffffu;
            bef += (wsum[1][k][tid >> 1] >> (16 * (tid & 1))) & 0xffffu;
        }
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (tid >= d) incl += t;
        }
        const uint32_t st = s_before + incl - tot;
        tile_start[tid] = st; tile_before[tid] = bef;
        if (part == 0) {
            const int tx = (s % a.SX) * GSR_SS_TILES + (tid & 3), ty = (s / a.SX) * GSR_SS_TILES + (tid >> 2);
            if (tx < a.gridx && ty < a.gridy) a.ranges[ty * a.gridx + tx] = make_uint2(st, st + tot);
        }
    }
    __syncthreads();
    uint32_t *sid = reinterpret_cast<uint32_t *>(buf);
    uint16_t *smask = reinterpret_cast<uint16_t *>(sid + CAP);
    {
        uint32_t rk[ITEMS_P], id[ITEMS_P], mm[ITEMS_P];
#pragma unroll
        for (int q = 0; q < ITEMS_P; q++) {
            const int j = tid + q * SS_THREADS;
            rk[q] = 0u; id[q] = 0u; mm[q] = 0u;
            if (j < np) {
                const uint64_t me = buf[j];
                const uint32_t sb = sub_of(me);
                const uint32_t a0 = start[sb] - base0, a1 = start[sb + 1] - base0;
                uint32_t r = a0, k = a0;
                for (; k + 8 <= a1; k += 8) {
                    uint64_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = buf[k + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) r += v[u] < me ? 1u : 0u;
                }
                for (; k < a1; k++) r += buf[k] < me ? 1u : 0u;
                rk[q] = r; id[q] = (uint32_t)me; mm[q] = mbuf[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ITEMS_P; q++)
            if (tid + q * SS_THREADS < np) { sid[rk[q]] = id[q]; smask[rk[q]] = (uint16_t)mm[q]; }
    }
    __syncthreads();
    uint32_t m[ITEMS_P], idv[ITEMS_P];
    const int jb = tid * ITEMS_P;
#pragma unroll
    for (int q = 0; q < ITEMS_P; q++) {
        const int j = jb + q;
        m[q] = j < np ? (uint32_t)smask[j] : 0u;
        idv[q] = j < np ? sid[j] : 0u;
    }
    uint32_t c[8];
    {
        uint32_t nlo = 0u, nhi = 0u;
#pragma unroll
        for (int q = 0; q < ITEMS_P; q++) { nlo += spread8(m[q] & 0xffu); nhi += spread8(m[q] >> 8); }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t src = k < 4 ? nlo : nhi;
            const int sh = 8 * (k & 3);
            c[k] = ((src >> sh) & 0xfu) | (((src >> (sh + 4)) & 0xfu) << 16);
        }
    }
    uint32_t ex[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t incl = ss_wave_incl_scan(c[k], lane);
        ex[k] = incl - c[k];
        if (lane == 63) wsum[0][w][k] = incl;
    }
    __syncthreads();
    for (int k = 0; k < w; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++) ex[u] += wsum[0][k][u];
    }
    uint32_t pos[16];
#pragma unroll
    for (int t = 0; t < 16; t++) pos[t] = tile_start[t] + tile_before[t] + ((ex[t >> 1] >> (16 * (t & 1))) & 0xffffu);
#pragma unroll
    for (int q = 0; q < ITEMS_P; q++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const uint32_t bitv = (m[q] >> t) & 1u;
            if (bitv && GSR_IDX_OK(pos[t], a.hdr[SS_HDR_N], a.hdr + GSR_DBG_GEOM_WORD, GSR_BOUND_POINT_LIST)) a.point_list[pos[t]] = idv[q];
            pos[t] += bitv;
        }
    }
}

static hipError_t ss_set_lds_attr(const void *fn, size_t bytes, std::atomic<uint64_t> &flags) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (flags.load() & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) flags.fetch_or(bit);
    return e;
}

static SsBinArgs ss_bin_args(const GeomView &g, const SuperSortPlan &pl, const SuperSortView &v, int P, int W, int H, int exact_cull) {
    SsBinArgs a;
    a.P = P; a.chunk = pl.chunk; a.SX = pl.SX; a.SY = pl.SY; a.W = W; a.H = H; a.exact_cull = exact_cull;
    a.ss_rec = g.ss_rec; a.rect = g.rect; a.depth_bits = reinterpret_cast<const uint32_t *>(g.depth); a.rec = g.rec;
    a.tiles = g.tiles; a.hot = g.hot; a.clamped = g.clamped; a.hot_cap = (uint32_t)acc_extra_rows(P);
    a.hdr = v.hdr; a.bin_cnt = v.bin_cnt; a.wg_cnt = g.ss_wg_cnt; a.bin_start = v.bin_start; a.bin_cur = v.bin_cur; a.bin_pairs = v.bin_pairs;
    a.entries = g.ss_entries;
    return a;
}

hipError_t launch_super_sort_count(const GeomView &g, int P, int W, int H, int exact_cull, uint32_t *host_out, uint32_t seq, hipStream_t s) {
    const SuperSortPlan pl = super_sort_plan(P, W, H);
    const SuperSortView v = super_sort_view(g);
    static std::atomic<uint64_t> attr{0};
    const size_t lds = ((size_t)pl.S + SS_LDS_LIST_WORDS(pl.chunk)) * sizeof(uint32_t);
    if (lds > 48 * 1024) {
        const hipError_t e = ss_set_lds_attr(reinterpret_cast<const void *>(ss_count_kernel), ((size_t)GSR_SS_MAXS + SS_LDS_LIST_WORDS(GSR_SS_MAX_CHUNK)) * sizeof(uint32_t), attr);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ss_count_kernel, dim3(pl.nblk), dim3(SS_BIN_THREADS), lds, s, ss_bin_args(g, pl, v, P, W, H, exact_cull));
    const uint32_t ecap = pl.ecap > 0xffffffffll ? 0xffffffffu : (uint32_t)pl.ecap;
    hipLaunchKernelGGL(ss_scan_kernel, dim3(1), dim3(SS_THREADS), 0, s, pl.S, ecap, v.bin_cnt, v.bin_start, v.hdr, host_out, seq);
    return hipGetLastError();
}

hipError_t launch_super_sort_scatter(const GeomView &g, int P, int W, int H, int exact_cull, hipStream_t s) {
    const SuperSortPlan pl = super_sort_plan(P, W, H);
    const SuperSortView v = super_sort_view(g);
    static std::atomic<uint64_t> attr{0};
    const size_t lds = ((size_t)3 * pl.S + SS_LDS_LIST_WORDS(pl.chunk)) * sizeof(uint32_t);
    if (lds > 48 * 1024) {
        const hipError_t e = ss_set_lds_attr(reinterpret_cast<const void *>(ss_scatter_kernel), ((size_t)3 * GSR_SS_MAXS + SS_LDS_LIST_WORDS(GSR_SS_MAX_CHUNK)) * sizeof(uint32_t), attr);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ss_scatter_kernel, dim3(pl.nblk), dim3(SS_BIN_THREADS), lds, s, ss_bin_args(g, pl, v, P, W, H, exact_cull));
    return hipGetLastError();
}

hipError_t launch_super_sort_expand(const GeomView &g, const ImageView &im, uint32_t *point_list, int P, int W, int H, uint32_t maxbin,
                                    hipStream_t s) {
    const SuperSortPlan pl = super_sort_plan(P, W, H);
    const SuperSortView v = super_sort_view(g);
    static std::atomic<uint64_t> attr_small{0}, attr_big{0};
    SsSortArgs a;
    a.S = pl.S; a.SX = pl.SX; a.gridx = (W + GSR_TILE - 1) / GSR_TILE; a.gridy = (H + GSR_TILE - 1) / GSR_TILE;
    a.hdr = v.hdr; a.bin_start = v.bin_start; a.bin_cur = v.bin_cur; a.bin_pairs = v.bin_pairs; a.entries = g.ss_entries;
    a.point_list = point_list; a.ranges = im.ranges;
    a.unsplit = g.ss_wg_cnt;          // the counting scratch is free once the scatter kernel has run (pl.S <= GSR_SS_WGCNT_WORDS)
    a.flagged_only = 0;
    {   // bins of up to GSR_SS_CAP entries: 70 KB of LDS, two workgroups per CU
        const size_t lds = (size_t)GSR_SS_CAP * 10;
        const hipError_t e = ss_set_lds_attr(reinterpret_cast<const void *>(ss_sort_expand_kernel<GSR_SS_CAP>), lds, attr_small);
        if (e != hipSuccess) return e;
        a.n_lo = 0; a.n_hi = GSR_SS_CAP;
        hipLaunchKernelGGL(ss_sort_expand_kernel<GSR_SS_CAP>, dim3(pl.S), dim3(SS_THREADS), lds, s, a);
    }
    if (maxbin > GSR_SS_CAP) {   // the crowded ones (a dense cloud centre): cut across four workgroups each ...
        static std::atomic<uint64_t> attr_split{0};
        const size_t lds_s = (size_t)GSR_SS_CAP * 10;
        const hipError_t es = ss_set_lds_attr(reinterpret_cast<const void *>(ss_sort_expand_split_kernel<SS_SPLIT_PARTS>), lds_s, attr_split);
        if (es != hipSuccess) return es;
        hipLaunchKernelGGL(ss_sort_expand_split_kernel<SS_SPLIT_PARTS>, dim3(pl.S * SS_SPLIT_PARTS), dim3(SS_THREADS), lds_s, s, a);
        // ... and what would not cut, in one workgroup with the 140 KB buffer (round 2 ran EVERY bin of such a frame this way)
        a.flagged_only = 1;
        const size_t lds = (size_t)GSR_SS_CAP_BIG * 10;
        const hipError_t e = ss_set_lds_attr(reinterpret_cast<const void *>(ss_sort_expand_kernel<GSR_SS_CAP_BIG>), lds, attr_big);
        if (e != hipSuccess) return e;
        a.n_lo = GSR_SS_CAP + 1; a.n_hi = 0x7fffffff;
        hipLaunchKernelGGL(ss_sort_expand_kernel<GSR_SS_CAP_BIG>, dim3(pl.S), dim3(SS_THREADS), lds, s, a);
    }
    return hipGetLastError();
}

}  // namespace gsr
