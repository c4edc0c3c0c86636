// tile_lists.hip -- per-tile, depth-ordered Gaussian lists (S7 + S8) without a sort of the (tile, Gaussian) pairs.
//
// Input: the depth-ordered list of emitting Gaussians (perm[0..Pl), binning.hip / depth_order.hip).
// Output: point_list (Gaussian ids, every tile's slice in depth order) and ranges[tile]; same per-tile lists as
// a stable sort of the pairs by tile id, laid out super-tile-major instead of tile-major (the compositing
// kernels only ever follow ranges[]; gsr_debug_read_binning re-linearises for the parity tests).
//
// The pair sort moves 8 bytes x N pairs four times through HBM behind ~10 rocPRIM launches (~230 us with key
// emission at 6.3 M pairs).  Here the unit of work is the (Gaussian, SUPER-TILE) entry -- a super-tile is 8 x 8
// tiles, so its tiles are the 64 bits of one mask and the 64 lanes of one wave -- and there are ~2 entries per
// Gaussian instead of ~6 pairs:
//   level 1  a stable multi-split of the entries by super-tile: the entries of a workgroup (512 consecutive
//            Gaussians of the depth order) are counted per super-tile ([super-tile][workgroup] matrix, row scan),
//            and placed with a per-wave 64-bit lane mask per super-tile in LDS: an entry's position is the popcount
//            of the lower lanes in its mask -- deterministic, in depth order, no sort.  The entry carries the
//            Gaussian id and its 64-bit tile mask, assembled from the row spans preprocess left in one word per
//            Gaussian (rectangles beyond 8 x 15 tiles re-evaluate the ellipse-vs-tile-row spans here).
//   level 2  every 128-entry segment of a super-tile's list is expanded by one wave: a 64 x 64 bit-matrix
//            transpose across the wave turns 64 entry masks into 64 tile columns (lane = tile), giving per-segment
//            tile counts, a column scan per super-tile, then each lane appends the ids of its column's set bits.
// All of it streams ~16 bytes per entry and 4 bytes per pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {

#define TL_L1_THREADS GSR_TL_L1
#define TL_L1_WAVES (TL_L1_THREADS / 64)
#define TL_SEG GSR_TL_SEG
#define TL_STAGE 1024             // ids one wave stages in LDS before writing them out (a 128-entry segment yields ~650)

TileListPlan tile_list_plan(int P, int64_t E, int W, int H) {
    TileListPlan p;
    const int gridx = (W + GSR_TILE - 1) / GSR_TILE, gridy = (H + GSR_TILE - 1) / GSR_TILE;
    p.SX = (gridx + 7) / 8;
    p.SY = (gridy + 7) / 8;
    p.S = p.SX * p.SY;
    p.nblk1 = (P > 0 ? P + TL_L1_THREADS - 1 : TL_L1_THREADS) / TL_L1_THREADS;    // matrix row stride: all P Gaussians
    p.nseg_max = (int64_t)(E > 0 ? E : 0) / TL_SEG + p.S;
    return p;
}

__device__ __forceinline__ uint32_t tl_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t tl_wave_max(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m));
    return v;
}
__device__ __forceinline__ uint32_t tl_wave_sum(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}

struct SuperRect { int x0, x1, y0, y1, sx0, sx1, sy0, sy1; };
__device__ __forceinline__ SuperRect super_rect(uint4 rc, int SX, int SY) {
    SuperRect r;
    r.x0 = (int)(rc.x & 0xffffu); r.x1 = (int)(rc.x >> 16); r.y0 = (int)(rc.y & 0xffffu); r.y1 = (int)(rc.y >> 16);
    // tile rectangles lie inside the grid; the clamp only bounds the loops should a record ever be garbage
    r.sx0 = r.x0 >> 3; r.sx1 = min((r.x1 + 7) >> 3, SX); r.sy0 = r.y0 >> 3; r.sy1 = min((r.y1 + 7) >> 3, SY);
    return r;
}

// ---- level 1a: entries per (super-tile, workgroup) ----
__global__ __launch_bounds__(TL_L1_THREADS) void tl_count_kernel(int P, const uint32_t *__restrict__ hdr, int SX, int SY, int nblk1,
                                                                 const uint4 *__restrict__ orect, uint32_t *__restrict__ mat1) {
    extern __shared__ uint32_t cnt[];
    const int S = SX * SY;
    if (hdr && hdr[DO_OVERFLOW]) return;                                 // grid-uniform: depth_order.hip gave up, the caller re-runs this
    const int Pl = hdr ? min((int)hdr[DO_PV], P) : P;                    // length of the depth-ordered list
    for (int b = threadIdx.x; b < S; b += TL_L1_THREADS) cnt[b] = 0u;
    __syncthreads();
    const int r = blockIdx.x * TL_L1_THREADS + threadIdx.x;
    if (r < Pl) {
        const SuperRect q = super_rect(orect[r], SX, SY);
        if (q.x1 > q.x0)
            for (int sy = q.sy0; sy < q.sy1; sy++)
                for (int sx = q.sx0; sx < q.sx1; sx++) atomicAdd(&cnt[sy * SX + sx], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < S; b += TL_L1_THREADS) mat1[(size_t)b * nblk1 + blockIdx.x] = cnt[b];
}

// ---- level 1b: one workgroup per super-tile: exclusive scan of its matrix row, total ----
__global__ __launch_bounds__(1024) void tl_binscan_kernel(int nblk1, uint32_t *__restrict__ mat1, uint32_t *__restrict__ bin_total) {
    __shared__ uint32_t wsum[16];
    uint32_t *row = mat1 + (size_t)blockIdx.x * nblk1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < nblk1; base += 1024) {                     // workgroup-uniform
        const int j = base + (int)threadIdx.x;
        const uint32_t v = j < nblk1 ? row[j] : 0u;
        const uint32_t incl = tl_wave_incl_scan(v, lane);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t ex = incl - v, tot = 0;
        for (int k = 0; k < 16; k++) { if (k < w) ex += wsum[k]; tot += wsum[k]; }
        if (j < nblk1) row[j] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) bin_total[blockIdx.x] = carry;
}

// ---- level 1c: place the entries (stable), computing their tile masks ----
struct TlScatterArgs {
    int Pl, SX, SY, S, nblk1, W, H, exact_cull;
    const uint32_t *perm;
    const uint4 *orect;
    const float *rec;
    const uint32_t *mat1, *bin_total;
    uint32_t *binstart, *segbase;    // [S + 1], written by workgroup 0
    uint32_t *seg_super;             // [segments] owning super-tile, written by workgroup 0
    uint4 *entries;
};
__global__ __launch_bounds__(TL_L1_THREADS) void tl_scatter_kernel(TlScatterArgs a) {
    extern __shared__ uint64_t lds64[];
    const int S = a.S;
    uint64_t *masks = lds64;                                             // [TL_L1_WAVES][S] lanes of the wave that touch the super-tile
    uint32_t *woff = reinterpret_cast<uint32_t *>(masks + TL_L1_WAVES * S);   // [TL_L1_WAVES][S] first slot of the wave's run
    uint32_t *bstart = woff + TL_L1_WAVES * S;                           // [S + 1]
    __shared__ uint32_t wsum[2][TL_L1_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // every global read this workgroup needs is issued up front: the kernel is a chain of round trips otherwise
    const int r = blockIdx.x * TL_L1_THREADS + tid;
    uint4 rc = make_uint4(0u, 0u, 0u, 0u);
    uint32_t id = 0u;
    if (r < a.Pl) { rc = a.orect[r]; id = a.perm[r]; }
    const uint32_t my_total = tid < S ? a.bin_total[tid] : 0u;
    const uint32_t my_before = tid < S ? a.mat1[(size_t)tid * a.nblk1 + blockIdx.x] : 0u;
    for (int i = tid; i < TL_L1_WAVES * S; i += TL_L1_THREADS) masks[i] = 0ull;
    {   // list start of every super-tile and its first segment (S <= TL_L1_THREADS: one per thread)
        const uint32_t v = my_total;
        const uint32_t sg = tid < S ? max(1u, (v + TL_SEG - 1) / TL_SEG) : 0u;
        const uint32_t iv = tl_wave_incl_scan(v, lane), is = tl_wave_incl_scan(sg, lane);
        if (lane == 63) { wsum[0][w] = iv; wsum[1][w] = is; }
        __syncthreads();
        uint32_t ev = iv - v, es = is - sg;
        for (int k = 0; k < w; k++) { ev += wsum[0][k]; es += wsum[1][k]; }
        if (tid < S) bstart[tid] = ev;
        if (tid == S - 1) bstart[S] = ev + v;
        if (blockIdx.x == 0 && tid < S) {
            a.binstart[tid] = ev; a.segbase[tid] = es;
            if (tid == S - 1) { a.binstart[S] = ev + v; a.segbase[S] = es + sg; }
            for (uint32_t k = 0; k < sg; k++) a.seg_super[es + k] = (uint32_t)tid;
        }
    }
    __syncthreads();
    const SuperRect q = super_rect(rc, a.SX, a.SY);
    const bool emits = q.x1 > q.x0;
    if (emits)
        for (int sy = q.sy0; sy < q.sy1; sy++)
            for (int sx = q.sx0; sx < q.sx1; sx++)
                atomicOr(reinterpret_cast<unsigned long long *>(&masks[w * S + sy * a.SX + sx]), 1ull << lane);
    __syncthreads();
    if (tid < S) {
        uint32_t run = bstart[tid] + my_before;
#pragma unroll
        for (int k = 0; k < TL_L1_WAVES; k++) { woff[k * S + tid] = run; run += (uint32_t)__popcll(masks[k * S + tid]); }
    }
    __syncthreads();
    if (!emits) return;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t sp = (uint64_t)rc.z | ((uint64_t)rc.w << 32);
    if (sp != ~0ull) {
        // small rectangle (<= 8 rows, <= 2 x 2 super-tiles): the masks come straight from preprocess's row spans.
        // Rows 0..7 of the rectangle are first packed one byte each (constant shifts: k is unrolled) per super-tile
        // column, then the 8-byte strips move down by y0 & 7 rows into the upper / lower super-tile.
        const int xs = q.x0 & 7, ys = q.y0 & 7;
        uint32_t c0lo = 0u, c0hi = 0u, c1lo = 0u, c1hi = 0u;                // strip of column j: rows 0-3 | rows 4-7
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t bb = k < 4 ? (rc.z >> (8 * k)) & 0xffu : (rc.w >> (8 * (k - 4))) & 0xffu;   // 0 beyond the last row
            const uint32_t lo4 = bb & 15u, hi4 = bb >> 4;
            const uint32_t w24 = (((1u << hi4) - 1u) & ~((1u << lo4) - 1u)) << xs;
            const uint32_t b0 = w24 & 0xffu, b1 = (w24 >> 8) & 0xffu;
            if (k < 4) { c0lo |= b0 << (8 * k); c1lo |= b1 << (8 * k); } else { c0hi |= b0 << (8 * (k - 4)); c1hi |= b1 << (8 * (k - 4)); }
        }
        const uint64_t strip[2] = {(uint64_t)c0lo | ((uint64_t)c0hi << 32), (uint64_t)c1lo | ((uint64_t)c1hi << 32)};
        const int sh = 8 * ys;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (q.sx0 + j < q.sx1) {
                const uint64_t up = strip[j] << sh;                          // rows that stay in super row sy0
                const uint64_t dn = sh ? strip[j] >> (64 - sh) : 0ull;       // rows that spill into sy0 + 1
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    if (q.sy0 + i < q.sy1) {
                        const uint64_t mm = i ? dn : up;
                        const int bin = (q.sy0 + i) * a.SX + q.sx0 + j;
                        const uint32_t slot = woff[w * S + bin] + (uint32_t)__popcll(masks[w * S + bin] & below);
                        a.entries[slot] = make_uint4(id, 0u, (uint32_t)mm, (uint32_t)(mm >> 32));
                    }
                }
            }
        }
        return;
    }
    // large rectangle: evaluate the ellipse-vs-tile-row spans here, 8 tile rows (one super row) at a time
    const float4 r0 = reinterpret_cast<const float4 *>(a.rec)[3 * (size_t)id];
    const float4 r1 = reinterpret_cast<const float4 *>(a.rec)[3 * (size_t)id + 1];
    const float4 r2 = reinterpret_cast<const float4 *>(a.rec)[3 * (size_t)id + 2];
    const CullParams cp = make_cull(r0.z, r0.w, r1.x, r2.z);
    for (int sy = q.sy0; sy < q.sy1; sy++) {
        uint32_t span[8];                                                // c0 | c1 << 16 of the 8 tile rows of this super row
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int ty = sy * 8 + k;
            span[k] = 0u;
            if (ty >= q.y0 && ty < q.y1) {
                int c0 = q.x0, c1 = q.x1;
                if (a.exact_cull) tile_row_span(cp, r0.x, r0.y, r0.z, r0.w, ty, a.W, a.H, q.x0, q.x1, c0, c1);
                span[k] = (uint32_t)c0 | ((uint32_t)c1 << 16);
            }
        }
        for (int sx = q.sx0; sx < q.sx1; sx++) {
            uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int lo = max((int)(span[k] & 0xffffu), sx * 8) - sx * 8, hi = min((int)(span[k] >> 16), sx * 8 + 8) - sx * 8;
                const uint32_t bits = hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
                if (k < 4) mlo |= bits << (8 * k); else mhi |= bits << (8 * (k - 4));
            }
            const int bin = sy * a.SX + sx;
            const uint32_t slot = woff[w * S + bin] + (uint32_t)__popcll(masks[w * S + bin] & below);
            a.entries[slot] = make_uint4(id, 0u, mlo, mhi);
        }
    }
}

// 64 x 64 bit-matrix transpose across the wave: lane l holds row l; afterwards lane t holds column t
// (bit l of the result = bit t of lane l's input).  Recursive exchange of the off-diagonal blocks.
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x, int lane) {
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
        const uint64_t M = j == 32 ? 0x00000000ffffffffull : j == 16 ? 0x0000ffff0000ffffull : j == 8 ? 0x00ff00ff00ff00ffull
                         : j == 4 ? 0x0f0f0f0f0f0f0f0full : j == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)x, j);
        x = (lane & j) ? ((x & ~M) | ((o & ~M) >> j)) : ((x & M) | ((o & M) << j));
    }
    return x;
}

// the same for the TL_SEG / 64 groups of a segment at once: independent exchange chains hide each other's latency
__device__ __forceinline__ void wave_transpose64x4(uint64_t x[TL_SEG / 64], int lane) {
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
        const uint64_t M = j == 32 ? 0x00000000ffffffffull : j == 16 ? 0x0000ffff0000ffffull : j == 8 ? 0x00ff00ff00ff00ffull
                         : j == 4 ? 0x0f0f0f0f0f0f0f0full : j == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
        uint64_t o[TL_SEG / 64];
#pragma unroll
        for (int q = 0; q < TL_SEG / 64; q++) o[q] = (uint64_t)__shfl_xor((unsigned long long)x[q], j);
#pragma unroll
        for (int q = 0; q < TL_SEG / 64; q++) x[q] = (lane & j) ? ((x[q] & ~M) | ((o[q] & ~M) >> j)) : ((x[q] & M) | ((o[q] & M) << j));
    }
}

// ---- level 2a: one wave per segment, lane = tile: how many of the segment's entries reach the tile ----
__global__ __launch_bounds__(256) void tl_segcount_kernel(int S, const uint32_t *__restrict__ segbase, const uint32_t *__restrict__ binstart,
                                                          const uint32_t *__restrict__ seg_super, const uint4 *__restrict__ entries,
                                                          uint32_t *__restrict__ segcnt) {
    const int lane = threadIdx.x & 63;
    const uint32_t seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= segbase[S]) return;                                       // wave-uniform
    const int s = (int)seg_super[seg];
    const uint32_t e0 = binstart[s] + (seg - segbase[s]) * TL_SEG, eend = binstart[s + 1];
    const int n = eend > e0 ? (int)min((uint32_t)TL_SEG, eend - e0) : 0;
    uint64_t m[TL_SEG / 64];
#pragma unroll
    for (int qq = 0; qq < TL_SEG / 64; qq++) {
        const int j = qq * 64 + lane;
        m[qq] = 0ull;
        if (j < n) { const uint4 e = entries[e0 + j]; m[qq] = (uint64_t)e.z | ((uint64_t)e.w << 32); }
    }
    wave_transpose64x4(m, lane);                                         // lane = tile: bit l of m[qq] = entry qq*64+l reaches it
    uint32_t c = 0;
#pragma unroll
    for (int qq = 0; qq < TL_SEG / 64; qq++) c += (uint32_t)__popcll(m[qq]);
    segcnt[(size_t)seg * 64 + lane] = c;
}

// ---- level 2b: one workgroup per super-tile: exclusive scan of the segment counts per tile, tile totals and
//      their offsets inside the super-tile's pair region ----
__global__ __launch_bounds__(1024) void tl_tilescan_kernel(const uint32_t *__restrict__ segbase, uint32_t *__restrict__ segcnt,
                                                           uint32_t *__restrict__ tile_off, uint32_t *__restrict__ tile_tot,
                                                           uint32_t *__restrict__ st_pairs) {
    __shared__ uint32_t part[16][64];
    __shared__ uint32_t tot[64];
    const int s = blockIdx.x, t = threadIdx.x & 63, g = threadIdx.x >> 6;
    const uint32_t sg0 = segbase[s];
    const int ns = (int)(segbase[s + 1] - sg0);
    const int rows = (ns + 15) / 16, r0 = g * rows, r1 = min(ns, r0 + rows);
    uint32_t sum = 0;
    for (int r = r0; r < r1; r++) sum += segcnt[(size_t)(sg0 + r) * 64 + t];
    part[g][t] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < g; k++) run += part[k][t];
    for (int r = r0; r < r1; r++) {
        const size_t idx = (size_t)(sg0 + r) * 64 + t;
        const uint32_t v = segcnt[idx];
        segcnt[idx] = run;
        run += v;
    }
    if (g == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) total += part[k][t];
        tot[t] = total;
        tile_tot[(size_t)s * 64 + t] = total;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t v = tot[t];
        const uint32_t incl = tl_wave_incl_scan(v, t);
        tile_off[(size_t)s * 64 + t] = incl - v;
        if (t == 63) st_pairs[s] = incl;
    }
}

// ---- level 2c: one wave per segment, lane = tile: append the ids of the entries whose mask has the lane's bit ----
__global__ __launch_bounds__(256) void tl_expand_kernel(int S, int SX, int gridx, int gridy, const uint32_t *__restrict__ segbase,
                                                        const uint32_t *__restrict__ binstart, const uint32_t *__restrict__ seg_super,
                                                        const uint4 *__restrict__ entries,
                                                        const uint32_t *__restrict__ segcnt, const uint32_t *__restrict__ tile_off,
                                                        const uint32_t *__restrict__ tile_tot, const uint32_t *__restrict__ st_pairs,
                                                        uint32_t *__restrict__ point_list, uint2 *__restrict__ ranges) {
    const int lane = threadIdx.x & 63;
    const uint32_t seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= segbase[S]) return;                                       // wave-uniform
    const int s = (int)seg_super[seg];
    uint32_t before = 0;                                                 // pairs of the super-tiles before this one
    for (int k = lane; k < s; k += 64) before += st_pairs[k];
    before = tl_wave_sum(before);
    const uint32_t off = before + tile_off[(size_t)s * 64 + lane];
    if (seg == segbase[s]) {                                             // the first segment publishes the tile ranges
        const int tx = (s % SX) * 8 + (lane & 7), ty = (s / SX) * 8 + (lane >> 3);
        if (tx < gridx && ty < gridy) ranges[ty * gridx + tx] = make_uint2(off, off + tile_tot[(size_t)s * 64 + lane]);
    }
    const uint32_t e0 = binstart[s] + (seg - segbase[s]) * TL_SEG, eend = binstart[s + 1];
    const int n = eend > e0 ? (int)min((uint32_t)TL_SEG, eend - e0) : 0;
    uint32_t cursor = off + segcnt[(size_t)seg * 64 + lane];
    uint64_t col[TL_SEG / 64];
    uint32_t eid[TL_SEG / 64];
#pragma unroll
    for (int qq = 0; qq < TL_SEG / 64; qq++) {
        const int j = qq * 64 + lane;
        col[qq] = 0ull; eid[qq] = 0u;
        if (j < n) { const uint4 e = entries[e0 + j]; eid[qq] = e.x; col[qq] = (uint64_t)e.z | ((uint64_t)e.w << 32); }
    }
    wave_transpose64x4(col, lane);                                       // bit l of col[qq]: entry qq*64+l reaches tile `lane`
    // The ids first go to a wave-private LDS slice, grouped by tile, and leave as one contiguous run per tile:
    // appended straight to global memory they are 4-byte stores into ~250 open cache lines per wave, which L2
    // evicts half empty (3.4 x write amplification measured).
    __shared__ uint32_t stage[4][TL_STAGE];
    __shared__ uint8_t stage_tile[4][TL_STAGE];                          // owning tile (= lane) of every staged id
    uint32_t *my = stage[threadIdx.x >> 6];
    uint8_t *my_tile = stage_tile[threadIdx.x >> 6];
    uint32_t pc[TL_SEG / 64], cnt = 0, most = 0;
#pragma unroll
    for (int qq = 0; qq < TL_SEG / 64; qq++) { pc[qq] = (uint32_t)__popcll(col[qq]); cnt += pc[qq]; most = max(most, pc[qq]); }
    const uint32_t lincl = tl_wave_incl_scan(cnt, lane);
    const uint32_t lstart = lincl - cnt;                                 // this tile's run inside the slice
    const uint32_t total = (uint32_t)__shfl((int)lincl, 63);
    const bool staged = total <= TL_STAGE;                               // wave-uniform
    // the groups append in order, so each has its own cursor and the four streams advance together
    uint32_t cur[TL_SEG / 64];
    {
        uint32_t run = staged ? lstart : cursor;
#pragma unroll
        for (int qq = 0; qq < TL_SEG / 64; qq++) { cur[qq] = run; run += pc[qq]; }
    }
    const int iters = (int)tl_wave_max(most);
    for (int it = 0; it < iters; it++) {                                 // wave-uniform trip count: every lane feeds the shuffles
#pragma unroll
        for (int qq = 0; qq < TL_SEG / 64; qq++) {
            const bool has = col[qq] != 0ull;
            const int l = has ? __ffsll((unsigned long long)col[qq]) - 1 : 0;
            const uint32_t gid = (uint32_t)__shfl((int)eid[qq], l);
            if (has) {
                if (staged) { my[cur[qq]] = gid; my_tile[cur[qq]] = (uint8_t)lane; } else point_list[cur[qq]] = gid;
                cur[qq]++;
            }
            col[qq] &= col[qq] - 1ull;
        }
    }
    if (!staged) return;
    __builtin_amdgcn_wave_barrier();
    // flattened copy-out: element i of the slice belongs to tile my_tile[i] and goes to i + (cursor - lstart) of that tile,
    // so consecutive lanes write consecutive words of a tile's run
    const uint32_t delta = cursor - lstart;
    for (uint32_t i = lane; i < ((total + 63u) & ~63u); i += 64) {       // wave-uniform trip count: every lane feeds the shuffle
        const bool in = i < total;
        const int t = in ? (int)my_tile[i] : 0;
        const uint32_t d = (uint32_t)__shfl((int)delta, t);
        if (in) point_list[i + d] = my[i];
    }
}

hipError_t launch_tile_lists_count(const GeomView &g, int P, const uint32_t *hdr, int W, int H, hipStream_t s) {
    const TileListPlan pl = tile_list_plan(P, 0, W, H);
    hipLaunchKernelGGL(tl_count_kernel, dim3(pl.nblk1), dim3(TL_L1_THREADS), pl.S * sizeof(uint32_t), s, P, hdr, pl.SX, pl.SY, pl.nblk1,
                       g.orect, g.tl_mat1);
    hipLaunchKernelGGL(tl_binscan_kernel, dim3(pl.S), dim3(1024), 0, s, pl.nblk1, g.tl_mat1, g.tl_bin_total);
    return hipGetLastError();
}

hipError_t launch_tile_lists(const GeomView &g, const TileListView &v, const ImageView &im, uint32_t *point_list, int P, int P_list,
                             int64_t E, int W, int H, int exact_cull, hipStream_t s) {
    const TileListPlan pl = tile_list_plan(P, E, W, H);
    const int gridx = (W + GSR_TILE - 1) / GSR_TILE, gridy = (H + GSR_TILE - 1) / GSR_TILE;
    TlScatterArgs a;
    a.Pl = P_list; a.SX = pl.SX; a.SY = pl.SY; a.S = pl.S; a.nblk1 = pl.nblk1; a.W = W; a.H = H; a.exact_cull = exact_cull;
    a.perm = g.perm; a.orect = g.orect; a.rec = g.rec; a.mat1 = g.tl_mat1; a.bin_total = g.tl_bin_total;
    a.binstart = v.binstart; a.segbase = v.segbase; a.seg_super = v.seg_super; a.entries = v.entries;
    const size_t lds = (size_t)TL_L1_WAVES * pl.S * (sizeof(uint64_t) + sizeof(uint32_t)) + (pl.S + 1) * sizeof(uint32_t);
    const int nblk_used = (P_list > 0 ? P_list + TL_L1_THREADS - 1 : TL_L1_THREADS) / TL_L1_THREADS;   // workgroup 0 always runs
    hipLaunchKernelGGL(tl_scatter_kernel, dim3(nblk_used), dim3(TL_L1_THREADS), lds, s, a);
    const unsigned sgrid = (unsigned)((pl.nseg_max + 3) / 4);
    hipLaunchKernelGGL(tl_segcount_kernel, dim3(sgrid), dim3(256), 0, s, pl.S, v.segbase, v.binstart, v.seg_super, v.entries, v.segcnt);
    hipLaunchKernelGGL(tl_tilescan_kernel, dim3(pl.S), dim3(1024), 0, s, v.segbase, v.segcnt, v.tile_off, v.tile_tot, v.st_pairs);
    hipLaunchKernelGGL(tl_expand_kernel, dim3(sgrid), dim3(256), 0, s, pl.S, pl.SX, gridx, gridy, v.segbase, v.binstart, v.seg_super, v.entries,
                       v.segcnt, v.tile_off, v.tile_tot, v.st_pairs, point_list, im.ranges);
    return hipGetLastError();
}

}  // namespace gsr
