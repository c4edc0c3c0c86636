// gsr_internal.h -- workspace layouts and kernel launchers shared by the translation units of
// libgsr_hip.so.  Nothing here is part of the public ABI (include/gsr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define GSR_TILE_HOST 16   // tile edge in pixels (== GSR_TILE of gsr_device.h)

namespace gsr {

// ---- capacity asserts of the debug build (python -m gaussian_transformer_amd.build --debug-bounds -> libgsr_hip_dbg.so) ----
// Every index into an LDS list, an entry run, a tile list, a unit list or the checkpoint pool whose bound is a PLAN made on the host
// or in another kernel (GSR_SS_CAP, the entry capacity, N, the list capacities ...) goes through GSR_IDX_OK in that build: an index
// at or beyond its capacity is recorded -- (code, index, capacity, count) in four spare words of the workspace headers, read by
// gsr_debug_read_bound_errors -- and the access is skipped, so the process survives and the test can say which bound broke.
// In the product build the macro is the constant `true` and costs nothing.
#define GSR_DBG_GEOM_WORD 12    // dord.hdr[12..15] (zeroed by preprocess with the other counters)
#define GSR_DBG_SEG_WORD 4      // seg.hdr[4..7]
#ifdef GSR_DEBUG_BOUNDS
__device__ __forceinline__ bool gsr_idx_ok(unsigned long long idx, unsigned long long cap, const uint32_t *words_c, uint32_t code) {
    if (idx < cap) return true;
    uint32_t *w = const_cast<uint32_t *>(words_c);
    if (atomicAdd(&w[3], 1u) == 0u) { w[0] = code; w[1] = (uint32_t)idx; w[2] = (uint32_t)cap; }
    return false;
}
#define GSR_IDX_OK(idx, cap, words, code) gsr_idx_ok((unsigned long long)(idx), (unsigned long long)(cap), (words), (code))
#else
#define GSR_IDX_OK(idx, cap, words, code) true
#endif
enum { GSR_BOUND_SS_BIG_LIST = 1, GSR_BOUND_SS_ENTRIES = 2, GSR_BOUND_SS_BIN_SIZE = 3, GSR_BOUND_SS_LDS_POS = 4, GSR_BOUND_POINT_LIST = 5,
       GSR_BOUND_FWD_LIST_READ = 6, GSR_BOUND_POOL_SLOT = 7, GSR_BOUND_BWD_LIST_READ = 8, GSR_BOUND_UNIT_LIST = 9, GSR_BOUND_UNIT_TICKET = 10,
       GSR_BOUND_SELFTEST = 99 };

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int ceil_log2_u32(uint32_t x) { int b = 0; while ((1u << b) < x && b < 31) b++; return b; }

// Per-Gaussian record gathered by the compositing kernels: 12 floats = 3 x float4.
//   [0] px  [1] py  [2] conic.xx  [3] conic.xy | [4] conic.yy [5] opacity [6] r [7] g | [8] b [9] depth [10] cull tau [11] replica code of the accumulator rows (bits; 0 = none)
#define GSR_REC_FLOATS 12
// Gradient accumulator of the reverse compositing pass: one 64-byte row per Gaussian so that the
// nine atomics of one (tile, Gaussian) pair fall into a single memory-side atomic request.
//   [0..2] dL/drgb  [3..4] sum s*d  [5..7] sum s*d d^T  [8] sum s,  s = opacity * G * dL/dalpha, d = centre - pixel
#define GSR_ACC_FLOATS 16
// A splat that covers hundreds of tiles receives one atomic per (wave, tile): they all meet in one 64-byte row and serialise
// in L2 (tiramisu ring cameras: 0.12 ms of a 0.42 ms reverse pass).  Such a splat gets 4..16 replica rows behind the P regular
// ones, the wave of tile t adds into replica t mod K, pergauss_bwd.hip sums them.  The code of a Gaussian: 0 = no replicas,
// else (first replica row - P) << 4 | log2 K; kept in hot[] (for pergauss_bwd, flagged by bit 7 of clamped[]) and in the record's spare float [11] (for the
// compositing kernel, which has the record in registers anyway).  Rows are handed out by supertile_sort.hip's counting kernel.
#define GSR_HOT_MIN_TILES 256   // fewer than 1 in 2000 Gaussians at config 3 (pergauss_bwd's fold loop diverges its wave)
static inline size_t acc_extra_rows(int P) { return (size_t)(P > 0 ? P : 1) / 32 + 1024; }
static inline size_t acc_rows(int P) { return (size_t)(P > 0 ? P : 1) + acc_extra_rows(P); }

// ---- depth_order.hip: bucketed depth order (replaces the rocPRIM depth sort + ordered scan) ----
#define GSR_DO_CAP 8192      // largest level-1 bucket one workgroup orders in LDS (64 KB of 64-bit keys)
#define GSR_DO_NSUB 512      // level-2 sub-buckets per bucket
#define GSR_DO_MAXB 8192     // level-1 buckets, at most (64 KB of LDS counters in the counting kernel)
#define GSR_DO_MAXBLK 512    // counting / scatter workgroups, at most
enum { DO_KMIN = 0, DO_KMAX = 1, DO_DONE = 2, DO_OVERFLOW = 3, DO_PV = 4, DO_NTOT = 5, DO_ETOT = 6, DO_HDR_WORDS = 16 };
#define GSR_DO_ZERO_WORDS (DO_HDR_WORDS + 3 * GSR_DO_MAXB)   // hdr | gpair (2 words each) | gcur, contiguous, zeroed by preprocess
struct DepthOrderPlan { int nb, nblk, chunk, npre; };
DepthOrderPlan depth_order_plan(int P, int log_map);   // log_map: bucket map linear in the depth bits, 4x the buckets
struct DepthOrderView {
    uint32_t *hdr;           // [DO_HDR_WORDS]; [DO_OVERFLOW, DO_PV, DO_NTOT] are read back by the host
    unsigned long long *gpair;   // [GSR_DO_MAXB] bucket size | pair-count sum << 32: one 64-bit atomic per workgroup and bucket
    uint32_t *gcur;              // [GSR_DO_MAXB] scatter cursors
    uint32_t *bstart, *tbase;    // [nb + 1] first position / pair-count base of every bucket
    uint32_t *blkmin, *blkmax;   // [npre] depth-bit extrema of the emitting Gaussians of every preprocess workgroup
    uint32_t *blkent;            // [npre] super-tile entries (tile_lists.hip) of every preprocess workgroup
    uint64_t *comp;              // [P] (depth bits << 32 | id), grouped by bucket
};

// ---- supertile_sort.hip: per-tile lists by binning entries per super-tile and ordering every bin in LDS ----
#define GSR_SS_TILES 4         // super-tile edge in tiles (64 x 64 pixels): 16 tiles = 16 mask bits = the 16 waves of a workgroup
#define GSR_SS_MAXS 8192       // super-tiles the LDS histograms cover (90 x 90: 5760 x 5760 pixels)
#define GSR_SS_CAP 7168        // entries one workgroup orders in 70 KB of LDS (two workgroups per CU)
#define GSR_SS_CAP_BIG 14336   // one workgroup per CU
#define GSR_SS_MAX_CHUNK 8192  // Gaussians per counting / scatter workgroup, at most (P <= 8 M; beyond that round 1's path runs)
#define GSR_SS_MIDCAP 1024     // medium rectangles one counting / scatter workgroup keeps as records in LDS (more: handled in place)
#define GSR_SS_ENT_PER_G 4     // capacity of the entry array in the geometry workspace, per Gaussian
#define GSR_SS_WGCNT_WORDS (4 << 20)   // per-(counting workgroup, super-tile) counts: 16 MB (1024 workgroups x 4096 super-tiles)
enum { SS_HDR_MAXBIN = 7, SS_HDR_N = 8, SS_HDR_E = 9, SS_HDR_HOT = 10 };   // words of the header next to DO_OVERFLOW (zeroed by preprocess)
struct SuperSortPlan { int SX, SY, S, chunk, nblk; int64_t ecap; };
SuperSortPlan super_sort_plan(int P, int W, int H);
struct SuperSortView {
    uint32_t *hdr;                       // depth_order.hip's header words (same zeroed region, used by one path at a time)
    uint32_t *bin_cnt, *bin_pairs, *bin_cur;   // [GSR_SS_MAXS] each, zeroed by preprocess (aliases gpair / gcur)
    uint32_t *bin_start;                 // [S + 1]
};

struct GeomView {          // per-Gaussian state, P entries each
    float *rec;            // [P][12]
    float *depth;          // [P]
    float *opac;           // [P] activated opacity (0 for a culled Gaussian): pergauss_bwd.hip scales the moment sums with it
    uint4 *rect;           // [P] (x0 | x1<<16, y0 | y1<<16, row spans lo, hi): tile rectangle + the tile-row spans of a
                           //     small rectangle (<= 8 rows, <= 15 columns, <= 2 super-tile columns) in one word:
                           //     byte k = (c0 - x0) | (c1 - x0) << 4 of row y0 + k; all ones = not representable
    uint32_t *tiles;       // [P] (Gaussian,tile) pairs emitted (== rect area when exact culling is off)
    uint32_t *offsets;     // [P] inclusive scan of tiles[perm[.]] (depth order)
    uint8_t *clamped;      // [P] bit ch (0-2) set iff SH colour channel ch was clamped at 0; bit 7: hot[] holds a replica code
    uint32_t *perm;        // [P] Gaussian ids in (depth, id) order
    uint32_t *depth_sorted;  // [P] sorted depth bits (by-product of the depth sort)
    uint4 *orect;          // [P] rect[perm[.]]: the same records in depth order (empty rectangle for a Gaussian that emits nothing)
    uint4 *ss_rec;         // [P] supertile_sort.hip's view of a Gaussian, written by preprocess: x = depth bits,
                           //     y = first bin (18 bits) | x0 & 3 << 18 | y0 & 3 << 20 | rows - 1 << 22 | cols - 1 << 25 | kind << 29,
                           //     kind 0: emits nothing; 1: zw = the 16-bit tile masks of the 2 x 2 super-tiles from that bin;
                           //     3: <= 8 rows x <= 15 columns: zw = row spans, byte k = (c0 - x0) | (c1 - x0) << 4 of row y0 + k;
                           //     2: larger rectangle (spans re-evaluated from rect / rec)
    uint8_t *touched;      // [P] composite_fwd stores this frame's mark (1..255, *touch_mark) for every Gaussian some wave staged with a
                           //     reachable 8x8 block.  Everything the reverse pass can give a gradient to is among those (it walks the
                           //     same entries up to each pixel's last contributor): pergauss_bwd writes plain zeros for a Gaussian whose
                           //     byte differs from the mark -- 91 % of the Gaussians at config 3, whose dense cloud is mostly occluded.
                           //     Never cleared: a stale or uninitialised byte that happens to equal the mark only costs the full
                           //     computation for that Gaussian
    uint32_t *touch_mark;  // [1] the mark of the forward pass that filled this workspace (written by preprocess)
    uint32_t *hot;         // [P] replica code of the gradient accumulator rows (GSR_HOT_MIN_TILES above); valid where clamped has bit 7
    uint4 *ss_entries;     // [GSR_SS_ENT_PER_G * P] supertile_sort.hip: (depth bits, id, 16-bit tile mask, -) grouped by super-tile
    uint32_t *ss_wg_cnt;   // [counting workgroups][S] entries per (workgroup of 4096 Gaussians, super-tile)
    uint32_t *tl_mat1;     // [GSR_TL_MAX_S][ceil(P / 512)] tile_lists.hip level-1 count matrix (filled before N is known)
    uint32_t *tl_bin_total;    // [GSR_TL_MAX_S]
    void *scan_temp;
    size_t scan_temp_bytes;
    void *dsort_temp;
    size_t dsort_temp_bytes;
    DepthOrderView dord;
    size_t total_bytes;
};
GeomView carve_geom(void *base, int P, size_t scan_temp_bytes, size_t dsort_temp_bytes);

// ---- segmented reverse pass: what composite_fwd.hip leaves for composite_bwd.hip besides final_T / n_contrib ----
// The reverse pass of a tile is a sequential chain over its list; one wave per half tile leaves the machine to a handful of
// waves for the second half of the kernel (wave timelines, scripts/wave_trace.py).  The forward wave therefore records, every
// `seg` list entries it is still alive at, a CHECKPOINT per pixel -- the transmittance in front of that entry and the colour
// composited behind it -- so that the reverse pass of [k seg, (k+1) seg) can start there on its own, and leaves per half tile
// how far its pixels got (plain stores, no atomics).  gsr_backward's first kernel (which also clears the accumulator rows) turns
// that into work units, one list per XCD band of tiles, longest units first; the reverse kernel is persistent: a wave draws the
// units of its own band (one ticket counter per band, 256 bytes apart: device-scope atomics on one line queue behind each other at
// several ns apiece) and helps the other bands -- those of its own XCD first -- when its own list is empty.  Long chains start first, short units fill the end.
#define GSR_SEG_BANDS 32        // unit lists: stripes of the image, stripe s served first by the waves of XCD s & 7 (workgroup b runs on XCD b & 7);
                                // four lists per XCD keep the ticket counters apart (1024 draws on one line at kernel start took 7 us)
#define GSR_SEG_MAXCK 7         // checkpoints one forward wave takes (lists beyond 8 seg entries keep one long last unit)
#define GSR_SEG_POOL_PER_UNIT 2 // checkpoint slots (128 pixels x 16 B) in the pool per half tile; a wave that finds none left stops segmenting
#define GSR_SEG_HDR_WORDS 8192  // the hot counters sit 256 bytes apart: device-scope atomics on lines that share a memory channel queue behind each other
#define GSR_SEG_LEN_CLASSES 18  // planner's counting sort: 0 = remainder longer than seg, 1 = full segment, 2..17 = shorter tops, longest first
enum { SEG_SEG = 0,             // entries per segment the forward pass used; 0: it left nothing (the reverse pass takes whole half tiles)
       SEG_BCOUNT = 16,         // [band] units in the band's list (written by the planner, read-only afterwards)
       SEG_POOL = 64,           // [band * GSR_SEG_CTR_STRIDE] checkpoint slots handed out of the band's share of the pool
       SEG_BTICKET = 64 * 33,   // [band * GSR_SEG_CTR_STRIDE] ticket counter of the reverse kernel (set by every gsr_backward)
       SEG_FTICKET = 64 * 65 }; // [band * GSR_SEG_CTR_STRIDE] ticket counter of the forward kernel
#define GSR_SEG_CTR_STRIDE 64
struct SegView {
    uint32_t *hdr;           // [GSR_SEG_HDR_WORDS], zeroed by preprocess
    float4 *pool;            // [pool_cap][128]: (T in front of the boundary, colour composited behind it) per pixel of a half tile
    uint32_t pool_cap;       // a multiple of GSR_SEG_BANDS: band x allocates from [x, x + 1) * pool_cap / 8
    uint2 *info;             // [units] (largest last-contributor position of the half tile's pixels, checkpoints taken)
    uint32_t *ck_slot;       // [units][8] pool slots of the half tile's checkpoints (boundary (j + 1) seg at [j])
    uint4 *bq;               // [GSR_SEG_BANDS][seg_list_cap] reverse units (half tile, first entry, end entry or ~0, checkpoint slot at the end entry or ~0);
                             // behind them the band's zero-fill units (~0, first Gaussian, end Gaussian, -)
    uint32_t units;          // 2 T half tiles
    uint32_t band_units;     // half tiles per band: band of half tile u = u / band_units
    uint32_t list_cap;       // entries per band list
};
#define GSR_SEG_FILL_CAP 2048   // (x 32 bands x 256 Gaussians = 16 M) zero-fill units a band's list can hold behind its compositing units (FillArgs below)
__host__ __device__ static inline size_t seg_list_base(const SegView &v, int band) { return (size_t)band * v.list_cap; }
// Gaussians per zero-fill unit: 256 (four rounds of a wave: a unit lasts a few us, so the fill really lies in the gaps of the
// compositing work); 0 = no fill units when P / 256 of them would not fit the lists (P > 16 M: pergauss_bwd writes the zeros)
#define GSR_SEG_FILL_CHUNK 256
static inline int seg_fill_chunk(int P) {
    const long long per_band = (((long long)P + GSR_SEG_FILL_CHUNK - 1) / GSR_SEG_FILL_CHUNK + GSR_SEG_BANDS - 1) / GSR_SEG_BANDS;
    return per_band <= GSR_SEG_FILL_CAP ? GSR_SEG_FILL_CHUNK : 0;
}

struct ImageView {
    uint2 *ranges;         // [T]
    float *final_T;        // [H*W]
    uint32_t *n_contrib;   // [H*W]
    SegView seg;
    size_t total_bytes;
};
ImageView carve_image(void *base, int W, int H);

struct BinningView {
    // one 24-byte-per-pair arena, two interpretations (regions A 4N | B 8N | C 8N | D 4N):
    //   global sort  : A point_list (sorted ids) | B keys_sorted u64 | C keys_unsorted u64 | D ids_unsorted
    //   two-level    : A point_list (ids sorted by tile, in depth order) | B tile keys unsorted (u32)
    //                  | C ids unsorted (u32) | D tile keys sorted
    uint32_t *point_list;        // [N] sorted Gaussian ids           (read by backward)
    uint8_t *contrib;            // [4][N] per 8x8 block of the tile: can this list entry reach the block (alpha >= 1/255)?
                                 //        written by composite_fwd for the entries it staged, read by composite_bwd
    uint64_t *keys_sorted;       // [N]
    uint64_t *keys_unsorted;     // [N]
    uint32_t *point_list_unsorted;  // [N]
    uint32_t *tkeys_unsorted;    // = A
    uint32_t *ids_sorted;        // = B
    uint32_t *ids_unsorted;      // = C
    uint32_t *tkeys_sorted;      // = D
    void *sort_temp;
    size_t sort_temp_bytes;
    size_t list_bytes;           // point_list + contrib: what the reverse pass needs of this workspace
    size_t total_bytes;
};
BinningView carve_binning(void *base, int64_t N, size_t sort_temp_bytes);

struct PreprocessArgs {
    int P, D, M, W, H, gridx, gridy;
    int raw_params;          // 1: opacities are logits, scales are log-scales, rotations are un-normalised
    const float *shs_rest;   // non-NULL: shs is the DC tensor [P,1,3], shs_rest is [P,M-1,3]
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *viewmatrix, *projmatrix, *campos;
    float scale_modifier, tanfovx, tanfovy;
    int *radii;
    int exact_cull;
    uint32_t touch_mark;     // this frame's mark for GeomView::touched (1..255)
    uint32_t *seg_hdr;       // SegView::hdr of the image workspace: zeroed here, with the other per-frame counters
    GeomView g;
};
hipError_t launch_preprocess_fwd(const PreprocessArgs &a, hipStream_t s);
hipError_t launch_composited_mask(int P, const uint8_t *touched, const uint32_t *mark, uint8_t *out, hipStream_t s);

hipError_t scan_temp_bytes(int P, size_t *bytes);

hipError_t sort_temp_bytes(int64_t N, int bits, size_t *bytes);
hipError_t launch_emit_keys(const GeomView &g, const BinningView &b, int P, int W, int H, int exact_cull, int two_level, hipStream_t s);
// histogram + bucket scan -> hdr totals; host_out (pinned, may be NULL) receives {overflow, Pv, N, E, seq}
hipError_t launch_depth_order_count(const GeomView &g, int P, int log_map, uint32_t *host_out, uint32_t seq, hipStream_t s);
// scatter, per-bucket order (+ scan of the pair counts when need_offsets) -> perm, orect, offsets
hipError_t launch_depth_order_place(const GeomView &g, int P, int log_map, int need_offsets, hipStream_t s);
hipError_t sort2_temp_bytes(int64_t N, int tile_bits, size_t *bytes);
hipError_t launch_sort2_by_tile(const BinningView &b, int64_t N, int tile_bits, hipStream_t s);
hipError_t depth_sort_temp_bytes(int P, size_t *bytes);
hipError_t launch_depth_sort(const GeomView &g, int P, hipStream_t s);
hipError_t launch_ordered_scan(const GeomView &g, int P, hipStream_t s);
hipError_t launch_sort(const BinningView &b, int64_t N, int bits, hipStream_t s);
hipError_t launch_entry_total(const GeomView &g, int P, hipStream_t s);          // hdr[DO_ETOT] and orect when depth_order.hip did not run

// ---- tile_lists.hip: per-tile depth-ordered lists through (Gaussian, super-tile) entries ----
#define GSR_TL_SEG 128       // entries per level-2 segment (64: 87 us, 128: 74 us, 256: 80 us, 512: 101 us for the tile lists at 1 M)
#define GSR_TL_L1 512        // Gaussians per level-1 workgroup
#define GSR_TL_MAX_S 512     // super-tiles (8 x 8 tiles) the LDS lane masks of level 1 cover: 4096 x 2176 pixels
struct TileListPlan { int SX, SY, S, nblk1; int64_t nseg_max; };
TileListPlan tile_list_plan(int P_list, int64_t E, int W, int H);
struct TileListView {
    uint32_t *binstart, *segbase;    // [S + 1] first entry / first segment of every super-tile
    uint32_t *seg_super;         // [nseg_max] super-tile of every segment
    uint4 *entries;              // [E] (Gaussian id, -, tile mask lo, hi), grouped by super-tile, depth order inside
    uint32_t *segcnt;            // [nseg_max][64] entries of the segment that reach the tile, scanned in place
    uint32_t *tile_off, *tile_tot;   // [S][64]
    uint32_t *st_pairs;          // [S]
    size_t total_bytes;
};
TileListView carve_tile_lists(void *base, const TileListPlan &pl, int64_t E);
// level-1 counting (needs neither N nor E: queued before the host reads them back); hdr: depth_order.hip's header
// (list length at [DO_PV], nothing to do when [DO_OVERFLOW]) or NULL (the list holds all P Gaussians)
hipError_t launch_tile_lists_count(const GeomView &g, int P, const uint32_t *hdr, int W, int H, hipStream_t s);
hipError_t launch_tile_lists(const GeomView &g, const TileListView &v, const ImageView &im, uint32_t *point_list, int P, int P_list,
                             int64_t E, int W, int H, int exact_cull, hipStream_t s);
hipError_t launch_ranges(const BinningView &b, const ImageView &im, int64_t N, int T, int two_level, hipStream_t s);

static inline SuperSortView super_sort_view(const GeomView &g) {
    SuperSortView v;
    v.hdr = g.dord.hdr;
    v.bin_cnt = reinterpret_cast<uint32_t *>(g.dord.gpair);                  // 2 x GSR_DO_MAXB zeroed words ...
    v.bin_pairs = reinterpret_cast<uint32_t *>(g.dord.gpair) + GSR_DO_MAXB;
    v.bin_cur = g.dord.gcur;                                                 // ... and GSR_DO_MAXB more
    v.bin_start = g.dord.bstart;
    return v;
}
// count + scan (totals to host_out: {overflow, largest bin, N, E, seq}); scatter (may be queued before the host has the totals);
// per-super-tile order + expansion into point_list / ranges
hipError_t launch_super_sort_count(const GeomView &g, int P, int W, int H, int exact_cull, uint32_t *host_out, uint32_t seq, hipStream_t s);
hipError_t launch_super_sort_scatter(const GeomView &g, int P, int W, int H, int exact_cull, hipStream_t s);
hipError_t launch_super_sort_expand(const GeomView &g, const ImageView &im, uint32_t *point_list, int P, int W, int H, uint32_t maxbin,
                                    hipStream_t s);

// lane-slot accounting of one compositing launch (instrumented kernels only; gsr_set_option("count_lanes", 1)).
// A "block visit" is one evaluation of one splat against one 8x8 pixel block = 64 lane slots; lanes_ok of them
// blended the splat, lanes_past_last sat on a pixel that was already saturated (forward) / stopped before this
// splat (reverse), lanes_below_alpha failed the alpha / power tests.
struct CompositeCounters {
    unsigned long long staged, visits, block_visits, lanes_ok, lanes_past_last, lanes_below_alpha, reductions,
        dead_block_visits, waves;
    uint4 *trace;                  // [trace_cap] per work unit: (start, end in 10 ns ticks of the constant-rate clock, entries staged, splat visits)
    unsigned long long trace_cap;
    unsigned long long pad[5];
};
#define GSR_TRACE_UNITS (1u << 20)

struct CompositeArgs {
    int W, H, gridx, gridy;
    CompositeCounters *counters;   // NULL: plain kernel
    int count_mode;                // with counters: 1 = lane-slot accounting, 2 = wave timeline only
    uint8_t *contrib;            // [4][contrib_stride]
    size_t contrib_stride;
    const uint2 *ranges;
    const uint32_t *point_list;
    const float *rec;
    const float *bg;
    float *final_T;
    uint32_t *n_contrib;
    float *out_color;
    uint8_t *touched;            // [P] GeomView::touched
    uint32_t touch_mark;         // value to store there
    SegView seg;                 // checkpoints + reverse work units (used by the 2-blocks-per-wave kernel when seg_len > 0)
    int seg_len;                 // entries per segment (multiple of 64); 0: no checkpoints, no units
    int asm_walk;                // 1: the written-out splat walk (composite_fwd.hip::walk_batch_2blocks) where it exists, 0: the C++ walk
    int pair_long_n;             // > 0: composite_fwd_pair_kernel -- half tiles with more list entries than this are walked by two waves, one per block
    int lpt_span;                // > 0 (and seg_len == 0): every half tile files how far its pixels got (SegView::info) and SEG_SEG = lpt_span, so that
                                 //    gsr_backward can order the reverse pass's half tiles by length (composite_bwd_lpt_kernel)
};
hipError_t launch_composite_fwd(const CompositeArgs &a, int npx, int exact_cull, int waves_per_block, hipStream_t s);

// Zero-fill of the gradient rows of Gaussians that receive no gradient (not composited by the forward pass, or culled): 248 B per
// Gaussian that gsr_backward's contract has written -- 91 % of the rows at config 3.  The persistent reverse kernel does it in
// units of its own, listed behind each band's compositing units: the waves that find no compositing work left write zeros while the
// last chains finish, and pergauss_bwd.hip (skip_unmarked) then touches only the Gaussians that have a gradient.
struct FillArgs {
    int P, M, chunk;             // chunk = Gaussians per fill unit (0: no fill units)
    const int *radii;
    const uint8_t *touched;
    const uint32_t *mark;
    float *means2D, *opacity, *colors, *means3D, *cov3D, *sh, *sh_rest, *scales, *rots;     // NULL: not an output of this call
};

struct CompositeBwdArgs {
    int W, H, gridx, gridy;
    CompositeCounters *counters;   // NULL: plain kernel
    int count_mode;                // with counters: 1 = lane-slot accounting, 2 = wave timeline only
    const uint8_t *contrib;      // [4][contrib_stride], from the forward pass
    size_t contrib_stride;
    const uint2 *ranges;
    const uint32_t *point_list;
    const float *rec;
    const float *bg;
    const float *final_T;
    const uint32_t *n_contrib;
    const float *dL_dpix;
    float *acc;   // [P + acc_extra_rows(P)][16], zeroed
    // deterministic mode only (det != NULL): one 16-float slot per (list entry, wave of the tile), zeroed; the geometry
    // det_reduce_kernel needs to find a Gaussian's entries again
    float *det;
    int P;
    const uint4 *rect;
    const uint32_t *tiles;
    const uint32_t *depth_bits;
    SegView seg;             // the forward pass's checkpoints and work units (persistent kernel)
    FillArgs fill;           // persistent kernel only
    int asm_walk;            // 1: the written-out walk (composite_bwd.hip::walk_batch_bwd_2blocks) where it exists; needs acc rows < 2^26
};
hipError_t launch_composite_bwd(const CompositeBwdArgs &a, int npx, int exact_cull, int waves_per_block, hipStream_t s);
hipError_t launch_bound_selftest(uint32_t *words, hipStream_t s);
// persistent reverse kernel (2 blocks per wave): `grid` waves draw the units the forward pass filed; the ticket counter must hold `grid`
hipError_t launch_composite_bwd_persistent(const CompositeBwdArgs &a, int grid, hipStream_t s);
hipError_t launch_composite_bwd_lpt(const CompositeBwdArgs &a, hipStream_t s);      // one wave per half tile, the longest first (lists by plan_units)
int composite_bwd_persistent_grid(int T, int det, int count_mode, int asm_walk);
// clears the accumulator rows the reverse pass can add into; with plan_grid > 0 its first GSR_SEG_BANDS workgroups also build the
// persistent reverse kernel's unit lists from what the forward pass left in `seg` and set the ticket counters for `plan_grid` waves
hipError_t launch_zero_marked_rows(int P, const uint8_t *touched, const uint32_t *mark, float *acc, size_t rows_total, const SegView &seg,
                                   int plan_grid, int fill_chunk, hipStream_t s);

struct PergaussBwdArgs {
    int P, D, M, W, H;
    int raw_params;
    int sh_tile;             // set by the launcher: dL/dshs rows leave through the LDS tile (M = 16, aligned, not split)
    const float *shs_rest;
    const float *rec;        // forward record (activated opacity at [5])
    float *dL_dsh_rest;
    const float *means3D, *shs, *colors_precomp, *scales, *rotations, *cov3D_precomp;
    const float *opac;       // [P] activated opacity as the forward pass used it (GeomView::opac)
    const float *viewmatrix, *projmatrix, *campos;
    float scale_modifier, tanfovx, tanfovy;
    const int *radii;
    const uint8_t *clamped;
    const float *acc;
    const uint32_t *hot;     // [P] replica codes (GeomView::hot)
    const uint8_t *touched;  // [P] GeomView::touched
    const uint32_t *touch_mark;
    int skip_unmarked;       // 1: the rows of Gaussians without a gradient have been zero-filled already (FillArgs): write nothing for them
    int dense;               // 1: pergauss_bwd_dense_kernel (needs skip_unmarked: launch_fill_zero and launch_gather_visible have run)
    uint32_t *vis_count;     // dense: number of Gaussians with a gradient (zeroed before launch_gather_visible)
    uint32_t *vis_list;      // [P] their indices, in the order the gathering workgroups arrived
    float4 *vis_rec;         // [vis_cap / 64][15][64] their inputs (pergauss_bwd.hip); entries from vis_cap on are gathered by the dense kernel itself
    uint32_t vis_cap;        // multiple of 64
    float *dL_dmeans2D, *dL_dopacity, *dL_dcolors, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drots;
};
hipError_t launch_pergauss_bwd(const PergaussBwdArgs &a, hipStream_t s);
hipError_t launch_fill_zero(const PergaussBwdArgs &a, hipStream_t s);      // zeros into every gradient output of `a`
hipError_t launch_gather_visible(const PergaussBwdArgs &a, hipStream_t s);  // vis_count / vis_list / vis_rec of `a` (vis_count zeroed by the caller)
bool pergauss_dense_eligible(const PergaussBwdArgs &a);
// the dense variant's share of the backward workspace: counter line, list, records for up to vis_cap(P) Gaussians
static inline uint32_t pergauss_vis_cap(int P) {
    const long long c = (long long)P / 4 > 131072 ? (long long)P / 4 : 131072;
    return (uint32_t)(((c < (long long)P ? c : (long long)P) + 63) / 64 * 64);
}
static inline size_t pergauss_vis_bytes(int P) {
    return 256 + (((size_t)P * 4 + 255) / 256 * 256) + (size_t)pergauss_vis_cap(P) * 15 * 16;
}

hipError_t launch_l1_ssim_forward(int C, int H, int W, const float *img, const float *gt, float lambda, float *dmaps,
                                  float *partial, float *out, hipStream_t s);
hipError_t launch_l1_ssim_backward(int C, int H, int W, const float *img, const float *gt, float lambda, const float *dmaps,
                                   const float *grad_loss, float *grad_img, hipStream_t s);

hipError_t knn_workspace_bytes(int N, size_t *bytes);
hipError_t launch_knn(int N, const float *pts, float *out, void *ws, hipStream_t s);

hipError_t launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present, hipStream_t s);

}  // namespace gsr
