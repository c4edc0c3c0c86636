/*
 * gsr.h -- C ABI of the MI355X-native differentiable Gaussian rasterizer (libgsr_hip.so).
 *
 * This is the drop-in boundary for the one hot path of stu214634/gaussian-transformer:
 * the native extension behind `diff_gaussian_rasterization.GaussianRasterizer`, which the
 * reference calls at gaussian_renderer/__init__.py:14,36-49,51,85-93.  The reference's own
 * native layer (submodules/diff-gaussian-rasterization, .gitmodules:4-6) is an empty directory
 * in the snapshot; the entry points below are what its Python front-end binds
 * (SURVEY.md 2b "pybind/torch glue", 8b "C ABI to export"):
 *
 *   gsr_forward       <->  _C.rasterize_gaussians           (RasterizeGaussiansCUDA)
 *   gsr_backward      <->  _C.rasterize_gaussians_backward  (RasterizeGaussiansBackwardCUDA)
 *   gsr_mark_visible  <->  _C.mark_visible                  (markVisible; unused by this reference)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  Every pointer is a DEVICE pointer
 *     on the current HIP device unless marked (host).  NULL means "not provided".
 *   - all floating point data is float32; the caller owns every buffer (PyTorch's caching
 *     allocator in the Python host); the library never frees or retains pointers across calls.
 *   - work is enqueued on `stream` (torch's current stream).  gsr_forward waits ONCE, in the middle of the call, for the device to
 *     have counted the (Gaussian,tile) pairs (the binning workspace is sized from that count): it polls a pinned host word the
 *     counting kernel writes -- no stream synchronisation, the kernels queued behind keep running -- and falls back to
 *     hipStreamSynchronize if nothing arrives within 2 s.  gsr_backward never waits unless debug != 0.
 *   - return value: 0 on success, a GSR_ERR_* code otherwise; gsr_last_error() gives the text
 *     (thread-local).  The library never aborts the process and leaves the device usable, because
 *     callers catch RuntimeError and continue (train_stacked_transformer.py:392-398).
 *   - matrices use the reference's memory layout (scene/cameras.py:54-57): the 16 floats of
 *     world_view_transform / full_proj_transform as stored, i.e. x' = m[0]x + m[4]y + m[8]z + m[12].
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 2

enum {
    GSR_OK = 0,
    GSR_ERR_INVALID_ARGUMENT = 1, /* bad sizes / missing or contradictory inputs            */
    GSR_ERR_ALLOC = 2,            /* the binning allocation callback returned NULL           */
    GSR_ERR_HIP = 3,              /* a HIP / rocPRIM call failed; text in gsr_last_error()   */
    GSR_ERR_WORKSPACE = 4         /* a caller-provided workspace is too small                */
};

typedef void *gsr_stream_t; /* hipStream_t */

/* Allocation callback for the binning workspace, whose size is only known after the
 * per-Gaussian stage has run (upstream: the resize-callback of its BinningState).  Must return
 * a device pointer to at least `bytes` bytes, 256-byte aligned, valid until the matching
 * gsr_backward has been enqueued, or NULL on failure. */
typedef void *(*gsr_alloc_fn)(void *user, size_t bytes);

int32_t gsr_abi_version(void);
const char *gsr_last_error(void);

/* Sizes (bytes) of the caller-allocated workspaces:
 *   geom_bytes : per-Gaussian state written by gsr_forward, read by gsr_backward.  Its contents need no initialisation, and
 *                gsr_backward must be given the workspace of the gsr_forward call it belongs to, unmodified: besides the
 *                projected records it carries which Gaussians the forward pass composited at all (the others only get
 *                zero gradients written) and which of them use replica accumulator rows
 *   img_bytes  : per-tile ranges + per-pixel final transmittance / last contributor
 *   bwd_bytes  : scratch used only inside gsr_backward (gradient accumulators: one 64-byte row per Gaussian plus
 *                P/32 + 1024 replica rows; behind them the dense per-Gaussian stage's list and record buffer, 4 P + 240 max(131072, P/4)
 *                bytes; no initialisation needed, gsr_backward clears what it uses) */
int32_t gsr_workspace_sizes(int32_t P, int32_t W, int32_t H, size_t *geom_bytes, size_t *img_bytes, size_t *bwd_bytes);

/* Size of gsr_backward's scratch workspace for a forward that rendered R pairs: the bwd_bytes of gsr_workspace_sizes,
 * plus, while the option "deterministic_bwd" is on, one 64-byte slot per (pair, wave of the tile).  A workspace that only covers the
 * accumulators is accepted: gsr_backward then keeps the streaming per-Gaussian kernel. */
int32_t gsr_backward_workspace_bytes(int32_t P, int64_t R, size_t *bytes);

/* Size of the binning workspace of the SORT path for N (Gaussian,tile) pairs (informational).  gsr_forward asks the
 * allocator callback for exactly what the path it takes needs: the default tile-list path wants 8 N + 16 E bytes plus
 * small tables, E = (Gaussian, super-tile) entries, which only the device knows. */
int32_t gsr_binning_bytes(int64_t N, int32_t W, int32_t H, size_t *bytes);

/* Forward: per-Gaussian projection + SH (S1-S6); per-tile depth-ordered lists (S7-S8: by default (Gaussian, super-tile) entries binned
 * per super-tile and ordered in LDS, csrc/supertile_sort.hip -- no global sort; round 1's depth order + tile lists or a rocPRIM radix
 * sort of the pairs when a frame does not fit that path or an option asks for it: the same lists); front-to-back compositing (S9).
 *   P Gaussians, D active SH degree (0..3), M stored SH coefficients per channel,
 *   W x H image.  Exactly one of shs / colors_precomp and exactly one of
 *   (scales, rotations) / cov3D_precomp must be non-NULL.
 *   out_color [3,H,W], radii [P] int32 are fully written.  *num_rendered (host) receives the
 *   number of (Gaussian,tile) pairs.  P == 0 writes a zero image (not background). */
int32_t gsr_forward(gsr_stream_t stream, int32_t P, int32_t D, int32_t M, int32_t W, int32_t H,
                    const float *bg /*[3]*/, const float *means3D /*[P,3]*/, const float *shs /*[P,M,3]*/,
                    const float *colors_precomp /*[P,3]*/, const float *opacities /*[P]*/,
                    const float *scales /*[P,3]*/, float scale_modifier, const float *rotations /*[P,4]*/,
                    const float *cov3D_precomp /*[P,6]*/, const float *viewmatrix /*[16]*/,
                    const float *projmatrix /*[16]*/, const float *campos /*[3]*/, float tanfovx, float tanfovy,
                    int32_t prefiltered, int32_t debug, float *out_color /*[3,H,W]*/, int32_t *radii /*[P]*/,
                    void *geom_ws, size_t geom_bytes, gsr_alloc_fn binning_alloc, void *binning_user,
                    void *img_ws, size_t img_bytes, int64_t *num_rendered /*host*/,
                    /* fused-step extension, pass NULL / 0 for the reference's behaviour: */
                    const float *shs_rest /* non-NULL: `shs` is features_dc [P,1,3], this is features_rest [P,M-1,3]
                                             (scene/gaussian_model.py:108-111 without the torch.cat) */,
                    int32_t raw_params /* 1: opacities are logits, scales log-scales, rotations un-normalised; the
                                          activations of scene/gaussian_model.py:33-41 are applied in-kernel */);

/* Backward: reverse compositing (S10) + per-Gaussian chain (S11-S13).
 *   R = num_rendered of the matching forward; geom_ws / binning_ws / img_ws as the forward left
 *   them (read-only here); radii as returned by the forward.
 *   Every gradient buffer is fully written (no pre-zeroing needed):
 *     dL_dmeans2D [P,3] (x,y = gradient w.r.t. NDC coordinates, z = 0), dL_dopacity [P],
 *     dL_dcolors [P,3] (may be NULL when shs is given: nobody reads it then), dL_dmeans3D [P,3],
 *     dL_dcov3D [P,6] (may be NULL when scales / rotations are given),
 *     dL_dsh [P,M,3] (NULL iff shs NULL), dL_dscales [P,3] / dL_drots [P,4] (NULL iff scales NULL). */
int32_t gsr_backward(gsr_stream_t stream, int32_t P, int32_t D, int32_t M, int64_t R, int32_t W, int32_t H,
                     const float *bg, const float *means3D, const int32_t *radii, const float *shs,
                     const float *colors_precomp, const float *scales, float scale_modifier,
                     const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                     const float *projmatrix, const float *campos, float tanfovx, float tanfovy,
                     const float *dL_dpix /*[3,H,W]*/, const void *geom_ws, size_t geom_bytes,
                     const void *binning_ws, size_t binning_bytes, const void *img_ws, size_t img_bytes,
                     void *bwd_ws, size_t bwd_bytes, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors,
                     float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drots,
                     int32_t debug,
                     /* fused-step extension: same meaning as in gsr_forward.  With shs_rest, dL_dsh is [P,1,3] and
                      * dL_dsh_rest [P,M-1,3]; with raw_params, dL_dopacity / dL_dscales / dL_drots are gradients
                      * w.r.t. the raw (pre-activation) parameters. */
                     const float *shs_rest, int32_t raw_params, float *dL_dsh_rest);

/* Announces the gradient outputs of the gsr_backward call that will follow the NEXT gsr_forward on the current device (same P, M;
 * pointers in gsr_backward's order, NULL where that call will pass NULL).  That gsr_forward then writes their zeros on the library's
 * second stream beside its own last kernels -- where CUs idle --, and the gsr_backward that names exactly these pointers skips its
 * own zero-fill.  The fill is ordered before that gsr_backward's writes, and before everything the next gsr_forward or gsr_backward
 * call on the device enqueues, whichever comes first; until one of them has been called the caller keeps the buffers allocated and
 * does not touch them (the fill may still be running when gsr_forward returns).  Purely optional and purely a matter of time: without it, with other pointers in
 * gsr_backward, with a second gsr_forward before that gsr_backward, or where gsr_backward would not zero-fill on the second stream
 * in the first place (see "dense_pergauss"), gsr_backward fills by itself as before.  An announcement is used by one gsr_forward
 * only; P <= 0 withdraws it.  Host-side state, no device work.  Extension: the reference allocates its gradients with torch::zeros
 * inside RasterizeGaussiansBackwardCUDA, i.e. pays the same fill on the critical path. */
int32_t gsr_backward_prefill(int32_t P, int32_t M, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors, float *dL_dmeans3D,
                             float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drots, float *dL_dsh_rest);

/* present[i] = 1 iff Gaussian i passes the near-plane test (view z > 0.2). */
int32_t gsr_mark_visible(gsr_stream_t stream, int32_t P, const float *means3D, const float *viewmatrix,
                         const float *projmatrix, uint8_t *present /*[P]*/);

/* out[i] = 1 iff Gaussian i was composited by the gsr_forward call that filled geom_ws (some wave staged it with a pixel block it
 * can reach), else 0.  Every Gaussian that can receive a non-zero gradient from gsr_backward is among the marked ones (a
 * superset: stale workspace bytes may mark a few more), so a data-parallel trainer needs to exchange only the gradient rows of
 * the union of the ranks' masks -- 9 % of the Gaussians for one camera of BASELINE config 3, against 100 % with radii > 0.
 * Extension: the reference has no counterpart.  Asynchronous on `stream`. */
int32_t gsr_composited_mask(gsr_stream_t stream, int32_t P, const void *geom_ws, size_t geom_bytes, uint8_t *out /*[P]*/);

/* Introspection for tests and the bench (device -> host copies, synchronising):
 * copies stage outputs out of the opaque workspaces.  Any destination may be NULL. */
int32_t gsr_debug_read_geom(gsr_stream_t stream, int32_t P, const void *geom_ws, float *depth /*[P]*/,
                            float *xy /*[P,2]*/, float *conic_opacity /*[P,4]*/, float *rgb /*[P,3]*/,
                            uint32_t *tiles_touched /*[P]*/, uint8_t *clamped /*[P,3]*/);
int32_t gsr_debug_read_binning(gsr_stream_t stream, int64_t N, int32_t W, int32_t H, const void *binning_ws,
                               const void *img_ws, uint64_t *keys_sorted /*[N]*/, uint32_t *point_list /*[N]*/,
                               uint32_t *ranges /*[T,2]*/);
int32_t gsr_debug_read_image_state(gsr_stream_t stream, int32_t W, int32_t H, const void *img_ws,
                                   float *final_T /*[H,W]*/, uint32_t *n_contrib /*[H,W]*/);

/* Capacity asserts of the debug build (libgsr_hip_dbg.so, compiled with -DGSR_DEBUG_BOUNDS: every index whose bound is a plan made on the
 * host or by another kernel is checked on the device; a violation is recorded and the access skipped instead of faulting).
 * out[0..3] = first violation in the list-building kernels (code, index, capacity, number of violations), out[4..7] = the same for the
 * compositing / planning kernels, out[8] = 1 in a debug build, out[9..11] = the mechanism's self test: (99, 5, 4) in a debug build,
 * (0xdead, 0, 0) in the product build, where the checks are compiled out.  Synchronises. */
int32_t gsr_debug_read_bound_errors(gsr_stream_t stream, int32_t P, const void *geom_ws, int32_t W, int32_t H, const void *img_ws,
                                    uint32_t *out /*[12] host*/);

/* What the forward pass left for the segmented reverse pass and what the last gsr_backward made of it (tests): summary[0] = entries
 * per segment the forward pass used (0: none), [1] = work units of the last gsr_backward's lists, [2] = checkpoint slots handed
 * out, [3] = slots in the pool, [4] = half tiles, [5] = tickets drawn by the last persistent reverse kernel.  Synchronises. */
int32_t gsr_debug_read_segments(gsr_stream_t stream, int32_t W, int32_t H, const void *img_ws, uint32_t *summary /*[8] host*/);

/* Lane-slot accounting of the two compositing kernels (pair evaluations against the FP32 vector peak).
 * After gsr_set_option("count_lanes", 1) every gsr_forward / gsr_backward on the current device runs instrumented
 * kernels (slower) that accumulate, per direction, 16 words:
 *   [0] list entries staged  [1] splat visits (a wave evaluates a splat)  [2] 8x8 block visits (64 lane slots each)
 *   [3] lanes that blended   [4] lanes on a pixel already finished / stopped earlier  [5] lanes failing the alpha tests
 *   [6] cross-lane reductions + atomics issued (reverse only)  [7] block visits in which no lane blended  [8] waves
 * This call synchronises the device, copies the words out (either pointer may be NULL) and resets them. */
int32_t gsr_debug_read_lane_counters(uint64_t *fwd /*[16] host*/, uint64_t *bwd /*[16] host*/);
/* Wave timeline of the last instrumented compositing launch (which: 0 forward, 1 reverse): four words per work unit
 * (unit = tile * waves-per-tile + wave): start and end of the wave in 10 ns ticks of the device's constant-rate clock (low
 * 32 bits), list entries staged, splat visits.  Synchronises.  Debug facility for load-balance studies (scripts/wave_trace.py). */
int32_t gsr_debug_read_wave_trace(int32_t which, uint32_t *out /*[4 * max_units] host*/, int64_t max_units);

/* Tuning knobs (process-wide).  Known options:
 *   "exact_tile_cull" (default 1): emit a (Gaussian,tile) pair only if the ellipse
 *        {alpha >= 1/255} can reach a pixel of the tile, instead of every tile of upstream's
 *        3-sigma bounding square.  Output-invariant (dropped pairs are skipped by every pixel's
 *        alpha test anyway); num_rendered and the internal lists shrink.  0 = upstream's rule.
 *   "two_level_sort" (default 1): put the Gaussians in (depth bits, id) order first, emit the pairs in that
 *        order and finish with a STABLE radix sort on the tile id only; identical resulting order to
 *        0 = one global radix sort on tile<<32|depth.  Speed only.
 *   "tile_lists" (0, 1 or 2; default 2): how the per-tile lists are built.  2 = csrc/supertile_sort.hip: (Gaussian, 64 x 64 px
 *        super-tile) entries binned per super-tile, every bin ordered by (depth, id) in LDS and expanded into its 16 tile lists --
 *        no global sort (needs two_level_sort = 1, <= 8192 super-tiles, a bin of <= 14336 entries and <= 4 entries per Gaussian on
 *        average; a frame that does not fit takes 1).  1 = round 1's path: bucketed global depth order (csrc/depth_order.hip) +
 *        csrc/tile_lists.hip (<= 512 super-tiles of 128 x 128 px, else 0).  0 = key emission + rocPRIM radix sort + range detection.
 *        Same per-tile lists in every case; with 1 and 2 point_list is laid out super-tile-major (ranges[] say where each tile's
 *        slice is).  Speed only.
 *   "depth_log_map" (default 0, set by the library itself): the depth buckets are cut linearly in depth (0) or in the
 *        depth's float bits, i.e. logarithmically, with 4x the buckets (1).  The library switches to 1 after a frame
 *        overflowed a bucket (far outliers); exposed for tests.
 *   "depth_buckets" (0, 1 or 2; default 1): how the Gaussians are put in depth order.  0 = rocPRIM radix sort
 *        + scan; 1 = the bucketed depth order of csrc/depth_order.hip when P >= 1024 (falls back to 0 by itself
 *        when a depth bucket does not fit in LDS); 2 = bucketed for every P (tests).  Same order.  Speed only.
 *   "composite_waves_per_block" (1, 2 or 4; default 1): wave64s per workgroup of the compositing
 *        kernels.  The waves never synchronise, so 1 lets every wave retire (and be replaced) alone.
 *   "fwd_blocks_per_wave", "bwd_blocks_per_wave" (1, 2 or 4; default 2): 8x8 pixel blocks one
 *        wave64 of the forward / reverse compositing kernel owns (4 = a whole 16x16 tile).  Speed only.
 *   "count_lanes" (0, 1 or 2; default 0): instrumented compositing kernels: 1 = lane-slot accounting (gsr_debug_read_lane_counters;
 *        several times slower), 2 = wave timeline only (gsr_debug_read_wave_trace; normal speed).
 *   "persistent_bwd" (0, 1 or 2; default 2): the reverse compositing kernel as persistent waves drawing work units -- (half tile, list
 *        segment) -- longest first from per-XCD lists (csrc/composite_bwd.hip): 0 never, 1 always, 2 on images of at most 6144 tiles,
 *        whose half tiles fill the chip less than 1.5 times and whose longest list, not the throughput, sets the kernel's time.
 *        Needs bwd_blocks_per_wave = 2.  Speed only.
 *   "segment_entries" (0 or a multiple of 64; default 256): while the persistent reverse kernel is in use, the forward pass records a
 *        checkpoint (transmittance so far, colour behind) per pixel every so many list entries, so that the reverse pass of a long
 *        list is split into pieces that different waves work off.  0 = whole half tiles.  Needs fwd_blocks_per_wave = 2.
 *        Gradients equal the unsegmented ones up to the order of float additions.
 *   "fill_in_tail" (default 0): let the persistent reverse kernel's idle waves write the zero gradient rows of Gaussians without a
 *        gradient (measured slower; kept for experiments).
 *   "asm_walk" (default 1): the innermost loop of both compositing kernels (the walk over a staged batch of splats) in hand-written
 *        gfx950 assembly; 0 = the C++ walks.  Same images bit for bit; gradients equal up to the order of float additions.  Speed only.
 *   "composite_lds_pad" (bytes, default 0): extra dynamic LDS per compositing workgroup (occupancy experiments only).
 *   "poll_timeouts" (read-only through gsr_get_option): read-backs of the pair count whose pinned-word poll timed out on this device.
 *   "fwd_pair_long" (default -1 = 64; 0 = off): on images of at most 6144 tiles (where "persistent_bwd" = 2 applies) the forward
 *        compositing kernel walks a half tile whose list has more entries than this with a workgroup of two waves, one per 8x8 block,
 *        instead of one wave over both blocks: the kernel's time there is its longest list walked alone, and a one-block visit is
 *        fewer instructions.  Same images bit for bit.  Needs fwd_blocks_per_wave = 2 and "asm_walk".  Speed only.
 *   "bwd_lpt" (default 1): on images of more than 6144 tiles the reverse compositing kernel takes its half tiles in order of decreasing
 *        length (the forward pass files how far each half tile's pixels got, the accumulator-clearing kernel sorts) instead of tile order:
 *        the kernel's last waves then have short lists.  Same gradients up to the order of float additions.  Speed only.
 *   "dense_pergauss" (0, 1, 2; default 2 = from 500 000 Gaussians): gsr_backward forks a second stream of the library's own (lowest
 *        priority, one per device, created on first use) on which the zeros of every gradient output are written and the Gaussians
 *        with a gradient are listed and their inputs copied into a compact buffer while the compositing kernel runs; the per-Gaussian
 *        kernel then runs on those only.  Applies to shs with M = 16 and scales + rotations (other layouts: the streaming kernel).
 *        The caller's stream sees one event record and one event wait; the join is enqueued before gsr_backward returns.  Same
 *        gradients (bit for bit under "deterministic_bwd").  Speed only.
 *   "prefill_at" (0, 1, 2; default 1): where a gsr_forward zero-fills gradient outputs announced through gsr_backward_prefill --
 *   1 beside its compositing kernel, 2 beside the list-ordering kernel already, 0 announcements are ignored.
 *   "dense_fork" (0, 1, 2; default 2): where that fork happens -- 1 after the accumulator rows are cleared, 0 before, 2 = after below
 *        2 000 000 Gaussians.
 *   Options that change what the forward pass leaves for the reverse pass ("persistent_bwd", "segment_entries", the blocks-per-wave
 *   settings) must not be changed between a gsr_forward and the gsr_backward that belongs to it.
 *   "deterministic_bwd" (default 0): the reverse compositing pass stores the partial gradients of every (wave, pair)
 *        into a slot of its own and a second kernel adds each Gaussian's slots in a fixed order, instead of float
 *        atomics whose order differs from run to run: bitwise reproducible gradients (race detection).
 *        Needs the larger scratch of gsr_backward_workspace_bytes; slower (debug mode).  While it is on the forward pass takes no
 *        checkpoints ("segment_entries" reads as 0): which half tiles get one when the pool runs out is a race.
 * Adaptive state (which depth-bucket map a device uses after it met depth outliers, "depth_log_map") is kept per
 * DEVICE, not per process; gsr_set_option("depth_log_map", v) sets it for the current device. */
int32_t gsr_set_option(const char *name, int32_t value);
int32_t gsr_get_option(const char *name, int32_t *value);

/* Per-stage timing of the last profiled gsr_forward / gsr_backward on the CURRENT DEVICE, milliseconds,
 * measured with hipEvents on `stream` when profiling was enabled by gsr_set_profiling(1).
 * names: array of GSR_NUM_STAGES const char*; ms: array of GSR_NUM_STAGES floats (host). */
#define GSR_NUM_STAGES 13
int32_t gsr_set_profiling(int32_t enable);
int32_t gsr_get_stage_times(const char **names, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
