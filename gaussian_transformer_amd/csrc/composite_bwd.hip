// composite_bwd.hip -- reverse compositing (S10): per pixel back-to-front over the tile's
// depth-sorted splat list, producing dL/d{rgb, mean2D, conic, opacity} per Gaussian.
//
// CDNA4 shape.  A wave64 owns NPX 8x8 pixel blocks of one 16x16 tile (NPX = 4: the whole tile,
// 2: its upper or lower half, 1: one quadrant); lane l holds pixel l of each block, so per-splat
// work that does not depend on the pixel (LDS record read, loop control, the cross-lane gradient
// reduction, the atomic) is paid once per NPX*64 pixels.  Waves never synchronise with each other
// (no workgroup barrier): each stages the tile's splat list 64 records at a time into a
// wave-private LDS slice, one record gathered per lane.  While staging, the lane also reads, per
// 8x8 block, whether the splat can reach alpha >= 1/255 anywhere in the block -- one byte per block
// and list entry that the forward pass wrote when it staged the same entry (exact min of the
// quadratic form over the block rectangle against the culling threshold of gsr_device.h); the
// wave then walks only the set bits of the 64-bit ballot, so dead splats cost two scalar
// instructions, and dead blocks of a live splat are skipped by a scalar branch.
// All lanes visit the same splat at the same step.  The nine partial gradients are reduced through
// LDS (the VALU is the bottleneck of this kernel, the LDS pipe is idle): every lane stores its nine
// values as rows of 68 floats (bank-conflict-free for the readers), 36 lanes each add 16 of them with
// four ds_read_b128, two DPP steps finish, and the nine totals leave as ONE global_atomic_add_f32
// instruction (9 lanes) into the splat's 64-byte accumulator row: one memory-side request per (wave, splat).
#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {
extern int g_composite_lds_pad;

#define LOG2E 1.4426950408889634f
#define RED_STRIDE 68
// LDS per wave: 64 staged records of 40 bytes, kept as three arrays (float4 | float4 | float2) so that every read stays aligned,
// + the reduction scratch, 9 rows of RED_STRIDE floats (153 float4): 5 008 B -- 32 waves fit a CU's 160 KB, 8 per SIMD (the
// kernel needs 53 VGPRs and is latency-sensitive: 48-byte records, 29 waves/CU, cost 4 %; 22 waves/CU cost another 9 %)
#define BWD_LDS_F4 (64 * 2 + 32 + (9 * RED_STRIDE + 3) / 4)

__device__ __forceinline__ int xcd_band_unit(int b, int nblocks_padded) {
    const int chunk = nblocks_padded >> 3;
    return (b & 7) * chunk + (b >> 3);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(moved);
}
// sum over the 16 lanes of each DPP row, result in every lane of the row
__device__ __forceinline__ float row_allreduce(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}
// lanes 0-31: a[l] + a[l+32];  lanes 32-63: b[l-32] + b[l]
__device__ __forceinline__ float fold32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// row0: a.r0 + a.r1;  row1: b.r0 + b.r1;  row2: a.r2 + a.r3;  row3: b.r2 + b.r3
__device__ __forceinline__ float fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// COUNT: instrumented instantiation (gsr_set_option("count_lanes", 1)): tallies staged records, splat visits, 8x8 block
// visits (= 64 lane slots each), blending lanes and the reason idle lanes were idle into a.counters (CompositeCounters)
// DET: deterministic mode (gsr_set_option("deterministic_bwd", 1)): instead of the float atomics, whose arrival order
// differs from run to run, every (wave, list entry) stores its nine sums into its own slot of a.det and
// det_reduce_kernel below adds each Gaussian's slots in a fixed order.  The in-wave reduction is order-fixed already.
template <int NPX, bool COUNT, bool DET>
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompositeBwdArgs a, int nblocks_padded, int /*exact_cull*/) {
    constexpr int UNITS_PER_TILE = 4 / NPX;          // waves per tile
    extern __shared__ __align__(16) float4 stage_dyn[];     // per wave: 64 records x 3 float4, then 9 x RED_STRIDE floats of reduction scratch
    const int T = a.gridx * a.gridy;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int unit = xcd_band_unit(blockIdx.x, nblocks_padded) * wpb + wave;
    const int tile = unit / UNITS_PER_TILE, sub = unit % UNITS_PER_TILE;
    if (tile >= T) return;                            // wave-uniform
    const int tx = tile % a.gridx, ty = tile / a.gridx;
    float4 *my = stage_dyn + wave * BWD_LDS_F4;
    float2 *myc = reinterpret_cast<float2 *>(my + 128);     // third array of the staged records
    float *red = reinterpret_cast<float *>(my + 128 + 32);  // [value 0..8][lane 0..63]
    const float4 *rec4 = reinterpret_cast<const float4 *>(a.rec);
    const uint2 range = a.ranges[tile];
    const size_t HW = (size_t)a.W * a.H;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

    // per-pixel state, one pixel per 8x8 block q:  Tr = transmittance behind the splats visited so far,
    // accd = <colour composited behind them (normalised by Tr), d>, d = dL/dpixel, tb = T_final * <bg, d>
    float fx[NPX], fy[NPX], Tr[NPX], accd[NPX];
    float d0[NPX], d1[NPX], d2[NPX], tb[NPX];
    int last[NPX];
    int max_last = 0;
#pragma unroll
    for (int q = 0; q < NPX; q++) {
        const int blk = NPX == 4 ? q : (NPX == 2 ? sub * 2 + q : sub);      // 0..3: (bx = blk&1, by = blk>>1)
        const int x0 = tx * GSR_TILE + (blk & 1) * 8, y0 = ty * GSR_TILE + (blk >> 1) * 8;
        const int x = x0 + (lane & 7), y = y0 + (lane >> 3);
        const bool inside = x < a.W && y < a.H;
        const size_t pix = (size_t)(inside ? y : 0) * a.W + (inside ? x : 0);
        fx[q] = (float)x; fy[q] = (float)y;
        const float Tf = inside ? a.final_T[pix] : 1.f;
        last[q] = inside ? (int)a.n_contrib[pix] : 0;
        d0[q] = inside ? a.dL_dpix[pix] : 0.f;
        d1[q] = inside ? a.dL_dpix[HW + pix] : 0.f;
        d2[q] = inside ? a.dL_dpix[2 * HW + pix] : 0.f;
        tb[q] = Tf * (bg0 * d0[q] + bg1 * d1[q] + bg2 * d2[q]);
        Tr[q] = Tf; accd[q] = 0.f;
        max_last = max(max_last, last[q]);
    }
    int blk_last[NPX];                                // last contributor over the 64 pixels of block q (wave-uniform)
#pragma unroll
    for (int q = 0; q < NPX; q++) {
        int m = last[q];
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) m = max(m, __shfl_xor(m, sft));
        blk_last[q] = __builtin_amdgcn_readfirstlane(m);
        max_last = max(max_last, blk_last[q]);
    }
    max_last = __builtin_amdgcn_readfirstlane(max_last);
    if (max_last == 0) return;                        // wave-uniform

    // Cross-lane reduction through LDS (the LDS pipe is idle otherwise, the VALU is the bottleneck):
    // every lane stores its 9 partial sums as red[value][lane]; lane 4*value + part then adds the 16
    // floats red[value][16*part .. 16*part+15] (4 x ds_read_b128), two DPP steps fold the 4 parts, and
    // lanes 0,4,...,32 hold the nine totals: ONE 9-lane global_atomic_add_f32 per (wave, splat).
    const int rvalue = lane >> 2, rpart = lane & 3;
    // rows of RED_STRIDE = 68 floats: with 64 the nine rows start in the same LDS bank and the 36 reading lanes collide 9-way
    const float4 *red_rd = reinterpret_cast<const float4 *>(red + (rvalue < 9 ? rvalue : 0) * RED_STRIDE + rpart * 16);
    const bool red_lane = rvalue < 9;
    const int slot = (red_lane && rpart == 0) ? rvalue : -1;

    unsigned long long c_staged = 0, c_visits = 0, c_blocks = 0, c_ok = 0, c_past = 0, c_alpha = 0, c_red = 0, c_dead = 0;
    for (int base = ((max_last - 1) >> 6) << 6; base >= 0; base -= 64) {
        const int cnt = min(64, max_last - base);
        __builtin_amdgcn_wave_barrier();
        bool live = false;
        if (lane < cnt) {
            const uint32_t g = a.point_list[range.x + base + lane];
            const float4 r0 = rec4[3 * (size_t)g], r1 = rec4[3 * (size_t)g + 1], r2 = rec4[3 * (size_t)g + 2];
            // blocks this entry can reach (exact ellipse-vs-block test, gsr_device.h), as the forward pass staged them
            uint32_t bits = 0u;
#pragma unroll
            for (int q = 0; q < NPX; q++) {
                const int blk = NPX == 4 ? q : (NPX == 2 ? sub * 2 + q : sub);
                bits |= a.contrib[(size_t)blk * a.contrib_stride + range.x + base + lane] ? (1u << q) : 0u;
            }
            live = bits != 0u;
            const StagedConic sc = stage_conic(r0.z, r0.w, r1.x);       // as the forward pass staged it
            my[lane] = make_float4(r0.x, r0.y, sc.a, sc.b);
            my[64 + lane] = make_float4(sc.c, r1.y, r1.z, r1.w);
            // accumulator row: the Gaussian's own, or -- for a splat with replica rows (gsr_internal.h) -- replica (tile mod K)
            const uint32_t hot = __float_as_uint(r2.w);
            const uint32_t row = hot ? (uint32_t)a.P + (hot >> 4) + ((uint32_t)tile & ((1u << (hot & 15u)) - 1u)) : g;
            myc[lane] = make_float2(r2.x, __uint_as_float(bits | (row << 4)));      // row < 2^28 (checked by gsr_backward)
        }
        uint64_t todo = __ballot(live);
        if (COUNT) { c_staged += cnt; c_visits += __builtin_popcountll(todo); }
        __builtin_amdgcn_wave_barrier();
        // The splats of the batch are visited back to front.  (Issuing the next record's LDS reads a visit ahead was measured:
        // no gain here, 10 % slower in the forward kernel -- the waves of a SIMD already cover that latency for each other.)
        auto visit = [&](const float4 r0, const float4 r1, const float4 r2, const int j) __attribute__((always_inline)) {
            const uint32_t bits = __builtin_amdgcn_readfirstlane(__float_as_uint(r2.z));
            // Per-lane partial sums over this lane's pixels: v0..v2 = sum w * dL/dpixel (colour gradient),
            // v8 = sum s, v3,v4 = sum s*dx, s*dy, v5..v7 = sum s*dx*dx, s*dx*dy, s*dy*dy with s = opacity * G * dL/dalpha
            // (the pre-cap alpha times dL/dalpha: the 0.99 cap is straight-through, S10).  Everything that is constant
            // per Gaussian (conic, W/2, H/2, -1/2, 1/opacity for dL/dopacity) is applied once in pergauss_bwd.hip.
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
            const int pos = base + j;
            unsigned long long any_ok = 0ull;         // 64-bit lane mask kept in SGPRs
            const StagedConic kc = {r0.z, r0.w, r1.x};
            // body for one 8x8 block
            auto block_body = [&](int q) __attribute__((always_inline)) {
                const float dx = r0.x - fx[q], dy = r0.y - fy[q];
                float araw;                                          // same expression, same bits as the forward pass
                const unsigned long long okm = splat_alpha(splat_power_log2(kc, dx, dy), r1.y, araw) &
                                               __builtin_amdgcn_ballot_w64(pos < last[q]);
                any_ok |= okm;
                if (COUNT) {
                    const unsigned long long mp = __ballot(!(pos < last[q]));
                    c_blocks += 1; c_ok += __builtin_popcountll(okm); c_past += __builtin_popcountll(mp);
                    c_alpha += __builtin_popcountll(~okm & ~mp); c_dead += okm == 0ull;
                }
#ifdef GSR_BWD_EXEC_MASK
                const float alpha = fminf(GSR_ALPHA_MAX, araw);
                if (__builtin_amdgcn_inverse_ballot_w64(okm)) {      // the other lanes are switched off (EXEC), not multiplied by zero
                    const float araw_ok = araw;
#else
                {   // lanes that do not blend carry alpha = 0 through the same arithmetic: every sum below gets an exact 0
                    const float araw_ok = __builtin_amdgcn_inverse_ballot_w64(okm) ? araw : 0.f;   // exp2 of a skipped lane may be inf
                    const float alpha = fminf(GSR_ALPHA_MAX, araw_ok);
#endif
                    const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
                    const float Tk = Tr[q] * inv;                    // transmittance in front of this splat
                    // <colour - colour behind, dL/dpixel>: the colour behind is only ever needed through this inner
                    // product, so the recurrence runs on accd = <colour behind, dL/dpixel> (one register, not three)
                    const float dot = (r1.z * d0[q] + r1.w * d1[q] + r2.x * d2[q]) - accd[q];
                    const float dL_dalpha = (dot * Tr[q] - tb[q]) * inv;
                    accd[q] += alpha * dot;                          // behind the next (nearer) splat
                    Tr[q] = Tk;
                    const float w = alpha * Tk;
                    v0 += w * d0[q]; v1 += w * d1[q]; v2 += w * d2[q];
                    const float sg = araw_ok * dL_dalpha;            // pre-cap alpha: the cap is straight-through (S10)
                    v8 += sg;
                    const float sx = sg * dx, sy = sg * dy;
                    v3 += sx; v4 += sy;
                    v5 += sx * dx; v6 += sx * dy; v7 += sy * dy;
                }
            };
            uint32_t todo_bits = bits;                // drop blocks whose pixels all stopped before this splat
#pragma unroll
            for (int q = 0; q < NPX; q++)
                if (pos >= blk_last[q]) todo_bits &= ~(1u << q);
#pragma unroll
            for (int q = 0; q < NPX; q++)
                if (todo_bits & (1u << q)) block_body(q);    // scalar branch per block (a fused all-blocks body: 7 % slower)
            if (any_ok == 0ull) return;                // wave-uniform: no pixel of this wave blends the splat
            if (COUNT) c_red += 1;
            red[0 * RED_STRIDE + lane] = v0; red[1 * RED_STRIDE + lane] = v1; red[2 * RED_STRIDE + lane] = v2;
            red[3 * RED_STRIDE + lane] = v3; red[4 * RED_STRIDE + lane] = v4; red[5 * RED_STRIDE + lane] = v5;
            red[6 * RED_STRIDE + lane] = v6; red[7 * RED_STRIDE + lane] = v7; red[8 * RED_STRIDE + lane] = v8;
            __builtin_amdgcn_wave_barrier();
            float sel = 0.f;
            if (red_lane) {
                const float4 p0 = red_rd[0], p1 = red_rd[1], p2 = red_rd[2], p3 = red_rd[3];       // (v_pk_add_f32 pairs here: 2 % slower)
                sel = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w)) +
                      (((p2.x + p2.y) + (p2.z + p2.w)) + ((p3.x + p3.y) + (p3.z + p3.w)));
            }
            __builtin_amdgcn_wave_barrier();
            sel = dpp_add<0xB1>(sel);   // quad_perm [1,0,3,2]
            sel = dpp_add<0x4E>(sel);   // quad_perm [2,3,0,1]
            if (slot >= 0) {
                if (DET) {
                    a.det[((size_t)(range.x + pos) * UNITS_PER_TILE + sub) * GSR_ACC_FLOATS + slot] = sel;
                } else {
                    const uint32_t row = __float_as_uint(r2.w);
                    atomicAdd(a.acc + GSR_ACC_FLOATS * (size_t)row + slot, sel);
                }
            }
        };
        while (todo) {
            const int j = 63 - __builtin_clzll(todo);
            todo &= ~(1ull << j);
            const float2 cj = myc[j];
            visit(my[j], my[64 + j], make_float4(cj.x, 0.f, __uint_as_float(__float_as_uint(cj.y) & 15u), __uint_as_float(__float_as_uint(cj.y) >> 4)), j);
        }
    }
    if (COUNT && lane == 0 && a.counters) {
        atomicAdd(&a.counters->staged, c_staged); atomicAdd(&a.counters->visits, c_visits);
        atomicAdd(&a.counters->block_visits, c_blocks); atomicAdd(&a.counters->lanes_ok, c_ok);
        atomicAdd(&a.counters->lanes_past_last, c_past); atomicAdd(&a.counters->lanes_below_alpha, c_alpha);
        atomicAdd(&a.counters->reductions, c_red); atomicAdd(&a.counters->dead_block_visits, c_dead);
        atomicAdd(&a.counters->waves, 1ull);
    }
}

// Deterministic mode, second half: one lane per Gaussian walks the tiles of its rectangle row-major, finds its entry in
// each tile's (depth, id)-sorted slice by binary search and adds the slots the waves of that tile left, wave 0 first.
// The summation order is a function of the scene alone.  Debug facility: no attempt at speed.
__global__ __launch_bounds__(256) void det_reduce_kernel(CompositeBwdArgs a, int units) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= a.P) return;
    float sum[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint4 rc = a.rect[g];
    const int x0 = (int)(rc.x & 0xffffu), x1 = (int)(rc.x >> 16), y0 = (int)(rc.y & 0xffffu), y1 = (int)(rc.y >> 16);
    const uint32_t dg = a.depth_bits[g];
    if (a.tiles[g] > 0u) {
        for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++) {
                const uint2 r = a.ranges[ty * a.gridx + tx];
                uint32_t lo = r.x, hi = r.y;
                while (lo < hi) {                         // first entry with (depth, id) >= (dg, g)
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    const uint32_t id = a.point_list[mid];
                    const uint32_t dm = a.depth_bits[id];
                    if (dm < dg || (dm == dg && id < (uint32_t)g)) lo = mid + 1; else hi = mid;
                }
                if (lo < r.y && a.point_list[lo] == (uint32_t)g)
                    for (int u = 0; u < units; u++) {
                        const float *sl = a.det + ((size_t)lo * units + u) * GSR_ACC_FLOATS;
#pragma unroll
                        for (int v = 0; v < 9; v++) sum[v] += sl[v];
                    }
            }
    }
    float *row = a.acc + GSR_ACC_FLOATS * (size_t)g;
#pragma unroll
    for (int v = 0; v < 9; v++) row[v] = sum[v];
}

template <int NPX>
static hipError_t launch_bwd(const CompositeBwdArgs &a, int exact_cull, int wpb, hipStream_t s) {
    const int T = a.gridx * a.gridy;
    const int units = T * (4 / NPX);
    const int blocks = (units + wpb - 1) / wpb;
    const int padded = (blocks + 7) / 8 * 8;
    const size_t lds = (size_t)wpb * BWD_LDS_F4 * sizeof(float4) + (size_t)g_composite_lds_pad;
    if (a.det) {
        hipLaunchKernelGGL((composite_bwd_kernel<NPX, false, true>), dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
        hipLaunchKernelGGL(det_reduce_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a, 4 / NPX);
    } else if (a.counters)
        hipLaunchKernelGGL((composite_bwd_kernel<NPX, true, false>), dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
    else
        hipLaunchKernelGGL((composite_bwd_kernel<NPX, false, false>), dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
    return hipGetLastError();
}

// The reverse pass only adds into the rows of Gaussians the forward pass marked (GeomView::touched) and into replica rows: only
// those are cleared -- one wave-wide 16-byte store per 4 marked rows instead of a 66 P byte memset (config 3: 9 % are marked).
__global__ __launch_bounds__(256) void zero_marked_rows_kernel(int P, const uint8_t *__restrict__ touched, const uint32_t *__restrict__ mark,
                                                               float4 *__restrict__ acc4, size_t rows_total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one thread per row
    if (i >= rows_total) return;
    if (i < (size_t)P && touched[i] != (uint8_t)*mark) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    acc4[4 * i] = z; acc4[4 * i + 1] = z; acc4[4 * i + 2] = z; acc4[4 * i + 3] = z;
}

hipError_t launch_zero_marked_rows(int P, const uint8_t *touched, const uint32_t *mark, float *acc, size_t rows_total, hipStream_t s) {
    if (rows_total == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_marked_rows_kernel, dim3((unsigned)((rows_total + 255) / 256)), dim3(256), 0, s, P, touched, mark,
                       reinterpret_cast<float4 *>(acc), rows_total);
    return hipGetLastError();
}

hipError_t launch_composite_bwd(const CompositeBwdArgs &a, int npx, int exact_cull, int wpb, hipStream_t s) {
    if (a.gridx * a.gridy <= 0) return hipSuccess;
    switch (npx) {
        case 1: return launch_bwd<1>(a, exact_cull, wpb, s);
        case 2: return launch_bwd<2>(a, exact_cull, wpb, s);
        default: return launch_bwd<4>(a, exact_cull, wpb, s);
    }
}

}  // namespace gsr
