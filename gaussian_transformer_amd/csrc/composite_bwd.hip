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
#include <atomic>

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

// COUNT: instrumented instantiations (gsr_set_option("count_lanes", 1 or 2)): 1 tallies staged records, splat visits, 8x8 block
// visits (= 64 lane slots each), blending lanes and the reason idle lanes were idle into a.counters (CompositeCounters); 2 only
// records the wave timeline (start / end clock per work unit), at the kernel's normal speed
// DET: deterministic mode (gsr_set_option("deterministic_bwd", 1)): instead of the float atomics, whose arrival order
// differs from run to run, every (wave, list entry) stores its nine sums into its own slot of a.det and
// det_reduce_kernel below adds each Gaussian's slots in a fixed order.  The in-wave reduction is order-fixed already.
struct BwdTally { unsigned long long staged = 0, visits = 0, blocks = 0, ok = 0, past = 0, alpha = 0, red = 0, dead = 0; };

// One work unit: the entries [lo, hi) of the list of `tile`, against the NPX blocks `sub` names.  hi < 0: up to the last contributor
// of these pixels (the whole half tile, the classic decomposition).  ckslot != ~0: pixels that blended anything at or behind entry
// hi start from the forward pass's checkpoint at that boundary instead of from the end of their list (SegView, gsr_internal.h).
// ---- the reverse walk of one staged batch, 2 blocks per wave, written out (round 3; see composite_fwd.hip::walk_batch_2blocks) ----
// The decisions (power, alpha, the tests on them) are formed exactly as in the C++ walk below and in the forward pass; the sums are
// not bit-identical to the C++ walk's (packed multiply-adds for the moments, a differently associated 16-float tree), within the
// order-of-addition tolerance the atomics impose anyway.  What differs in structure: which entries reach which block (and lie in
// front of the block's last contributor) is a lane mask per block made at staging time -- no v_readfirstlane, no per-visit
// comparisons against blk_last; the first block of a visit writes the nine sums, the second adds to them (no zero fill, no
// duplicated products); the moments s (dx, dy), s dx (dx, dy) are v_pk_mul/fma_f32 on the register pair (dx, dy); the reduction rows
// are written with immediate offsets and folded with v_pk_add_f32; the totals leave through global_atomic_add_f32 with the
// accumulator base in SGPRs and a 32-bit byte offset (rows < 2^26: gsr_backward checks).
// v36..v63 are used by name: record v36-45 (px py a b | c opacity r g | blue, bits|row<<4); sums v46-48 (colour), v[50:51] (s dx,
// s dy), v[52:53] (s dx dx, s dx dy), v49 (s dy dy), v54 (s); u w v56-57, (dx, dy) v[58:59], temporaries v55, v60-63; the reduction
// reads its 16 floats into v36-43 and v56-63.
struct BwdPx { float T, accd, fx, d0, d1, d2, tb; int last; };
__device__ __forceinline__ void walk_batch_bwd_2blocks(uint32_t lds, uint32_t redw, uint32_t redr, uint32_t lane, int base, float fy,
                                                       unsigned long long m, unsigned long long b0m, unsigned long long b1m,
                                                       const float *acc, BwdPx &p0, BwdPx &p1) {
    uint32_t j, pos, m0sv;
    unsigned long long m1, m3, any;
    const unsigned long long redm = 0xFFFFFFFFFull, slotm = 0x111111111ull;
    asm volatile(
        "s_mov_b32 %[m0sv], m0\n"                     // M0 is the compiler's: saved here, restored at the end
        "s_mov_b32 m0, %[redw]\n"                     // base of the reduction rows (ds_write_addtid_b32)
        "1:\n"
        "s_flbit_i32_b64 %[j], %[m]\n"
        "s_xor_b32 %[j], %[j], 63\n"                  // highest set bit: the batch is walked back to front
        "s_bitset0_b64 %[m], %[j]\n"
        "v_lshl_add_u32 v58, %[j], 4, %[lds]\n"
        "v_lshl_add_u32 v59, %[j], 3, %[lds]\n"
        "ds_read_b128 v[36:39], v58\n"                // px, py, a, b
        "ds_read_b128 v[40:43], v58 offset:1024\n"    // c, opacity, red, green
        "ds_read_b64 v[44:45], v59 offset:2048\n"     // blue, reachability bits | accumulator row << 4
        "s_add_i32 %[pos], %[j], %[base]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "v_sub_f32 v59, v37, %[fy]\n"                 // dy
        "v_mul_f32 v56, v59, v39\n"                   // u = b dy
        "s_waitcnt lgkmcnt(1)\n"
        "v_mul_f32 v57, v59, v40\n"
        "v_mul_f32 v57, v59, v57\n"                   // w = (c dy) dy
        "s_waitcnt lgkmcnt(0)\n"
        "s_bitcmp1_b64 %[b0m], %[j]\n"
        "s_cbranch_scc0 3f\n"
        "v_sub_f32 v58, v36, %[fx0]\n"
        "v_fma_f32 v55, v38, v58, v56\n"
        "v_fma_f32 v55, v55, v58, v57\n"
        "v_exp_f32 v60, v55\n"
        "v_cmp_nlt_f32_e64 %[m1], 0, v55\n"
        "v_cmp_lt_i32 vcc, %[pos], %[l0]\n"
        "v_mul_f32 v61, %[d01], v43\n"
        "v_mul_f32 v60, v41, v60\n"
        "v_cmp_ngt_f32_e64 %[m3], %[amin], v60\n"
        "s_and_b64 %[m3], %[m3], %[m1]\n"
        "s_and_b64 vcc, %[m3], vcc\n"
        "v_cndmask_b32 v60, 0, v60, vcc\n"
        "v_min_f32 v62, 0x3f7d70a4, v60\n"
        "v_sub_f32 v63, 1.0, v62\n"
        "v_rcp_f32 v63, v63\n"
        "v_fmac_f32 v61, %[d00], v42\n"
        "v_fmac_f32 v61, %[d02], v44\n"
        "v_sub_f32 v61, v61, %[A0]\n"
        "v_fma_f32 v55, %[T0], v61, -%[tb0]\n"
        "v_mul_f32 %[T0], %[T0], v63\n"
        "v_mul_f32 v55, v55, v63\n"
        "v_fmac_f32 %[A0], v61, v62\n"
        "v_mul_f32 v62, v62, %[T0]\n"
        "s_mov_b64 %[any], vcc\n"
        "v_mul_f32 v46, %[d00], v62\n"
        "v_mul_f32 v47, %[d01], v62\n"
        "v_mul_f32 v48, %[d02], v62\n"
        "v_mul_f32 v54, v60, v55\n"
        "v_pk_mul_f32 v[50:51], v[54:55], v[58:59] op_sel_hi:[0,1]\n"
        "v_pk_mul_f32 v[52:53], v[50:51], v[58:59] op_sel_hi:[0,1]\n"
        "v_mul_f32 v49, v59, v51\n"
        "s_bitcmp1_b64 %[b1m], %[j]\n"
        "s_cbranch_scc0 4f\n"
        "v_sub_f32 v58, v36, %[fx1]\n"
        "v_fma_f32 v55, v38, v58, v56\n"
        "v_fma_f32 v55, v55, v58, v57\n"
        "v_exp_f32 v60, v55\n"
        "v_cmp_nlt_f32_e64 %[m1], 0, v55\n"
        "v_cmp_lt_i32 vcc, %[pos], %[l1]\n"
        "v_mul_f32 v61, %[d11], v43\n"
        "v_mul_f32 v60, v41, v60\n"
        "v_cmp_ngt_f32_e64 %[m3], %[amin], v60\n"
        "s_and_b64 %[m3], %[m3], %[m1]\n"
        "s_and_b64 vcc, %[m3], vcc\n"
        "v_cndmask_b32 v60, 0, v60, vcc\n"
        "v_min_f32 v62, 0x3f7d70a4, v60\n"
        "v_sub_f32 v63, 1.0, v62\n"
        "v_rcp_f32 v63, v63\n"
        "v_fmac_f32 v61, %[d10], v42\n"
        "v_fmac_f32 v61, %[d12], v44\n"
        "v_sub_f32 v61, v61, %[A1]\n"
        "v_fma_f32 v55, %[T1], v61, -%[tb1]\n"
        "v_mul_f32 %[T1], %[T1], v63\n"
        "v_mul_f32 v55, v55, v63\n"
        "v_fmac_f32 %[A1], v61, v62\n"
        "v_mul_f32 v62, v62, %[T1]\n"
        "s_or_b64 %[any], %[any], vcc\n"
        "v_fmac_f32 v46, %[d10], v62\n"
        "v_fmac_f32 v47, %[d11], v62\n"
        "v_fmac_f32 v48, %[d12], v62\n"
        "v_fmac_f32 v54, v60, v55\n"
        "v_mul_f32 v60, v60, v55\n"
        "v_pk_mul_f32 v[62:63], v[60:61], v[58:59] op_sel_hi:[0,1]\n"
        "v_pk_fma_f32 v[50:51], v[60:61], v[58:59], v[50:51] op_sel_hi:[0,1,1]\n"
        "v_pk_fma_f32 v[52:53], v[62:63], v[58:59], v[52:53] op_sel_hi:[0,1,1]\n"
        "v_fmac_f32 v49, v59, v63\n"
        "s_branch 4f\n"
        "3:\n"                                        // block 0 takes no part: block 1 does (the entry is in one of the masks)
        "v_sub_f32 v58, v36, %[fx1]\n"
        "v_fma_f32 v55, v38, v58, v56\n"
        "v_fma_f32 v55, v55, v58, v57\n"
        "v_exp_f32 v60, v55\n"
        "v_cmp_nlt_f32_e64 %[m1], 0, v55\n"
        "v_cmp_lt_i32 vcc, %[pos], %[l1]\n"
        "v_mul_f32 v61, %[d11], v43\n"
        "v_mul_f32 v60, v41, v60\n"
        "v_cmp_ngt_f32_e64 %[m3], %[amin], v60\n"
        "s_and_b64 %[m3], %[m3], %[m1]\n"
        "s_and_b64 vcc, %[m3], vcc\n"
        "v_cndmask_b32 v60, 0, v60, vcc\n"
        "v_min_f32 v62, 0x3f7d70a4, v60\n"
        "v_sub_f32 v63, 1.0, v62\n"
        "v_rcp_f32 v63, v63\n"
        "v_fmac_f32 v61, %[d10], v42\n"
        "v_fmac_f32 v61, %[d12], v44\n"
        "v_sub_f32 v61, v61, %[A1]\n"
        "v_fma_f32 v55, %[T1], v61, -%[tb1]\n"
        "v_mul_f32 %[T1], %[T1], v63\n"
        "v_mul_f32 v55, v55, v63\n"
        "v_fmac_f32 %[A1], v61, v62\n"
        "v_mul_f32 v62, v62, %[T1]\n"
        "s_mov_b64 %[any], vcc\n"
        "v_mul_f32 v46, %[d10], v62\n"
        "v_mul_f32 v47, %[d11], v62\n"
        "v_mul_f32 v48, %[d12], v62\n"
        "v_mul_f32 v54, v60, v55\n"
        "v_pk_mul_f32 v[50:51], v[54:55], v[58:59] op_sel_hi:[0,1]\n"
        "v_pk_mul_f32 v[52:53], v[50:51], v[58:59] op_sel_hi:[0,1]\n"
        "v_mul_f32 v49, v59, v51\n"
        "4:\n"
        "s_cmp_eq_u64 %[any], 0\n"
        "s_cbranch_scc1 5f\n"                         // no pixel of this wave blends the splat
        // reduction through LDS: nine rows of the lanes' partial sums, 36 lanes add 16 floats each (packed adds), two DPP steps, one
        // 9-lane atomic
        // (ds_write_addtid_b32: address = M0 + offset + 4 lane, no address VGPR to move: 2 LDS-path cycles per row instead of 4 --
        //  MI355X_MICROARCH.md, LDS; the reduction's LDS traffic is what bounds this kernel once the vector work is trimmed)
        "ds_write_addtid_b32 v46\n"
        "ds_write_addtid_b32 v47 offset:272\n"
        "ds_write_addtid_b32 v48 offset:544\n"
        "ds_write_addtid_b32 v50 offset:816\n"
        "ds_write_addtid_b32 v51 offset:1088\n"
        "ds_write_addtid_b32 v52 offset:1360\n"
        "ds_write_addtid_b32 v53 offset:1632\n"
        "ds_write_addtid_b32 v49 offset:1904\n"
        "ds_write_addtid_b32 v54 offset:2176\n"
        "s_mov_b64 exec, %[redm]\n"
        "ds_read_b128 v[36:39], %[redr]\n"
        "ds_read_b128 v[40:43], %[redr] offset:16\n"
        "ds_read_b128 v[56:59], %[redr] offset:32\n"
        "ds_read_b128 v[60:63], %[redr] offset:48\n"
        "s_waitcnt lgkmcnt(3)\n"
        "v_pk_add_f32 v[36:37], v[36:37], v[38:39]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_pk_add_f32 v[56:57], v[56:57], v[58:59]\n"
        "v_pk_add_f32 v[36:37], v[36:37], v[40:41]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_pk_add_f32 v[60:61], v[60:61], v[62:63]\n"
        "v_pk_add_f32 v[56:57], v[56:57], v[60:61]\n"
        "v_pk_add_f32 v[36:37], v[36:37], v[56:57]\n"
        "v_add_f32 v36, v36, v37\n"
        "s_mov_b64 exec, -1\n"
        "s_nop 1\n"                                   // a DPP read of a VGPR two wait states after its write
        "v_add_f32_dpp v36, v36, v36 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "s_nop 1\n"
        "v_add_f32_dpp v36, v36, v36 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "s_mov_b64 exec, %[slotm]\n"                  // lanes 0, 4, ..., 32 hold the nine totals
        "v_and_b32 v37, -16, v45\n"
        "v_lshl_add_u32 v37, v37, 2, %[lane]\n"       // 64 row + 4 value
        "global_atomic_add_f32 v37, v36, %[acc]\n"
        "s_mov_b64 exec, -1\n"
        "5:\n"
        "s_cmp_lg_u64 %[m], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_mov_b32 m0, %[m0sv]\n"
        : [m] "+s"(m), [j] "=&s"(j), [pos] "=&s"(pos), [m1] "=&s"(m1), [m3] "=&s"(m3), [any] "=&s"(any), [m0sv] "=&s"(m0sv),
          [T0] "+v"(p0.T), [A0] "+v"(p0.accd), [T1] "+v"(p1.T), [A1] "+v"(p1.accd)
        : [b0m] "s"(b0m), [b1m] "s"(b1m), [base] "s"(base), [amin] "s"(GSR_ALPHA_MIN), [redm] "s"(redm), [slotm] "s"(slotm), [acc] "s"(acc), [redw] "s"(redw),
          [lds] "v"(lds), [redr] "v"(redr), [lane] "v"(lane), [fy] "v"(fy),
          [fx0] "v"(p0.fx), [d00] "v"(p0.d0), [d01] "v"(p0.d1), [d02] "v"(p0.d2), [tb0] "v"(p0.tb), [l0] "v"(p0.last),
          [fx1] "v"(p1.fx), [d10] "v"(p1.d0), [d11] "v"(p1.d1), [d12] "v"(p1.d2), [tb1] "v"(p1.tb), [l1] "v"(p1.last)
        : "memory", "scc", "vcc", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",
          "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

template <int NPX, int COUNT, bool DET, bool ASMW = false>
__device__ __forceinline__ void bwd_unit(const CompositeBwdArgs &a, float4 *my, const int lane, const int tile, const int sub, const int lo,
                                         int hi, const uint32_t ckslot, BwdTally &tl) {
    constexpr int UNITS_PER_TILE = 4 / NPX;          // waves per tile
    const int tx = tile % a.gridx, ty = tile / a.gridx;
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
    if (hi < 0 || hi > max_last) hi = max_last;
    if (hi <= lo) return;                             // wave-uniform
    if (NPX == 2 && ckslot != ~0u) {
        // a pixel whose last contributor lies behind this unit's end starts from what the forward pass recorded there: the transmittance
        // in front of entry hi and the colour composited from it on (a sum of per-segment colours, no cancellation)
        const float4 *ck = a.seg.pool + (size_t)ckslot * 128 + lane;
#pragma unroll
        for (int q = 0; q < NPX; q++) {
            const float4 c = ck[q * 64];
            if (last[q] > hi) { Tr[q] = c.x; accd[q] = (c.y * d0[q] + c.z * d1[q] + c.w * d2[q]) / c.x; }
        }
    }

    // Cross-lane reduction through LDS (the LDS pipe is idle otherwise, the VALU is the bottleneck):
    // every lane stores its 9 partial sums as red[value][lane]; lane 4*value + part then adds the 16
    // floats red[value][16*part .. 16*part+15] (4 x ds_read_b128), two DPP steps fold the 4 parts, and
    // lanes 0,4,...,32 hold the nine totals: ONE 9-lane global_atomic_add_f32 per (wave, splat).
    const int rvalue = lane >> 2, rpart = lane & 3;
    // rows of RED_STRIDE = 68 floats: with 64 the nine rows start in the same LDS bank and the 36 reading lanes collide 9-way
    const float4 *red_rd = reinterpret_cast<const float4 *>(red + (rvalue < 9 ? rvalue : 0) * RED_STRIDE + rpart * 16);
    const bool red_lane = rvalue < 9;
    const int slot = (red_lane && rpart == 0) ? rvalue : -1;

    for (int base = ((hi - 1) >> 6) << 6; base >= lo; base -= 64) {
        const int cnt = min(64, hi - base);
        __builtin_amdgcn_wave_barrier();
        bool live = false;
        uint32_t mybits = 0u;                         // blocks of this wave the lane's entry can reach
        if (lane < cnt && GSR_IDX_OK((size_t)range.x + base + lane, a.contrib_stride, a.seg.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_BWD_LIST_READ)) {
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
            mybits = bits;
            const StagedConic sc = stage_conic(r0.z, r0.w, r1.x);       // as the forward pass staged it
            my[lane] = make_float4(r0.x, r0.y, sc.a, sc.b);
            my[64 + lane] = make_float4(sc.c, r1.y, r1.z, r1.w);
            // accumulator row: the Gaussian's own, or -- for a splat with replica rows (gsr_internal.h) -- replica (tile mod K)
            const uint32_t hot = __float_as_uint(r2.w);
            const uint32_t row = hot ? (uint32_t)a.P + (hot >> 4) + ((uint32_t)tile & ((1u << (hot & 15u)) - 1u)) : g;
            myc[lane] = make_float2(r2.x, __uint_as_float(bits | (row << 4)));      // row < 2^28 (checked by gsr_backward)
        }
        uint64_t todo = __ballot(live);
        if (COUNT) { tl.staged += cnt; tl.visits += __builtin_popcountll(todo); }
        __builtin_amdgcn_wave_barrier();
        if constexpr (ASMW) {
            static_assert(NPX == 2 && COUNT != 1 && !DET, "the written-out walk exists for 2 blocks per wave, without lane counting, atomic sums");
            // entries that reach block q and lie in front of its last contributor (positions base .. blk_last[q] - 1), as lane masks
            auto below = [](int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : (1ull << n) - 1ull); };
            const unsigned long long b0m = __builtin_amdgcn_ballot_w64((mybits & 1u) != 0u) & below(blk_last[0] - base);
            const unsigned long long b1m = __builtin_amdgcn_ballot_w64((mybits & 2u) != 0u) & below(blk_last[1] - base);
            const unsigned long long m = b0m | b1m;
            if (m != 0ull) {
                BwdPx p0 = {Tr[0], accd[0], fx[0], d0[0], d1[0], d2[0], tb[0], last[0]}, p1 = {Tr[1], accd[1], fx[1], d0[1], d1[1], d2[1], tb[1], last[1]};
                walk_batch_bwd_2blocks((uint32_t)(uintptr_t)my, (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)red), (uint32_t)(uintptr_t)red_rd, (uint32_t)lane, base,
                                       fy[0], m, b0m, b1m, a.acc, p0, p1);
                Tr[0] = p0.T; accd[0] = p0.accd; Tr[1] = p1.T; accd[1] = p1.accd;
            }
            todo = 0;
        }
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
            const RowTerms rt = splat_row_terms(kc, r0.y - fy[0]);   // blocks side by side share dy (gsr_device.h); the forward pass forms the exponent the same way
            const RowTerms rt1 = NPX == 4 ? splat_row_terms(kc, r0.y - fy[NPX - 1]) : rt;
            // body for one 8x8 block
            auto block_body = [&](int q) __attribute__((always_inline)) {
                const float dx = r0.x - fx[q], dy = r0.y - fy[q];
                float araw;                                          // same expression, same bits as the forward pass
                const unsigned long long okm = splat_alpha(splat_power_log2_row(kc, (NPX == 4 && q >= 2) ? rt1 : rt, dx), r1.y, araw) &
                                               __builtin_amdgcn_ballot_w64(pos < last[q]);
                any_ok |= okm;
                if (COUNT == 1) {
                    const unsigned long long mp = __ballot(!(pos < last[q]));
                    tl.blocks += 1; tl.ok += __builtin_popcountll(okm); tl.past += __builtin_popcountll(mp);
                    tl.alpha += __builtin_popcountll(~okm & ~mp); tl.dead += okm == 0ull;
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
            if (COUNT == 1) tl.red += 1;
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
}

template <int COUNT>
__device__ __forceinline__ void bwd_flush_tally(const CompositeBwdArgs &a, const BwdTally &tl, const int lane, const unsigned trace_idx,
                                                const unsigned long long t_start) {
    if (COUNT && lane == 0 && a.counters) {
        if (COUNT == 1) {
            atomicAdd(&a.counters->staged, tl.staged); atomicAdd(&a.counters->visits, tl.visits);
            atomicAdd(&a.counters->block_visits, tl.blocks); atomicAdd(&a.counters->lanes_ok, tl.ok);
            atomicAdd(&a.counters->lanes_past_last, tl.past); atomicAdd(&a.counters->lanes_below_alpha, tl.alpha);
            atomicAdd(&a.counters->reductions, tl.red); atomicAdd(&a.counters->dead_block_visits, tl.dead);
            atomicAdd(&a.counters->waves, 1ull);
        }
        if (a.counters->trace && (unsigned long long)trace_idx < a.counters->trace_cap)
            a.counters->trace[trace_idx] = make_uint4((uint32_t)t_start, (uint32_t)wall_clock64(), (uint32_t)tl.staged,
                                                      (uint32_t)min(tl.visits, 4095ull) | (__builtin_amdgcn_s_getreg(30724) & 0xffffu) << 12 |   // HW_ID[15:0]: wave, simd, pipe, cu, sh, se
                                                      (__builtin_amdgcn_s_getreg(6164) & 0xfu) << 28);                                           // XCC_ID
    }
}

// classic decomposition: one wave per NPX blocks of a tile, XCD-banded
template <int NPX, int COUNT, bool DET, bool ASMW>
__device__ __forceinline__ void bwd_kernel_body(const CompositeBwdArgs &a, int nblocks_padded) {
    constexpr int UNITS_PER_TILE = 4 / NPX;          // waves per tile
    extern __shared__ __align__(16) float4 stage_dyn[];     // per wave: 64 records x 3 float4, then 9 x RED_STRIDE floats of reduction scratch
    const int T = a.gridx * a.gridy;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int unit = xcd_band_unit(blockIdx.x, nblocks_padded) * wpb + wave;
    const int tile = unit / UNITS_PER_TILE, sub = unit % UNITS_PER_TILE;
    if (tile >= T) return;                            // wave-uniform
    const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;
    BwdTally tl;
    bwd_unit<NPX, COUNT, DET, ASMW>(a, stage_dyn + wave * BWD_LDS_F4, lane, tile, sub, 0, -1, ~0u, tl);
    bwd_flush_tally<COUNT>(a, tl, lane, (unsigned)unit, t_start);
}
template <int NPX, int COUNT, bool DET>
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompositeBwdArgs a, int nblocks_padded, int /*exact_cull*/) {
    bwd_kernel_body<NPX, COUNT, DET, false>(a, nblocks_padded);
}
// the default instantiation: written-out walk (8 waves per SIMD asked for explicitly: the walk names 28 registers)
template <int COUNT>      // 0, or 2 = with the wave timeline
__global__ __launch_bounds__(256, 8) void composite_bwd_walk_kernel(CompositeBwdArgs a, int nblocks_padded, int /*exact_cull*/) {
    bwd_kernel_body<2, COUNT, false, true>(a, nblocks_padded);
}

// One zero-fill unit (FillArgs): the gradient rows of the Gaussians [g0, g1) that pergauss_bwd.hip will not write -- culled, or not
// composited by the forward pass -- set to zero, 64 Gaussians per round: their rows are contiguous in every output tensor, so the
// wave stores whole lines, switching off the lanes whose element belongs to a Gaussian that does have a gradient.
__device__ __forceinline__ void fill_unit(const FillArgs &f, const int lane, const int g0, const int g1) {
    const uint32_t mark = *f.mark;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // a unit is at most four rounds of 64 Gaussians: their flags are requested together (one memory round trip per unit, not per round)
    bool un[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int g = g0 + 64 * r + lane;
        const bool in = g < g1 && g < f.P;
        const int rad = in ? f.radii[g] : 1;
        const uint32_t t = in ? (uint32_t)f.touched[g] : mark;
        un[r] = in && !(rad > 0 && t == mark);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int base = g0 + 64 * r;
        if (base >= g1) break;
        const int g = base + lane;
        const bool unmarked = un[r];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(unmarked);
        if (m == 0ull) continue;
        auto bit = [&](int local) { return (m >> local) & 1ull; };
        // three floats per Gaussian: element e of the round's 192 belongs to Gaussian e / 3
        auto fill3 = [&](float *p) {
            if (!p) return;
            float *q = p + (size_t)base * 3;
#pragma unroll
            for (int t = 0; t < 3; t++) { const int e = lane + 64 * t; if (bit((e * 171) >> 9)) q[e] = 0.f; }
        };
        fill3(f.means2D); fill3(f.means3D); fill3(f.scales); fill3(f.colors);
        if (unmarked) {
            f.opacity[g] = 0.f;
            if (f.rots) reinterpret_cast<float4 *>(f.rots)[g] = z4;
        }
        if (f.cov3D) {
            float *q = f.cov3D + (size_t)base * 6;
#pragma unroll
            for (int t = 0; t < 6; t++) { const int e = lane + 64 * t; if (bit((e * 43691) >> 18)) q[e] = 0.f; }
        }
        auto fill_rows = [&](float *p, int fl) {     // fl floats per Gaussian
            if (!p) return;
            if (fl == 48 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {       // M = 16: twelve 16-byte chunks per Gaussian
                float4 *q = reinterpret_cast<float4 *>(p) + (size_t)base * 12;
#pragma unroll
                for (int t = 0; t < 12; t++) { const int c = lane + 64 * t; if (bit((c * 43691) >> 19)) q[c] = z4; }
            } else if (fl == 3) {
                fill3(p);
            } else {
                float *q = p + (size_t)base * fl;
                for (int e = lane; e < 64 * fl; e += 64) if (bit(e / fl)) q[e] = 0.f;
            }
        };
        if (f.sh_rest) { fill_rows(f.sh, 3); fill_rows(f.sh_rest, 3 * (f.M - 1)); }
        else fill_rows(f.sh, 3 * f.M);
    }
}

// Persistent decomposition (2 blocks per wave): the grid's waves draw work units -- (half tile, entry range, checkpoint), listed per
// XCD band of tiles in order of decreasing length by plan_units() below -- from the band's ticket counter; a wave whose band is
// finished helps the next band.  Long chains start first, the short units fill the end of the kernel; neighbouring tiles stay on one
// XCD (their splat records share its L2) as long as that XCD has work of its own.
template <int COUNT, bool DET, bool ASMW>
__device__ __forceinline__ void bwd_pk_body(const CompositeBwdArgs &a) {
    extern __shared__ __align__(16) float4 stage_dyn[];
    const int lane = threadIdx.x;
    uint32_t *hdr = a.seg.hdr;
    // bands in the order this wave serves them: its own first, then the other lists of its XCD (band & 7), then the other XCDs'
    const int home = blockIdx.x & (GSR_SEG_BANDS - 1);
    auto band_at = [&](int e) { return (((home & 7) + (e >> 2)) & 7) + 8 * (((home >> 3) + (e & 3)) & 3); };
    int band = home;
    // every ticket is drawn, the first one too: a ticket tied to the block index belongs to a workgroup that may not be resident yet
    // (measured: the launch was one wave per SIMD larger than what the chip held, and those 1024 units waited for the end)
    uint32_t ticket = 0u;
    if (lane == 0) ticket = atomicAdd(&hdr[SEG_BTICKET + GSR_SEG_CTR_STRIDE * band], 1u);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    while (true) {                                    // wave-uniform
        if (ticket >= hdr[SEG_BCOUNT + band]) {       // this band's list is used up (for good)
            // Look before drawing, and look at every band at once (lane e reads the counter of the e-th band of this wave's order):
            // when a band runs dry thousands of waves arrive here together, and their ticket atomics would queue up in one memory
            // channel behind each other (measured: 150 us of nothing, and the waves still working stalled with them); probing the
            // bands one after the other is a chain of 31 round trips at the end of every wave's life (measured: 10 us on a
            // 65 us kernel).  A plain coherent read can only lag behind the counter, never run ahead of it.
            bool has = false;
            if (lane < GSR_SEG_BANDS) {
                const int b = band_at(lane);
                has = __hip_atomic_load(&hdr[SEG_BTICKET + GSR_SEG_CTR_STRIDE * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < hdr[SEG_BCOUNT + b];
            }
            const unsigned long long open = __ballot(has);
            if (open == 0ull) break;
            band = band_at(__builtin_ctzll(open));
            uint32_t t = 0u;
            if (lane == 0) t = atomicAdd(&hdr[SEG_BTICKET + GSR_SEG_CTR_STRIDE * band], 1u);
            ticket = __builtin_amdgcn_readfirstlane(t);
            continue;
        }
        const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;
        const size_t at = seg_list_base(a.seg, band) + ticket;
        if (!GSR_IDX_OK(ticket, a.seg.list_cap, hdr + GSR_DBG_SEG_WORD, GSR_BOUND_UNIT_TICKET)) break;
        uint4 u = a.seg.bq[at];
        u.x = __builtin_amdgcn_readfirstlane(u.x); u.y = __builtin_amdgcn_readfirstlane(u.y);
        u.z = __builtin_amdgcn_readfirstlane(u.z); u.w = __builtin_amdgcn_readfirstlane(u.w);
        BwdTally tl;
        if (!ASMW && u.x == ~0u) fill_unit(a.fill, lane, (int)u.y, (int)u.z);       // wave-uniform (the walk variant is launched without fill units)
        else bwd_unit<2, COUNT, DET, ASMW>(a, stage_dyn, lane, (int)(u.x >> 1), (int)(u.x & 1u), (int)u.y, (int)u.z, u.w, tl);
        bwd_flush_tally<COUNT>(a, tl, lane, (unsigned)at, t_start);
        // the next ticket is drawn only now: a ticket drawn ahead of time is a unit nobody else can take while this wave is still busy --
        // measured: once all tickets were handed out, half the waves left and the rest worked off two units each
        uint32_t nxt = 0u;
        if (lane == 0) nxt = atomicAdd(&hdr[SEG_BTICKET + GSR_SEG_CTR_STRIDE * band], 1u);
        ticket = __builtin_amdgcn_readfirstlane(nxt);
    }
}

template <int COUNT, bool DET>
__global__ __launch_bounds__(64) void composite_bwd_pk_kernel(CompositeBwdArgs a) {
    bwd_pk_body<COUNT, DET, false>(a);
}
template <int COUNT>
__global__ __launch_bounds__(64, 8) void composite_bwd_pk_walk_kernel(CompositeBwdArgs a) {     // the default: written-out walk
    bwd_pk_body<COUNT, false, true>(a);
}

// Large images (round 3): one wave per half tile as in the classic kernel, but in the order of plan_units' lists -- per band the longest
// half tiles first (lengths filed by the forward pass, CompositeArgs::lpt_span) -- so that what the second generation of waves has left
// to do at the end of the kernel is short work.  Static assignment (workgroup b takes entry b / 32 of band b % 32): no tickets.
__global__ __launch_bounds__(64, 8) void composite_bwd_lpt_kernel(CompositeBwdArgs a) {
    extern __shared__ __align__(16) float4 stage_dyn[];
    const int lane = threadIdx.x;
    const uint32_t band = blockIdx.x & (GSR_SEG_BANDS - 1), idx = blockIdx.x / GSR_SEG_BANDS;
    if (idx >= a.seg.hdr[SEG_BCOUNT + band]) return;             // workgroup-uniform
    if (!GSR_IDX_OK(idx, a.seg.list_cap, a.seg.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_UNIT_TICKET)) return;
    uint4 u = a.seg.bq[seg_list_base(a.seg, (int)band) + idx];
    if (!GSR_IDX_OK(u.x, a.seg.units, a.seg.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_UNIT_LIST)) return;
    u.x = __builtin_amdgcn_readfirstlane(u.x); u.y = __builtin_amdgcn_readfirstlane(u.y);
    u.z = __builtin_amdgcn_readfirstlane(u.z); u.w = __builtin_amdgcn_readfirstlane(u.w);
    BwdTally tl;
    bwd_unit<2, 0, false, true>(a, stage_dyn, lane, (int)(u.x >> 1), (int)(u.x & 1u), (int)u.y, (int)u.z, u.w, tl);
}
hipError_t launch_composite_bwd_lpt(const CompositeBwdArgs &a, hipStream_t s) {
    if (a.gridx * a.gridy <= 0) return hipSuccess;
    const unsigned grid = GSR_SEG_BANDS * a.seg.band_units;
    hipLaunchKernelGGL(composite_bwd_lpt_kernel, dim3(grid), dim3(64), (size_t)BWD_LDS_F4 * sizeof(float4) + (size_t)g_composite_lds_pad, s, a);
    return hipGetLastError();
}

// The unit list of one band (one workgroup of 256 threads): every half tile's pieces -- [k seg, (k+1) seg) below each checkpoint the
// forward wave took, and the top piece up to the last contributor -- counting-sorted by length, longest first.  Runs next to the
// clearing of the accumulator rows and has to be done when that is (~9 us): everything it reads is requested up front (eight half
// tiles per thread cover 2048 per band: 4K images), LDS atomics are issued once per wave and class.
#define PLAN_ROUNDS 8
__device__ void plan_units(const SegView &v, const int band, const int fillP, const int fill_chunk) {
    __shared__ uint32_t hist[GSR_SEG_LEN_CLASSES], cur[GSR_SEG_LEN_CLASSES];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t u0 = (uint32_t)band * v.band_units, u1 = min(v.units, u0 + v.band_units);
    const uint32_t span = (u1 > u0 ? u1 - u0 : 0u);
    uint4 *list = v.bq + seg_list_base(v, band);
    // requested before anything depends on them
    uint2 in[PLAN_ROUNDS]; uint4 cka[PLAN_ROUNDS], ckb[PLAN_ROUNDS];
#pragma unroll
    for (int r = 0; r < PLAN_ROUNDS; r++) {
        const uint32_t u = u0 + r * 256u + tid;
        const bool ok = u < u1;
        in[r] = ok ? v.info[u] : make_uint2(0u, 0u);
        const uint4 *cs = reinterpret_cast<const uint4 *>(v.ck_slot + (size_t)(ok ? u : u0) * 8);
        cka[r] = cs[0]; ckb[r] = cs[1];
    }
    const uint32_t seg = v.hdr[SEG_SEG];
    if (tid == 0) v.hdr[SEG_BTICKET + GSR_SEG_CTR_STRIDE * band] = 0u;
    // the band's share of the zero-fill units (FillArgs), behind its compositing units: chunks [c0, c1) of fill_chunk Gaussians
    uint32_t c0 = 0u, c1 = 0u;
    if (fill_chunk > 0 && fillP > 0) {
        const uint32_t nc = ((uint32_t)fillP + fill_chunk - 1) / fill_chunk, per = (nc + GSR_SEG_BANDS - 1) / GSR_SEG_BANDS;
        c0 = min(nc, (uint32_t)band * per); c1 = min(nc, c0 + min(per, (uint32_t)GSR_SEG_FILL_CAP));
    }
    auto append_fill = [&](uint32_t at) {
        for (uint32_t c = c0 + tid; c < c1; c += 256)
            list[at + (c - c0)] = make_uint4(~0u, c * fill_chunk, min((uint32_t)fillP, (c + 1u) * fill_chunk), 0u);
    };
    if (seg == 0u) {                                  // the forward pass left no lengths: the half tiles themselves, in tile order
        for (uint32_t u = u0 + tid; u < u1; u += 256) list[u - u0] = make_uint4(u, 0u, ~0u, ~0u);
        append_fill(span);
        if (tid == 0) v.hdr[SEG_BCOUNT + band] = span + (c1 - c0);
        return;
    }
    if (tid < GSR_SEG_LEN_CLASSES) { hist[tid] = 0u; cur[tid] = 0u; }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t rounds = (span + 255u) / 256u;
    auto classify = [&](uint32_t u, const uint2 inf, uint32_t &ml, uint32_t &kt, int &cls) {
        ml = u < u1 ? inf.x : 0u;
        kt = ml ? min((ml - 1u) / seg, min(inf.y, (uint32_t)GSR_SEG_MAXCK)) : 0u;
        const uint32_t top = ml - kt * seg;
        cls = top > seg ? 0 : (top == seg ? 1 : 2 + (int)min((uint32_t)(GSR_SEG_LEN_CLASSES - 3), ((seg - top) * (GSR_SEG_LEN_CLASSES - 2)) / seg));
    };
    auto count_round = [&](uint32_t ml, uint32_t kt, int cls) {
        uint32_t ks = kt;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) ks += __shfl_xor(ks, sft);
        if (lane == 0 && ks) atomicAdd(&hist[1], ks);
        unsigned long long todo = __ballot(ml != 0u);
        while (todo) {                                // wave-uniform: one round per class present in the wave
            const int leader = __builtin_ctzll(todo);
            const int c = __shfl(cls, leader);
            const unsigned long long same = __ballot(ml != 0u && cls == c);
            if (lane == leader) atomicAdd(&hist[c], (uint32_t)__builtin_popcountll(same));
            todo &= ~same;
        }
    };
    auto place_round = [&](uint32_t u, uint32_t ml, uint32_t kt, int cls, const uint4 ca, const uint4 cb) {
        // full segments: the wave reserves the sum of its kt with one atomic, every lane takes its prefix
        uint32_t incl = kt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const uint32_t wsum = __shfl(incl, 63);
        uint32_t wbase = 0u;
        if (lane == 0 && wsum) wbase = atomicAdd(&cur[1], wsum);
        wbase = __shfl(wbase, 0);
        if (kt) {
            const uint32_t p = hist[1] + wbase + (incl - kt);
            const uint32_t slots[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
            for (uint32_t k = 0; k < GSR_SEG_MAXCK; k++)
                if (k < kt && GSR_IDX_OK(p + k, v.list_cap, v.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_UNIT_LIST)) list[p + k] = make_uint4(u, k * seg, (k + 1u) * seg, slots[k]);
        }
        unsigned long long todo = __ballot(ml != 0u);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int c = __shfl(cls, leader);
            const unsigned long long same = __ballot(ml != 0u && cls == c);
            uint32_t cbase = 0u;
            if (lane == leader) cbase = atomicAdd(&cur[c], (uint32_t)__builtin_popcountll(same));
            cbase = __shfl(cbase, leader);
            if (ml != 0u && cls == c) {
                const uint32_t at = hist[c] + cbase + (uint32_t)__builtin_popcountll(same & below);
                if (GSR_IDX_OK(at, v.list_cap, v.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_UNIT_LIST)) list[at] = make_uint4(u, kt * seg, ml, ~0u);
            }
            todo &= ~same;
        }
    };
    uint32_t ml[PLAN_ROUNDS], kt[PLAN_ROUNDS]; int cls[PLAN_ROUNDS];
#pragma unroll
    for (int r = 0; r < PLAN_ROUNDS; r++) {
        classify(u0 + r * 256u + tid, in[r], ml[r], kt[r], cls[r]);
        if ((uint32_t)r < rounds) count_round(ml[r], kt[r], cls[r]);
    }
    for (uint32_t r = PLAN_ROUNDS; r < rounds; r++) {         // bands of more than 2048 half tiles (beyond 4K)
        const uint32_t u = u0 + r * 256u + tid;
        uint32_t m, k; int c;
        classify(u, u < u1 ? v.info[u] : make_uint2(0u, 0u), m, k, c);
        count_round(m, k, c);
    }
    __syncthreads();
    __shared__ uint32_t n_units;
    if (tid == 0) {
        uint32_t run = 0u;
        for (int c = 0; c < GSR_SEG_LEN_CLASSES; c++) { const uint32_t h = hist[c]; hist[c] = run; run += h; }
        n_units = run;                                // <= band_units * (1 + GSR_SEG_MAXCK) by construction
        v.hdr[SEG_BCOUNT + band] = run + (c1 - c0);
    }
    __syncthreads();
    append_fill(n_units);
#pragma unroll
    for (int r = 0; r < PLAN_ROUNDS; r++)
        if ((uint32_t)r < rounds) place_round(u0 + r * 256u + tid, ml[r], kt[r], cls[r], cka[r], ckb[r]);
    for (uint32_t r = PLAN_ROUNDS; r < rounds; r++) {
        const uint32_t u = u0 + r * 256u + tid;
        uint32_t m, k; int c;
        classify(u, u < u1 ? v.info[u] : make_uint2(0u, 0u), m, k, c);
        const uint4 *cs = reinterpret_cast<const uint4 *>(v.ck_slot + (size_t)(u < u1 ? u : u0) * 8);
        place_round(u, m, k, c, cs[0], cs[1]);
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
        hipLaunchKernelGGL((composite_bwd_kernel<NPX, 0, true>), dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
        hipLaunchKernelGGL(det_reduce_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a, 4 / NPX);
    } else if (a.counters && a.count_mode == 2 && NPX == 2 && a.asm_walk)
        hipLaunchKernelGGL(composite_bwd_walk_kernel<2>, dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
    else if (a.counters && a.count_mode == 2)
        hipLaunchKernelGGL((composite_bwd_kernel<NPX, 2, false>), dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
    else if (a.counters)
        hipLaunchKernelGGL((composite_bwd_kernel<NPX, 1, false>), dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
    else if (NPX == 2 && a.asm_walk)
        hipLaunchKernelGGL(composite_bwd_walk_kernel<0>, dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
    else
        hipLaunchKernelGGL((composite_bwd_kernel<NPX, 0, false>), dim3(padded), dim3(64 * wpb), lds, s, a, padded, exact_cull);
    return hipGetLastError();
}

// The reverse pass only adds into the rows of Gaussians the forward pass marked (GeomView::touched) and into replica rows: only
// those are cleared -- one wave-wide 16-byte store per 4 marked rows instead of a 66 P byte memset (config 3: 9 % are marked).
__global__ __launch_bounds__(256) void zero_marked_rows_kernel(int P, const uint8_t *__restrict__ touched, const uint32_t *__restrict__ mark,
                                                               float4 *__restrict__ acc4, size_t rows_total, SegView seg, int plan_grid,
                                                               int fill_chunk) {
    // the first GSR_SEG_BANDS workgroups build the unit lists of the persistent reverse kernel instead (independent of the clearing,
    // dispatched first so that they run alongside it)
    const unsigned nplan = plan_grid > 0 ? GSR_SEG_BANDS : 0;
    if (blockIdx.x < nplan) {
        plan_units(seg, (int)blockIdx.x, P, fill_chunk);
        return;
    }
    const size_t i = (size_t)(blockIdx.x - nplan) * 256 + threadIdx.x;          // one thread per row
    if (i >= rows_total) return;
    if (i < (size_t)P && touched[i] != (uint8_t)*mark) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    acc4[4 * i] = z; acc4[4 * i + 1] = z; acc4[4 * i + 2] = z; acc4[4 * i + 3] = z;
}

hipError_t launch_zero_marked_rows(int P, const uint8_t *touched, const uint32_t *mark, float *acc, size_t rows_total, const SegView &seg,
                                   int plan_grid, int fill_chunk, hipStream_t s) {
    const size_t blocks = (rows_total + 255) / 256 + (plan_grid > 0 ? GSR_SEG_BANDS : 0);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_marked_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, P, touched, mark, reinterpret_cast<float4 *>(acc), rows_total, seg,
                       plan_grid, fill_chunk);
    return hipGetLastError();
}

// waves of the persistent kernel: exactly what the chip holds of this instantiation (occupancy query; a workgroup that is not resident
// from the start would sit on its first ticket until another wave retires), or fewer when the image cannot have that many units
template <typename K>
static int resident_waves(K kernel, size_t lds, std::atomic<int> &cache) {
    int n = cache.load();
    if (n > 0) return n;
    int dev = 0, cus = 256, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 16;
    n = cus * per_cu / GSR_SEG_BANDS * GSR_SEG_BANDS;
    if (n < GSR_SEG_BANDS) n = GSR_SEG_BANDS;
    cache.store(n);
    return n;
}
static size_t pk_lds_bytes() { return (size_t)BWD_LDS_F4 * sizeof(float4) + (size_t)g_composite_lds_pad; }

int composite_bwd_persistent_grid(int T, int det, int count_mode, int asm_walk) {
    static std::atomic<int> c_plain{0}, c_det{0}, c_cnt{0}, c_trace{0}, c_walk{0}, c_walk_trace{0};
    int n;
    if (det) n = resident_waves(composite_bwd_pk_kernel<0, true>, pk_lds_bytes(), c_det);
    else if (count_mode == 2 && asm_walk) n = resident_waves(composite_bwd_pk_walk_kernel<2>, pk_lds_bytes(), c_walk_trace);
    else if (count_mode == 2) n = resident_waves(composite_bwd_pk_kernel<2, false>, pk_lds_bytes(), c_trace);
    else if (count_mode == 1) n = resident_waves(composite_bwd_pk_kernel<1, false>, pk_lds_bytes(), c_cnt);
    else if (asm_walk) n = resident_waves(composite_bwd_pk_walk_kernel<0>, pk_lds_bytes(), c_walk);
    else n = resident_waves(composite_bwd_pk_kernel<0, false>, pk_lds_bytes(), c_plain);
    long long most = 2ll * T * (1 + GSR_SEG_MAXCK);              // units a frame can have at all
    most = (most + GSR_SEG_BANDS - 1) / GSR_SEG_BANDS * GSR_SEG_BANDS;     // a multiple of the bands: wave b starts in band b & 7
    return (int)(most < n ? (most > 0 ? most : GSR_SEG_BANDS) : n);
}

hipError_t launch_composite_bwd_persistent(const CompositeBwdArgs &a, int grid, hipStream_t s) {
    if (a.gridx * a.gridy <= 0 || grid <= 0) return hipSuccess;
    const size_t lds = pk_lds_bytes();
    if (a.det) {
        hipLaunchKernelGGL((composite_bwd_pk_kernel<0, true>), dim3(grid), dim3(64), lds, s, a);
        hipLaunchKernelGGL(det_reduce_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a, 2);
    } else if (a.counters && a.count_mode == 2 && a.asm_walk && a.fill.chunk == 0)
        hipLaunchKernelGGL(composite_bwd_pk_walk_kernel<2>, dim3(grid), dim3(64), lds, s, a);
    else if (a.counters && a.count_mode == 2)
        hipLaunchKernelGGL((composite_bwd_pk_kernel<2, false>), dim3(grid), dim3(64), lds, s, a);
    else if (a.counters)
        hipLaunchKernelGGL((composite_bwd_pk_kernel<1, false>), dim3(grid), dim3(64), lds, s, a);
    else if (a.asm_walk && a.fill.chunk == 0)
        hipLaunchKernelGGL(composite_bwd_pk_walk_kernel<0>, dim3(grid), dim3(64), lds, s, a);
    else
        hipLaunchKernelGGL((composite_bwd_pk_kernel<0, false>), dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

// debug-build self test: one deliberate out-of-range index; the words must then read (GSR_BOUND_SELFTEST, 5, 4, 1)
__global__ void bound_selftest_kernel(uint32_t *words) {
    if (GSR_IDX_OK(5, 4, words, GSR_BOUND_SELFTEST)) words[0] = 0xdeadu;
}
hipError_t launch_bound_selftest(uint32_t *words, hipStream_t s) {
    hipLaunchKernelGGL(bound_selftest_kernel, dim3(1), dim3(1), 0, s, words);
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
