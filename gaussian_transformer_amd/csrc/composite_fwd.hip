// composite_fwd.hip -- front-to-back alpha compositing (S9).
//
// CDNA4 shape: a wave64 owns NPX 8x8 pixel blocks of one 16x16 tile (4 = whole tile, 2 = upper or
// lower half, 1 = one quadrant); lane l holds pixel l of each block.  Waves are fully independent
// (no workgroup barrier): each stages the tile's depth-sorted splat list 64 records at a time into a
// wave-private 3 KiB LDS slice (one record gathered per lane, read back as wave-uniform broadcast
// ds_read_b128) and stops as soon as all of ITS pixels are saturated (64-bit ballot), so a finished
// block never waits for the slowest pixel of the tile.  While staging, each lane decides per 8x8
// block whether its splat can reach alpha >= 1/255 anywhere in the block (exact minimum of the
// quadratic form over the block rectangle, gsr_device.h); the wave then walks only the set bits of
// the ballot -- dead splats cost two scalar instructions -- and skips dead blocks of a live splat
// with a scalar branch.  The per-block bits are also stored (one byte per block and list entry) for the
// reverse pass, which stages the same entries against the same blocks.  The conic is pre-scaled by -0.5*log2(e) / -log2(e) at staging time so the
// per-pixel exponent feeds v_exp_f32 directly.
// blockIdx is remapped so that each XCD (blocks b, b+8, ... share one) walks a contiguous band
// of tiles: neighbouring tiles share splats, which keeps the record gathers in that XCD's L2.
#include "gsr_device.h"
#include "gsr_internal.h"

namespace gsr {
extern int g_composite_lds_pad;

#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int xcd_band_unit_f(int b, int nblocks_padded) {
    const int chunk = nblocks_padded >> 3;
    return (b & 7) * chunk + (b >> 3);
}

// ---- the splat walk of one staged batch, 2 blocks per wave, written out (round 3) ----
// What the compiler makes of the C++ walk below is bound by the SCALAR unit as much as by the vector one: 31 scalar instructions per
// splat visit (mask algebra for the per-pair decisions, the set-bit walk, structured-control-flow bookkeeping) against 39 vector
// ones, and a gfx950 SIMD issues one scalar instruction per 4.2 cycles whatever the number of waves (scripts/valu_rate.hip: s_add_u32
// 4.24 cycles, v_fma_f32 2.71) -- 131 against ~129 cycles per visit.  Here the decisions narrow EXEC directly (v_cmpx), finished pixels
// are taken out by one s_not, "some pixel stops at this splat" leaves the straight-line path through s_cbranch_vccnz, and which
// entries reach which block is a lane mask per block made at staging time (s_bitcmp1_b64 against the entry's index: no
// v_readfirstlane of the record's bits): 14 scalar and 5 + 16 per block vector instructions per visit.  Arithmetic, operand order
// and comparison opcodes are those of the C++ walk (same fma contraction): the two produce identical images, bit for bit
// (tests/test_gpu_parity.py::test_forward_walkers_agree).
// Staged record of entry j at lds + 48 j: (px, py, a, b) (c, opacity, r, g) (b, ...).  v52..v63 are used by name (the b128 reads need
// register tuples whose components can be addressed).  Hazards (gfx940 family): v_exp result first read one instruction later;
// no VALU reads an SGPR a VALU wrote; EXEC is restored by SALU long before the next v_readlane-class instruction (there is none).
struct WalkState { float T, C0, C1, C2; uint32_t last; };
__device__ __forceinline__ void walk_batch_2blocks(uint32_t lds, uint32_t pos1, float fx0, float fx1, float fy, unsigned long long &m,
                                                   unsigned long long &b0m, unsigned long long &b1m, unsigned long long &d0,
                                                   unsigned long long &d1, WalkState &p0, WalkState &p1) {
    uint32_t j, pos;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n"
        "s_ff1_i32_b64 %[j], %[m]\n"
        "s_bitset0_b64 %[m], %[j]\n"
        "v_mad_u32_u24 v61, %[j], 48, %[lds]\n"
        "ds_read_b128 v[52:55], v61\n"               // px, py, a, b
        "ds_read_b128 v[56:59], v61 offset:16\n"     // c, opacity, red, green
        "ds_read_b32 v60, v61 offset:32\n"           // blue
        "s_add_u32 %[pos], %[j], %[pos1]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "v_sub_f32 v61, v53, %[fy]\n"                // dy
        "v_mul_f32 v62, v61, v55\n"                  // u = b dy
        "s_waitcnt lgkmcnt(1)\n"
        "v_mul_f32 v63, v61, v56\n"                  // c dy
        "v_mul_f32 v63, v61, v63\n"                  // w = (c dy) dy
        // (py, b, c and dy are dead from here: v53, v55, v56, v61 are the blocks' temporaries)
        "s_bitcmp1_b64 %[b0m], %[j]\n"
        "s_cbranch_scc0 2f\n"
        "s_not_b64 exec, %[d0]\n"
        "v_sub_f32 v61, v52, %[fx0]\n"             // dx
        "v_fma_f32 v53, v54, v61, v62\n"             // a dx + u
        "v_fma_f32 v61, v53, v61, v63\n"             // power (log2 units)
        "v_exp_f32 v53, v61\n"
        "v_cmpx_nlt_f32 vcc, 0, v61\n"               // !(power > 0)
        "v_mul_f32 v53, v57, v53\n"                  // opacity * G  (the v_exp result is first read one instruction after it)
        "v_cmpx_ngt_f32 vcc, %[amin], v53\n"         // !(alpha < 1/255)
        "v_min_f32 v55, 0x3f7d70a4, v53\n"           // min(0.99, .)
        "v_fma_f32 v56, -%[T0], v55, %[T0]\n"        // T (1 - alpha)
        "v_cmp_gt_f32 vcc, %[tmin], v56\n"           // it would fall below 1e-4: the pixel stops in front of this splat
        "s_cbranch_vccnz 20f\n"
        "21:\n"
        "v_mul_f32 %[T0], %[T0], v55\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_fmac_f32 %[C02], v60, %[T0]\n"
        "v_fmac_f32 %[C01], v59, %[T0]\n"
        "v_fmac_f32 %[C00], v58, %[T0]\n"
        "v_mov_b32 %[T0], v56\n"
        "v_mov_b32 %[L0], %[pos]\n"
        "2:\n"
        "s_bitcmp1_b64 %[b1m], %[j]\n"
        "s_cbranch_scc0 3f\n"
        "s_not_b64 exec, %[d1]\n"
        "v_sub_f32 v61, v52, %[fx1]\n"             // dx
        "v_fma_f32 v53, v54, v61, v62\n"             // a dx + u
        "v_fma_f32 v61, v53, v61, v63\n"             // power (log2 units)
        "v_exp_f32 v53, v61\n"
        "v_cmpx_nlt_f32 vcc, 0, v61\n"               // !(power > 0)
        "v_mul_f32 v53, v57, v53\n"                  // opacity * G  (the v_exp result is first read one instruction after it)
        "v_cmpx_ngt_f32 vcc, %[amin], v53\n"         // !(alpha < 1/255)
        "v_min_f32 v55, 0x3f7d70a4, v53\n"           // min(0.99, .)
        "v_fma_f32 v56, -%[T1], v55, %[T1]\n"        // T (1 - alpha)
        "v_cmp_gt_f32 vcc, %[tmin], v56\n"           // it would fall below 1e-4: the pixel stops in front of this splat
        "s_cbranch_vccnz 30f\n"
        "31:\n"
        "v_mul_f32 %[T1], %[T1], v55\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_fmac_f32 %[C12], v60, %[T1]\n"
        "v_fmac_f32 %[C11], v59, %[T1]\n"
        "v_fmac_f32 %[C10], v58, %[T1]\n"
        "v_mov_b32 %[T1], v56\n"
        "v_mov_b32 %[L1], %[pos]\n"
        "3:\n"
        "s_mov_b64 exec, -1\n"
        "s_waitcnt lgkmcnt(0)\n"                      // (a visit that blended nothing must not leave the read of v60 in flight)
        "s_cmp_lg_u64 %[m], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_branch 9f\n"
        // some pixel of the block stops here: it is finished and does not blend; a block with no pixel left takes no further entries
        "20:\n"
        "s_or_b64 %[d0], %[d0], vcc\n"
        "s_andn2_b64 exec, exec, vcc\n"
        "s_cmp_eq_u64 %[d0], -1\n"
        "s_cbranch_scc0 21b\n"
        "s_mov_b64 %[b0m], 0\n"
        "s_and_b64 %[m], %[m], %[b1m]\n"
        "s_branch 21b\n"
        "30:\n"
        "s_or_b64 %[d1], %[d1], vcc\n"
        "s_andn2_b64 exec, exec, vcc\n"
        "s_cmp_eq_u64 %[d1], -1\n"
        "s_cbranch_scc0 31b\n"
        "s_mov_b64 %[b1m], 0\n"
        "s_and_b64 %[m], %[m], %[b0m]\n"
        "s_branch 31b\n"
        "9:\n"
        : [m] "+s"(m), [b0m] "+s"(b0m), [b1m] "+s"(b1m), [d0] "+s"(d0), [d1] "+s"(d1), [j] "=&s"(j), [pos] "=&s"(pos),
          [T0] "+v"(p0.T), [C00] "+v"(p0.C0), [C01] "+v"(p0.C1), [C02] "+v"(p0.C2), [L0] "+v"(p0.last),
          [T1] "+v"(p1.T), [C10] "+v"(p1.C0), [C11] "+v"(p1.C1), [C12] "+v"(p1.C2), [L1] "+v"(p1.last)
        : [lds] "v"(lds), [pos1] "s"(pos1), [fx0] "v"(fx0), [fx1] "v"(fx1), [fy] "v"(fy), [amin] "s"(GSR_ALPHA_MIN), [tmin] "s"(GSR_T_MIN)
        : "memory", "scc", "vcc", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// COUNT: instrumented instantiations (gsr_set_option("count_lanes", 1 or 2)): 1 = lane-slot accounting (several times slower), see
// CompositeCounters; 2 = wave timeline only (two clock reads and one store per wave: the kernel runs at its normal speed)
// One work unit: the NPX blocks `sub` names of `tile`, the whole list.  `trace_id`: where the instrumented builds file the wave's timeline.
// ASMW: the splat walk is walk_batch_2blocks (NPX == 2, uninstrumented only) instead of the C++ loop
template <int NPX, int COUNT, bool ASMW = false>
__device__ __forceinline__ void fwd_unit(const CompositeArgs &a, float4 *my, const int lane, const int tile, const int sub, const int trace_id,
                                         const int exact_cull) {
    static_assert(!ASMW || (NPX == 2 && COUNT != 1), "the written-out walk exists for 2 blocks per wave, without lane counting");
    constexpr int UNITS_PER_TILE = 4 / NPX;
    const int unit = tile * UNITS_PER_TILE + sub;
    const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;
    const int tx = tile % a.gridx, ty = tile / a.gridx;
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);
    const float4 *rec4 = reinterpret_cast<const float4 *>(a.rec);

    // Per-pixel state: Tr = transmittance so far (it simply stops changing once the pixel is finished), C = colour,
    // last = 1-based list position of the last blended splat.  Which pixels are finished is kept as one 64-bit lane
    // mask per block in SGPRs (done[q]): the per-pair decisions are ballots combined with scalar logic, so the
    // bookkeeping of finished pixels costs no vector instructions.
    float fx[NPX], fy[NPX], Tr[NPX], C0[NPX], C1[NPX], C2[NPX];
    uint32_t last[NPX];
    bool inside[NPX];
    float bxa[NPX], bya[NPX], bxb[NPX], byb[NPX];
    unsigned long long done[NPX];                     // lanes whose pixel is saturated or outside the image
    uint32_t blk_done = 0;                            // bit q: every pixel of block q is finished (wave-uniform)
#pragma unroll
    for (int q = 0; q < NPX; q++) {
        const int blk = NPX == 4 ? q : (NPX == 2 ? sub * 2 + q : sub);
        const int x0 = tx * GSR_TILE + (blk & 1) * 8, y0 = ty * GSR_TILE + (blk >> 1) * 8;
        const int x = x0 + (lane & 7), y = y0 + (lane >> 3);
        inside[q] = x < a.W && y < a.H;
        fx[q] = (float)x; fy[q] = (float)y;
        bxa[q] = (float)x0; bya[q] = (float)y0;
        bxb[q] = (float)min(x0 + 7, a.W - 1); byb[q] = (float)min(y0 + 7, a.H - 1);
        Tr[q] = 1.f; C0[q] = C1[q] = C2[q] = 0.f; last[q] = 0u;
        done[q] = __builtin_amdgcn_ballot_w64(!inside[q]);
        if (done[q] == ~0ull) blk_done |= 1u << q;
    }
    const uint32_t all_blocks = (1u << NPX) - 1u;

    // Checkpoints for the segmented reverse pass (gsr_internal.h, SegView): every seg_len entries the wave is still alive at, it
    // stores per pixel the transmittance so far and the colour accumulated SINCE THE PREVIOUS CHECKPOINT, then restarts the colour
    // sums: the colour behind a boundary is later formed as a sum of per-segment colours, never as a difference of large sums.
    const bool seg_on = NPX == 2 && a.seg_len > 0;
    const bool file_info = NPX == 2 && (seg_on || a.lpt_span > 0);                      // lengths for the reverse pass's planner
    if (file_info && unit == 0 && lane == 0) a.seg.hdr[SEG_SEG] = (uint32_t)(seg_on ? a.seg_len : a.lpt_span);     // tells the reverse pass that units were filed
    int next_ck = seg_on ? a.seg_len : 0x7fffffff, n_ck = 0;
    uint32_t ck_slots = 0u;                           // lane j holds the pool slot of checkpoint j (boundary (j + 1) seg_len)
    unsigned long long c_staged = 0, c_visits = 0, c_blocks = 0, c_ok = 0, c_past = 0, c_alpha = 0, c_dead = 0;
    for (int base = 0; base < n && blk_done != all_blocks; base += 64) {
        const int cnt = min(64, n - base);
        if (NPX == 2 && base == next_ck) {            // wave-uniform
            next_ck = 0x7fffffff;
            if (n_ck < GSR_SEG_MAXCK) {
                // a slot of the band's share of the pool (one counter per band, each on its own cache line)
                const uint32_t band = (uint32_t)unit / a.seg.band_units, share = a.seg.pool_cap / GSR_SEG_BANDS;
                uint32_t slot = 0u;
                if (lane == 0) slot = atomicAdd(&a.seg.hdr[SEG_POOL + GSR_SEG_CTR_STRIDE * band], 1u);
                slot = __builtin_amdgcn_readfirstlane(slot);
                if (slot < share) {                   // else: the share is used up, this half tile stays one unit from here on
                    slot += band * share;
                    if (!GSR_IDX_OK(slot, a.seg.pool_cap, a.seg.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_POOL_SLOT)) slot = 0u;
                    float4 *ck = a.seg.pool + (size_t)slot * 128 + lane;
#pragma unroll
                    for (int q = 0; q < NPX; q++) {
                        ck[q * 64] = make_float4(Tr[q], C0[q], C1[q], C2[q]);
                        C0[q] = 0.f; C1[q] = 0.f; C2[q] = 0.f;
                    }
                    if (lane == n_ck) ck_slots = slot;
                    n_ck++;
                    next_ck = base + a.seg_len;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        bool live = false;
        uint32_t mybits = 0u;                         // blocks of this wave the lane's entry can reach
        if (lane < cnt && GSR_IDX_OK((size_t)range.x + base + lane, a.contrib_stride, a.seg.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_FWD_LIST_READ)) {
            const uint32_t g = a.point_list[range.x + base + lane];
            const float4 r0 = rec4[3 * (size_t)g], r1 = rec4[3 * (size_t)g + 1], r2 = rec4[3 * (size_t)g + 2];
            uint32_t bits = (1u << NPX) - 1u;
            if (exact_cull && r2.z > 0.f) {           // tau == 0: forward ran with culling off -> no information
                const float invA = 1.f / r0.z, invC = 1.f / r1.x;
                bits = 0u;
#pragma unroll
                for (int q = 0; q < NPX; q++)
                    bits |= block_reachable(r0.x, r0.y, r0.z, r0.w, r1.x, invA, invC, r2.z, bxa[q], bxb[q], bya[q], byb[q]) ? (1u << q) : 0u;
            }
            live = bits != 0u;
            mybits = bits;
            // "staged with a reachable block": pergauss_bwd.hip writes plain zeros for the Gaussians nobody marks.  A plain byte store:
            // an atomicOr into a shared flag word serialises on the splats that thousands of waves stage (config 4: 0.17 -> 0.60 ms)
            if (live) a.touched[g] = (uint8_t)a.touch_mark;
            // the reverse pass stages the same entries against the same blocks: hand it the reachability bits
#pragma unroll
            for (int q = 0; q < NPX; q++) {
                const int blk = NPX == 4 ? q : (NPX == 2 ? sub * 2 + q : sub);
                a.contrib[(size_t)blk * a.contrib_stride + range.x + base + lane] = (uint8_t)((bits >> q) & 1u);
            }
            // (px, py, -0.5*log2e*A, -log2e*B) (-0.5*log2e*C, opacity, r, g) (b, -, block bits, -)
            // (built as new vectors: overwriting components of the loaded ones sent them through scratch memory, 16 bytes of private
            //  segment per lane and its set-up at every wave's start)
            const StagedConic sc = stage_conic(r0.z, r0.w, r1.x);
            my[lane * 3 + 0] = make_float4(r0.x, r0.y, sc.a, sc.b);
            my[lane * 3 + 1] = make_float4(sc.c, r1.y, r1.z, r1.w);
            my[lane * 3 + 2] = make_float4(r2.x, 0.f, __uint_as_float(bits), 0.f);
        }
        uint64_t todo = __ballot(live);
        if (COUNT) c_staged += cnt;
        __builtin_amdgcn_wave_barrier();
        if constexpr (ASMW) {
            // which entries reach which block, as lane masks (a finished block takes none); the walk visits the union
            unsigned long long b0m = (blk_done & 1u) ? 0ull : __builtin_amdgcn_ballot_w64((mybits & 1u) != 0u);
            unsigned long long b1m = (blk_done & 2u) ? 0ull : __builtin_amdgcn_ballot_w64((mybits & 2u) != 0u);
            unsigned long long m = b0m | b1m;
            if (m != 0ull) {
                WalkState p0 = {Tr[0], C0[0], C1[0], C2[0], last[0]}, p1 = {Tr[1], C0[1], C1[1], C2[1], last[1]};
                walk_batch_2blocks((uint32_t)(uintptr_t)my, (uint32_t)(base + 1), fx[0], fx[1], fy[0], m, b0m, b1m, done[0], done[1], p0, p1);
                Tr[0] = p0.T; C0[0] = p0.C0; C1[0] = p0.C1; C2[0] = p0.C2; last[0] = p0.last;
                Tr[1] = p1.T; C0[1] = p1.C0; C1[1] = p1.C1; C2[1] = p1.C2; last[1] = p1.last;
                if (done[0] == ~0ull) blk_done |= 1u;
                if (done[1] == ~0ull) blk_done |= 2u;
            }
            todo = 0;
        }
        while (todo) {
            if (COUNT) c_visits += 1;
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 *mj = my + (uint32_t)j * 3u;
            const float4 r0 = mj[0], r1 = mj[1], r2 = mj[2];
            const uint32_t bits = __builtin_amdgcn_readfirstlane(__float_as_uint(r2.z));
            const uint32_t pos = (uint32_t)(base + j + 1);
            const StagedConic kc = {r0.z, r0.w, r1.x};
            // blocks side by side share dy: the terms of the exponent without dx once per row of blocks (gsr_device.h)
            const RowTerms rt = splat_row_terms(kc, r0.y - fy[0]);
            const RowTerms rt1 = NPX == 4 ? splat_row_terms(kc, r0.y - fy[NPX - 1]) : rt;
            unsigned long long any_stop = 0ull;       // lanes finishing at this splat
            auto block_body = [&](int q) __attribute__((always_inline)) {
                const float dx = r0.x - fx[q];
                float araw;                                          // the one evaluation both passes share (gsr_device.h)
                const unsigned long long okm = splat_alpha(splat_power_log2_row(kc, (NPX == 4 && q >= 2) ? rt1 : rt, dx), r1.y, araw);
                const float alpha = fminf(GSR_ALPHA_MAX, araw);
                const float aT = alpha * Tr[q];
                const float Tn = Tr[q] - aT;                         // = T (1 - alpha)
                // decisions as lane masks: a live pixel that passes the alpha tests either blends the splat or,
                // if that would take T below 1e-4, stops in front of it (S9)
                const unsigned long long livem = okm & ~done[q];
                const unsigned long long stopm = livem & __builtin_amdgcn_ballot_w64(Tn < GSR_T_MIN);
                const unsigned long long blendm = livem & ~stopm;
                done[q] |= stopm; any_stop |= stopm;
                if (COUNT == 1) {
                    c_blocks += 1; c_ok += __builtin_popcountll(blendm); c_past += __builtin_popcountll(done[q] & ~stopm);
                    c_alpha += __builtin_popcountll(~blendm & ~(done[q] & ~stopm)); c_dead += blendm == 0ull;
                }
                if (__builtin_amdgcn_inverse_ballot_w64(blendm)) {
                    C0[q] += r1.z * aT; C1[q] += r1.w * aT; C2[q] += r2.x * aT;
                    Tr[q] = Tn;
                    last[q] = pos;
                }
            };
            const uint32_t need = bits & ~blk_done;   // blocks the splat can reach and that still have a live pixel
            // (one straight-line body for "every block needed" was measured: 25 % slower -- larger code and live ranges buy
            // nothing, the SIMD's other waves already fill the issue slots)
#pragma unroll
            for (int q = 0; q < NPX; q++)
                if (need & (1u << q)) block_body(q);         // scalar branch: unreachable or saturated block
            if (any_stop != 0ull) {                   // wave-uniform: some pixel finished, maybe a whole block
#pragma unroll
                for (int q = 0; q < NPX; q++)
                    if (done[q] == ~0ull) blk_done |= 1u << q;
                if (blk_done == all_blocks) todo = 0;
            }
        }
    }
    if (COUNT && lane == 0 && a.counters) {
        if (COUNT == 1) {
        atomicAdd(&a.counters->staged, c_staged); atomicAdd(&a.counters->visits, c_visits);
        atomicAdd(&a.counters->block_visits, c_blocks); atomicAdd(&a.counters->lanes_ok, c_ok);
        atomicAdd(&a.counters->lanes_past_last, c_past); atomicAdd(&a.counters->lanes_below_alpha, c_alpha);
        atomicAdd(&a.counters->dead_block_visits, c_dead); atomicAdd(&a.counters->waves, 1ull);
        }
        if (a.counters->trace && (unsigned long long)trace_id < a.counters->trace_cap)
            a.counters->trace[trace_id] = make_uint4((uint32_t)t_start, (uint32_t)wall_clock64(), (uint32_t)c_staged,
                                                 (uint32_t)min(c_visits, 4095ull) | (__builtin_amdgcn_s_getreg(30724) & 0xffffu) << 12 |   // HW_ID[15:0]: wave, simd, pipe, cu, sh, se
                                                 (__builtin_amdgcn_s_getreg(6164) & 0xfu) << 28);                                           // XCC_ID
    }
    if (file_info) {
        // back over the checkpoints: each one's colour (its own segment's) is replaced by the colour composited BEHIND its
        // boundary, what the reverse pass starts from; the running sum ends as the pixel's whole colour
        for (int j = n_ck - 1; j >= 0; j--) {
            const uint32_t slot = __builtin_amdgcn_readlane(ck_slots, j);
            float4 *ck = a.seg.pool + (size_t)slot * 128 + lane;
#pragma unroll
            for (int q = 0; q < NPX; q++) {
                const float4 x = ck[q * 64];
                ck[q * 64] = make_float4(x.x, C0[q], C1[q], C2[q]);
                C0[q] += x.y; C1[q] += x.z; C2[q] += x.w;
            }
        }
        // what the reverse pass's planner needs of this half tile: how far its pixels got, the checkpoints taken and their slots
        int ml = 0;
#pragma unroll
        for (int q = 0; q < NPX; q++) ml = max(ml, (int)last[q]);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) ml = max(ml, __shfl_xor(ml, sft));
        if (lane < 8) a.seg.ck_slot[(size_t)unit * 8 + lane] = ck_slots;
        if (lane == 0) a.seg.info[unit] = make_uint2((uint32_t)ml, (uint32_t)n_ck);
    }
    const size_t HW = (size_t)a.W * a.H;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
#pragma unroll
    for (int q = 0; q < NPX; q++) {
        if (inside[q]) {
            const size_t pix = (size_t)fy[q] * a.W + (size_t)fx[q];
            const float Tf = Tr[q];
            a.final_T[pix] = Tf;
            a.n_contrib[pix] = last[q];
            a.out_color[pix] = C0[q] + Tf * bg0;
            a.out_color[HW + pix] = C1[q] + Tf * bg1;
            a.out_color[2 * HW + pix] = C2[q] + Tf * bg2;
        }
    }
}

// ---- long lists on small images: one wave per 8x8 BLOCK, two of them (a workgroup) per half tile (round 3) ----
// On images whose half tiles do not fill the chip twice over, the forward kernel lasts as long as its longest list: every wave is
// resident from the start and the wave of the longest half tile walks its chain alone (profiles/r03/wave_trace_default.jsonl: 1 458
// of 4 863 waves alive at half of config 2's kernel, 46 at three quarters).  A lone wave issues one instruction per ~5 cycles whatever
// its kind, so the chain's speed is the number of instructions per visit: 57 for a two-block visit, 36 for a one-block visit.  Half
// tiles whose list is longer than `pair_long_n` are therefore walked by a WORKGROUP of two waves, one per block, each with the one-block
// walk below (same exponent form -- row terms -- as the two-block walk: the reverse pass must take the same decisions).  The two
// share what the reverse pass expects per half tile: the checkpoint slots (first wave to reach a boundary draws the slot, through an
// LDS word), and `info` (largest last contributor, checkpoints taken), written by whichever finishes last.
__device__ __forceinline__ void walk_batch_1block(uint32_t lds, uint32_t pos1, float fx, float fy, unsigned long long &m,
                                                  unsigned long long &d, WalkState &p) {
    uint32_t j, pos;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n"
        "s_ff1_i32_b64 %[j], %[m]\n"
        "s_bitset0_b64 %[m], %[j]\n"
        "v_mad_u32_u24 v61, %[j], 48, %[lds]\n"
        "ds_read_b128 v[52:55], v61\n"               // px, py, a, b
        "ds_read_b128 v[56:59], v61 offset:16\n"     // c, opacity, red, green
        "ds_read_b32 v60, v61 offset:32\n"           // blue
        "s_add_u32 %[pos], %[j], %[pos1]\n"
        "s_not_b64 exec, %[d]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "v_sub_f32 v61, v53, %[fy]\n"                // dy
        "v_mul_f32 v62, v61, v55\n"                  // u = b dy
        "s_waitcnt lgkmcnt(1)\n"
        "v_mul_f32 v63, v61, v56\n"                  // c dy
        "v_mul_f32 v63, v61, v63\n"                  // w = (c dy) dy
        "v_sub_f32 v61, v52, %[fx]\n"                // dx
        "v_fma_f32 v53, v54, v61, v62\n"             // a dx + u
        "v_fma_f32 v61, v53, v61, v63\n"             // power (log2 units)
        "v_exp_f32 v53, v61\n"
        "v_cmpx_nlt_f32 vcc, 0, v61\n"               // !(power > 0)
        "v_mul_f32 v53, v57, v53\n"                  // opacity * G
        "v_cmpx_ngt_f32 vcc, %[amin], v53\n"         // !(alpha < 1/255)
        "v_min_f32 v55, 0x3f7d70a4, v53\n"           // min(0.99, .)
        "v_fma_f32 v56, -%[T], v55, %[T]\n"          // T (1 - alpha)
        "v_cmp_gt_f32 vcc, %[tmin], v56\n"
        "s_cbranch_vccnz 20f\n"
        "21:\n"
        "v_mul_f32 %[T], %[T], v55\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_fmac_f32 %[C2], v60, %[T]\n"
        "v_fmac_f32 %[C1], v59, %[T]\n"
        "v_fmac_f32 %[C0], v58, %[T]\n"
        "v_mov_b32 %[T], v56\n"
        "v_mov_b32 %[L], %[pos]\n"
        "s_mov_b64 exec, -1\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_cmp_lg_u64 %[m], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_branch 9f\n"
        "20:\n"
        "s_or_b64 %[d], %[d], vcc\n"
        "s_andn2_b64 exec, exec, vcc\n"
        "s_cmp_eq_u64 %[d], -1\n"
        "s_cbranch_scc0 21b\n"
        "s_mov_b64 %[m], 0\n"                        // no pixel left: no further entries
        "s_branch 21b\n"
        "9:\n"
        : [m] "+s"(m), [d] "+s"(d), [j] "=&s"(j), [pos] "=&s"(pos),
          [T] "+v"(p.T), [C0] "+v"(p.C0), [C1] "+v"(p.C1), [C2] "+v"(p.C2), [L] "+v"(p.last)
        : [lds] "v"(lds), [pos1] "s"(pos1), [fx] "v"(fx), [fy] "v"(fy), [amin] "s"(GSR_ALPHA_MIN), [tmin] "s"(GSR_T_MIN)
        : "memory", "scc", "vcc", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

struct PairShared {                 // one per workgroup (LDS), zeroed before the waves part ways
    uint32_t slot[GSR_SEG_MAXCK + 1];   // checkpoint k: 0 = nobody was here yet, 1 = being drawn, 2 + s = pool slot s, ~0u = the pool's share is used up
    uint32_t ml[2], nck[2];             // what each wave found: largest last contributor, checkpoints it wrote
    uint32_t finished;                  // waves done (the second one files the half tile)
};

// block wave `w` (0 = left, 1 = right block) of half tile `sub` of `tile`
template <int COUNT>
__device__ __forceinline__ void fwd_unit_pair(const CompositeArgs &a, float4 *my, PairShared *ps, const int lane, const int tile, const int sub,
                                              const int w, const int trace_id, const int exact_cull) {
    const int unit = tile * 2 + sub;
    const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;
    const int tx = tile % a.gridx, ty = tile / a.gridx;
    const uint2 range = a.ranges[tile];
    const int n = __builtin_amdgcn_readfirstlane((int)(range.y - range.x));
    const float4 *rec4 = reinterpret_cast<const float4 *>(a.rec);
    const int blk = sub * 2 + w;
    const int x0 = tx * GSR_TILE + (blk & 1) * 8, y0 = ty * GSR_TILE + (blk >> 1) * 8;
    const int x = x0 + (lane & 7), y = y0 + (lane >> 3);
    const bool inside = x < a.W && y < a.H;
    const float fx = (float)x, fy = (float)y;
    const float bxa = (float)x0, bya = (float)y0, bxb = (float)min(x0 + 7, a.W - 1), byb = (float)min(y0 + 7, a.H - 1);
    WalkState p = {1.f, 0.f, 0.f, 0.f, 0u};
    unsigned long long done = __builtin_amdgcn_ballot_w64(!inside);
    const bool seg_on = a.seg_len > 0;
    if (seg_on && unit == 0 && w == 0 && lane == 0) a.seg.hdr[SEG_SEG] = (uint32_t)a.seg_len;
    int next_ck = seg_on ? a.seg_len : 0x7fffffff, n_ck = 0;
    uint32_t ck_slots = 0u;                           // lane j: pool slot of checkpoint j
    unsigned long long c_staged = 0;
    for (int base = 0; base < n && done != ~0ull; base += 64) {
        const int cnt = min(64, n - base);
        if (base == next_ck) {                        // wave-uniform; both waves of the pair pass the same boundaries while they live
            next_ck = 0x7fffffff;
            const int k = base / a.seg_len - 1;       // checkpoint index of this boundary
            if (k < GSR_SEG_MAXCK) {
                uint32_t got = 0u;
                if (lane == 0) {
                    uint32_t old = atomicCAS(&ps->slot[k], 0u, 1u);
                    if (old == 0u) {                  // first of the pair here: draw the slot for both
                        const uint32_t band = (uint32_t)unit / a.seg.band_units, share = a.seg.pool_cap / GSR_SEG_BANDS;
                        uint32_t sl = atomicAdd(&a.seg.hdr[SEG_POOL + GSR_SEG_CTR_STRIDE * band], 1u);
                        old = sl < share ? 2u + band * share + sl : ~0u;
                        __hip_atomic_store(&ps->slot[k], old, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        while (old == 1u) { __builtin_amdgcn_s_sleep(2); old = __hip_atomic_load(&ps->slot[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    }
                    got = old;
                }
                got = __builtin_amdgcn_readfirstlane(got);
                if (got != ~0u) {
                    uint32_t slot = got - 2u;
                    if (!GSR_IDX_OK(slot, a.seg.pool_cap, a.seg.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_POOL_SLOT)) slot = 0u;
                    a.seg.pool[(size_t)slot * 128 + w * 64 + lane] = make_float4(p.T, p.C0, p.C1, p.C2);
                    p.C0 = 0.f; p.C1 = 0.f; p.C2 = 0.f;
                    if (lane == k) ck_slots = slot;
                    n_ck = k + 1;
                    next_ck = base + a.seg_len;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        bool reach = false;
        if (lane < cnt && GSR_IDX_OK((size_t)range.x + base + lane, a.contrib_stride, a.seg.hdr + GSR_DBG_SEG_WORD, GSR_BOUND_FWD_LIST_READ)) {
            const uint32_t g = a.point_list[range.x + base + lane];
            const float4 r0 = rec4[3 * (size_t)g], r1 = rec4[3 * (size_t)g + 1], r2 = rec4[3 * (size_t)g + 2];
            reach = true;
            if (exact_cull && r2.z > 0.f) {
                const float invA = 1.f / r0.z, invC = 1.f / r1.x;
                reach = block_reachable(r0.x, r0.y, r0.z, r0.w, r1.x, invA, invC, r2.z, bxa, bxb, bya, byb);
            }
            if (reach) a.touched[g] = (uint8_t)a.touch_mark;
            a.contrib[(size_t)blk * a.contrib_stride + range.x + base + lane] = (uint8_t)(reach ? 1u : 0u);
            const StagedConic sc = stage_conic(r0.z, r0.w, r1.x);
            my[lane * 3 + 0] = make_float4(r0.x, r0.y, sc.a, sc.b);
            my[lane * 3 + 1] = make_float4(sc.c, r1.y, r1.z, r1.w);
            my[lane * 3 + 2] = make_float4(r2.x, 0.f, 0.f, 0.f);
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(reach);
        if (COUNT) c_staged += cnt;
        __builtin_amdgcn_wave_barrier();
        if (m != 0ull) {
            walk_batch_1block((uint32_t)(uintptr_t)my, (uint32_t)(base + 1), fx, fy, m, done, p);
            // (wave-uniform, said so for the compiler: carried around this loop as an asm output it would be given vector registers)
            done = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(done >> 32)) << 32) |
                   (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)done);
        }
    }
    if (COUNT && lane == 0 && a.counters && a.counters->trace && (unsigned long long)trace_id < a.counters->trace_cap)
        a.counters->trace[trace_id] = make_uint4((uint32_t)t_start, (uint32_t)wall_clock64(), (uint32_t)c_staged,
                                                 (__builtin_amdgcn_s_getreg(30724) & 0xffffu) << 12 | (__builtin_amdgcn_s_getreg(6164) & 0xfu) << 28);
    if (seg_on) {
        n_ck = __builtin_amdgcn_readfirstlane(n_ck);      // (wave-uniform; said so for the compiler: readlane's index below)
        // back over the checkpoints this wave wrote: its own segment's colour -> the colour composited behind the boundary
        for (int j = n_ck - 1; j >= 0; j--) {
            const uint32_t slot = __builtin_amdgcn_readlane(ck_slots, j);
            float4 *ck = a.seg.pool + (size_t)slot * 128 + w * 64 + lane;
            const float4 c = *ck;
            *ck = make_float4(c.x, p.C0, p.C1, p.C2);
            p.C0 += c.y; p.C1 += c.z; p.C2 += c.w;
        }
        int ml = (int)p.last;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) ml = max(ml, __shfl_xor(ml, sft));
        if (lane == 0) {
            ps->ml[w] = (uint32_t)ml; ps->nck[w] = (uint32_t)n_ck;
            __threadfence_block();
            if (atomicAdd(&ps->finished, 1u) == 1u) {             // the other wave is done too: file the half tile
                __threadfence_block();
                const uint32_t nk = max(ps->nck[0], ps->nck[1]);
                a.seg.info[unit] = make_uint2(max(ps->ml[0], ps->ml[1]), nk);
                for (uint32_t k = 0; k < 8u; k++)
                    a.seg.ck_slot[(size_t)unit * 8 + k] = (k < nk && ps->slot[k] >= 2u && ps->slot[k] != ~0u) ? ps->slot[k] - 2u : 0u;
            }
        }
    }
    if (inside) {
        const size_t HW = (size_t)a.W * a.H;
        const size_t pix = (size_t)y * a.W + (size_t)x;
        a.final_T[pix] = p.T;
        a.n_contrib[pix] = p.last;
        a.out_color[pix] = p.C0 + p.T * a.bg[0];
        a.out_color[HW + pix] = p.C1 + p.T * a.bg[1];
        a.out_color[2 * HW + pix] = p.C2 + p.T * a.bg[2];
    }
}

// classic decomposition: one wave per NPX blocks of a tile, XCD-banded
template <int NPX, int COUNT, bool ASMW>
__device__ __forceinline__ void fwd_kernel_body(const CompositeArgs &a, int nblocks_padded, int exact_cull) {
    constexpr int UNITS_PER_TILE = 4 / NPX;
    extern __shared__ __align__(16) float4 stage_dyn[];     // [waves per block][64 * 3]
    const int T = a.gridx * a.gridy;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int unit = xcd_band_unit_f(blockIdx.x, nblocks_padded) * wpb + wave;
    const int tile = unit / UNITS_PER_TILE, sub = unit % UNITS_PER_TILE;
    if (tile >= T) return;                            // wave-uniform
    fwd_unit<NPX, COUNT, ASMW>(a, stage_dyn + wave * (64 * 3), lane, tile, sub, unit, exact_cull);
}
template <int NPX, int COUNT>
__global__ __launch_bounds__(256) void composite_fwd_kernel(CompositeArgs a, int nblocks_padded, int exact_cull) {
    fwd_kernel_body<NPX, COUNT, false>(a, nblocks_padded, exact_cull);
}
// the default instantiation: written-out walk; 8 waves per SIMD asked for explicitly (the walk's twelve named registers + the loop
// invariants the compiler hoists came to 65 VGPRs without it)
template <int COUNT>      // 0, or 2 = with the wave timeline (count_lanes = 2: splat visits are not counted in this walk)
__global__ __launch_bounds__(256, 8) void composite_fwd_walk_kernel(CompositeArgs a, int nblocks_padded, int exact_cull) {
    fwd_kernel_body<2, COUNT, true>(a, nblocks_padded, exact_cull);
}

// small images: a workgroup of two waves per half tile; a long list is walked by both (one block each), a short one by the first
// (both blocks, the walk above) while the second leaves at once
template <int COUNT>
__global__ __launch_bounds__(128, 8) void composite_fwd_pair_kernel(CompositeArgs a, int nblocks_padded, int exact_cull) {
    extern __shared__ __align__(16) float4 stage_dyn[];     // [2 waves][64 * 3]
    __shared__ PairShared ps;
    const int T = a.gridx * a.gridy;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int unit = xcd_band_unit_f(blockIdx.x, nblocks_padded);
    const int tile = unit >> 1, sub = unit & 1;
    if (tile >= T) return;                            // workgroup-uniform
    if (threadIdx.x < sizeof(PairShared) / 4) reinterpret_cast<uint32_t *>(&ps)[threadIdx.x] = 0u;
    __syncthreads();
    const uint2 range = a.ranges[tile];
    const int n_list = __builtin_amdgcn_readfirstlane((int)(range.y - range.x));      // (told to the compiler: wave-uniform)
    if (n_list > a.pair_long_n) fwd_unit_pair<COUNT>(a, stage_dyn + w * (64 * 3), &ps, lane, tile, sub, w, unit * 2 + w, exact_cull);
    else if (w == 0) fwd_unit<2, COUNT, true>(a, stage_dyn, lane, tile, sub, unit * 2, exact_cull);
}

template <int NPX>
static hipError_t launch_fwd(const CompositeArgs &a, int exact_cull, int wpb, hipStream_t s) {
    const int T = a.gridx * a.gridy;
    const int units = T * (4 / NPX);
    const int blocks = (units + wpb - 1) / wpb;
    const int padded = (blocks + 7) / 8 * 8;
    if (NPX == 2 && a.asm_walk && a.pair_long_n > 0 && !(a.counters && a.count_mode == 1)) {
        const int pblocks = (units + 7) / 8 * 8;      // one workgroup per half tile
        const size_t plds = (size_t)2 * 64 * 3 * sizeof(float4) + (size_t)g_composite_lds_pad;
        if (a.counters && a.count_mode == 2) hipLaunchKernelGGL(composite_fwd_pair_kernel<2>, dim3(pblocks), dim3(128), plds, s, a, pblocks, exact_cull);
        else hipLaunchKernelGGL(composite_fwd_pair_kernel<0>, dim3(pblocks), dim3(128), plds, s, a, pblocks, exact_cull);
        return hipGetLastError();
    }
    if (a.counters && a.count_mode == 2 && NPX == 2 && a.asm_walk)
        hipLaunchKernelGGL(composite_fwd_walk_kernel<2>, dim3(padded), dim3(64 * wpb), (size_t)wpb * 64 * 3 * sizeof(float4) + (size_t)g_composite_lds_pad, s, a,
                           padded, exact_cull);
    else if (a.counters && a.count_mode == 2)
        hipLaunchKernelGGL((composite_fwd_kernel<NPX, 2>), dim3(padded), dim3(64 * wpb), (size_t)wpb * 64 * 3 * sizeof(float4) + (size_t)g_composite_lds_pad, s, a,
                           padded, exact_cull);
    else if (a.counters)
        hipLaunchKernelGGL((composite_fwd_kernel<NPX, 1>), dim3(padded), dim3(64 * wpb), (size_t)wpb * 64 * 3 * sizeof(float4) + (size_t)g_composite_lds_pad, s, a,
                           padded, exact_cull);
    else if (NPX == 2 && a.asm_walk)
        hipLaunchKernelGGL(composite_fwd_walk_kernel<0>, dim3(padded), dim3(64 * wpb), (size_t)wpb * 64 * 3 * sizeof(float4) + (size_t)g_composite_lds_pad, s, a,
                           padded, exact_cull);
    else
        hipLaunchKernelGGL((composite_fwd_kernel<NPX, 0>), dim3(padded), dim3(64 * wpb), (size_t)wpb * 64 * 3 * sizeof(float4) + (size_t)g_composite_lds_pad, s, a,
                           padded, exact_cull);
    return hipGetLastError();
}

hipError_t launch_composite_fwd(const CompositeArgs &a, int npx, int exact_cull, int wpb, hipStream_t s) {
    if (a.gridx * a.gridy <= 0) return hipSuccess;
    switch (npx) {
        case 1: return launch_fwd<1>(a, exact_cull, wpb, s);
        case 2: return launch_fwd<2>(a, exact_cull, wpb, s);
        default: return launch_fwd<4>(a, exact_cull, wpb, s);
    }
}

}  // namespace gsr
