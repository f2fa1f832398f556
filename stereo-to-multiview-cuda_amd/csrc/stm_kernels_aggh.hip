// stm_kernels_aggh.hip -- the last horizontal aggregation pass + WTA with the row's window range held in REGISTERS.
//
// Reference stages replaced: ca_cross_hsum_kernel_3, d_ca_cross_sum.cu:243-293 (window
// [x - armL, x + armR), ascending float32 adds from 0.0f) and dc_wta_kernel, d_dc_wta.cu:9-35 (first strictly lowest cost).
//
// The construction of stm_k_pq_v12r (stm_kernels_aggv.hip) applied to a row walk: what bounds the matrix-pipe kernels is the
// pipe the f32 MFMAs share with the vector ALU, and the round-3 row walk stm_k_pq_hs spends 35 % of its
// pipe time on vector instructions that are not MFMAs (window arithmetic, LDS addressing, WTA through exec-masked branches)
// and idles behind its barriers.  Here
//  * a wave owns a part of an image row and walks its tiles of 16 pixels.  The PQ layout's float4 [chunk][y][g][hypothesis]
//    IS four B operands (lane 16 b + n = chunk b, hypothesis n; the four floats = four consecutive window steps), so a pixel
//    group is four registers loaded by ONE buffer_load_dwordx4; the 22 groups a tile's sweep can touch are a ring of 88 registers,
//    the groups of the tiles ahead arrive in two alternating sets of 16 landing registers: ~165 registers, THREE waves per SIMD,
//    no LDS, no barrier;
//  * the window masks come from a per-frame table (stm_k_hwin_table: per tile of 16 pixels the sweep's first group, its length
//    and one 64-bit lane mask per group -- lanes 16 a + m = step a of the group, pixel m: the A operand of four MFMAs through
//    CBSZ / ABID); the 24 mask pairs of a tile are three s_load_dwordx16 issued a ring update ahead of the sweep, so the sweep
//    itself contains no load, no wait and no branch: 22 blocks of 56 bytes, entered at block 22 - n by one computed jump;
//  * WTA on the accumulators in place: per pixel the minimum over the four chunks (lowest chunk on ties), then over the 16
//    hypothesis lanes on the costs' bit patterns (aggregated costs are sums of non-negative terms), ties to the lowest d.
// One asm block with its own register allocation, like stm_k_pq_v12r.  Limits: usd <= 36, D <= 64 (one chunk set); everything
// else runs stm_k_pq_hs.  Results are bit-identical to it.
#include "stm_hwin.h"
#include <cstdlib>

namespace stm {

// The horizontal window table of `nviews` views from the arm planes (the frame path builds it inside stm_k_cross_arms instead).
__global__ __launch_bounds__(256) void stm_k_hwin_table(PQViews v, uint32_t *__restrict__ tab, int H, int W, int nTx)
{
    __shared__ uint32_t ev_all[4][4 * 96];
    const int wv = threadIdx.x >> 6, y = blockIdx.y, view = blockIdx.z;
    const int x0 = (blockIdx.x * 4 + wv) * 64, x = x0 + (threadIdx.x & 63);
    const u8 *__restrict__ armL = view ? v.armL[1] : v.armL[0], *__restrict__ armR = view ? v.armR[1] : v.armR[0];
    int aL = 0, aR = 0;
    if (x < W) {
        aL = armL[(size_t)y * W + x];
        aR = armR[(size_t)y * W + x];
    }
    hwin_build(tab + (size_t)view * H * nTx * HR_REC, ev_all[wv], y, x0, aL, aR, W, nTx);
}

// Register map of the asm block.
//   v[0:87]    the ring: pixel group g in registers 4 (g mod 22) .. + 3 (groups [4 t - 9, 4 t + 13) are live during tile t)
//   v[88:103], v[104:119] landing registers of the four groups loaded during an even / odd step (for the tile two steps ahead)
//   v[120:135] accumulators (register 4 b + i of lane 16 q + n = pixel 4 q + i, chunk b, hypothesis n); v136 / v137 the A operand of
//   the even / odd blocks; v[138:141] winning hypothesis per pixel, v[142:149] temporaries, v150 the constant 0x7fffffff
//   s[16:19] / s[20:23] buffer descriptors (volume, disparity row); s24 tile, s25 last tile + 1, s26 ring index of the tile's
//   range start, s29 ring size, s30 ring index of the running block, s[36:83] the 24 mask pairs of the tile (pair p = block p),
//   s84 / s85 first group and groups of the tile's sweep, s[86:87] the next tile's, s[88:89] mask address of block 0, s90 entry
//   block, s[92:93] jump target, s[94:99] temporaries / compare masks.
template <int EXP> // timing experiments (libstm_hip_timing.so only; results invalid): 1 no sweeps, 2 no volume loads, 4 no WTA, 8 no mask loads
__global__ __launch_bounds__(64, 3) void stm_k_pq_hsr(PQViews pv, const uint32_t *__restrict__ htab, int D, int zd, int H, int W, int G, int NC, int nTx,
                                                      int parts, int nviews)
{
    // block (one wave) -> (view, row, part of the row)
    const int part = blockIdx.x % parts, rest = blockIdx.x / parts, y = rest % H, view = rest / H;
    if (view >= nviews) return;
    const int per = (nTx + parts - 1) / parts, t0 = part * per, t1 = min(nTx, t0 + per);
    if (t0 >= t1) return;
    const int l = threadIdx.x;
    const float *in = view ? pv.a[1] : pv.a[0];
    float *disp_row = (view ? pv.disp[1] : pv.disp[0]) + (size_t)y * W;
    const uint32_t *hrec = htab + ((size_t)(view * H + y) * nTx + t0) * HR_REC; // record of tile t0; tile t at + (t - t0) * HR_REC
    const uint32_t cstride = (uint32_t)H * (uint32_t)G * 256u;                    // bytes between chunks
    const uint32_t nrec = (uint32_t)NC * cstride;                                // bytes of the volume (chunks past the last: out of range -> zeros)
    const int vload = (l >> 4) * (int)cstride + (l & 15) * 16;                   // lane 16 b + n reads chunk b, hypothesis n of a group
    const int vn = l & 15;
    const int vst = (l & 15) == 0 ? (l >> 4) * 16 : 0x7ffffff0;                   // lane n == 0 of a pixel quad stores its four pixels
    // hypotheses d >= D (padding of the last chunk, chunks past the last) must never win: their costs become FLT_MAX
    float vbig[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) vbig[b] = (16 * b + (l & 15)) >= D ? 3.402823466e+38f : 0.0f;
    const int yG = y * G, Gm1 = G - 1, W4 = W * 4;
    const float zdf = (float)zd;
    asm volatile(R"ASM(
        .set HR_EXP, %[exp]
        ; ---------------------------------------------------------------- macros
        ; one block of a sweep: four MFMAs on one pixel group (its four registers = four window steps), the A operand (lanes 16 a + m =
        ; step a, pixel m) shared through CBSZ / ABID.  p = block index, acur / anxt = the A registers of this / the next block
        .macro HR_BLOCK p, acur, anxt
        s_set_gpr_idx_idx s30
        v_mfma_f32_16x16x1_4b_f32 v[120:135], v[\acur], v0, v[120:135] cbsz:2 abid:0
        v_cndmask_b32_e64 v[\anxt], 0, 1.0, s[36+2*((\p)+1):37+2*((\p)+1)]
        v_mfma_f32_16x16x1_4b_f32 v[120:135], v[\acur], v1, v[120:135] cbsz:2 abid:1
        s_add_u32 s30, s30, 4
        s_cmp_eq_u32 s30, s29
        s_cselect_b32 s30, 0, s30
        v_mfma_f32_16x16x1_4b_f32 v[120:135], v[\acur], v2, v[120:135] cbsz:2 abid:2
        v_mfma_f32_16x16x1_4b_f32 v[120:135], v[\acur], v3, v[120:135] cbsz:2 abid:3
        .endm
        .macro HR_PAIR b
        HR_BLOCK 2*(\b), 136, 137
        HR_BLOCK 2*(\b)+1, 137, 136
        .endm
        ; in: s84 = q0 (first group of the sweep inside the range), s85 = n (1 .. 22), s26 = ring index of the range start.
        ; out: s90 = entry block 22 - n, s30 = ring index of the sweep's first group
        .macro HR_SWEEP_ISSUE
        s_sub_u32 s90, 22, s85
        s_lshl_b32 s31, s84, 2
        s_add_u32 s30, s26, s31
        s_sub_u32 s31, s30, s29
        s_cmp_ge_u32 s30, s29
        s_cselect_b32 s30, s31, s30
        .endm
        ; the header and the 24 mask pairs (block p reads pair p + 1 for its successor, the entry block's own comes through M0) of
        ; tile s31 (index relative to t0) start travelling
        .macro HR_NEXT_MASKS
        s_lshl_b32 s31, s31, 8
        s_add_u32 s88, %[hrec_lo], s31
        s_addc_u32 s89, %[hrec_hi], 0
        s_load_dwordx2 s[86:87], s[88:89], 0x0
        .if (HR_EXP & 8) == 0
        s_load_dwordx16 s[36:51], s[88:89], 0x20
        s_load_dwordx16 s[52:67], s[88:89], 0x60
        s_load_dwordx16 s[68:83], s[88:89], 0xa0
        .endif
        .endm
        .macro HR_SWEEP_RUN
        s_lshl_b32 m0, s90, 1
        s_nop 0
        s_movrels_b64 s[94:95], s[36:37]      ; the entry block's mask
        v_cndmask_b32_e64 v136, 0, 1.0, s[94:95]
        v_cndmask_b32_e64 v137, 0, 1.0, s[94:95]
        s_set_gpr_idx_on s30, 0x2
        s_mul_i32 s31, s90, 56
        s_getpc_b64 s[92:93]
HR_pc_%=_\@:
        s_add_u32 s31, s31, HR_blk0_%=_\@-HR_pc_%=_\@+12
        s_add_u32 s92, s92, s31
        s_addc_u32 s93, s93, 0
        v_mfma_f32_16x16x1_4b_f32 v[120:135], v136, v0, 0 cbsz:2 abid:0
        s_setpc_b64 s[92:93]
HR_blk0_%=_\@:
        HR_PAIR 0
        HR_PAIR 1
        HR_PAIR 2
        HR_PAIR 3
        HR_PAIR 4
        HR_PAIR 5
        HR_PAIR 6
        HR_PAIR 7
        HR_PAIR 8
        HR_PAIR 9
        HR_PAIR 10
HR_end_%=_\@:
        .if (HR_end_%=_\@-HR_blk0_%=_\@) != 22*56
        .error "sweep blocks are not 56 bytes each"
        .endif
        s_set_gpr_idx_off
        s_nop 15
        s_nop 3
        .endm
        .macro HR_ZERO_ACC
        v_mov_b32 v120, 0
        v_mov_b32 v121, 0
        v_mov_b32 v122, 0
        v_mov_b32 v123, 0
        v_mov_b32 v124, 0
        v_mov_b32 v125, 0
        v_mov_b32 v126, 0
        v_mov_b32 v127, 0
        v_mov_b32 v128, 0
        v_mov_b32 v129, 0
        v_mov_b32 v130, 0
        v_mov_b32 v131, 0
        v_mov_b32 v132, 0
        v_mov_b32 v133, 0
        v_mov_b32 v134, 0
        v_mov_b32 v135, 0
        .endm
        ; four registers src.. -> ring registers index.. (relative destination), then index += 4 with wrap
        .macro HR_PUT4 src
        v_mov_b32 v0, v[\src]
        v_mov_b32 v1, v[\src+1]
        v_mov_b32 v2, v[\src+2]
        v_mov_b32 v3, v[\src+3]
        s_add_u32 s96, s96, 4
        s_cmp_eq_u32 s96, s29
        s_cselect_b32 s96, 0, s96
        s_set_gpr_idx_idx s96
        .endm
        ; group s97 (clamped into the row: a group outside it is never inside a window, any finite values do) -> landing registers
        .macro HR_LOAD_GROUP lb, k
        s_max_i32 s98, s97, 0
        s_min_i32 s98, s98, %[Gm1]
        s_add_u32 s98, s98, %[yG]
        s_lshl_b32 s98, s98, 8
        .if (HR_EXP & 2) == 0
        buffer_load_dwordx4 v[\lb+4*\k:\lb+4*\k+3], %[vload], s[16:19], s98 offen
        .endif
        s_add_i32 s97, s97, 1
        .endm
        ; chunk b against the running minimum of pixel i (registers 120 + i, winner 138 + i): strictly lower wins (d_dc_wta.cu:28)
        .macro HR_WTA_CHUNK b, dsrc
        v_cmp_gt_f32_e64 s[94:95], v120, v[120+4*\b]
        v_cmp_gt_f32_e64 s[96:97], v121, v[121+4*\b]
        v_cmp_gt_f32_e64 s[98:99], v122, v[122+4*\b]
        v_cmp_gt_f32_e64 s[90:91], v123, v[123+4*\b]
        v_cndmask_b32_e64 v120, v120, v[120+4*\b], s[94:95]
        v_cndmask_b32_e64 v121, v121, v[121+4*\b], s[96:97]
        v_cndmask_b32_e64 v122, v122, v[122+4*\b], s[98:99]
        v_cndmask_b32_e64 v123, v123, v[123+4*\b], s[90:91]
        .if \b == 1
        v_cndmask_b32_e64 v138, %[vn], v[\dsrc], s[94:95]
        v_cndmask_b32_e64 v139, %[vn], v[\dsrc], s[96:97]
        v_cndmask_b32_e64 v140, %[vn], v[\dsrc], s[98:99]
        v_cndmask_b32_e64 v141, %[vn], v[\dsrc], s[90:91]
        .else
        v_cndmask_b32_e64 v138, v138, v[\dsrc], s[94:95]
        v_cndmask_b32_e64 v139, v139, v[\dsrc], s[96:97]
        v_cndmask_b32_e64 v140, v140, v[\dsrc], s[98:99]
        v_cndmask_b32_e64 v141, v141, v[\dsrc], s[90:91]
        .endif
        .endm
        ; minimum over the 16 lanes of a pixel quad (a DPP row), four independent registers interleaved (no hazard padding)
        .macro HR_ROWMIN4 d0, d1, d2, d3, s0, s1, s2, s3
        v_min_i32_dpp v[\d0], v[\s0], v[\s0] row_ror:8 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d1], v[\s1], v[\s1] row_ror:8 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d2], v[\s2], v[\s2] row_ror:8 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d3], v[\s3], v[\s3] row_ror:8 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d0], v[\d0], v[\d0] row_ror:4 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d1], v[\d1], v[\d1] row_ror:4 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d2], v[\d2], v[\d2] row_ror:4 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d3], v[\d3], v[\d3] row_ror:4 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d0], v[\d0], v[\d0] row_ror:2 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d1], v[\d1], v[\d1] row_ror:2 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d2], v[\d2], v[\d2] row_ror:2 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d3], v[\d3], v[\d3] row_ror:2 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d0], v[\d0], v[\d0] row_ror:1 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d1], v[\d1], v[\d1] row_ror:1 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d2], v[\d2], v[\d2] row_ror:1 row_mask:0xf bank_mask:0xf
        v_min_i32_dpp v[\d3], v[\d3], v[\d3] row_ror:1 row_mask:0xf bank_mask:0xf
        .endm

        ; one step of the walk: tile s24; lb = the landing registers this step empties and refills
        .macro HR_STEP lb
        ; the groups loaded two steps ago (4 t + 9 .. 4 t + 12) -> ring (the first tile's came with the prologue)
        s_cmp_eq_u32 s24, %[t0]
        s_cbranch_scc1 HR_nomove_%=_\@
        .if (HR_EXP & 2) == 0
        s_waitcnt vmcnt(12)                   ; issued since: 4 stores, 4 loads, 4 stores
        .endif
        s_add_u32 s96, s26, 72                ; group 4 t + 9 = range start + 18 groups
        s_sub_u32 s31, s96, s29
        s_cmp_ge_u32 s96, s29
        s_cselect_b32 s96, s31, s96
        s_set_gpr_idx_on s96, 0x8
        HR_PUT4 \lb
        HR_PUT4 \lb+4
        HR_PUT4 \lb+8
        HR_PUT4 \lb+12
        s_set_gpr_idx_off
HR_nomove_%=_\@:
        ; groups 4 t + 17 .. 4 t + 20 (the new groups of tile t + 2) -> the landing registers just emptied
        s_lshl_b32 s97, s24, 2
        s_add_i32 s97, s97, 17
        HR_LOAD_GROUP \lb, 0
        HR_LOAD_GROUP \lb, 1
        HR_LOAD_GROUP \lb, 2
        HR_LOAD_GROUP \lb, 3
        ; ------------------------------------------------------------ the sweep of tile t (header and masks left after the last sweep)
        s_waitcnt lgkmcnt(0)
        s_mov_b32 s84, s86
        s_min_u32 s85, s87, 22
        s_cmp_eq_u32 s85, 0
        s_cbranch_scc1 HR_zero_%=_\@
        .if (HR_EXP & 1) == 0
        HR_SWEEP_ISSUE
        HR_SWEEP_RUN
        s_branch HR_wta_%=_\@
        .endif
HR_zero_%=_\@:
        HR_ZERO_ACC
HR_wta_%=_\@:
        ; the next tile's header and masks travel during WTA and the next ring update (the last tile reads its own again)
        s_add_i32 s31, s24, 1
        s_sub_u32 s91, %[t1], 1
        s_min_i32 s31, s31, s91
        s_sub_i32 s31, s31, %[t0]
        HR_NEXT_MASKS
        .if (HR_EXP & 4) == 0
        ; ------------------------------------------------------------ WTA (d_dc_wta.cu:19-34)
        s_cmp_eq_u32 s27, 0
        s_cbranch_scc1 HR_nofix_%=_\@
        v_max_f32 v120, v120, %[vb0]
        v_max_f32 v121, v121, %[vb0]
        v_max_f32 v122, v122, %[vb0]
        v_max_f32 v123, v123, %[vb0]
        v_max_f32 v124, v124, %[vb1]
        v_max_f32 v125, v125, %[vb1]
        v_max_f32 v126, v126, %[vb1]
        v_max_f32 v127, v127, %[vb1]
        v_max_f32 v128, v128, %[vb2]
        v_max_f32 v129, v129, %[vb2]
        v_max_f32 v130, v130, %[vb2]
        v_max_f32 v131, v131, %[vb2]
        v_max_f32 v132, v132, %[vb3]
        v_max_f32 v133, v133, %[vb3]
        v_max_f32 v134, v134, %[vb3]
        v_max_f32 v135, v135, %[vb3]
HR_nofix_%=_\@:
        ; per lane: the lowest of its four chunks, the lower chunk on ties; v[138:141] = that hypothesis (d = 16 b + n)
        HR_WTA_CHUNK 1, 151
        HR_WTA_CHUNK 2, 152
        HR_WTA_CHUNK 3, 153
        ; over the 16 hypothesis lanes: lowest cost (bit patterns: costs are never negative), then the lowest d among its holders
        HR_ROWMIN4 142, 143, 144, 145, 120, 121, 122, 123
        v_cmp_eq_u32_e64 s[94:95], v142, v120
        v_cmp_eq_u32_e64 s[96:97], v143, v121
        v_cmp_eq_u32_e64 s[98:99], v144, v122
        v_cmp_eq_u32_e64 s[90:91], v145, v123
        v_cndmask_b32_e64 v146, v150, v138, s[94:95]
        v_cndmask_b32_e64 v147, v150, v139, s[96:97]
        v_cndmask_b32_e64 v148, v150, v140, s[98:99]
        v_cndmask_b32_e64 v149, v150, v141, s[90:91]
        HR_ROWMIN4 142, 143, 144, 145, 146, 147, 148, 149
        v_cvt_f32_i32 v142, v142
        v_cvt_f32_i32 v143, v143
        v_cvt_f32_i32 v144, v144
        v_cvt_f32_i32 v145, v145
        v_subrev_f32 v142, %[zdf], v142
        v_subrev_f32 v143, %[zdf], v143
        v_subrev_f32 v144, %[zdf], v144
        v_subrev_f32 v145, %[zdf], v145
        .endif
        ; lane n == 0 of a pixel quad stores pixels 16 t + 4 q + i; pixels past the row are out of range (the whole offset is in
        ; the vector register, so the range check sees it).  Always four stores: a step's loads are counted against them
        s_lshl_b32 s31, s24, 6
        v_add_u32 v146, s31, %[vst]
        s_nop 0
        buffer_store_dword v142, v146, s[20:23], 0 offen
        buffer_store_dword v143, v146, s[20:23], 0 offen offset:4
        buffer_store_dword v144, v146, s[20:23], 0 offen offset:8
        buffer_store_dword v145, v146, s[20:23], 0 offen offset:12
        ; next tile
        s_add_u32 s26, s26, 16
        s_sub_u32 s31, s26, s29
        s_cmp_ge_u32 s26, s29
        s_cselect_b32 s26, s31, s26
        s_add_i32 s24, s24, 1
        .endm

        ; ---------------------------------------------------------------- setup
        s_mov_b64 s[16:17], %[in]
        s_and_b32 s17, s17, 0xffff
        s_mov_b32 s18, %[nrec]
        s_mov_b32 s19, 0x20000
        s_mov_b64 s[20:21], %[disp]
        s_and_b32 s21, s21, 0xffff
        s_mov_b32 s22, 0
        s_mov_b32 s23, 0x20000
        s_mov_b32 s24, %[t0]
        s_mov_b32 s25, %[t1]
        s_mov_b32 s26, 0                      ; the ring starts at the first tile's range start: group 4 t0 - 9 + k in registers 4 k ..
        s_movk_i32 s29, 88
        s_mov_b32 s31, 0
        HR_NEXT_MASKS
        v_mov_b32 v150, 0x7fffffff
        v_add_u32 v151, 16, %[vn]
        v_add_u32 v152, 32, %[vn]
        v_add_u32 v153, 48, %[vn]
        v_or_b32 v146, %[vb0], %[vb1]
        v_or3_b32 v146, v146, %[vb2], %[vb3]
        v_cmp_ne_u32 vcc, 0, v146
        s_cmp_lg_u64 vcc, 0
        s_cselect_b32 s27, 1, 0               ; D < 64: some lanes hold hypotheses that must not win
        ; the first tile's 22 groups straight into the ring, the second tile's new groups into the odd step's landing registers
        s_lshl_b32 s97, %[t0], 2
        s_sub_i32 s97, s97, 9
        HR_LOAD_GROUP 0, 0
        HR_LOAD_GROUP 0, 1
        HR_LOAD_GROUP 0, 2
        HR_LOAD_GROUP 0, 3
        HR_LOAD_GROUP 0, 4
        HR_LOAD_GROUP 0, 5
        HR_LOAD_GROUP 0, 6
        HR_LOAD_GROUP 0, 7
        HR_LOAD_GROUP 0, 8
        HR_LOAD_GROUP 0, 9
        HR_LOAD_GROUP 0, 10
        HR_LOAD_GROUP 0, 11
        HR_LOAD_GROUP 0, 12
        HR_LOAD_GROUP 0, 13
        HR_LOAD_GROUP 0, 14
        HR_LOAD_GROUP 0, 15
        HR_LOAD_GROUP 0, 16
        HR_LOAD_GROUP 0, 17
        HR_LOAD_GROUP 0, 18
        HR_LOAD_GROUP 0, 19
        HR_LOAD_GROUP 0, 20
        HR_LOAD_GROUP 0, 21
        HR_LOAD_GROUP 104, 0
        HR_LOAD_GROUP 104, 1
        HR_LOAD_GROUP 104, 2
        HR_LOAD_GROUP 104, 3
        ; four stores into an empty buffer keep the steps' count of outstanding operations uniform (4 loads + 4 stores each)
        buffer_store_dword v142, v146, s[20:23], 0 offen
        buffer_store_dword v142, v146, s[20:23], 0 offen
        buffer_store_dword v142, v146, s[20:23], 0 offen
        buffer_store_dword v142, v146, s[20:23], 0 offen
        s_mov_b32 s22, %[W4]
        .if (HR_EXP & 2) == 0
        s_waitcnt vmcnt(8)                    ; the ring is in
        .endif
HR_loop_%=:
        HR_STEP 88
        s_cmp_ge_i32 s24, s25
        s_cbranch_scc1 HR_done_%=
        HR_STEP 104
        s_cmp_lt_i32 s24, s25
        s_cbranch_scc1 HR_loop_%=
HR_done_%=:
        s_waitcnt vmcnt(0) lgkmcnt(0)         ; nothing may be in flight when the wave ends
        .purgem HR_BLOCK
        .purgem HR_PAIR
        .purgem HR_SWEEP_ISSUE
        .purgem HR_NEXT_MASKS
        .purgem HR_SWEEP_RUN
        .purgem HR_ZERO_ACC
        .purgem HR_PUT4
        .purgem HR_LOAD_GROUP
        .purgem HR_WTA_CHUNK
        .purgem HR_ROWMIN4
        .purgem HR_STEP
        )ASM"
                 :
                 : [exp] "n"(EXP), [in] "s"(in), [disp] "s"(disp_row), [hrec_lo] "s"((uint32_t)(uintptr_t)hrec), [hrec_hi] "s"((uint32_t)((uintptr_t)hrec >> 32)),
                   [yG] "s"(yG), [Gm1] "s"(Gm1), [nrec] "s"(nrec), [W4] "s"(W4), [t0] "s"(t0), [t1] "s"(t1), [zdf] "s"(zdf),
                   [vload] "v"(vload), [vn] "v"(vn), [vst] "v"(vst), [vb0] "v"(vbig[0]),
                   [vb1] "v"(vbig[1]), [vb2] "v"(vbig[2]), [vb3] "v"(vbig[3])
                 : "memory", "scc", "vcc",
                   "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s29", "s30", "s31", "s36", "s37", "s38", "s39",
                   "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59",
                   "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
                   "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
                   "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                   "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",
                   "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
                   "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                   "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99",
                   "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
                   "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133",
                   "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150",
                   "v151", "v152", "v153");
}

bool aggh_supports(int usd, int D) { return usd >= 1 && usd <= HR_TOP && D >= 1 && D <= 64; }

size_t aggh_table_dwords(int nviews, int H, int W) { return (size_t)nviews * H * cdiv(W, 16) * HR_REC + 64; }

// the horizontal window table of `nviews` views (arms in v); `tab` must hold aggh_table_dwords
void launch_hwin_table(PQViews &v, int nviews, uint32_t *tab, int H, int W)
{
    const int nTx = cdiv(W, 16);
    STM_LAUNCH(stm_k_hwin_table, dim3(cdiv(nTx, 16), H, nviews), dim3(256), 0, stream(), v, tab, H, W, nTx);
    STM_CHECK_LAUNCH();
}

// last horizontal pass + WTA, vol_a -> disparities, for `nviews` views
void launch_pq_hsr(PQViews &v, int nviews, const uint32_t *tab, int D, int zd, int H, int W)
{
    const int G = (W + 3) / 4, NC = (D + 15) / 16, nTx = cdiv(W, 16);
    // parts per row: about ten tiles per wave measured best at 1080p (0.275 ms against 0.303 with 30 and 0.342 with 5: a wave's
    // start costs one round trip for the 26 loads of its first ring, its end leaves its slot empty for the rest of the launch)
    int parts = (nTx + 5) / 10;
    parts = parts < 1 ? 1 : parts;
    const dim3 grid(nviews * H * parts);
#ifdef STM_TIMING
    if (const char *e = getenv("STM_HSR_PARTS")) parts = atoi(e);
    const dim3 gridt(nviews * H * parts);
    switch (timing_knobs()) {
    case 1: STM_LAUNCH(stm_k_pq_hsr<1>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    case 2: STM_LAUNCH(stm_k_pq_hsr<2>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    case 3: STM_LAUNCH(stm_k_pq_hsr<3>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    case 4: STM_LAUNCH(stm_k_pq_hsr<4>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    case 5: STM_LAUNCH(stm_k_pq_hsr<5>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    case 6: STM_LAUNCH(stm_k_pq_hsr<6>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    case 7: STM_LAUNCH(stm_k_pq_hsr<7>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    case 8: STM_LAUNCH(stm_k_pq_hsr<8>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    default: STM_LAUNCH(stm_k_pq_hsr<0>, gridt, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews); break;
    }
#else
    STM_LAUNCH(stm_k_pq_hsr<0>, grid, dim3(64), 0, stream(), v, tab, D, zd, H, W, G, NC, nTx, parts, nviews);
#endif
    STM_CHECK_LAUNCH();
}

} // namespace stm
