// stm_kernels_aggv.hip -- both vertical aggregation passes of the frame pipeline, the rows of a strip held in REGISTERS.
//
// Reference stage replaced: ca_cross_vhsum_kernel_2 x 2 + both transposes,
// d_ca_cross_sum.cu:148-198 (window [y - armU, y + armD), ascending float32 adds), order d_ca_cross.cu:258-267.
//
// Round 3's fused vertical kernel (stm_k_pq_v12t, stm_kernels_aggm.hip) streams a strip through two LDS rings shared by the
// six waves of a block, two barriers per step; its matrix pipe was busy 48 % of the time (profiles/r03_pmc_sq_aggm.txt).
// Here a strip of 4 columns x 16 hypotheses belongs to ONE wave and nothing is shared:
//  * a row of the strip is 64 floats = ONE vector register, laid out exactly as the B operand of v_mfma_f32_16x16x1_4B_f32
//    wants it (lane 16 b + n = column b, hypothesis n).  The CU's register file (512 KB) is three times its LDS: the 88 input
//    rows a tile's sweep can touch and the 100 first-pass rows the second pass needs are two rings of registers per wave;
//  * no LDS, no barrier: a wave walks down its strip, per tile of 16 rows one first-pass sweep (ring 1 -> accumulators), a
//    16-instruction transposition of the accumulators into row registers (v_permlane32_swap / 16_swap), and one second-pass
//    sweep three tiles behind (ring 2 -> accumulators -> HBM);
//  * the B operand of an MFMA is addressed through the VGPR INDEX MODE (s_set_gpr_idx_on, src1 relative: measured to apply to
//    v_mfma on gfx950, tools/gpridx_probe.hip), so a sweep is ONE sequence of 22 quad blocks for every ring position;
//  * a sweep contains NO branch.  What a sweep costs is the length of the wave's own instruction stream: its MFMAs form one
//    dependent chain, issue is in order, and only what sits BETWEEN two MFMAs hides behind them.  A never-taken branch per
//    quad costs 5-14 cycles per MFMA, selects / scalar work / waits bunched between quads are added to the chain
//    (tools/quad_probe.hip: 38.6 cycles per MFMA for the block below against 53 with guards, waits and selects between the
//    quads; the compiler-scheduled C++ version of this kernel ran at 61).  So a sweep of n quads ENTERS the sequence at block
//    22 - n with one computed jump and runs to its end; a block is
//        s_set_gpr_idx_idx | MFMA | [odd block: wait for the other mask set] 4 selects for the NEXT block | MFMA |
//        [odd block: reload the set that became free, two batches ahead] | ring index += 4 (three scalar instructions) | MFMA | MFMA
//    and every block has the same size (92 bytes), so the entry point is base + 92 p + 12 (the first MFMA of a sweep is issued
//    by the prologue with the constant 0 as accumulator input: nothing is cleared);
//  * input rows arrive by buffer_load_dword in landing registers TWO steps ahead (32 rows = 8 KB in flight per wave, 64 KB per
//    CU: with one step ahead the waves spent 19 % of their time waiting for them) and are moved into the ring with the index
//    mode's relative destination.
// The whole walk is one asm block with its own register allocation (the compiler cannot be made to produce the block above:
// it merges, hoists and re-orders around guards, and re-materialises accumulators at merge points).
// Masks: the window table of stm_k_vwin_table in its static layout (one 64-bit lane mask per tile and window row, at the
// row's position inside the tile's range), read in batches of two quads (s_load_dwordx16) a batch ahead.
// Results are bit-identical to stm_k_pq_v12t (same chains, same order).  Limits: usd <= 36 (sweep range of 88 rows); longer
// arms run stm_k_pq_v12t.
#include "stm_common.h"

namespace stm {

constexpr int VR_TOP = 36; // a tile's sweep range starts this many rows above the tile: rows [16 u - 36, 16 u + 52)
constexpr int VR_NQ = 22;  // quads of rows in that range

// Register map of the asm block.
//   v[0:87]    ring 1: input row y in register y mod 88 (rows [16 u - 36, 16 u + 52) are live during step u)
//   v[88:103], v[228:243] landing registers of the 16 rows loaded during an even / odd step (rows [16 u + 68, 16 u + 84): two
//              steps ahead of their move into the ring -- 8 KB in flight per wave, 64 KB per CU)
//   v[104:203] ring 2: first-pass row y in register 104 + y mod 100 (the second pass of tile u - 3 reads [16 u - 84, 16 u + 4))
//   v[204:219] accumulators;  v[220:223], v[224:227] the A operands (1.0 / 0.0 per lane) of the even / odd blocks
//   v[244:245] store addresses (the store tuples are the A registers)
//   s[16:19] / s[20:23] buffer descriptors (input strip, output strip); s24 u (step), s25 / s26 ring index of row 16 u + 36 in
//   ring 1 / of row 16 u in ring 2, s27 last step + 1, s29 ring size of the running sweep, s30 ring index of the running block,
//   s[36:51] / s[52:67] mask sets of the even / odd batches, s[68:75] masks of a sweep's first quad, s[76:77] mask address of
//   block 0, s[78:79] jump target, s80 entry block, s81..s84 this step's headers (q0, n of either pass), s[86:89] the next
//   step's, s90.. temporaries, s94..s99 sweep parameters (q0, n, ring index of the range start, -, mask base), s28 u - 3
//   (s32..s35 are left alone: the ABI's stack registers).
// EXP: timing experiments (libstm_hip_timing.so only; results NOT valid): 1 = no sweeps, 2 = no loads / stores, 4 = sweeps without
// their mask waits and loads, 8 = no transposition / ring moves.  The product library instantiates EXP = 0 only.
template <int EXP>
__global__ __launch_bounds__(64, 2) void stm_k_pq_v12r(PQViews pv, const uint32_t *__restrict__ wtab, int rec, int H, int G, int NC, int nviews)
{
    // block (one wave) -> (view, group, chunk); the NC chunk waves of a strip read the same window records: consecutive blocks
    // Consecutive blocks go to different XCDs (8, each with its own L2), but the NC chunk waves of a strip read the same window
    // records: the work items are dealt out so that a strip's chunks follow each other on ONE XCD
    int wi = blockIdx.x;
    {
        const int per_xcd = gridDim.x >> 3;
        if (wi < 8 * per_xcd) wi = (wi & 7) * per_xcd + (wi >> 3);
    }
    const int c = wi % NC, sidx = wi / NC;
    if (sidx >= G * nviews) return;
    const int g = sidx % G, view = sidx / G;
    const int l = threadIdx.x;
    const int nT = (H + 15) >> 4;
    const int rsb = G * 256; // bytes between consecutive rows of the strip
    const size_t strip = ((size_t)c * H * G + g) * 64; // float index of (chunk c, row 0, group g, hypothesis 0)
    const float *in = (const float *)(view ? pv.b[1] : pv.b[0]) + strip;
    float *out = (float *)(view ? pv.a[1] : pv.a[0]) + strip;
    const uint32_t range = (uint32_t)(H - 1) * (uint32_t)rsb + 256u; // bytes of a strip up to the end of its last row
    const uint32_t *trow = wtab + ((size_t)view * nT * G + g) * rec; // record of tile 0; tile u at + u * G * rec dwords
    const int tstep = G * rec * 4;
    const int voff = (l & 15) * 16 + (l >> 4) * 4;           // lane 16 b + n loads column b of hypothesis n: float4 n, element b
    const int vst = 4 * (l >> 4) * rsb + (l & 15) * 16;      // lane 16 q + n stores row 4 q (+ i) of hypothesis n
    asm volatile(R"ASM(
        .set VR_EXP, %[exp]
        ; ---------------------------------------------------------------- macros
        ; one block of a sweep.  p = block index, rb = first register of the ring, acur / anxt = first A register this block
        ; uses / prepares, cur / nxt = first SGPR of the mask set of this block's batch / of the other set
        .macro VR_BLOCK p, rb, acur, anxt, cur, nxt
        s_set_gpr_idx_idx s30
        v_mfma_f32_16x16x1_4b_f32 v[204:219], v[\acur], v[\rb], v[204:219]
        .if (\p) & 1
        .if (VR_EXP & 4) == 0
        s_waitcnt lgkmcnt(0)
        .else
        s_nop 0
        .endif
        v_cndmask_b32_e64 v[\anxt], 0, 1.0, s[\nxt:\nxt+1]
        v_cndmask_b32_e64 v[\anxt+1], 0, 1.0, s[\nxt+2:\nxt+3]
        v_cndmask_b32_e64 v[\anxt+2], 0, 1.0, s[\nxt+4:\nxt+5]
        v_cndmask_b32_e64 v[\anxt+3], 0, 1.0, s[\nxt+6:\nxt+7]
        .else
        v_cndmask_b32_e64 v[\anxt], 0, 1.0, s[\cur+8:\cur+9]
        v_cndmask_b32_e64 v[\anxt+1], 0, 1.0, s[\cur+10:\cur+11]
        v_cndmask_b32_e64 v[\anxt+2], 0, 1.0, s[\cur+12:\cur+13]
        v_cndmask_b32_e64 v[\anxt+3], 0, 1.0, s[\cur+14:\cur+15]
        .endif
        v_mfma_f32_16x16x1_4b_f32 v[204:219], v[\acur+1], v[\rb+1], v[204:219]
        .if (\p) & 1
        .if ((\p) <= 17) && ((VR_EXP & 4) == 0)
        s_load_dwordx16 s[\cur:\cur+15], s[76:77], 64*((\p)/2+2)
        .else
        s_nop 0
        s_nop 0
        .endif
        .endif
        s_add_u32 s30, s30, 4
        s_cmp_eq_u32 s30, s29
        s_cselect_b32 s30, 0, s30
        v_mfma_f32_16x16x1_4b_f32 v[204:219], v[\acur+2], v[\rb+2], v[204:219]
        v_mfma_f32_16x16x1_4b_f32 v[204:219], v[\acur+3], v[\rb+3], v[204:219]
        .if ((\p) & 1) == 0
        s_nop 0
        s_nop 0
        s_nop 0
        .endif
        .endm
        ; a pair of blocks = batch b: even block (A operands v220.., prepares v224..), odd block (the reverse)
        .macro VR_PAIR b, rb
        .if (\b) & 1
        VR_BLOCK 2*(\b), \rb, 220, 224, 52, 36
        VR_BLOCK 2*(\b)+1, \rb, 224, 220, 52, 36
        .else
        VR_BLOCK 2*(\b), \rb, 220, 224, 36, 52
        VR_BLOCK 2*(\b)+1, \rb, 224, 220, 36, 52
        .endif
        .endm
        ; the loads a sweep starts from.  in: s94 = q0 (first quad of the sweep inside the tile's range), s95 = n (quads, >= 1),
        ; s96 = ring index of the range's first row, s29 = ring size, s[98:99] = masks of range quad 0.
        ; out: s80 = entry block 22 - n, s30 = ring index of the sweep's first row, s[76:77] = mask address of block 0;
        ; in flight: the even batch of {b0, b0 + 1} -> s[36:51], the odd one -> s[52:67] (b0 = batch of the entry block), the
        ; first quad's own masks -> s[68:75]
        .macro VR_SWEEP_ISSUE
        s_min_u32 s95, s95, 22                ; (the table cannot hold more: the jump below must stay inside the sequence)
        s_sub_u32 s80, 22, s95
        s_lshl_b32 s31, s94, 2
        s_add_u32 s30, s96, s31
        s_sub_u32 s31, s30, s29
        s_cmp_ge_u32 s30, s29
        s_cselect_b32 s30, s31, s30
        s_sub_i32 s31, s94, s80
        s_lshl_b32 s31, s31, 5
        s_ashr_i32 s97, s31, 31
        s_add_u32 s76, s98, s31
        s_addc_u32 s77, s99, s97
        s_lshr_b32 s31, s80, 1
        s_add_u32 s97, s31, 1
        s_and_b32 s97, s97, -2
        s_lshl_b32 s97, s97, 6
        s_load_dwordx16 s[36:51], s[76:77], s97
        s_or_b32 s97, s31, 1
        s_lshl_b32 s97, s97, 6
        s_load_dwordx16 s[52:67], s[76:77], s97
        s_lshl_b32 s97, s80, 5
        s_load_dwordx8 s[68:75], s[76:77], s97
        .endm
        ; the sweep: prologue (first A operands, first MFMA with the constant 0 as accumulator input), computed jump into the
        ; sequence of blocks, epilogue (the accumulators are read by vector instructions next: the last MFMA must have left
        ; the pipe)
        .macro VR_SWEEP_RUN rb
        s_waitcnt lgkmcnt(0)
        v_cndmask_b32_e64 v220, 0, 1.0, s[68:69]
        v_cndmask_b32_e64 v221, 0, 1.0, s[70:71]
        v_cndmask_b32_e64 v222, 0, 1.0, s[72:73]
        v_cndmask_b32_e64 v223, 0, 1.0, s[74:75]
        v_cndmask_b32_e64 v224, 0, 1.0, s[68:69]
        v_cndmask_b32_e64 v225, 0, 1.0, s[70:71]
        v_cndmask_b32_e64 v226, 0, 1.0, s[72:73]
        v_cndmask_b32_e64 v227, 0, 1.0, s[74:75]
        s_set_gpr_idx_on s30, 0x2
        s_mul_i32 s31, s80, 92
        s_getpc_b64 s[78:79]
VR_pc_%=_\@:
        s_add_u32 s31, s31, VR_blk0_%=_\@-VR_pc_%=_\@+12
        s_add_u32 s78, s78, s31
        s_addc_u32 s79, s79, 0
        v_mfma_f32_16x16x1_4b_f32 v[204:219], v220, v[\rb], 0
        s_setpc_b64 s[78:79]
VR_blk0_%=_\@:
        VR_PAIR 0, \rb
        VR_PAIR 1, \rb
        VR_PAIR 2, \rb
        VR_PAIR 3, \rb
        VR_PAIR 4, \rb
        VR_PAIR 5, \rb
        VR_PAIR 6, \rb
        VR_PAIR 7, \rb
        VR_PAIR 8, \rb
        VR_PAIR 9, \rb
        VR_PAIR 10, \rb
VR_end_%=_\@:
        .if (VR_end_%=_\@-VR_blk0_%=_\@) != 22*92
        .error "sweep blocks are not 92 bytes each"
        .endif
        s_set_gpr_idx_off
        s_nop 15
        s_nop 3
        .endm
        .macro VR_ZERO_ACC
        v_mov_b32 v204, 0
        v_mov_b32 v205, 0
        v_mov_b32 v206, 0
        v_mov_b32 v207, 0
        v_mov_b32 v208, 0
        v_mov_b32 v209, 0
        v_mov_b32 v210, 0
        v_mov_b32 v211, 0
        v_mov_b32 v212, 0
        v_mov_b32 v213, 0
        v_mov_b32 v214, 0
        v_mov_b32 v215, 0
        v_mov_b32 v216, 0
        v_mov_b32 v217, 0
        v_mov_b32 v218, 0
        v_mov_b32 v219, 0
        .endm
        ; four registers src.. -> ring registers rb + index.. (relative destination), then index += 4 with wrap at `size` (s31)
        .macro VR_PUT4 rb, src
        v_mov_b32 v[\rb], v[\src]
        v_mov_b32 v[\rb+1], v[\src+1]
        v_mov_b32 v[\rb+2], v[\src+2]
        v_mov_b32 v[\rb+3], v[\src+3]
        s_add_u32 s92, s92, 4
        s_cmp_eq_u32 s92, s31
        s_cselect_b32 s92, 0, s92
        s_set_gpr_idx_idx s92
        .endm
        .macro VR_LOAD_ROW lb, k
        s_cmp_lt_u32 s90, %[H]
        s_cselect_b32 s18, %[range], 0
        s_mul_i32 s91, s90, %[rsb]
        .if (VR_EXP & 2) == 0
        buffer_load_dword v[\lb+\k], %[voff], s[16:19], s91 offen nt
        .endif
        s_add_u32 s90, s90, 1
        .endm
        ; (the store tuples are the A registers, free between two sweeps; a tuple is rewritten eight instructions after its store)
        .macro VR_LOAD_FAST lb, k
        .if (VR_EXP & 2) == 0
        buffer_load_dword v[\lb+\k], %[voff], s[16:19], s91 offen nt
        .endif
        s_add_u32 s91, s91, %[rsb]
        .endm
        .macro VR_STORE_ROWS i
        v_mov_b32 v[220+4*((\i)&1)], v[204+\i]
        v_mov_b32 v[221+4*((\i)&1)], v[208+\i]
        v_mov_b32 v[222+4*((\i)&1)], v[212+\i]
        v_mov_b32 v[223+4*((\i)&1)], v[216+\i]
        v_add_u32 v[244+((\i)&1)], s85, %[vst]
        s_add_u32 s85, s85, %[rsb]
        s_nop 0
        .if (VR_EXP & 2) == 0
        buffer_store_dwordx4 v[220+4*((\i)&1):223+4*((\i)&1)], v[244+((\i)&1)], s[20:23], 0 offen nt
        .endif
        .endm

        ; ---------------------------------------------------------------- setup
        s_mov_b64 s[16:17], %[in]
        s_and_b32 s17, s17, 0xffff
        s_mov_b32 s18, 0
        s_mov_b32 s19, 0x20000
        s_mov_b64 s[20:21], %[out]
        s_and_b32 s21, s21, 0xffff
        s_mov_b32 s22, 0
        s_mov_b32 s23, 0x20000
        s_mov_b32 s24, -5                     ; u: five steps that only bring rows [0, 52) into ring 1
        s_mov_b32 s25, 44                     ; ring-1 index of row 16 u + 36 = -44
        s_mov_b32 s26, 20                     ; ring-2 index of row 16 u = -80
        s_add_u32 s27, %[nT], 3               ; the second pass runs three tiles behind
        s_mov_b64 s[86:87], 0
        s_mov_b64 s[88:89], 0
        s_mov_b32 s81, 0                      ; the first step has no tiles
        s_mov_b32 s82, 0
        s_mov_b32 s83, 0
        s_mov_b32 s84, 0
        s_sub_i32 s28, s24, 3
        .macro VR_STEP lb
        ; s81..s84 = (q0, n) of this step's first-pass tile u and second-pass tile u - 3, s28 = u - 3: set at the end of the step
        ; before, where the first masks of this step's first pass were requested too
        ; ------------------------------------------------------------ the rows loaded during the last step -> ring 1
        s_cmp_lt_i32 s24, -3
        s_cbranch_scc1 VR_nocopy_%=_\@
        .if (VR_EXP & 2) == 0
        s_waitcnt vmcnt(24)                   ; the 16 loads of the step before the last (issued since: 4 + 16 + 4 stores and loads)
        .endif
        .if (VR_EXP & 8) == 0
        s_mov_b32 s92, s25
        s_movk_i32 s31, 88
        s_set_gpr_idx_on s92, 0x8
        VR_PUT4 0, \lb
        VR_PUT4 0, \lb+4
        VR_PUT4 0, \lb+8
        VR_PUT4 0, \lb+12
        s_set_gpr_idx_off
        .endif
VR_nocopy_%=_\@:
        s_cmp_lt_i32 s24, 0
        s_cbranch_scc1 VR_loads_%=_\@
        ; ------------------------------------------------------------ first pass of tile u
        s_cmp_eq_u32 s82, 0
        s_cbranch_scc1 VR_zero1_%=_\@
        VR_SWEEP_RUN 0
        s_branch VR_p1done_%=_\@
VR_zero1_%=_\@:
        VR_ZERO_ACC
VR_p1done_%=_\@:
        ; first masks of the second pass (they travel during the transposition)
        s_cmp_eq_u32 s84, 0
        s_cbranch_scc1 VR_noissue2_%=_\@
        s_mov_b32 s94, s83
        s_mov_b32 s95, s84
        s_add_u32 s96, s26, 16                ; row 16 u - 84, and -84 = 16 (mod 100)
        s_sub_u32 s31, s96, 100
        s_cmp_ge_u32 s96, 100
        s_cselect_b32 s96, s31, s96
        s_movk_i32 s29, 100
        s_mul_i32 s92, s28, %[tstep]
        s_add_u32 s92, s92, 32
        s_add_u32 s98, %[trow_lo], s92
        s_addc_u32 s99, %[trow_hi], 0
        VR_SWEEP_ISSUE
VR_noissue2_%=_\@:
        ; accumulators (register 4 b + i of lane 16 q + n = [row 4 q + i][column b][hypothesis n]) -> row registers (register
        ; 4 q + i of lane 16 b + n): for each i a 4 x 4 transposition of (register b, lane group q)
        .if (VR_EXP & 8) == 0
        v_permlane32_swap_b32 v204, v212
        v_permlane32_swap_b32 v208, v216
        v_permlane32_swap_b32 v205, v213
        v_permlane32_swap_b32 v209, v217
        v_permlane32_swap_b32 v206, v214
        v_permlane32_swap_b32 v210, v218
        v_permlane32_swap_b32 v207, v215
        v_permlane32_swap_b32 v211, v219
        s_nop 1
        v_permlane16_swap_b32 v204, v208
        v_permlane16_swap_b32 v212, v216
        v_permlane16_swap_b32 v205, v209
        v_permlane16_swap_b32 v213, v217
        v_permlane16_swap_b32 v206, v210
        v_permlane16_swap_b32 v214, v218
        v_permlane16_swap_b32 v207, v211
        v_permlane16_swap_b32 v215, v219
        s_nop 1
        ; rows [16 u, 16 u + 16) of the first pass -> ring 2
        s_mov_b32 s92, s26
        s_movk_i32 s31, 100
        s_set_gpr_idx_on s92, 0x8
        VR_PUT4 104, 204
        VR_PUT4 104, 208
        VR_PUT4 104, 212
        VR_PUT4 104, 216
        s_set_gpr_idx_off
        .endif
VR_loads_%=_\@:
        ; ------------------------------------------------------------ rows [16 u + 68, 16 u + 84) -> the landing registers just emptied
        ; (rows above or below the image read zeros from an empty buffer; the row offset is scalar, and whether or not the
        ; range check sees it, a row of the image passes)
        s_lshl_b32 s90, s24, 4
        s_add_i32 s90, s90, 68
        s_add_i32 s91, s90, 15
        s_cmp_lt_u32 s91, %[H]                ; (unsigned: also false for rows above the image)
        s_cbranch_scc0 VR_slowloads_%=_\@
        s_cmp_lt_u32 s90, %[H]
        s_cbranch_scc0 VR_slowloads_%=_\@
        ; all sixteen rows inside the image (every step but the first and the last few): one scalar add per row
        s_mov_b32 s18, %[range]
        s_mul_i32 s91, s90, %[rsb]
        VR_LOAD_FAST \lb, 0
        VR_LOAD_FAST \lb, 1
        VR_LOAD_FAST \lb, 2
        VR_LOAD_FAST \lb, 3
        VR_LOAD_FAST \lb, 4
        VR_LOAD_FAST \lb, 5
        VR_LOAD_FAST \lb, 6
        VR_LOAD_FAST \lb, 7
        VR_LOAD_FAST \lb, 8
        VR_LOAD_FAST \lb, 9
        VR_LOAD_FAST \lb, 10
        VR_LOAD_FAST \lb, 11
        VR_LOAD_FAST \lb, 12
        VR_LOAD_FAST \lb, 13
        VR_LOAD_FAST \lb, 14
        VR_LOAD_FAST \lb, 15
        s_branch VR_loaded_%=_\@
VR_slowloads_%=_\@:
        VR_LOAD_ROW \lb, 0
        VR_LOAD_ROW \lb, 1
        VR_LOAD_ROW \lb, 2
        VR_LOAD_ROW \lb, 3
        VR_LOAD_ROW \lb, 4
        VR_LOAD_ROW \lb, 5
        VR_LOAD_ROW \lb, 6
        VR_LOAD_ROW \lb, 7
        VR_LOAD_ROW \lb, 8
        VR_LOAD_ROW \lb, 9
        VR_LOAD_ROW \lb, 10
        VR_LOAD_ROW \lb, 11
        VR_LOAD_ROW \lb, 12
        VR_LOAD_ROW \lb, 13
        VR_LOAD_ROW \lb, 14
        VR_LOAD_ROW \lb, 15
VR_loaded_%=_\@:
        s_cmp_lt_i32 s24, 0
        s_cbranch_scc1 VR_end_of_step_%=_\@
        ; ------------------------------------------------------------ second pass of tile u - 3
        s_cmp_eq_u32 s84, 0
        s_cbranch_scc1 VR_zero2_%=_\@
        VR_SWEEP_RUN 104
        s_branch VR_end_of_step_%=_\@
VR_zero2_%=_\@:
        VR_ZERO_ACC
VR_end_of_step_%=_\@:
        ; registers 4b..4b+3 of lane 16q + n = out[rows 16 v + 4q .. + 3][column b][hypothesis n]: one float4 (four columns) per
        ; row.  Always four stores (a step's loads are counted against them); before tile 0 into an empty buffer
        s_cmp_ge_i32 s28, 0
        s_cselect_b32 s22, %[range], 0
        s_lshl_b32 s85, s28, 4
        s_mul_i32 s85, s85, %[rsb]
        ; ------------------------------------------------------------ the next step: ring indices, tiles, first masks -- in front of
        ; the stores, so that the masks travel behind them and behind the next step's row moves
        s_add_u32 s25, s25, 16
        s_sub_u32 s31, s25, 88
        s_cmp_ge_u32 s25, 88
        s_cselect_b32 s25, s31, s25
        s_add_u32 s26, s26, 16
        s_sub_u32 s31, s26, 100
        s_cmp_ge_u32 s26, 100
        s_cselect_b32 s26, s31, s26
        s_add_i32 s24, s24, 1
        s_waitcnt lgkmcnt(0)                  ; the headers of its tiles, requested a step ago
        s_mov_b32 s81, s86
        s_cmp_ge_i32 s24, 0
        s_cselect_b32 s82, s87, 0
        s_cmp_lt_i32 s24, %[nT]
        s_cselect_b32 s82, s82, 0             ; first-pass tiles past the image: no window rows (their rows are zeros)
        s_sub_i32 s28, s24, 3
        s_mov_b32 s83, s88
        s_cmp_ge_i32 s28, 0
        s_cselect_b32 s84, s89, 0
        .if VR_EXP & 1
        s_mov_b32 s82, 0
        s_mov_b32 s84, 0
        .endif
        ; headers of the tiles of the step after it (tile indices clamped into the table; unused entries are masked above)
        s_sub_u32 s93, %[nT], 1
        s_add_i32 s92, s24, 1
        s_max_i32 s92, s92, 0
        s_min_i32 s92, s92, s93
        s_mul_i32 s92, s92, %[tstep]
        s_load_dwordx2 s[86:87], %[trow], s92
        s_sub_i32 s92, s24, 2
        s_max_i32 s92, s92, 0
        s_min_i32 s92, s92, s93
        s_mul_i32 s92, s92, %[tstep]
        s_load_dwordx2 s[88:89], %[trow], s92
        s_cmp_eq_u32 s82, 0
        s_cbranch_scc1 VR_noissue1_%=_\@
        s_mov_b32 s94, s81
        s_mov_b32 s95, s82
        s_add_u32 s96, s25, 16                ; row 16 u - 36 = row 16 u + 36 - 72, and -72 = 16 (mod 88)
        s_sub_u32 s31, s96, 88
        s_cmp_ge_u32 s96, 88
        s_cselect_b32 s96, s31, s96
        s_movk_i32 s29, 88
        s_mul_i32 s92, s24, %[tstep]
        s_add_u32 s92, s92, 32
        s_add_u32 s98, %[trow_lo], s92
        s_addc_u32 s99, %[trow_hi], 0
        VR_SWEEP_ISSUE
VR_noissue1_%=_\@:
        VR_STORE_ROWS 0
        VR_STORE_ROWS 1
        VR_STORE_ROWS 2
        VR_STORE_ROWS 3
        .endm
VR_loop_%=:
        VR_STEP 88
        s_cmp_ge_i32 s24, s27
        s_cbranch_scc1 VR_done_%=
        VR_STEP 228
        s_cmp_lt_i32 s24, s27
        s_cbranch_scc1 VR_loop_%=
VR_done_%=:
        s_waitcnt vmcnt(0) lgkmcnt(0)         ; the last rows and headers: nothing may be in flight when the wave ends
        .purgem VR_BLOCK
        .purgem VR_PAIR
        .purgem VR_SWEEP_ISSUE
        .purgem VR_SWEEP_RUN
        .purgem VR_ZERO_ACC
        .purgem VR_PUT4
        .purgem VR_LOAD_ROW
        .purgem VR_LOAD_FAST
        .purgem VR_STORE_ROWS
        .purgem VR_STEP
        )ASM"
                 :
                 : [in] "s"(in), [out] "s"(out), [trow] "s"(trow), [trow_lo] "s"((uint32_t)(uintptr_t)trow), [trow_hi] "s"((uint32_t)((uintptr_t)trow >> 32)),
                   [tstep] "s"(tstep), [rsb] "s"(rsb), [H] "s"(H), [nT] "s"(nT), [range] "s"(range), [voff] "v"(voff), [vst] "v"(vst), [exp] "n"(EXP)
                 : "memory", "scc", "vcc",
                   "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s29", "s30", "s31", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s28",
                   "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                   "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",
                   "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
                   "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                   "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99",
                   "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
                   "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133",
                   "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150",
                   "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167",
                   "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184",
                   "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201",
                   "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218",
                   "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235",
                   "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
}

bool aggv_supports(int usd) { return usd >= 1 && usd <= VR_TOP; }
int aggv_table_top() { return VR_TOP; }
int aggv_table_rec() { return 8 + 8 * (VR_NQ + 2); } // header + 22 quads + the batch read-ahead

// both vertical passes, vol_b -> vol_a, for `nviews` views; `tab` / `rec`: the window table of stm_k_vwin_table (static layout)
void launch_pq_v12r(PQViews &v, int nviews, const uint32_t *tab, int rec, int H, int W, int G, int NC)
{
    (void)W;
    const dim3 grid(G * nviews * NC);
#ifdef STM_TIMING
    switch (timing_knobs()) {
    case 1: STM_LAUNCH(stm_k_pq_v12r<1>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    case 2: STM_LAUNCH(stm_k_pq_v12r<2>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    case 3: STM_LAUNCH(stm_k_pq_v12r<3>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    case 4: STM_LAUNCH(stm_k_pq_v12r<4>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    case 6: STM_LAUNCH(stm_k_pq_v12r<6>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    case 8: STM_LAUNCH(stm_k_pq_v12r<8>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    case 9: STM_LAUNCH(stm_k_pq_v12r<9>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    default: STM_LAUNCH(stm_k_pq_v12r<0>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews); break;
    }
#else
    STM_LAUNCH(stm_k_pq_v12r<0>, grid, dim3(64), 0, stream(), v, tab, rec, H, G, NC, nviews);
#endif
    STM_CHECK_LAUNCH();
}

} // namespace stm
