// stm_kernels_hslo.hip -- four-direction scanline optimisation (HSLO) for gfx950.
//
// The reference ships only a stub for this stage (d_dc_hslo.cu:9-29 empty path-cost kernels,
// :97-221 driver that never writes `disp`, call site commented out at image_io.cpp:310-316), so
// parity is UNPINNED: the algorithm is Mei et al. section 3.3 with the penalty rule the reference
// does express (dc_hslo_h_cdiff_kernel, d_dc_hslo.cu:73-93; constants :124-127).  The definition the
// oracle and this file share is written out in oracle/stm_oracle.c (orc_dc_hslo_slab2).
//
// Mapping.  The recurrence is sequential along a scan line and parallel over lines x hypotheses, so a wave's lanes are
// the hypotheses d (D <= 64 * DPL, d = lane + 64 j): Cr(p-r, d+-1) come from DPP wave shifts, min_k Cr(p-r, k) from a
// 6-step DPP reduction.  The frame pipeline's PQ volume layout, float4 [chunk = d / 16][y][g = x / 4][d % 16] with the
// float4 = the four pixels 4g..4g+3 of one hypothesis (stm_kernels_aggm.hip), already IS hypothesis-major inside a
// group of four pixels: lane d reads its own 16 bytes (256 B contiguous per 16 lanes) and no transposition, no LDS
// tile and no transposed copy of the volume is needed in either orientation.
//   horizontal passes: one wave = one image row; the four pixels of a float4 are four consecutive steps;
//   vertical passes:   one wave = one group of four columns; the four pixels of a float4 are four independent lines
//                      (their reductions are interleaved, which also hides the DPP latencies).
// The four direction volumes are never materialised: left->right writes C_lr, right->left adds to it in place, top->bottom
// adds to that in place and bottom->top forms ((C_lr + C_rl) + C_tb) + C_bt) * 0.25 in registers and does WTA (10 V of
// traffic per view instead of 16 V + two transposes + an 8 V combine).
// The penalty class of a step needs D1 (own image, the same for all hypotheses) and D2 (other image at the matched pixel
// x + osign (d - zd)): both are classified once per frame into byte planes (stm_k_hslo_classes); a lane fetches the four
// D2 classes of a group with one unaligned dword load, the wave fetches the four D1 classes with another, and the pair
// (P1, P2) of a step is one 8-byte LDS read from a 9-entry table at offset 24 class(D1) + 8 class(D2).
#include "stm_common.h"
#include <type_traits>

namespace stm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));

namespace {

__device__ __forceinline__ f4 nt_load4(const f4 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void nt_store4(f4 *p, f4 v) { __builtin_nontemporal_store(v, p); }

// ------------------------------------------------------------------ class planes
// own image = integer mean as u8 (d_dc_hslo.cu:57-58), other image = float mean (:66-67)
__device__ __forceinline__ float avg_own(const u8 *p) { return (float)(u8)(((int)p[0] + (int)p[1] + (int)p[2]) / 3); }
__device__ __forceinline__ float avg_other(const u8 *p) { return (float)((double)(float)((int)p[0] + (int)p[1] + (int)p[2]) / 3.0); }
// 0: step < T, 1: step > T, 2: neither (d_dc_hslo.cu:73-93 compares with '<' and '>')
__device__ __forceinline__ int step_class(float a, float b, float T)
{
    const float s = fabsf(a - b);
    return s < T ? 0 : (s > T ? 1 : 2);
}

struct HsloArgs {
    const f4 *cost[2]; // per view: aggregated costs, PQ layout
    f4 *acc[2];        // per view: C_lr, then C_lr + C_rl, then + C_tb (PQ layout)
    u8 *u[2][4];       // per view: 24 * class(D1) of the step INTO (y, x): [H][WU] for left->right, right->left, top->bottom, bottom->top
    u8 *c2[2][2];      // per view: 8 * class(D2) [H][WP] at index PAD + x' for the horizontal / vertical predecessor of x'
    float *disp[2];
    const u8 *img_a[2], *img_b[2]; // own / other image of the view
    int osign[2];      // +1 left view (matched pixel x + d - zd), -1 right view
    f4 *inf;           // 256 B of +inf: what the lanes of absent hypotheses (d >= D) read, and where their (+inf) results go
    float p1[9], p2[9]; // penalties by 3 class(D1) + class(D2)
};

// grid (cdiv(WP, 256), H, views)
__global__ __launch_bounds__(256) void stm_k_hslo_classes(HsloArgs a, float T, int H, int W, int WU, int WP, int PAD, int elem_sz)
{
    const int i = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, view = blockIdx.z;
    if (blockIdx.x == 0 && y == 0 && view == 0 && threadIdx.x < 64) ((float *)a.inf)[threadIdx.x] = __builtin_inff();
    if (i >= WP) return;
    const u8 *__restrict__ A = a.img_a[view], *__restrict__ B = a.img_b[view];
    const size_t row = (size_t)y * W;
    {
        const int xq = min(max(i - PAD, 0), W - 1), xp = min(max(i - PAD - 1, 0), W - 1);
        const float b = avg_other(B + (row + xq) * elem_sz);
        a.c2[view][0][(size_t)y * WP + i] = (u8)(8 * step_class(b, avg_other(B + (row + xp) * elem_sz), T));
        a.c2[view][1][(size_t)y * WP + i] = y > 0 ? (u8)(8 * step_class(b, avg_other(B + (row - W + xq) * elem_sz), T)) : 0;
    }
    if (i < WU) {
        const int x = i;
        int lr = 0, rl = 0, tb = 0, bt = 0;
        if (x < W) {
            const float c = avg_own(A + (row + x) * elem_sz);
            if (x > 0) lr = step_class(c, avg_own(A + (row + x - 1) * elem_sz), T);
            if (x + 1 < W) rl = step_class(c, avg_own(A + (row + x + 1) * elem_sz), T);
            if (y > 0) tb = step_class(c, avg_own(A + (row - W + x) * elem_sz), T);
            if (y + 1 < H) bt = step_class(c, avg_own(A + (row + W + x) * elem_sz), T);
        }
        const size_t p = (size_t)y * WU + x;
        a.u[view][0][p] = (u8)(24 * lr); a.u[view][1][p] = (u8)(24 * rl);
        a.u[view][2][p] = (u8)(24 * tb); a.u[view][3][p] = (u8)(24 * bt);
    }
}

// any layout -> PQ (only needed when the caller's volume is a plane table / slab / quads volume):
// thread = (group g, quad qq of the chunk); grid (cdiv(4 G, 256), H, NC)
// `odd` (may be null): set to 1 when the volume holds an element that is not an ordinary number -- infinite, NaN or denormal.  The
// matrix-pipe aggregation adds window elements as acc += mask * b: a masked element must be finite for 0 * b to be 0, and
// denormal operands are not guaranteed to survive the matrix instruction; the per-stage ca_cross then runs the vector-ALU
// kernels, which touch an element only inside the windows that contain it (d_ca_cross_sum.cu:284-289).
__device__ __forceinline__ bool pq_odd(float v)
{
    const uint32_t e = __builtin_bit_cast(uint32_t, v) & 0x7f800000u, m = __builtin_bit_cast(uint32_t, v) & 0x007fffffu;
    return e == 0x7f800000u || (e == 0u && m != 0u);
}
template <bool QUAD> __global__ __launch_bounds__(256) void stm_k_to_pq(Vol in, f4 *__restrict__ out, int D, int H, int W, int G, uint32_t *__restrict__ odd)
{
    const int t = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    const int g = t >> 2, q = 4 * c + (t & 3);
    if (g >= G) return;
    float4 px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = 4 * g + k;
        px[k] = (x < W && 4 * q < D) ? load_quad<QUAD>(in, q, D, (size_t)y * W + x) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (odd) {
        bool o = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) o = o || pq_odd(px[k].x) || pq_odd(px[k].y) || pq_odd(px[k].z) || pq_odd(px[k].w);
        if (o) *odd = 1u; // same value from every writer
    }
    f4 *o = out + (((size_t)c * H + y) * G + g) * 16 + 4 * (t & 3);
    o[0] = f4{px[0].x, px[1].x, px[2].x, px[3].x};
    o[1] = f4{px[0].y, px[1].y, px[2].y, px[3].y};
    o[2] = f4{px[0].z, px[1].z, px[2].z, px[3].z};
    o[3] = f4{px[0].w, px[1].w, px[2].w, px[3].w};
}

// ------------------------------------------------------------------ the recurrence
#define STM_DPP(old, v, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, 0xf, 0xf, false))

// min over the 64 lanes as a scalar.  Six v_min_f32 with a DPP source operand: a lane whose DPP source does not exist is
// simply not written (bound_ctrl off), and min is idempotent, so the row masks of the classic reduction are not needed;
// lane 63 ends up with the minimum.  Inline asm because clang expands fminf(x, dpp(x)) into mov-immediate + v_mov_dpp + a
// canonicalising v_max + v_min; the s_nop covers the VALU-write -> DPP-read hazard the assembler does not pad for us.
__device__ __forceinline__ float wave_min(float v)
{
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// the same for four independent values: the four chains are interleaved, so no step waits for its predecessor
#define STM_MIN4_STAGE(ctrl)                                                 \
    "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"       \
    "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"       \
    "v_min_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"       \
    "v_min_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_min4(float (&v)[4])
{
    asm volatile("s_nop 1\n\t" STM_MIN4_STAGE("row_shr:1") STM_MIN4_STAGE("row_shr:2") STM_MIN4_STAGE("row_shr:4")
                     STM_MIN4_STAGE("row_shr:8") STM_MIN4_STAGE("row_bcast:15") STM_MIN4_STAGE("row_bcast:31") "s_nop 1"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 63));
}
__device__ __forceinline__ float min2(float a, float b) // the operands are never NaN: plain v_min_f32, no canonicalisation
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// One step of one line: prev = Cr(p - r, .), cc = C(p, .), m = min_k prev[k] (wave-uniform), (P1, P2) per hypothesis.
//   Cr(p, d) = C(p, d) + min(Cr(p-r, d), Cr(p-r, d-1) + P1, Cr(p-r, d+1) + P1, m + P2) - m
// evaluated in the order written in oracle/stm_oracle.c (hslo_dir): the sum, then the subtraction.
// A lane without a neighbour keeps +inf, which covers d = 0 and d = D - 1 (hypotheses >= D hold +inf).
template <int DPL>
__device__ __forceinline__ void hslo_step(const float (&prev)[DPL], const float (&cc)[DPL], const float (&P1)[DPL],
                                          const float (&P2)[DPL], float m, float (&cur)[DPL], int lane)
{
    const float inf = __builtin_inff();
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        float tb = inf, ta = inf;
        if (DPL == 1) { // the wave shift rides on the add (DPP operand)
            asm("s_nop 1\n\tv_add_f32_dpp %0, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %1, %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf"
                : "+v"(tb), "+v"(ta)
                : "v"(prev[j]), "v"(P1[j]));
        } else { // several hypotheses per lane: patch the 64-lane borders
            float below = STM_DPP(inf, prev[j], 0x138); // wave_shr:1 : lane i <- lane i-1
            float above = STM_DPP(inf, prev[j], 0x130); // wave_shl:1 : lane i <- lane i+1
            if (j > 0) {
                const float edge = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prev[j > 0 ? j - 1 : 0]), 63));
                if (lane == 0) below = edge;
            }
            if (j + 1 < DPL) {
                const float edge = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prev[j + 1 < DPL ? j + 1 : j]), 0));
                if (lane == 63) above = edge;
            }
            tb = below + P1[j];
            ta = above + P1[j];
        }
        const float best = min2(min3(prev[j], tb, ta), m + P2[j]);
        float v = cc[j] + best;
        v = v - m;
        cur[j] = v;
    }
}

template <int DPL> __device__ __forceinline__ float lane_min(const float (&v)[DPL])
{
    float m = v[0];
#pragma unroll
    for (int j = 1; j < DPL; ++j) m = min2(m, v[j]);
    return m;
}

// (P1, P2) of a step: the byte `k` of the D2 class word + the wave-uniform D1 class offset index the LDS table
__device__ __forceinline__ float2 penalties(const float2 *ptab, uint32_t c2w, int k, uint32_t ub)
{
    const uint32_t idx = ((c2w >> (8 * k)) & 0xffu) + ub;
    return *(const float2 *)((const char *)ptab + idx);
}

// ------------------------------------------------------------------ horizontal passes
// grid (cdiv(H, 4), views); block = 4 waves, one image row each.  BWD = false: left->right, acc = C_lr;
// BWD = true: right->left, acc += C_rl.  PF = groups of four pixels loaded ahead of the walk.
constexpr int HH_WPB = 1; // waves (image rows) per block in the horizontal passes: 2160 one-wave blocks spread over the 256 CUs as 8 or 9 each; four-row blocks gave some CUs 12 waves and others 8, and every pass ran at the pace of the fullest CU (0.49 + 0.73 -> 0.43 + 0.64 ms)
template <int DPL, bool BWD, int PF>
__global__ __launch_bounds__(64 * HH_WPB) void stm_k_hslo_h(HsloArgs a, int D, int zd, int H, int W, int G, int WU, int WP, int PAD, int dbg)
{
    __shared__ float2 ptab[9];
    if (threadIdx.x < 9) ptab[threadIdx.x] = make_float2(a.p1[threadIdx.x], a.p2[threadIdx.x]);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int y = blockIdx.x * HH_WPB + wave, view = blockIdx.y;
    if (y >= H) return; // whole wave; no block barrier below
    const int osign = a.osign[view];
    const f4 *cp[DPL];
    f4 *ap[DPL];
    const u8 *c2p[DPL];
    int gs[DPL]; // float4 elements between consecutive groups (0 for an absent hypothesis: it keeps reading the +inf block)
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        const int d = lane + 64 * j;
        const bool ok = d < D;
        const size_t off = (((size_t)(d >> 4) * H + y) * G) * 16 + (d & 15);
        cp[j] = ok ? a.cost[view] + off : a.inf;
        ap[j] = ok ? a.acc[view] + off : a.inf;
        gs[j] = ok ? 16 : 0;
        c2p[j] = a.c2[view][0] + (size_t)y * WP + PAD + (ok ? osign * (d - zd) : 0) + (BWD ? 1 : 0);
    }
    const u8 *up = a.u[view][BWD ? 1 : 0] + (size_t)y * WU;

    struct Row {
        f4 c[DPL], s[DPL];
        uint32_t c2[DPL], u;
    };
    auto load_row = [&](Row &r, int v) { // v = position of the group in walking order
        const int vv = min(v, G - 1), g = BWD ? G - 1 - vv : vv;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            r.c[j] = nt_load4(cp[j] + (size_t)g * gs[j]);
            if (BWD) r.s[j] = *(ap[j] + (size_t)g * gs[j]); // plain load: the end of the row is what left->right wrote last (4 % faster)
            r.c2[j] = *(const u32_unaligned *)(c2p[j] + 4 * g);
        }
        r.u = *(const uint32_t *)(up + 4 * g);
    };

    float prev[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) prev[j] = 0.f;
    const int x_first = BWD ? W - 1 : 0;
    auto process = [&](const Row &r, int v) {
        const int g = BWD ? G - 1 - v : v;
        const uint32_t uw = __builtin_amdgcn_readfirstlane(r.u);
        f4 o[DPL];
#pragma unroll
        for (int j = 0; j < DPL; ++j) o[j] = BWD ? r.s[j] : r.c[j];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = BWD ? 3 - kk : kk, x = 4 * g + k;
            if (x >= W) continue; // only in the last group
            float cc[DPL], cur[DPL];
#pragma unroll
            for (int j = 0; j < DPL; ++j) cc[j] = r.c[j][k];
            if (x == x_first || STM_DBG(dbg, 1)) { // first pixel of the line: Cr(p0, d) = C(p0, d)
#pragma unroll
                for (int j = 0; j < DPL; ++j) cur[j] = cc[j];
            } else {
                const uint32_t ub = (uw >> (8 * k)) & 0xffu;
                float P1[DPL], P2[DPL];
#pragma unroll
                for (int j = 0; j < DPL; ++j) {
                    const float2 p = penalties(ptab, r.c2[j], k, ub);
                    P1[j] = p.x; P2[j] = p.y;
                }
                const float m = wave_min(lane_min<DPL>(prev));
                hslo_step<DPL>(prev, cc, P1, P2, m, cur, lane);
            }
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                prev[j] = cur[j];
                o[j][k] = BWD ? r.s[j][k] + cur[j] : cur[j];
            }
        }
#pragma unroll
        for (int j = 0; j < DPL; ++j)
            if (!STM_DBG(dbg, 2)) {
                // left->right: plain stores (8 % faster than non-temporal ones: right->left starts where this pass ended);
                // right->left: non-temporal (plain ones slow the vertical pass that follows by 5 %)
                if (BWD) nt_store4(ap[j] + (size_t)g * gs[j], o[j]);
                else *(ap[j] + (size_t)g * gs[j]) = o[j];
            }
    };

    Row buf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(buf[i], i);
    for (int v0 = 0; v0 < G; v0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (v0 + i < G) process(buf[i], v0 + i);
            load_row(buf[i], v0 + PF + i);
        }
    }
}

// Both horizontal directions of a row in ONE walk (round 4, D <= 64): the walk of one direction is a chain of dependent
// steps (a 6-stage DPP minimum, the step, one pixel after the other: 0.43 + 0.64 ms for two launches at two waves per SIMD),
// so a wave walks its row from BOTH ends at once -- two independent recurrences whose instructions interleave.  Iteration v
// takes group v left -> right and group G - 1 - v right -> left.  In the first half of the walk either side is the first
// to visit its group and stores its path costs; in the second half it finds the other side's there and stores the sum.  The sum
// C_lr + C_rl is one float add in either order, so the result is the one the two-launch form (and the oracle) produce.
// A group's stored costs are written and read by the same lane of the same wave, in program order; they are requested PF
// groups ahead only when the store is already behind (2 v >= G - 1 + PF), else at the step itself.
__device__ __forceinline__ void wave_min2(float &a, float &b)
{
#define STM_MIN2_STAGE(ctrl)                                              \
    "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"    \
    "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"    \
    "s_nop 0\n\t"
    asm volatile("s_nop 1\n\t" STM_MIN2_STAGE("row_shr:1") STM_MIN2_STAGE("row_shr:2") STM_MIN2_STAGE("row_shr:4") STM_MIN2_STAGE("row_shr:8")
                     STM_MIN2_STAGE("row_bcast:15") STM_MIN2_STAGE("row_bcast:31") "s_nop 0"
                 : "+v"(a), "+v"(b));
#undef STM_MIN2_STAGE
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
}

template <int DPL, int PF>
__global__ __launch_bounds__(64) void stm_k_hslo_h2(HsloArgs a, int D, int zd, int H, int W, int G, int WU, int WP, int PAD, int dbg)
{
    __shared__ float2 ptab[9];
    if (threadIdx.x < 9) ptab[threadIdx.x] = make_float2(a.p1[threadIdx.x], a.p2[threadIdx.x]);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x, view = blockIdx.y;
    const int osign = a.osign[view];
    const f4 *cp[DPL];
    f4 *ap[DPL];
    const u8 *c2f[DPL], *c2b[DPL];
    int gs[DPL]; // float4 elements between consecutive groups (0 for an absent hypothesis: it keeps reading the +inf block)
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        const int d = lane + 64 * j;
        const bool ok = d < D;
        const size_t off = (((size_t)(d >> 4) * H + y) * G) * 16 + (d & 15);
        cp[j] = ok ? a.cost[view] + off : a.inf;
        ap[j] = ok ? a.acc[view] + off : a.inf;
        gs[j] = ok ? 16 : 0;
        c2f[j] = a.c2[view][0] + (size_t)y * WP + PAD + (ok ? osign * (d - zd) : 0);
        c2b[j] = c2f[j] + 1;
    }
    const u8 *upf = a.u[view][0] + (size_t)y * WU, *upb = a.u[view][1] + (size_t)y * WU;

    struct Row {
        f4 c[DPL], s[DPL];
        uint32_t c2[DPL], u;
    };
    auto load_row = [&](Row &r, int v, auto bwd_) { // v = iteration
        constexpr bool BWD = decltype(bwd_)::value;
        const int vv = min(v, G - 1), g = BWD ? G - 1 - vv : vv;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            r.c[j] = nt_load4(cp[j] + (size_t)g * gs[j]);
            if (2 * vv >= G - 1 + PF) r.s[j] = *(ap[j] + (size_t)g * gs[j]); // the other side's store is behind us
            r.c2[j] = *(const u32_unaligned *)((BWD ? c2b[j] : c2f[j]) + 4 * g);
        }
        r.u = *(const uint32_t *)((BWD ? upb : upf) + 4 * g);
    };

    float prevF[DPL], prevB[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) prevF[j] = prevB[j] = 0.f;
    bool firstF = true, firstB = true; // the next valid pixel starts the line: Cr(p0, d) = C(p0, d)
    auto process = [&](const Row &rf, const Row &rb, int v) {
        const int gF = v, gB = G - 1 - v;
        const bool second = 2 * v > G - 1, mid = 2 * v == G - 1;
        const uint32_t uwF = __builtin_amdgcn_readfirstlane(rf.u), uwB = __builtin_amdgcn_readfirstlane(rb.u);
        f4 sF[DPL], sB[DPL], oF[DPL], oB[DPL];
        if (second) {
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                if (2 * v >= G - 1 + PF) { sF[j] = rf.s[j]; sB[j] = rb.s[j]; }
                else { sF[j] = *(ap[j] + (size_t)gF * gs[j]); sB[j] = *(ap[j] + (size_t)gB * gs[j]); }
            }
        }
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            oF[j] = second ? sF[j] : rf.c[j];
            oB[j] = second ? sB[j] : (mid ? (f4){0.f, 0.f, 0.f, 0.f} : rb.c[j]);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kF = kk, kB = 3 - kk, xF = 4 * gF + kF, xB = 4 * gB + kB;
            const bool vF = xF < W, vB = xB < W; // false only in the last group
            float ccF[DPL], ccB[DPL], curF[DPL], curB[DPL];
#pragma unroll
            for (int j = 0; j < DPL; ++j) { ccF[j] = rf.c[j][kF]; ccB[j] = rb.c[j][kB]; }
            if (vF && vB && !firstF && !firstB && !STM_DBG(dbg, 1)) { // both recurrences, interleaved
                const uint32_t ubF = (uwF >> (8 * kF)) & 0xffu, ubB = (uwB >> (8 * kB)) & 0xffu;
                float P1F[DPL], P2F[DPL], P1B[DPL], P2B[DPL];
#pragma unroll
                for (int j = 0; j < DPL; ++j) {
                    const float2 pf = penalties(ptab, rf.c2[j], kF, ubF), pb = penalties(ptab, rb.c2[j], kB, ubB);
                    P1F[j] = pf.x; P2F[j] = pf.y; P1B[j] = pb.x; P2B[j] = pb.y;
                }
                float mF = lane_min<DPL>(prevF), mB = lane_min<DPL>(prevB);
                wave_min2(mF, mB);
                hslo_step<DPL>(prevF, ccF, P1F, P2F, mF, curF, lane);
                hslo_step<DPL>(prevB, ccB, P1B, P2B, mB, curB, lane);
            } else { // the ends of the line
                if (vF) {
                    if (firstF || STM_DBG(dbg, 1)) {
#pragma unroll
                        for (int j = 0; j < DPL; ++j) curF[j] = ccF[j];
                        firstF = false;
                    } else {
                        const uint32_t ubF = (uwF >> (8 * kF)) & 0xffu;
                        float P1[DPL], P2[DPL];
#pragma unroll
                        for (int j = 0; j < DPL; ++j) { const float2 pq = penalties(ptab, rf.c2[j], kF, ubF); P1[j] = pq.x; P2[j] = pq.y; }
                        const float m = wave_min(lane_min<DPL>(prevF));
                        hslo_step<DPL>(prevF, ccF, P1, P2, m, curF, lane);
                    }
                }
                if (vB) {
                    if (firstB || STM_DBG(dbg, 1)) {
#pragma unroll
                        for (int j = 0; j < DPL; ++j) curB[j] = ccB[j];
                        firstB = false;
                    } else {
                        const uint32_t ubB = (uwB >> (8 * kB)) & 0xffu;
                        float P1[DPL], P2[DPL];
#pragma unroll
                        for (int j = 0; j < DPL; ++j) { const float2 pq = penalties(ptab, rb.c2[j], kB, ubB); P1[j] = pq.x; P2[j] = pq.y; }
                        const float m = wave_min(lane_min<DPL>(prevB));
                        hslo_step<DPL>(prevB, ccB, P1, P2, m, curB, lane);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                if (vF) { prevF[j] = curF[j]; oF[j][kF] = second ? sF[j][kF] + curF[j] : curF[j]; }
                if (vB) { prevB[j] = curB[j]; oB[j][kB] = second ? sB[j][kB] + curB[j] : curB[j]; }
            }
        }
        if (STM_DBG(dbg, 2)) return;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            if (mid) { // one group, both sides in this iteration: C_lr + C_rl (pixels past the row keep the cost, as in the two-launch form)
                f4 o = oF[j];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * gF + k < W) o[k] = oF[j][k] + oB[j][k];
                nt_store4(ap[j] + (size_t)gF * gs[j], o);
            } else if (second) { // final values: non-temporal (plain ones slow the vertical pass that follows)
                nt_store4(ap[j] + (size_t)gF * gs[j], oF[j]);
                nt_store4(ap[j] + (size_t)gB * gs[j], oB[j]);
            } else {
                *(ap[j] + (size_t)gF * gs[j]) = oF[j];
                *(ap[j] + (size_t)gB * gs[j]) = oB[j];
            }
        }
    };

    Row bf[PF], bb[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) { load_row(bf[i], i, std::false_type()); load_row(bb[i], i, std::true_type()); }
    for (int v0 = 0; v0 < G; v0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (v0 + i < G) process(bf[i], bb[i], v0 + i);
            load_row(bf[i], v0 + PF + i, std::false_type());
            load_row(bb[i], v0 + PF + i, std::true_type());
        }
    }
}

// ------------------------------------------------------------------ vertical passes
// grid (cdiv(G, 4), views); block = 4 waves, one group of four columns each (adjacent groups: 1 KB contiguous per row and
// chunk).  BWD = false: top->bottom, acc += C_tb;  BWD = true: bottom->top, ((acc + C_bt) * 0.25) -> WTA -> disp.
template <int DPL, bool BWD, int PF>
__global__ __launch_bounds__(256) void stm_k_hslo_v(HsloArgs a, int D, int zd, int H, int W, int G, int WU, int WP, int PAD, int dbg)
{
    __shared__ float2 ptab[9];
    if (threadIdx.x < 9) ptab[threadIdx.x] = make_float2(a.p1[threadIdx.x], a.p2[threadIdx.x]);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave, view = blockIdx.y;
    if (g >= G) return;
    const int osign = a.osign[view];
    const f4 *cp[DPL];
    f4 *ap[DPL];
    const u8 *c2p[DPL];
    size_t rs[DPL]; // float4 elements between consecutive rows (0 for an absent hypothesis)
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        const int d = lane + 64 * j;
        const bool ok = d < D;
        const size_t off = (((size_t)(d >> 4) * H) * G + g) * 16 + (d & 15);
        cp[j] = ok ? a.cost[view] + off : a.inf;
        ap[j] = ok ? a.acc[view] + off : a.inf;
        rs[j] = ok ? (size_t)G * 16 : 0;
        c2p[j] = a.c2[view][1] + PAD + 4 * g + (ok ? osign * (d - zd) : 0);
    }
    const u8 *up = a.u[view][BWD ? 3 : 2] + 4 * g;
    float *__restrict__ disp = a.disp[view];

    struct Row {
        f4 c[DPL], s[DPL];
        uint32_t c2[DPL], u;
    };
    auto load_row = [&](Row &r, int v) { // v = position of the row in walking order
        const int vv = min(v, H - 1), y = BWD ? H - 1 - vv : vv;
        const int yc = BWD ? min(y + 1, H - 1) : y; // the D2 class of a bottom->top step into row y is stored with row y + 1
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            r.c[j] = nt_load4(cp[j] + (size_t)y * rs[j]);
            r.s[j] = nt_load4(ap[j] + (size_t)y * rs[j]);
            r.c2[j] = *(const u32_unaligned *)(c2p[j] + (size_t)yc * WP);
        }
        r.u = *(const uint32_t *)(up + (size_t)y * WU);
    };

    float prev[4][DPL];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < DPL; ++j) prev[k][j] = 0.f;
    auto process = [&](const Row &r, int v) {
        const int y = BWD ? H - 1 - v : v;
        float cur[4][DPL];
        if (v == 0 || STM_DBG(dbg, 1)) { // first pixel of the four lines
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < DPL; ++j) cur[k][j] = r.c[j][k];
        } else {
            const uint32_t uw = __builtin_amdgcn_readfirstlane(r.u);
            float m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = lane_min<DPL>(prev[k]);
            wave_min4(m);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t ub = (uw >> (8 * k)) & 0xffu;
                float cc[DPL], P1[DPL], P2[DPL];
#pragma unroll
                for (int j = 0; j < DPL; ++j) {
                    const float2 p = penalties(ptab, r.c2[j], k, ub);
                    P1[j] = p.x; P2[j] = p.y;
                    cc[j] = r.c[j][k];
                }
                hslo_step<DPL>(prev[k], cc, P1, P2, m[k], cur[k], lane);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < DPL; ++j) prev[k][j] = cur[k][j];
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                f4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = r.s[j][k] + cur[k][j];
                if (!STM_DBG(dbg, 2)) nt_store4(ap[j] + (size_t)y * rs[j], o);
            }
        } else {
            // C2(p, d) = (((C_lr + C_rl) + C_tb) + C_bt) * 0.25f, then first-lowest-wins WTA (d_dc_wta.cu:19-34).  The scaling
            // by 0.25 is exact and order-preserving, so the argmin is taken on the unscaled sums: the lane's best (ascending d,
            // strict <), the wave minimum, then the smallest d among the lanes that hold it (with one hypothesis per lane that
            // is the first set bit of the equality ballot)
            float bv[4], bd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bv[k] = r.s[0][k] + cur[k][0];
                bd[k] = (float)lane;
#pragma unroll
                for (int j = 1; j < DPL; ++j) {
                    const float val = r.s[j][k] + cur[k][j];
                    if (val < bv[k]) { bv[k] = val; bd[k] = (float)(lane + 64 * j); }
                }
            }
            float mv[4] = {bv[0], bv[1], bv[2], bv[3]};
            wave_min4(mv);
            float cd[4];
            if (DPL == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) cd[k] = (float)__builtin_ctzll(__ballot(bv[k] == mv[k]));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) cd[k] = bv[k] == mv[k] ? bd[k] : 1.0e9f;
                wave_min4(cd);
            }
            if (lane < 4 && 4 * g + lane < W) {
                const float best = lane == 0 ? cd[0] : (lane == 1 ? cd[1] : (lane == 2 ? cd[2] : cd[3]));
                disp[(size_t)y * W + 4 * g + lane] = best - (float)zd;
            }
        }
    };

    Row buf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(buf[i], i);
    for (int v0 = 0; v0 < H; v0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (v0 + i < H) process(buf[i], v0 + i);
            load_row(buf[i], v0 + PF + i);
        }
    }
}

template <int DPL, int PF>
void hslo_passes(const HsloArgs &a, int nviews, int D, int zd, int H, int W, int G, int WU, int WP, int PAD)
{
    const int dbg = timing_knobs(); // timing build only (stm_common.h): 1 = no recurrence, 2 = no stores
    bool both = false;
    if constexpr (DPL == 1) {
        if ((agg_variant() / 100) % 10 != 4) { // round 4: both horizontal directions in one walk; 400: two launches
            ProfScope p("hslo_lr");
            STM_LAUNCH((stm_k_hslo_h2<DPL, (PF > 4 ? 4 : PF)>), dim3(H, nviews), dim3(64), 0, stream(), a, D, zd, H, W, G, WU, WP, PAD, dbg);
            STM_CHECK_LAUNCH();
            both = true;
        }
    }
    if (!both) {
        {
            ProfScope p("hslo_lr");
            STM_LAUNCH((stm_k_hslo_h<DPL, false, PF>), dim3(cdiv(H, HH_WPB), nviews), dim3(64 * HH_WPB), 0, stream(), a, D, zd, H, W, G, WU, WP, PAD, dbg);
            STM_CHECK_LAUNCH();
        }
        {
            ProfScope p("hslo_rl");
            STM_LAUNCH((stm_k_hslo_h<DPL, true, PF>), dim3(cdiv(H, HH_WPB), nviews), dim3(64 * HH_WPB), 0, stream(), a, D, zd, H, W, G, WU, WP, PAD, dbg);
            STM_CHECK_LAUNCH();
        }
    }
    {
        ProfScope p("hslo_tb");
        STM_LAUNCH((stm_k_hslo_v<DPL, false, PF>), dim3(cdiv(G, 4), nviews), dim3(256), 0, stream(), a, D, zd, H, W, G, WU, WP, PAD, dbg);
        STM_CHECK_LAUNCH();
    }
    {
        ProfScope p("hslo_bt");
        STM_LAUNCH((stm_k_hslo_v<DPL, true, PF>), dim3(cdiv(G, 4), nviews), dim3(256), 0, stream(), a, D, zd, H, W, G, WU, WP, PAD, dbg);
        STM_CHECK_LAUNCH();
    }
}

} // namespace

void launch_to_pq(Vol in, float *pq, int D, int H, int W, uint32_t *odd)
{
    const int G = (W + 3) / 4, NC = (D + 15) / 16;
    if (in.quad) STM_LAUNCH(stm_k_to_pq<true>, dim3(cdiv(4 * G, 256), H, NC), dim3(256), 0, stream(), in, (f4 *)pq, D, H, W, G, odd);
    else STM_LAUNCH(stm_k_to_pq<false>, dim3(cdiv(4 * G, 256), H, NC), dim3(256), 0, stream(), in, (f4 *)pq, D, H, W, G, odd);
    STM_CHECK_LAUNCH();
}


// ------------------------------------------------------------------ drivers
// Scanline optimisation + WTA for 1 or 2 views whose aggregated costs are PQ volumes (pq_volume_floats each).
// cost_pq[v] is read only; acc_pq[v] is a scratch volume of the same size.
// img_a[v] = the view's own image, img_b[v] = the other image; osign[v] = +1 (left view) / -1 (right view).
// Scratch from the current Workspace scope: six byte planes per view.
void launch_hslo_wta_pq(int nviews, float *const *cost_pq, float *const *acc_pq, const u8 *const *img_a, const u8 *const *img_b,
                        const int *osign, float *const *disp, float T, float H1, float H2, int D, int zd, int H, int W, int elem_sz)
{
    if (D > 256 || D < 1 || nviews < 1 || nviews > 2) {
        fail("hslo: num_disp must be 1..256 and views 1..2", "D", __FILE__, __LINE__);
        return;
    }
    const float P1[3] = {H1, (float)((double)H1 / 4.0), (float)((double)H1 / 10.0)}; // d_dc_hslo.cu:124-127
    const float P2[3] = {H2, (float)((double)H2 / 4.0), (float)((double)H2 / 10.0)};
    const int G = (W + 3) / 4, WU = 4 * G;
    const int PAD = max(max(zd, D - 1 - zd), 0);
    const int WP = (4 * G + 2 * PAD + 8 + 3) & ~3;
    HsloArgs a;
    a.inf = (f4 *)Workspace::get<float>(64);
    for (int v = 0; v < 2; ++v) {
        const int s = v < nviews ? v : 0;
        a.cost[v] = (const f4 *)cost_pq[s]; a.acc[v] = (f4 *)acc_pq[s]; a.disp[v] = disp[s];
        a.img_a[v] = img_a[s]; a.img_b[v] = img_b[s]; a.osign[v] = osign[s];
        if (v >= nviews) {
            for (int i = 0; i < 4; ++i) a.u[v][i] = a.u[0][i];
            a.c2[v][0] = a.c2[0][0]; a.c2[v][1] = a.c2[0][1];
            continue;
        }
        for (int i = 0; i < 4; ++i) a.u[v][i] = Workspace::get<u8>((size_t)H * WU + 16);
        for (int i = 0; i < 2; ++i) a.c2[v][i] = Workspace::get<u8>((size_t)H * WP + 16);
    }
    // both below the threshold -> full penalty; exactly one below and the other above -> / 4; everything else -> / 10
    for (int u = 0; u < 3; ++u)
        for (int c = 0; c < 3; ++c) {
            const int cls = (u == 0 && c == 0) ? 0 : ((u == 0 && c == 1) || (u == 1 && c == 0)) ? 1 : 2;
            a.p1[3 * u + c] = P1[cls];
            a.p2[3 * u + c] = P2[cls];
        }
    {
        ProfScope p("hslo_classes");
        STM_LAUNCH(stm_k_hslo_classes, dim3(cdiv(WP, 256), H, nviews), dim3(256), 0, stream(), a, T, H, W, WU, WP, PAD, elem_sz);
        STM_CHECK_LAUNCH();
    }
    if (D <= 64) hslo_passes<1, 8>(a, nviews, D, zd, H, W, G, WU, WP, PAD);
    else if (D <= 128) hslo_passes<2, 4>(a, nviews, D, zd, H, W, G, WU, WP, PAD);
    else hslo_passes<4, 2>(a, nviews, D, zd, H, W, G, WU, WP, PAD);
}

// The same for cost volumes in any other layout (the per-stage API, the vector-ALU aggregation path): converted to PQ
// first.  Scratch from the current Workspace scope, per view: two PQ volumes + the byte planes.
void launch_hslo_wta(int nviews, const Vol *cost, const u8 *const *img_a, const u8 *const *img_b, const int *osign,
                     float *const *disp, float T, float H1, float H2, int D, int zd, int H, int W, int elem_sz)
{
    if (D > 256 || D < 1 || nviews < 1 || nviews > 2) {
        fail("hslo: num_disp must be 1..256 and views 1..2", "D", __FILE__, __LINE__);
        return;
    }
    const int G = (W + 3) / 4, NC = (D + 15) / 16;
    const size_t VP = pq_volume_floats(D, H, W);
    float *cpq[2] = {nullptr, nullptr}, *apq[2] = {nullptr, nullptr};
    for (int v = 0; v < nviews; ++v) {
        cpq[v] = Workspace::get<float>(VP);
        apq[v] = Workspace::get<float>(VP);
        ProfScope p("hslo_to_pq");
        if (cost[v].quad) STM_LAUNCH(stm_k_to_pq<true>, dim3(cdiv(4 * G, 256), H, NC), dim3(256), 0, stream(), cost[v], (f4 *)cpq[v], D, H, W, G, (uint32_t *)nullptr);
        else STM_LAUNCH(stm_k_to_pq<false>, dim3(cdiv(4 * G, 256), H, NC), dim3(256), 0, stream(), cost[v], (f4 *)cpq[v], D, H, W, G, (uint32_t *)nullptr);
        STM_CHECK_LAUNCH();
    }
    launch_hslo_wta_pq(nviews, cpq, apq, img_a, img_b, osign, disp, T, H1, H2, D, zd, H, W, elem_sz);
}

} // namespace stm
