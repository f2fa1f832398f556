// stm_kernels_hslo.hip -- four-direction scanline optimisation (HSLO) for gfx950.
//
// The reference ships only a stub for this stage (d_dc_hslo.cu:9-29 empty path-cost kernels,
// :97-221 driver that never writes `disp`, call site commented out at image_io.cpp:310-316), so
// parity is UNPINNED: the algorithm is Mei et al. section 3.3 with the penalty rule the reference
// does express (dc_hslo_h_cdiff_kernel, d_dc_hslo.cu:73-93; constants :124-127).  The definition the
// oracle and this file share is written out in oracle/stm_oracle.c (orc_dc_hslo_slab2).
//
// Mapping.  The recurrence is sequential along a scan line and parallel over lines x hypotheses, so ONE WAVE
// OWNS ONE LINE and, while it walks the line, its lanes are the hypotheses d (D <= 64 * DPL): Cr(p-r, d+-1) come
// from DPP wave shifts, min_k Cr(p-r, k) from a 6-step DPP reduction -- no LDS, no barrier on the critical path.
// Memory, however, wants lanes along the line.  So a line is processed in chunks of 32 pixels through an LDS tile
// [pixel][d] that the wave fills with coalesced 16-byte quads (lane = pixel), walks (lane = d, in place) and writes
// back coalesced: a transposition through LDS private to the wave (no block barriers).
// Vertical lines are made horizontal first: the cost volume is transposed once per call (quads [q][W][H]), the
// top->bottom / bottom->top passes run on the transposed volume, and the combine kernel reads their results back
// through an LDS tile transposition.  All line passes (2 directions x views, per orientation) share a launch.
#include "stm_common.h"
#include <cstdlib>

namespace stm {

// ------------------------------------------------------------------ small helpers
// colour averages used by the penalty rule, in both orientations:
// own image = integer mean as u8 (d_dc_hslo.cu:57-58), other image = float mean (:66-67)
__global__ __launch_bounds__(256) void stm_k_hslo_avg(const u8 *__restrict__ img_a, const u8 *__restrict__ img_b,
                                                      float *__restrict__ avg_a, float *__restrict__ avg_b,
                                                      float *__restrict__ avg_at, float *__restrict__ avg_bt, int H, int W,
                                                      int elem_sz)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t p = (size_t)y * W + x, pt = (size_t)x * H + y;
    const u8 *l = img_a + p * elem_sz, *r = img_b + p * elem_sz;
    const float va = (float)(u8)(((int)l[0] + (int)l[1] + (int)l[2]) / 3);
    const float vb = (float)((double)(float)((int)r[0] + (int)r[1] + (int)r[2]) / 3.0);
    avg_a[p] = va; avg_b[p] = vb;
    avg_at[pt] = va; avg_bt[pt] = vb;
}

// any layout -> quads float4 [NQ][H][W] (only needed when the caller's volume is a plane table / slab)
__global__ __launch_bounds__(256) void stm_k_to_quads(Vol in, float4 *__restrict__ out, int D, size_t HW)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int q = blockIdx.y;
    if (p >= HW) return;
    out[(size_t)q * HW + p] = load_quad<false>(in, q, D, p);
}

// quads [q][A][B] -> quads [q][B][A], 32x32 tiles of float4 through LDS
__global__ __launch_bounds__(256) void stm_k_transpose_quads(const float4 *__restrict__ in, float4 *__restrict__ out, int A, int B)
{
    __shared__ float4 tile[32][33];
    const int q = blockIdx.z, b0 = blockIdx.x * 32, a0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const float4 *src = in + (size_t)q * A * B;
    float4 *dst = out + (size_t)q * A * B;
    for (int r = ty; r < 32; r += 8)
        if (a0 + r < A && b0 + tx < B) tile[r][tx] = src[(size_t)(a0 + r) * B + b0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (b0 + r < B && a0 + tx < A) dst[(size_t)(b0 + r) * A + a0 + tx] = tile[tx][r];
}

#define STM_DPP(old, v, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, 0xf, 0xf, false))

// min over the 64 lanes, result broadcast from lane 63.  Six v_min_f32 with a DPP source operand: a lane whose
// DPP source does not exist is simply not written (bound_ctrl off), and min is idempotent, so the row masks of the
// classic reduction are not needed.  Written as inline asm because clang expands fminf(x, dpp(x)) into
// mov-immediate + v_mov_dpp + a canonicalising v_max + v_min (4 instructions per step on the critical path of every
// pixel of every line); the s_nop covers the VALU-write -> DPP-read hazard the assembler does not pad for us.
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

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ line passes
struct HsloLineArgs {
    const float4 *cost[2]; // per view: quads [NQ][nlines][len] in the orientation of this launch
    float4 *out[2];        // per view: two direction volumes, quads [2][NQ][nlines][len]
    const float *avg_a[2]; // per view: own-image averages, [nlines][len]
    const float *avg_b[2]; // per view: other-image averages, [nlines][len]
    int osign[2];          // +1 left view (matched pixel x + d - zd), -1 right view (x - (d - zd))
};


// grid = (lines / WPB, 2 directions, views); block = WPB waves, one line each.
// PERP = false: the line runs along x (matched pixels slide along the line);
// PERP = true : the volume is transposed, the line runs along y and the matched pixel of hypothesis d sits in line
//               `line + osign (d - zd)` at the same position.
// CW = pixels per chunk (8, 16 or 32): the smaller the tile, the more lines are resident per CU.
template <int DPL, int WPB, bool PERP, int CW>
__global__ __launch_bounds__(64 * WPB) void stm_k_hslo_lines(HsloLineArgs a, float T, float P1a, float P1b, float P1c, float P2a,
                                                             float P2b, float P2c, int D, int zd, int nlines, int len)
{
    constexpr int DP = 64 * DPL + 4; // tile row pitch in floats: conflict-free for b128 rows and for lane = d columns
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int line = blockIdx.x * WPB + wave;
    if (line >= nlines) return; // whole wave; there are no block barriers in this kernel
    const int bwd = blockIdx.y, view = blockIdx.z;
    const int NQ = (D + 3) >> 2;
    const int PAD = max(max(zd, D - 1 - zd), 0);
    const int seg_len = ((PERP ? D * (CW + 1) : CW + 2 * PAD) + 3) & ~3; // keep every wave's tile 16-byte aligned
    float *tile = lds + (size_t)wave * (CW * DP + seg_len);
    float *seg = tile + CW * DP;
    const float4 *__restrict__ cost = a.cost[view];
    float4 *__restrict__ out = a.out[view] + (size_t)bwd * NQ * nlines * len;
    const float *__restrict__ avg_a = a.avg_a[view], *__restrict__ avg_b = a.avg_b[view];
    const int osign = a.osign[view];
    constexpr int G = 64 / CW; // pixel groups per wave instruction in the fill and drain
    const int px = lane & (CW - 1), h = lane / CW;
    const float inf = __builtin_inff();

    float prev[DPL], rprev[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) { prev[j] = inf; rprev[j] = 0.f; }
    float lcarry = 0.f; // own-image average at the last position of the previous chunk
    bool started = false;
    const int nchunks = (len + CW - 1) / CW;
    for (int c = 0; c < nchunks; ++c) {
        const int lo = bwd ? max(len - CW * (c + 1), 0) : c * CW;
        const int hi = bwd ? len - CW * c : min(lo + CW, len);
        const int n = hi - lo;
        // ---- fill: lane = pixel, 16-byte quads, two quads per wave instruction
        wave_lds_fence();
        if (px < n) {
            for (int q = h; q < NQ; q += G)
                *(float4 *)(tile + px * DP + 4 * q) = cost[((size_t)q * nlines + line) * len + lo + px];
        }
        if (PERP) { // seg[d][t] = other-image average of line `line + osign (d - zd)` at position lo + t
            for (int d = h; d < D; d += G) {
                const int ln = min(max(line + osign * (d - zd), 0), nlines - 1);
                if (px < n) seg[d * (CW + 1) + px] = avg_b[(size_t)ln * len + lo + px];
            }
        } else { // seg[i] = other-image average of this line at position clamp(lo - PAD + i)
            for (int i = lane; i < n + 2 * PAD; i += 64)
                seg[i] = avg_b[(size_t)line * len + min(max(lo - PAD + i, 0), len - 1)];
        }
        // own-image colour steps of the chunk, classified once per chunk with lane = position: D1 = |avg(p) - avg(p - r)|
        // is the same for every hypothesis, so its class (0: D1 < T, 1: D1 > T, 2: neither) is broadcast per step
        // with one v_readlane and the penalty candidates are picked on the scalar unit
        const float lrow = lane < n ? avg_a[(size_t)line * len + lo + lane] : 0.f;
        int d1cls;
        {
            float pred = bwd ? __shfl_down(lrow, 1) : __shfl_up(lrow, 1); // the position walked just before this one
            if (lane == (bwd ? n - 1 : 0)) pred = lcarry;                   // ... which may be the previous chunk's last
            const float D1 = fabsf(lrow - pred);
            d1cls = D1 < T ? 0 : (D1 > T ? 1 : 2);
            lcarry = __shfl(lrow, bwd ? 0 : n - 1);
        }
        wave_lds_fence();
        // ---- walk: lane = hypothesis, tile updated in place
        for (int k = 0; k < n; ++k) {
            const int t = bwd ? n - 1 - k : k;
            float cc[DPL], r0[DPL];
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                const int d = min(lane + 64 * j, D - 1);
                cc[j] = tile[t * DP + d];
                r0[j] = PERP ? seg[d * (CW + 1) + t] : seg[t + PAD + osign * (d - zd)];
            }
            if (!started) { // first pixel of the line: Cr(p0, d) = C(p0, d)
#pragma unroll
                for (int j = 0; j < DPL; ++j) {
                    prev[j] = lane + 64 * j < D ? cc[j] : inf;
                    rprev[j] = r0[j];
                }
                started = true;
                continue;
            }
            float mloc = prev[0];
#pragma unroll
            for (int j = 1; j < DPL; ++j) mloc = fminf(mloc, prev[j]);
            const float m = wave_min(mloc); // min_k Cr(p-r, k); absent hypotheses hold +inf
            // penalty pairs by the class of D2, given the (uniform) class of D1 -- d_dc_hslo.cu:73-93:
            //   both < T -> a;  exactly one < T and the other > T -> b;  everything else -> c
            const int u = __builtin_amdgcn_readlane(d1cls, t);
            const float P1lt = u == 0 ? P1a : (u == 1 ? P1b : P1c), P2lt = u == 0 ? P2a : (u == 1 ? P2b : P2c); // D2 < T
            const float P1gt = u == 0 ? P1b : P1c, P2gt = u == 0 ? P2b : P2c;                                     // D2 > T
            float cur[DPL];
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                const int d = lane + 64 * j;
                cur[j] = inf;
                if (d < D) {
                    const float D2 = fabsf(r0[j] - rprev[j]);
                    const float P1 = D2 < T ? P1lt : (D2 > T ? P1gt : P1c);
                    const float P2 = D2 < T ? P2lt : (D2 > T ? P2gt : P2c);
                    // Cr(p-r, d-1) + P1 and Cr(p-r, d+1) + P1: the wave shift rides on the add (DPP operand); a lane
                    // without a neighbour keeps +inf, which also covers d = 0 and d = D-1 (hypotheses >= D hold +inf)
                    float tb = inf, ta = inf;
                    if (DPL == 1) {
                        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                                     "v_add_f32_dpp %1, %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf"
                                     : "+v"(tb), "+v"(ta)
                                     : "v"(prev[j]), "v"(P1));
                    } else { // several hypotheses per lane: patch the 64-lane chunk borders
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
                        tb = below + P1;
                        ta = above + P1;
                    }
                    float best = prev[j];
                    if (tb < best) best = tb;
                    if (ta < best) best = ta;
                    { const float tt = m + P2; if (tt < best) best = tt; }
                    float v = cc[j] + best;
                    v = v - m;
                    cur[j] = v;
                    tile[t * DP + d] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < DPL; ++j) { prev[j] = cur[j]; rprev[j] = r0[j]; }
        }
        // ---- drain: lane = pixel again
        wave_lds_fence();
        if (px < n) {
            for (int q = h; q < NQ; q += G)
                out[((size_t)q * nlines + line) * len + lo + px] = *(const float4 *)(tile + px * DP + 4 * q);
        }
    }
}

// ------------------------------------------------------------------ combine + WTA
// C2(p, d) = (((C_lr + C_rl) + C_tb) + C_bt) * 0.25f, then first-lowest-wins WTA (d_dc_wta.cu:19-34).  The two
// vertical results live in the transposed orientation and come back through a 32x32 LDS tile per quad.
struct HsloCombineArgs {
    const float4 *out_h[2], *out_v[2]; // per view: [2][NQ][H][W] and [2][NQ][W][H]
    float *disp[2];
    float *vol[2]; // optional dense [D][H*W] copy of the combined volume
};

__global__ __launch_bounds__(256) void stm_k_hslo_combine_wta(HsloCombineArgs a, int D, int zd, int H, int W)
{
    __shared__ float4 t2[32][33], t3[32][33];
    const int view = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8, each thread owns pixels (x0+tx, y0+ty+8i)
    const int NQ = (D + 3) >> 2;
    const size_t HW = (size_t)H * W, VQ = (size_t)NQ * HW;
    const float4 *__restrict__ oh = a.out_h[view], *__restrict__ ov = a.out_v[view];
    float lowest[4];
    int best[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lowest[i] = 3.402823466e+38f; best[i] = 0; }
    for (int q = 0; q < NQ; ++q) {
        __syncthreads();
        for (int r = ty; r < 32; r += 8) { // transposed volumes: row = x, contiguous along y
            const int x = x0 + r, y = y0 + tx;
            if (x < W && y < H) {
                t2[r][tx] = ov[((size_t)q * W + x) * H + y];
                t3[r][tx] = ov[VQ + ((size_t)q * W + x) * H + y];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = x0 + tx, y = y0 + ty + 8 * i;
            if (x < W && y < H) {
                const size_t p = (size_t)y * W + x;
                const float4 a0 = oh[(size_t)q * HW + p], a1 = oh[VQ + (size_t)q * HW + p];
                const float4 a2 = t2[tx][ty + 8 * i], a3 = t3[tx][ty + 8 * i];
                float s[4];
                s[0] = a0.x + a1.x; s[0] = s[0] + a2.x; s[0] = s[0] + a3.x; s[0] = s[0] * 0.25f;
                s[1] = a0.y + a1.y; s[1] = s[1] + a2.y; s[1] = s[1] + a3.y; s[1] = s[1] * 0.25f;
                s[2] = a0.z + a1.z; s[2] = s[2] + a2.z; s[2] = s[2] + a3.z; s[2] = s[2] * 0.25f;
                s[3] = a0.w + a1.w; s[3] = s[3] + a2.w; s[3] = s[3] + a3.w; s[3] = s[3] * 0.25f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = 4 * q + e;
                    if (d < D) {
                        if (a.vol[view]) a.vol[view][(size_t)d * HW + p] = s[e];
                        if (lowest[i] > s[e]) { lowest[i] = s[e]; best[i] = d; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x0 + tx, y = y0 + ty + 8 * i;
        if (x < W && y < H) a.disp[view][(size_t)y * W + x] = (float)best[i] - (float)zd;
    }
}

// ------------------------------------------------------------------ driver
#ifndef HSLO_CW_H
#define HSLO_CW_H 16
#endif
#ifndef HSLO_CW_V
#define HSLO_CW_V 8
#endif
template <int DPL, int WPB, bool PERP, int CW>
static void hslo_lines_launch(const HsloLineArgs &a, int nviews, float T, const float *P1, const float *P2, int D, int zd, int nlines,
                              int len)
{
    const int PAD = D - 1 - zd > zd ? D - 1 - zd : (zd > 0 ? zd : 0);
    const size_t per_wave = (size_t)CW * (64 * DPL + 4) + (((PERP ? (size_t)D * (CW + 1) : (size_t)CW + 2 * PAD) + 3) & ~(size_t)3);
    const size_t smem = per_wave * WPB * 4;
    if (smem > 64 * 1024)
        STM_CHECK(hipFuncSetAttribute((const void *)stm_k_hslo_lines<DPL, WPB, PERP, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    STM_LAUNCH((stm_k_hslo_lines<DPL, WPB, PERP, CW>), dim3(cdiv(nlines, WPB), 2, nviews), dim3(64 * WPB), smem, stream(), a, T, P1[0],
                       P1[1], P1[2], P2[0], P2[1], P2[2], D, zd, nlines, len);
    STM_CHECK_LAUNCH();
}

template <bool PERP, int CW>
static void hslo_lines_cw(const HsloLineArgs &a, int nviews, float T, const float *P1, const float *P2, int D, int zd, int nlines, int len)
{
    if (D <= 64) hslo_lines_launch<1, 4, PERP, CW>(a, nviews, T, P1, P2, D, zd, nlines, len);
    else if (D <= 128) hslo_lines_launch<2, 2, PERP, CW>(a, nviews, T, P1, P2, D, zd, nlines, len);
    else hslo_lines_launch<4, 1, PERP, CW>(a, nviews, T, P1, P2, D, zd, nlines, len);
}
template <bool PERP>
static void hslo_lines(const HsloLineArgs &a, int nviews, float T, const float *P1, const float *P2, int D, int zd, int nlines, int len)
{
    static const int cw_env = [] { const char *e = getenv(PERP ? "STM_HSLO_CW_V" : "STM_HSLO_CW_H"); return e ? atoi(e) : 0; }();
    const int cw = cw_env ? cw_env : (PERP ? HSLO_CW_V : HSLO_CW_H);
    if (cw == 8) hslo_lines_cw<PERP, 8>(a, nviews, T, P1, P2, D, zd, nlines, len);
    else if (cw == 16) hslo_lines_cw<PERP, 16>(a, nviews, T, P1, P2, D, zd, nlines, len);
    else hslo_lines_cw<PERP, 32>(a, nviews, T, P1, P2, D, zd, nlines, len);
}

// Scanline optimisation + WTA for 1 or 2 views.
// img_a[v] = the view's own image, img_b[v] = the other image; osign[v] = +1 (left view) / -1 (right view).
// Scratch from the current Workspace scope, per view: 5 quad volumes (6 when the input is not already in quads)
// + 4 planes.
void launch_hslo_wta(int nviews, const Vol *cost, const u8 *const *img_a, const u8 *const *img_b, const int *osign,
                     float *const *disp, float *const *vol_out, float T, float H1, float H2, int D, int zd, int H, int W,
                     int elem_sz)
{
    const float P1[3] = {H1, (float)((double)H1 / 4.0), (float)((double)H1 / 10.0)}; // d_dc_hslo.cu:124-127
    const float P2[3] = {H2, (float)((double)H2 / 4.0), (float)((double)H2 / 10.0)};
    const size_t HW = (size_t)H * W;
    const int NQ = (D + 3) / 4;
    const size_t VQ = (size_t)NQ * HW; // float4 elements of one quad volume
    if (D > 256 || D < 1 || nviews < 1 || nviews > 2) {
        fail("hslo: num_disp must be 1..256 and views 1..2", "D", __FILE__, __LINE__);
        return;
    }
    ProfScope p("hslo");
    HsloLineArgs ah, av;
    HsloCombineArgs ac;
    for (int v = 0; v < 2; ++v) {
        const int s = v < nviews ? v : 0;
        ah.osign[v] = av.osign[v] = osign[s];
        if (v >= nviews) {
            ah.cost[v] = ah.cost[0]; ah.out[v] = ah.out[0]; ah.avg_a[v] = ah.avg_a[0]; ah.avg_b[v] = ah.avg_b[0];
            av.cost[v] = av.cost[0]; av.out[v] = av.out[0]; av.avg_a[v] = av.avg_a[0]; av.avg_b[v] = av.avg_b[0];
            ac.out_h[v] = ac.out_h[0]; ac.out_v[v] = ac.out_v[0]; ac.disp[v] = ac.disp[0]; ac.vol[v] = ac.vol[0];
            continue;
        }
        float *planes = Workspace::get<float>(4 * HW);
        STM_LAUNCH(stm_k_hslo_avg, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), img_a[s], img_b[s], planes, planes + HW,
                           planes + 2 * HW, planes + 3 * HW, H, W, elem_sz);
        STM_CHECK_LAUNCH();
        const float4 *cq;
        if (cost[s].quad) cq = (const float4 *)cost[s].base;
        else {
            float4 *conv = Workspace::get<float4>(VQ);
            STM_LAUNCH(stm_k_to_quads, dim3((unsigned)((HW + 255) / 256), NQ), dim3(256), 0, stream(), cost[s], conv, D, HW);
            STM_CHECK_LAUNCH();
            cq = conv;
        }
        float4 *ct = Workspace::get<float4>(VQ);
        STM_LAUNCH(stm_k_transpose_quads, dim3(cdiv(W, 32), cdiv(H, 32), NQ), dim3(256), 0, stream(), cq, ct, H, W);
        STM_CHECK_LAUNCH();
        float4 *oh = Workspace::get<float4>(2 * VQ), *ov = Workspace::get<float4>(2 * VQ);
        ah.cost[v] = cq; ah.out[v] = oh; ah.avg_a[v] = planes; ah.avg_b[v] = planes + HW;
        av.cost[v] = ct; av.out[v] = ov; av.avg_a[v] = planes + 2 * HW; av.avg_b[v] = planes + 3 * HW;
        ac.out_h[v] = oh; ac.out_v[v] = ov; ac.disp[v] = disp[s]; ac.vol[v] = vol_out ? vol_out[s] : nullptr;
    }
    hslo_lines<false>(ah, nviews, T, P1, P2, D, zd, H, W); // left->right and right->left
    hslo_lines<true>(av, nviews, T, P1, P2, D, zd, W, H);  // top->bottom and bottom->top on the transposed volume
    STM_LAUNCH(stm_k_hslo_combine_wta, dim3(cdiv(W, 32), cdiv(H, 32), nviews), dim3(256), 0, stream(), ac, D, zd, H, W);
    STM_CHECK_LAUNCH();
}

} // namespace stm
