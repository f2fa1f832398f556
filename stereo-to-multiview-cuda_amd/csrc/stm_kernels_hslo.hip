// stm_kernels_hslo.hip -- four-direction scanline optimisation (HSLO) for gfx950.
//
// The reference ships only a stub for this stage (d_dc_hslo.cu:9-29 empty path-cost kernels,
// :97-221 driver that never writes `disp`, call site commented out at image_io.cpp:310-316), so
// parity is UNPINNED: the algorithm is Mei et al. section 3.3 with the penalty rule the reference
// does express (dc_hslo_h_cdiff_kernel, d_dc_hslo.cu:73-93; constants :124-127).  The definition the
// oracle and this file share is written out in oracle/stm_oracle.c (orc_dc_hslo_slab).
//
// Mapping: the recurrence is sequential along a scan line and parallel over lines and hypotheses, so
// ONE WAVE OWNS ONE LINE and its lanes are the hypotheses d (D <= 64 * DPL, DPL values per lane).
// Per pixel of the line: the neighbours Cr(p-r, d+-1) come from DPP wave shifts (no LDS), the minimum
// over d of the new path costs is a 6-step DPP reduction, the right-image colour step D2 is a
// coalesced load (lane d looks at column x + d - zd).  Cost / accumulator accesses touch 64 planes
// at one pixel (one 4-byte element per plane); horizontal lines re-use those cache lines for the next
// 31 pixels, vertical lines share them with the neighbouring columns' waves of the same block.
#include "stm_common.h"

namespace stm {

// colour averages used by the penalty rule: left = integer mean as u8 (d_dc_hslo.cu:57-58), right = float mean (:66-67)
__global__ __launch_bounds__(256) void stm_k_hslo_avg(const u8 *__restrict__ img_l, const u8 *__restrict__ img_r,
                                                      float *__restrict__ avg_l, float *__restrict__ avg_r, size_t HW, int elem_sz)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const u8 *l = img_l + p * elem_sz, *r = img_r + p * elem_sz;
    avg_l[p] = (float)(u8)(((int)l[0] + (int)l[1] + (int)l[2]) / 3);
    avg_r[p] = (float)((double)(float)((int)r[0] + (int)r[1] + (int)r[2]) / 3.0);
}

#define STM_DPP(old, v, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, 0xf, 0xf, false))

// min over the 64 lanes (min is idempotent, so the row masks of the classic reduction are not needed);
// lanes a source does not reach keep +inf.  The result is broadcast from lane 63.
__device__ __forceinline__ float wave_min(float v)
{
    const float inf = __builtin_inff();
    v = fminf(v, STM_DPP(inf, v, 0x111)); // row_shr:1
    v = fminf(v, STM_DPP(inf, v, 0x112)); // row_shr:2
    v = fminf(v, STM_DPP(inf, v, 0x114)); // row_shr:4
    v = fminf(v, STM_DPP(inf, v, 0x118)); // row_shr:8   -> lane 15 of every row holds the row minimum
    v = fminf(v, STM_DPP(inf, v, 0x142)); // row_bcast:15 -> lanes 31 and 63 hold the minimum of two rows
    v = fminf(v, STM_DPP(inf, v, 0x143)); // row_bcast:31 -> lane 63 holds the wave minimum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <bool QUAD> __device__ __forceinline__ float *elem_ptr(const Vol &v, int d, size_t p)
{
    if (QUAD) return v.base + ((size_t)(d >> 2) * v.plane_stride + p) * 4 + (d & 3);
    return v.plane(d) + p;
}

struct HsloArgs {
    Vol cost[2];          // per view
    float *out[2];        // per view: 4 direction volumes as quads, float4 [4][ceil(D/4)][H*W]
    const float *avg_a[2]; // per view: integer-mean plane of the view's own image ("left" role, d_dc_hslo.cu:57-58)
    const float *avg_b[2]; // per view: float-mean plane of the other image ("right" role, :66-67)
    int osign[2];          // +1 left view (matched pixel x + d - zd), -1 right view (x - (d - zd))
};

// Everything a step needs that does NOT depend on the recurrence (cost, the colour averages behind the penalty
// class) is fetched a whole CHUNK of K steps ahead, so only the DPP chain (min over d, neighbours, compares) is on
// the critical path of a line.
template <int DPL> struct HsloRaw {
    float c[DPL], r0[DPL], r1[DPL];
    float l0, l1;
    size_t p;
};

// grid = (lines / 4, 4 directions, views): the four directions and both views are independent, so they run in
// ONE launch (8x the waves of a single direction: a line is a long dependent chain and only ~1-2K lines exist).
// Every direction writes its own volume; the fixed-order sum ((lr + rl) + tb) + bt happens in the combine kernel.
// dir: 0 = left->right, 1 = right->left, 2 = top->bottom, 3 = bottom->top
template <int DPL, int K, bool QUAD>
__global__ __launch_bounds__(256) void stm_k_hslo_dir(HsloArgs a, float T, float P1a, float P1b, float P1c, float P2a,
                                                      float P2b, float P2c, int D, int zd, int H, int W)
{
    const int dir = blockIdx.y, view = blockIdx.z;
    const Vol cost = a.cost[view];
    const size_t HW = (size_t)H * W;
    const int NQ = (D + 3) >> 2;
    float *__restrict__ out = a.out[view] + (size_t)dir * NQ * HW * 4; // element (d, p) at ((d >> 2) * HW + p) * 4 + (d & 3)
    const float *__restrict__ avg_l = a.avg_a[view], *__restrict__ avg_r = a.avg_b[view];
    const int osign = a.osign[view];
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool horiz = dir < 2;
    const int nlines = horiz ? H : W, len = horiz ? W : H;
    if (line >= nlines) return; // whole wave
    const int dx = dir == 0 ? 1 : (dir == 1 ? -1 : 0), dy = dir == 2 ? 1 : (dir == 3 ? -1 : 0);
    const float inf = __builtin_inff();

    auto fetch = [&](int i, HsloRaw<DPL> &in) {
        const int ii = min(i, len - 1); // steps past the end are fetched (in range) and never used
        const int x = horiz ? (dx > 0 ? ii : W - 1 - ii) : line;
        const int y = horiz ? line : (dy > 0 ? ii : H - 1 - ii);
        in.p = (size_t)y * W + x;
        const int px = ii > 0 ? x - dx : x, py = ii > 0 ? y - dy : y; // previous pixel of the line
        in.l0 = avg_l[in.p];
        in.l1 = avg_l[(size_t)py * W + px];
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            const int d = min(lane + 64 * j, D - 1);
            in.c[j] = *elem_ptr<QUAD>(cost, d, in.p);
            const int o = osign * (d - zd);
            const int qx = min(max(x + o, 0), W - 1), qpx = min(max(px + o, 0), W - 1);
            in.r0[j] = avg_r[(size_t)y * W + qx];
            in.r1[j] = avg_r[(size_t)py * W + qpx];
        }
    };

    float prev[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) prev[j] = inf;
    HsloRaw<DPL> nxt[K];
#pragma unroll
    for (int k = 0; k < K; ++k) fetch(k, nxt[k]);
    for (int i0 = 0; i0 < len; i0 += K) {
        HsloRaw<DPL> cur_in[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cur_in[k] = nxt[k];
        if (i0 + K < len) {
#pragma unroll
            for (int k = 0; k < K; ++k) fetch(i0 + K + k, nxt[k]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = i0 + k;
            if (i >= len) break; // wave-uniform
            const HsloRaw<DPL> &in = cur_in[k];
            if (i == 0) {
#pragma unroll
                for (int j = 0; j < DPL; ++j) {
                    const int d = lane + 64 * j;
                    if (d < D) {
                        prev[j] = in.c[j];
                        out[((size_t)(d >> 2) * HW + in.p) * 4 + (d & 3)] = in.c[j];
                    }
                }
                continue;
            }
            float mloc = prev[0];
#pragma unroll
            for (int j = 1; j < DPL; ++j) mloc = fminf(mloc, prev[j]);
            const float m = wave_min(mloc); // min_k Cr(p-r, k); inactive lanes hold +inf
            const float D1 = fabsf(in.l0 - in.l1);
            float cur[DPL];
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                const int d = lane + 64 * j;
                // Cr(p-r, d-1) and Cr(p-r, d+1): wave shifts, patched at the 64-lane chunk borders
                float below = STM_DPP(inf, prev[j], 0x138); // wave_shr:1 : lane i <- lane i-1
                float above = STM_DPP(inf, prev[j], 0x130); // wave_shl:1 : lane i <- lane i+1
                if (j > 0) {
                    const float edge = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prev[j - 1]), 63));
                    if (lane == 0) below = edge;
                }
                if (j + 1 < DPL) {
                    const float edge = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prev[j + 1]), 0));
                    if (lane == 63) above = edge;
                }
                cur[j] = inf;
                if (d < D) {
                    const float D2 = fabsf(in.r0[j] - in.r1[j]);
                    float P1, P2;
                    if (D1 < T && D2 < T) { P1 = P1a; P2 = P2a; }
                    else if ((D1 < T && D2 > T) || (D1 > T && D2 < T)) { P1 = P1b; P2 = P2b; }
                    else { P1 = P1c; P2 = P2c; }
                    float best = prev[j];
                    if (d > 0) { const float t = below + P1; if (t < best) best = t; }
                    if (d < D - 1) { const float t = above + P1; if (t < best) best = t; }
                    { const float t = m + P2; if (t < best) best = t; }
                    float v = in.c[j] + best;
                    v = v - m;
                    cur[j] = v;
                    out[((size_t)(d >> 2) * HW + in.p) * 4 + (d & 3)] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < DPL; ++j) prev[j] = cur[j];
        }
    }
}

// C2(p, d) = (((C_lr + C_rl) + C_tb) + C_bt) * 0.25f, then first-lowest-wins WTA (d_dc_wta.cu:19-34); optionally the
// combined volume is written (dense [D][H*W]) for callers that want it
__global__ __launch_bounds__(256) void stm_k_hslo_combine_wta(HsloArgs a, float *disp0, float *disp1, float *vol0, float *vol1,
                                                              int D, int zd, size_t HW)
{
    const int view = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float4 *__restrict__ o = (const float4 *)a.out[view];
    float *__restrict__ disp = view ? disp1 : disp0;
    float *__restrict__ vol = view ? vol1 : vol0;
    const int NQ = (D + 3) >> 2;
    const size_t V4 = (size_t)NQ * HW;
    float lowest = 3.402823466e+38f;
    int best = 0;
    for (int q = 0; q < NQ; ++q) {
        const size_t i = (size_t)q * HW + p;
        const float4 a0 = o[i], a1 = o[V4 + i], a2 = o[2 * V4 + i], a3 = o[3 * V4 + i];
        float s[4];
        s[0] = a0.x + a1.x; s[0] = s[0] + a2.x; s[0] = s[0] + a3.x; s[0] = s[0] * 0.25f;
        s[1] = a0.y + a1.y; s[1] = s[1] + a2.y; s[1] = s[1] + a3.y; s[1] = s[1] * 0.25f;
        s[2] = a0.z + a1.z; s[2] = s[2] + a2.z; s[2] = s[2] + a3.z; s[2] = s[2] * 0.25f;
        s[3] = a0.w + a1.w; s[3] = s[3] + a2.w; s[3] = s[3] + a3.w; s[3] = s[3] * 0.25f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int d = 4 * q + e;
            if (d < D) {
                if (vol) vol[(size_t)d * HW + p] = s[e];
                if (lowest > s[e]) { lowest = s[e]; best = d; }
            }
        }
    }
    disp[p] = (float)best - (float)zd;
}

template <int DPL>
static void hslo_launch(const HsloArgs &a, int nviews, float T, const float *P1, const float *P2, int D, int zd, int H, int W)
{
    constexpr int K = DPL == 1 ? 8 : (DPL == 2 ? 4 : 2);
    const int nl = H > W ? H : W;
    if (a.cost[0].quad)
        hipLaunchKernelGGL((stm_k_hslo_dir<DPL, K, true>), dim3(cdiv(nl, 4), 4, nviews), dim3(256), 0, stream(), a, T, P1[0], P1[1],
                           P1[2], P2[0], P2[1], P2[2], D, zd, H, W);
    else
        hipLaunchKernelGGL((stm_k_hslo_dir<DPL, K, false>), dim3(cdiv(nl, 4), 4, nviews), dim3(256), 0, stream(), a, T, P1[0], P1[1],
                           P1[2], P2[0], P2[1], P2[2], D, zd, H, W);
    STM_CHECK_LAUNCH();
}

// Scanline optimisation + WTA for 1 or 2 views in three launches (colour averages, 4 directions x views, combine).
// img_a[v] = the view's own image, img_b[v] = the other image; osign[v] = +1 (left view) / -1 (right view).
// Scratch from the current Workspace scope: (4 D + 2) H W floats per view.
void launch_hslo_wta(int nviews, const Vol *cost, const u8 *const *img_a, const u8 *const *img_b, const int *osign,
                     float *const *disp, float *const *vol_out, float T, float H1, float H2, int D, int zd, int H, int W,
                     int elem_sz)
{
    const float P1[3] = {H1, (float)((double)H1 / 4.0), (float)((double)H1 / 10.0)}; // d_dc_hslo.cu:124-127
    const float P2[3] = {H2, (float)((double)H2 / 4.0), (float)((double)H2 / 10.0)};
    const size_t HW = (size_t)H * W;
    if (D > 256 || nviews < 1 || nviews > 2 || (nviews == 2 && cost[0].quad != cost[1].quad)) {
        fail("hslo: num_disp > 256 or bad view count / layouts", "D", __FILE__, __LINE__);
        return;
    }
    ProfScope p("hslo");
    HsloArgs a;
    for (int v = 0; v < 2; ++v) {
        const int s = v < nviews ? v : 0;
        a.cost[v] = cost[s];
        a.osign[v] = osign[s];
        if (v < nviews) {
            a.out[v] = Workspace::get<float>(4 * (size_t)((D + 3) / 4) * 4 * HW);
            float *av = Workspace::get<float>(2 * HW);
            a.avg_a[v] = av;
            a.avg_b[v] = av + HW;
            hipLaunchKernelGGL(stm_k_hslo_avg, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), img_a[s], img_b[s], av,
                               av + HW, HW, elem_sz);
            STM_CHECK_LAUNCH();
        } else {
            a.out[v] = a.out[0]; a.avg_a[v] = a.avg_a[0]; a.avg_b[v] = a.avg_b[0];
        }
    }
    if (D <= 64) hslo_launch<1>(a, nviews, T, P1, P2, D, zd, H, W);
    else if (D <= 128) hslo_launch<2>(a, nviews, T, P1, P2, D, zd, H, W);
    else hslo_launch<4>(a, nviews, T, P1, P2, D, zd, H, W);
    hipLaunchKernelGGL(stm_k_hslo_combine_wta, dim3((unsigned)((HW + 255) / 256), nviews), dim3(256), 0, stream(), a, disp[0],
                       nviews > 1 ? disp[1] : disp[0], vol_out ? vol_out[0] : nullptr,
                       (vol_out && nviews > 1) ? vol_out[1] : nullptr, D, zd, HW);
    STM_CHECK_LAUNCH();
}

__global__ __launch_bounds__(256) void stm_k_scale_volume(Vol v, float s, int D, size_t HW)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    if (v.quad) {
        for (int q = 0; q * 4 < D; ++q) {
            float4 *a = (float4 *)v.base + (size_t)q * v.plane_stride + p;
            float4 t = *a;
            t.x = t.x * s; t.y = t.y * s; t.z = t.z * s; t.w = t.w * s;
            *a = t;
        }
        return;
    }
    for (int d = 0; d < D; ++d) {
        float *a = v.plane(d) + p;
        *a = *a * s;
    }
}
void launch_scale_volume(Vol v, float s, int D, int H, int W)
{
    size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(stm_k_scale_volume, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), v, s, D, HW);
    STM_CHECK_LAUNCH();
}

} // namespace stm
