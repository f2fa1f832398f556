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

// dir: 0 = left->right, 1 = right->left, 2 = top->bottom, 3 = bottom->top
template <int DPL, bool QUAD>
__global__ __launch_bounds__(256) void stm_k_hslo_dir(Vol cost, Vol acc, const float *__restrict__ avg_l,
                                                      const float *__restrict__ avg_r, float T, float P1a, float P1b,
                                                      float P1c, float P2a, float P2b, float P2c, int D, int zd, int H,
                                                      int W, int dir, int first, int osign)
{
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool horiz = dir < 2;
    const int nlines = horiz ? H : W, len = horiz ? W : H;
    if (line >= nlines) return; // whole wave
    const int dx = dir == 0 ? 1 : (dir == 1 ? -1 : 0), dy = dir == 2 ? 1 : (dir == 3 ? -1 : 0);
    const float inf = __builtin_inff();

    float prev[DPL];
    for (int i = 0; i < len; ++i) {
        const int x = horiz ? (dx > 0 ? i : W - 1 - i) : line;
        const int y = horiz ? line : (dy > 0 ? i : H - 1 - i);
        const size_t p = (size_t)y * W + x;
        if (i == 0) {
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
                const int d = lane + 64 * j;
                prev[j] = inf;
                if (d < D) {
                    const float v = *elem_ptr<QUAD>(cost, d, p);
                    prev[j] = v;
                    float *a = elem_ptr<QUAD>(acc, d, p);
                    *a = first ? v : *a + v;
                }
            }
            continue;
        }
        const int px = x - dx, py = y - dy;
        const size_t pp = (size_t)py * W + px;
        float mloc = prev[0];
#pragma unroll
        for (int j = 1; j < DPL; ++j) mloc = fminf(mloc, prev[j]);
        const float m = wave_min(mloc); // min_k Cr(p-r, k); inactive lanes hold +inf
        const float D1 = fabsf(avg_l[p] - avg_l[pp]);
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
                const int o = osign * (d - zd); // matched pixel: x + (d - zd) for the left view, x - (d - zd) for the right
                const int qx = min(max(x + o, 0), W - 1), qpx = min(max(px + o, 0), W - 1);
                const float D2 = fabsf(avg_r[(size_t)y * W + qx] - avg_r[(size_t)py * W + qpx]);
                float P1, P2;
                if (D1 < T && D2 < T) { P1 = P1a; P2 = P2a; }
                else if ((D1 < T && D2 > T) || (D1 > T && D2 < T)) { P1 = P1b; P2 = P2b; }
                else { P1 = P1c; P2 = P2c; }
                float best = prev[j];
                if (d > 0) { const float t = below + P1; if (t < best) best = t; }
                if (d < D - 1) { const float t = above + P1; if (t < best) best = t; }
                { const float t = m + P2; if (t < best) best = t; }
                float v = *elem_ptr<QUAD>(cost, d, p) + best;
                v = v - m;
                cur[j] = v;
                float *a = elem_ptr<QUAD>(acc, d, p);
                *a = first ? v : *a + v;
            }
        }
#pragma unroll
        for (int j = 0; j < DPL; ++j) prev[j] = cur[j];
    }
}

template <int DPL>
static void hslo_launch(Vol cost, Vol acc, const float *avg_l, const float *avg_r, float T, const float *P1, const float *P2, int D,
                        int zd, int H, int W, int dir, int osign)
{
    const int nlines = dir < 2 ? H : W;
    if (cost.quad != acc.quad) fail("hslo: cost and accumulator must share a layout", "quad", __FILE__, __LINE__);
    if (cost.quad)
        hipLaunchKernelGGL((stm_k_hslo_dir<DPL, true>), dim3(cdiv(nlines, 4)), dim3(256), 0, stream(), cost, acc, avg_l, avg_r, T,
                           P1[0], P1[1], P1[2], P2[0], P2[1], P2[2], D, zd, H, W, dir, dir == 0 ? 1 : 0, osign);
    else
        hipLaunchKernelGGL((stm_k_hslo_dir<DPL, false>), dim3(cdiv(nlines, 4)), dim3(256), 0, stream(), cost, acc, avg_l, avg_r, T,
                           P1[0], P1[1], P1[2], P2[0], P2[1], P2[2], D, zd, H, W, dir, dir == 0 ? 1 : 0, osign);
    STM_CHECK_LAUNCH();
}

// avg_l / avg_r: scratch planes of H*W floats each
// osign = +1: `cost` is a LEFT-view volume (hypothesis d pairs x with x + d - zd in img_r); -1: a right-view volume
// (img_l = the right image, img_r = the left image, matched pixel x - (d - zd))
void launch_hslo(Vol cost, Vol acc, const u8 *img_l, const u8 *img_r, float *avg_l, float *avg_r, float T, float H1, float H2,
                 int D, int zd, int H, int W, int elem_sz, int osign)
{
    const float P1[3] = {H1, (float)((double)H1 / 4.0), (float)((double)H1 / 10.0)}; // d_dc_hslo.cu:124-127
    const float P2[3] = {H2, (float)((double)H2 / 4.0), (float)((double)H2 / 10.0)};
    const size_t HW = (size_t)H * W;
    if (D > 256) {
        fail("hslo: num_disp > 256 is not supported", "D", __FILE__, __LINE__);
        return;
    }
    ProfScope p("hslo");
    hipLaunchKernelGGL(stm_k_hslo_avg, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), img_l, img_r, avg_l, avg_r, HW, elem_sz);
    STM_CHECK_LAUNCH();
    for (int dir = 0; dir < 4; ++dir) {
        if (D <= 64) hslo_launch<1>(cost, acc, avg_l, avg_r, T, P1, P2, D, zd, H, W, dir, osign);
        else if (D <= 128) hslo_launch<2>(cost, acc, avg_l, avg_r, T, P1, P2, D, zd, H, W, dir, osign);
        else hslo_launch<4>(cost, acc, avg_l, avg_r, T, P1, P2, D, zd, H, W, dir, osign);
    }
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
