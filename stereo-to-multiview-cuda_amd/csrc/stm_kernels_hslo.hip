// stm_kernels_hslo.hip -- four-direction scanline optimisation (HSLO) for gfx950.
//
// The reference ships only a stub for this stage (d_dc_hslo.cu:9-29 empty path-cost kernels,
// :97-221 driver that never writes `disp`, call site commented out at image_io.cpp:310-316), so
// parity is UNPINNED: the algorithm is Mei et al. section 3.3 with the penalty rule the reference
// does express (dc_hslo_h_cdiff_kernel, d_dc_hslo.cu:73-93; constants :124-127).  The definition the
// oracle and this file share is written out in oracle/stm_oracle.c (orc_dc_hslo_slab).
//
// Mapping: one thread owns one scan line and walks it sequentially; the D path costs of the previous
// pixel live in LDS as prev[d][lane] (each lane touches only its own column -> no barriers, no bank
// conflicts).  Lanes of a wave are neighbouring lines, so for the vertical directions every plane
// access is a coalesced row segment; the horizontal directions stride by a row (served from L2).
#include "stm_common.h"

namespace stm {

constexpr int HS_T = 64;

__device__ __forceinline__ float avg_l_int(const u8 *p) { return (float)(u8)(((int)p[0] + (int)p[1] + (int)p[2]) / 3); }
__device__ __forceinline__ float avg_r_flt(const u8 *p) { return (float)((double)(float)((int)p[0] + (int)p[1] + (int)p[2]) / 3.0); }

// dir: 0 = left->right, 1 = right->left, 2 = top->bottom, 3 = bottom->top
__global__ __launch_bounds__(HS_T) void stm_k_hslo_dir(Vol cost, Vol acc, const u8 *__restrict__ img_l,
                                                       const u8 *__restrict__ img_r, float T, float P1a, float P1b,
                                                       float P1c, float P2a, float P2b, float P2c, int D, int zd, int H,
                                                       int W, int elem_sz, int dir, int first)
{
    extern __shared__ float prev[]; // [D][HS_T]
    const int lane = threadIdx.x;
    const int line = blockIdx.x * HS_T + lane;
    const bool horiz = dir < 2;
    const int nlines = horiz ? H : W, len = horiz ? W : H;
    if (line >= nlines) return;
    const int dx = dir == 0 ? 1 : (dir == 1 ? -1 : 0), dy = dir == 2 ? 1 : (dir == 3 ? -1 : 0);

    for (int i = 0; i < len; ++i) {
        int x, y;
        if (horiz) { y = line; x = dx > 0 ? i : W - 1 - i; }
        else       { x = line; y = dy > 0 ? i : H - 1 - i; }
        const size_t p = (size_t)y * W + x;
        if (i == 0) {
            for (int d = 0; d < D; ++d) {
                float v = cost.plane(d)[p];
                prev[d * HS_T + lane] = v;
                float *a = acc.plane(d) + p;
                *a = first ? v : *a + v;
            }
            continue;
        }
        const int px = x - dx, py = y - dy;
        float m = prev[lane];
        for (int d = 1; d < D; ++d) { float t = prev[d * HS_T + lane]; if (t < m) m = t; }
        const float D1 = fabsf(avg_l_int(img_l + p * elem_sz) - avg_l_int(img_l + ((size_t)py * W + px) * elem_sz));
        float below = 0.f; // old prev[d-1]
        for (int d = 0; d < D; ++d) {
            const int o = d - zd;
            const int qx = min(max(x + o, 0), W - 1), qpx = min(max(px + o, 0), W - 1);
            const float D2 = fabsf(avg_r_flt(img_r + ((size_t)y * W + qx) * elem_sz) -
                                   avg_r_flt(img_r + ((size_t)py * W + qpx) * elem_sz));
            float P1, P2;
            if (D1 < T && D2 < T) { P1 = P1a; P2 = P2a; }
            else if ((D1 < T && D2 > T) || (D1 > T && D2 < T)) { P1 = P1b; P2 = P2b; }
            else { P1 = P1c; P2 = P2c; }
            const float here = prev[d * HS_T + lane];
            float best = here;
            if (d > 0) { float t = below + P1; if (t < best) best = t; }
            if (d < D - 1) { float t = prev[(d + 1) * HS_T + lane] + P1; if (t < best) best = t; }
            { float t = m + P2; if (t < best) best = t; }
            float v = cost.plane(d)[p] + best;
            v = v - m;
            below = here;
            prev[d * HS_T + lane] = v;
            float *a = acc.plane(d) + p;
            *a = first ? v : *a + v;
        }
    }
}

void launch_hslo(Vol cost, Vol acc, const u8 *img_l, const u8 *img_r, float T, float H1, float H2, int D, int zd, int H,
                 int W, int elem_sz)
{
    float P1a = H1, P1b = (float)((double)H1 / 4.0), P1c = (float)((double)H1 / 10.0); // d_dc_hslo.cu:124-127
    float P2a = H2, P2b = (float)((double)H2 / 4.0), P2c = (float)((double)H2 / 10.0);
    size_t smem = (size_t)D * HS_T * 4;
    if (smem > 64 * 1024)
        STM_CHECK(hipFuncSetAttribute((const void *)stm_k_hslo_dir, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int dir = 0; dir < 4; ++dir) {
        int nlines = dir < 2 ? H : W;
        ProfScope p("hslo_dir");
        hipLaunchKernelGGL(stm_k_hslo_dir, dim3(cdiv(nlines, HS_T)), dim3(HS_T), smem, stream(), cost, acc, img_l, img_r, T,
                           P1a, P1b, P1c, P2a, P2b, P2c, D, zd, H, W, elem_sz, dir, dir == 0 ? 1 : 0);
        STM_CHECK_LAUNCH();
    }
}

__global__ __launch_bounds__(256) void stm_k_scale_volume(Vol v, float s, int D, size_t HW)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
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
