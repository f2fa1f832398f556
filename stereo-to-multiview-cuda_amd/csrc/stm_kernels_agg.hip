// stm_kernels_agg.hip -- cross construction, cross-based cost aggregation and WTA for gfx950.
//
// Reference stages replaced:
//   ca_cross_construction_kernel  d_ca_cross.cu:17-172
//   ca_cross_hsum_kernel_3        d_ca_cross_sum.cu:243-293   (horizontal window sum)
//   ca_cross_vhsum_kernel_2       d_ca_cross_sum.cu:148-198   (vertical window sum, run on a transposed volume)
//   cost_transpose_kernel_4       d_ca_cross_sum.cu:29-58     (deleted: no transposes here)
//   dc_wta_kernel                 d_dc_wta.cu:9-35
//
// Numerics: every window is summed exactly like the reference -- ascending index, float32, starting
// from 0.0f (d_ca_cross_sum.cu:284-289) -- so the aggregated volume and the WTA indices are bit-identical
// to the CPU oracle.  (A prefix-sum formulation would be cheaper but changes float results: SURVEY
// section 7, hard part 1.)
//
// MI355X mapping: a thread owns one pixel and FOUR consecutive disparity hypotheses.  The four planes'
// values of a pixel sit in one 16-byte LDS slot, so one ds_read_b128 feeds four independent accumulators:
// the per-lane window loop (data-dependent trip count, the same for all four hypotheses because the arms
// do not depend on d) costs one LDS instruction per window element instead of four, at the full
// 256 B/clk/CU LDS rate.  Global traffic is row-contiguous per plane (256 B per wave instruction).
//   H pass: one block = one image row x QPB disparity quads; the whole row lives in LDS (W * 16 B).
//   V pass: one block = a strip of VTX columns x a band of rows x one quad; rows stream top to bottom
//           through an LDS ring, each input row is read from HBM exactly once per band (+ usd halo).
#include "stm_common.h"

namespace stm {

// ------------------------------------------------------------------ cross arms
__device__ __forceinline__ int mad_bgrx(uint32_t a, uint32_t b)
{
    int d0 = abs((int)(a & 0xff) - (int)(b & 0xff));
    int d1 = abs((int)((a >> 8) & 0xff) - (int)((b >> 8) & 0xff));
    int d2 = abs((int)((a >> 16) & 0xff) - (int)((b >> 16) & 0xff));
    return max(max(d0, d1), d2);
}

// arm along (dx,dy): value recorded BEFORE the colour test (SURVEY A-Q9, d_ca_cross.cu:41-69)
__device__ __forceinline__ int one_arm(const uint32_t *__restrict__ img, int W, int H, int tx, int ty, int dx, int dy,
                                       float ucd, float lcd, int usd, int lsd, uint32_t anchor)
{
    uint32_t prev = anchor;
    int arm = 0;
    for (int k = 1; k <= usd; ++k) {
        int cx = tx + dx * k, cy = ty + dy * k;
        if (cx < 0 || cy < 0 || cx > W - 1 || cy > H - 1) break;
        arm = k;
        uint32_t c = img[(size_t)cy * W + cx];
        int ac = mad_bgrx(c, anchor), cp = mad_bgrx(c, prev);
        if (k > lsd) {
            if ((float)ac > ucd) break;
        } else {
            if ((float)ac > lcd || (float)cp > lcd) break;
        }
        prev = c;
    }
    return arm;
}

__global__ __launch_bounds__(256) void stm_k_cross_arms(const uint32_t *__restrict__ img, u8 *__restrict__ up,
                                                        u8 *__restrict__ down, u8 *__restrict__ left,
                                                        u8 *__restrict__ right, float ucd, float lcd, int usd, int lsd,
                                                        int H, int W)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    size_t p = (size_t)y * W + x;
    uint32_t a = img[p];
    up[p] = (u8)one_arm(img, W, H, x, y, 0, -1, ucd, lcd, usd, lsd, a);
    down[p] = (u8)one_arm(img, W, H, x, y, 0, 1, ucd, lcd, usd, lsd, a);
    left[p] = (u8)one_arm(img, W, H, x, y, -1, 0, ucd, lcd, usd, lsd, a);
    right[p] = (u8)one_arm(img, W, H, x, y, 1, 0, ucd, lcd, usd, lsd, a);
}

void launch_cross_arms(const uint32_t *packed, u8 *up, u8 *down, u8 *left, u8 *right, float ucd, float lcd, int usd,
                       int lsd, int H, int W)
{
    ProfScope p("cross_arms");
    hipLaunchKernelGGL(stm_k_cross_arms, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), packed, up, down, left, right,
                       ucd, lcd, usd, lsd, H, W);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ float4 load_quad(const Vol &v, int d0, int D, size_t idx)
{
    float4 r;
    r.x = v.plane(d0)[idx];
    r.y = d0 + 1 < D ? v.plane(d0 + 1)[idx] : 0.f;
    r.z = d0 + 2 < D ? v.plane(d0 + 2)[idx] : 0.f;
    r.w = d0 + 3 < D ? v.plane(d0 + 3)[idx] : 0.f;
    return r;
}
__device__ __forceinline__ void store_quad(const Vol &v, int d0, int D, size_t idx, float4 s)
{
    v.plane(d0)[idx] = s.x;
    if (d0 + 1 < D) v.plane(d0 + 1)[idx] = s.y;
    if (d0 + 2 < D) v.plane(d0 + 2)[idx] = s.z;
    if (d0 + 3 < D) v.plane(d0 + 3)[idx] = s.w;
}

// ------------------------------------------------------------------ horizontal pass
constexpr int AH_T = 256;   // threads per block
constexpr int AH_QPB = 4;   // disparity quads per block

// LDS: float4 tile[W] + u16 arms[W]  (armL | armR << 8)
template <bool WTA>
__global__ __launch_bounds__(AH_T) void stm_k_agg_h(Vol in, Vol out, const u8 *__restrict__ armL,
                                                    const u8 *__restrict__ armR, float *__restrict__ disp,
                                                    int D, int zd, int H, int W, int qpb)
{
    extern __shared__ float4 smem4[];
    float4 *tile = smem4;
    uint16_t *arms = (uint16_t *)(tile + W);
    const int y = blockIdx.x, tid = threadIdx.x;
    const size_t row = (size_t)y * W;
    const int nq = (D + 3) >> 2;
    const int q0 = blockIdx.y * qpb, q1 = min(q0 + qpb, nq);

    for (int x = tid; x < W; x += AH_T) arms[x] = (uint16_t)armL[row + x] | ((uint16_t)armR[row + x] << 8);

    // WTA state for up to 16 pixels per thread would need registers per pixel; instead WTA blocks
    // (qpb == nq) keep the running minimum in LDS next to the tile: best cost + best index per pixel.
    float *best_c = (float *)(arms + ((W + 1) & ~1));
    int *best_d = (int *)(best_c + W);
    if (WTA)
        for (int x = tid; x < W; x += AH_T) { best_c[x] = 3.402823466e+38f; best_d[x] = 0; }

    for (int q = q0; q < q1; ++q) {
        const int d0 = q * 4;
        __syncthreads(); // previous quad's readers are done with the tile (and arms are visible)
        for (int x = tid; x < W; x += AH_T) tile[x] = load_quad(in, d0, D, row + x);
        __syncthreads();
        for (int x = tid; x < W; x += AH_T) {
            uint32_t ar = arms[x];
            int a = x - (int)(ar & 0xff), b = x + (int)(ar >> 8);
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = a; k < b; ++k) {
                float4 v = tile[k];
                s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w;
            }
            if (WTA) {
                // first strictly-lowest cost wins, ascending d (d_dc_wta.cu:19-34)
                float bc = best_c[x]; int bd = best_d[x];
                if (bc > s.x) { bc = s.x; bd = d0; }
                if (d0 + 1 < D && bc > s.y) { bc = s.y; bd = d0 + 1; }
                if (d0 + 2 < D && bc > s.z) { bc = s.z; bd = d0 + 2; }
                if (d0 + 3 < D && bc > s.w) { bc = s.w; bd = d0 + 3; }
                best_c[x] = bc; best_d[x] = bd;
            } else {
                store_quad(out, d0, D, row + x, s);
            }
        }
    }
    if (WTA)
        for (int x = tid; x < W; x += AH_T) disp[row + x] = (float)best_d[x] - (float)zd; // own pixels only: no barrier needed
}

// dynamic LDS above the 64 KiB default needs an explicit opt-in (gfx950 has 160 KiB per CU)
static void allow_lds(const void *func, size_t bytes)
{
    if (bytes > 64 * 1024) STM_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

static size_t agg_h_smem(int W, bool wta)
{
    size_t s = (size_t)W * 16 + (size_t)((W + 1) & ~1) * 2;
    if (wta) s += (size_t)W * 8;
    return s;
}

void launch_agg_h(Vol in, Vol out, const u8 *armL, const u8 *armR, int D, int H, int W)
{
    int nq = (D + 3) / 4;
    allow_lds((const void *)stm_k_agg_h<false>, agg_h_smem(W, false));
    ProfScope p("agg_h");
    hipLaunchKernelGGL(stm_k_agg_h<false>, dim3(H, cdiv(nq, AH_QPB)), dim3(AH_T), agg_h_smem(W, false), stream(), in, out,
                       armL, armR, (float *)nullptr, D, 0, H, W, AH_QPB);
    STM_CHECK_LAUNCH();
}

// last horizontal pass fused with WTA: the aggregated volume is consumed in LDS and never written
void launch_agg_h_wta(Vol in, const u8 *armL, const u8 *armR, float *disp, int D, int zd, int H, int W)
{
    int nq = (D + 3) / 4;
    Vol none = vol_slab(nullptr, 0);
    allow_lds((const void *)stm_k_agg_h<true>, agg_h_smem(W, true));
    ProfScope p("agg_hw");
    hipLaunchKernelGGL(stm_k_agg_h<true>, dim3(H, 1), dim3(AH_T), agg_h_smem(W, true), stream(), in, none, armL, armR,
                       disp, D, zd, H, W, nq);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ vertical pass
constexpr int AV_TX = 32;   // columns per block
constexpr int AV_TY = 8;    // thread rows per block (256 threads)
constexpr int AV_CH = 8;    // output rows per step (= AV_TY: one row per thread row)
static int av_band(int H) { int b = (cdiv(H, 4) + AV_CH - 1) / AV_CH * AV_CH; return b < AV_CH ? AV_CH : b; } // output rows per block

// ring[R][AV_TX] of float4, R = power of two >= 2*usd + AV_CH.  Row r of the plane lives in slot r & (R-1).
__global__ __launch_bounds__(AV_TX *AV_TY) void stm_k_agg_v(Vol in, Vol out, const u8 *__restrict__ armU,
                                                            const u8 *__restrict__ armD, int D, int H, int W, int usd,
                                                            int R, int band)
{
    extern __shared__ float4 ring[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * AV_TX + tx;
    const int yb0 = blockIdx.y * band, yb1 = min(yb0 + band, H);
    const int d0 = blockIdx.z * 4;
    const int mask = R - 1;
    const bool xin = x < W;

    int loaded = max(yb0 - usd, 0); // rows [max(yb0-usd,0), loaded) are in the ring
    for (int y0 = yb0; y0 < yb1; y0 += AV_CH) {
        // rows needed by this step: up to y0 + AV_CH - 1 + usd - 1 (window is half-open at the bottom)
        const int need = min(y0 + AV_CH + usd - 1, H);
        __syncthreads(); // everyone finished the previous step before its oldest rows are overwritten
        for (int r = loaded + ty; r < need; r += AV_TY)
            ring[(r & mask) * AV_TX + tx] = xin ? load_quad(in, d0, D, (size_t)r * W + x) : make_float4(0, 0, 0, 0);
        loaded = max(loaded, need);
        __syncthreads();
        const int y = y0 + ty;
        if (xin && y < yb1) {
            const size_t p = (size_t)y * W + x;
            int a = y - (int)armU[p], b = y + (int)armD[p];
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = a; k < b; ++k) {
                float4 v = ring[(k & mask) * AV_TX + tx];
                s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w;
            }
            store_quad(out, d0, D, p, s);
        }
    }
}

void launch_agg_v(Vol in, Vol out, const u8 *armU, const u8 *armD, int D, int H, int W, int usd)
{
    int nq = (D + 3) / 4;
    int R = 16;
    while (R < 2 * usd + AV_CH) R <<= 1;
    size_t smem = (size_t)R * AV_TX * 16;
    int band = av_band(H);
    allow_lds((const void *)stm_k_agg_v, smem);
    ProfScope p("agg_v");
    hipLaunchKernelGGL(stm_k_agg_v, dim3(cdiv(W, AV_TX), cdiv(H, band), nq), dim3(AV_TX, AV_TY), smem, stream(), in,
                       out, armU, armD, D, H, W, usd, R, band);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ WTA (un-fused, per-stage API)
__global__ __launch_bounds__(256) void stm_k_wta(Vol cost, float *__restrict__ disp, int D, int zd, size_t HW)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    float lowest = 3.402823466e+38f;
    int best = 0;
    for (int d = 0; d < D; ++d) {
        float c = cost.plane(d)[p];
        if (lowest > c) { lowest = c; best = d; }
    }
    disp[p] = (float)best - (float)zd;
}

void launch_wta(Vol cost, float *disp, int D, int zd, int H, int W)
{
    size_t HW = (size_t)H * W;
    ProfScope p("wta");
    hipLaunchKernelGGL(stm_k_wta, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), cost, disp, D, zd, HW);
    STM_CHECK_LAUNCH();
}

} // namespace stm
