// stm_kernels_cost.hip -- AD-Census cost initialisation for gfx950 (wave64).
//
// Reference stages replaced:
//   mux_average_kernel        d_mux_common.cu:7-21     grey
//   tx_census_9x7_kernel_3    d_ci_census.cu:18-50     census transform
//   alu_hamdist_64            d_alu.cu:7-15            "Hamming" distance (low 32 bits, bit 31 weighs 33)
//   ci_ad_kernel_5            d_ci_ad.cu:73-159        absolute-difference cost
//   ci_census_kernel_6        d_ci_census.cu:197-254   census cost
//   ci_adcensus_kernel        d_ci_adcensus.cu:10-36   robust combine
// The reference materialises AD and census volumes (4 V of traffic) and combines them in a third
// pass; here one kernel writes the combined volume once.  Design notes:
//   * images are repacked once to one dword per pixel (B | G<<8 | R<<16): every later access is
//     a single aligned dword and |dB|+|dG|+|dR| is one v_sad_u8.
//   * alu_hamdist_64 only ever looks at the low 32 bits of the 48-bit census (int c = a ^ b), so
//     only that word (window rows y = -1, +1, +2, +3) is computed and stored.
//   * rho() is a table lookup (766 + 65 entries in LDS) -> results are bit-identical to the CPU.
#include "stm_common.h"

namespace stm {

// ---------------------------------------------------------------- pack BGR -> BGRX dword
__global__ __launch_bounds__(256) void stm_k_pack_bgrx(const u8 *__restrict__ bgr, uint32_t *__restrict__ packed,
                                                       int n, int elem_sz)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u8 *p = bgr + (size_t)i * elem_sz;
    packed[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

void launch_pack_bgrx(const u8 *bgr, uint32_t *packed, int H, int W, int elem_sz)
{
    int n = H * W;
    STM_LAUNCH(stm_k_pack_bgrx, dim3(cdiv(n, 256)), dim3(256), 0, stream(), bgr, packed, n, elem_sz);
    STM_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- grey + census (low word)
// grey = (u8)(b*c + g*c + r*c), three rounded products, left-to-right adds (d_mux_common.cu:16-20)
__device__ __forceinline__ uint32_t grey_of(uint32_t px)
{
    const float c = 0.33333334f;
    float b = (float)(px & 0xff) * c;
    float g = (float)((px >> 8) & 0xff) * c;
    float r = (float)((px >> 16) & 0xff) * c;
    float s = b + g;
    s = s + r;
    return (uint32_t)s;
}

constexpr int CEN_TX = 64, CEN_TY = 4;
// tile rows y0-1 .. y0+CEN_TY-1+3, cols x0-4 .. x0+CEN_TX-1+4, clamp-to-edge (d_ci_census.cu:39-40)
__global__ __launch_bounds__(CEN_TX *CEN_TY) void stm_k_census32(const uint32_t *__restrict__ packed_a,
                                                                uint32_t *__restrict__ census_a,
                                                                const uint32_t *__restrict__ packed_b,
                                                                uint32_t *__restrict__ census_b, int H, int W)
{
    const uint32_t *__restrict__ packed = blockIdx.z ? packed_b : packed_a; // blockIdx.z = view
    uint32_t *__restrict__ census = blockIdx.z ? census_b : census_a;
    constexpr int TW = CEN_TX + 8, TH = CEN_TY + 4;
    __shared__ u8 g[TH][TW + 4];
    int x0 = blockIdx.x * CEN_TX, y0 = blockIdx.y * CEN_TY;
    int tid = threadIdx.y * CEN_TX + threadIdx.x;
    for (int i = tid; i < TW * TH; i += CEN_TX * CEN_TY) {
        int ty = i / TW, tx = i - ty * TW;
        int gx = min(max(x0 + tx - 4, 0), W - 1);
        int gy = min(max(y0 + ty - 1, 0), H - 1);
        g[ty][tx] = (u8)grey_of(packed[(size_t)gy * W + gx]);
    }
    __syncthreads();
    int gx = x0 + threadIdx.x, gy = y0 + threadIdx.y;
    if (gx >= W || gy >= H) return;
    int cx = threadIdx.x + 4, cy = threadIdx.y + 1;
    uint32_t cmp = g[cy][cx];
    uint32_t w = 0;
    // bit order: y outer (-1, +1, +2, +3 survive the truncation to 32 bits), x inner -4..4 skipping 0,
    // appended MSB first (d_ci_census.cu:35-47)
    const int ys[4] = {-1, 1, 2, 3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int x = -4; x <= 4; ++x) {
            if (x == 0) continue;
            w = (w << 1) | (g[cy + ys[j]][cx + x] < cmp ? 1u : 0u);
        }
    }
    census[(size_t)gy * W + gx] = w;
}

// both images of a pair in one launch
void launch_census32_pair(const uint32_t *packed_l, uint32_t *census_l, const uint32_t *packed_r, uint32_t *census_r, int H, int W)
{
    STM_LAUNCH(stm_k_census32, dim3(cdiv(W, CEN_TX), cdiv(H, CEN_TY), 2), dim3(CEN_TX, CEN_TY), 0, stream(),
                       packed_l, census_l, packed_r, census_r, H, W);
    STM_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- combined cost volume
// popc(x & 0x7fffffff) + 33 * (x >> 31)  ==  the 64-iteration loop of d_alu.cu:7-15 (SURVEY A-Q1)
__device__ __forceinline__ int hamdist_ref(uint32_t a, uint32_t b)
{
    uint32_t x = a ^ b;
    return __popc(x & 0x7fffffffu) + 33 * (int)(x >> 31);
}

constexpr int CI_TX = 256;
// one block = CI_TX pixels of one row, all D hypotheses.
// left  cost: L(x) vs R(clamp(x + o)), right cost: R(x) vs L(clamp(x - o)), o = d - zd  (A-Q6, clean A-Q7)
template <bool QUAD>
__global__ __launch_bounds__(CI_TX) void stm_k_cost_init(const uint32_t *__restrict__ pk_l, const uint32_t *__restrict__ pk_r,
                                                         const uint32_t *__restrict__ cen_l, const uint32_t *__restrict__ cen_r,
                                                         Vol cost_l, Vol cost_r,
                                                         const float *__restrict__ lut_ad_g, const float *__restrict__ lut_census_g,
                                                         int D, int zd, int H, int W, int pad)
{
    extern __shared__ uint32_t sm[];
    const int span = CI_TX + 2 * pad;
    uint32_t *s_pl = sm, *s_pr = sm + span, *s_cl = sm + 2 * span, *s_cr = sm + 3 * span;
    float *s_lut_ad = (float *)(sm + 4 * span);
    float *s_lut_c = s_lut_ad + 768;

    int y = blockIdx.y, x0 = blockIdx.x * CI_TX, tid = threadIdx.x;
    size_t row = (size_t)y * W;
    for (int i = tid; i < span; i += CI_TX) {
        int gx = min(max(x0 + i - pad, 0), W - 1);
        s_pl[i] = pk_l[row + gx];
        s_pr[i] = pk_r[row + gx];
        s_cl[i] = cen_l[row + gx];
        s_cr[i] = cen_r[row + gx];
    }
    for (int i = tid; i < 766; i += CI_TX) s_lut_ad[i] = lut_ad_g[i];
    if (tid < 65) s_lut_c[tid] = lut_census_g[tid];
    __syncthreads();

    int x = x0 + tid;
    if (x >= W) return;
    int c = tid + pad;
    uint32_t pl0 = s_pl[c], pr0 = s_pr[c], cl0 = s_cl[c], cr0 = s_cr[c];
    const int nq = (D + 3) >> 2;
    for (int q = 0; q < nq; ++q) {
        float vl[4], vr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = q * 4 + j;
            vl[j] = 0.f;
            vr[j] = 0.f;
            if (d < D) {
                const int o = d - zd;
                // clamp-to-edge is in GLOBAL coordinates; the tile was filled with clamped pixels, so a plain
                // tile offset reproduces it as long as |o| <= pad (pad = max(zd, D-1-zd)).
                uint32_t pr1 = s_pr[c + o], cr1 = s_cr[c + o];
                uint32_t pl1 = s_pl[c - o], cl1 = s_cl[c - o];
                int ad_l = (int)__builtin_amdgcn_sad_u8(pl0, pr1, 0u);
                int ad_r = (int)__builtin_amdgcn_sad_u8(pr0, pl1, 0u);
                int h_l = hamdist_ref(cl0, cr1);
                int h_r = hamdist_ref(cr0, cl1);
                vl[j] = s_lut_ad[ad_l] + s_lut_c[h_l];
                vr[j] = s_lut_ad[ad_r] + s_lut_c[h_r];
            }
        }
        store_quad<QUAD, false>(cost_l, q, D, row + x, make_float4(vl[0], vl[1], vl[2], vl[3]));
        store_quad<QUAD, false>(cost_r, q, D, row + x, make_float4(vr[0], vr[1], vr[2], vr[3]));
    }
}

void launch_cost_init(const uint32_t *pk_l, const uint32_t *pk_r, const uint32_t *cen_l, const uint32_t *cen_r,
                      Vol cost_l, Vol cost_r, const float *lut_ad, const float *lut_census,
                      int D, int zd, int H, int W)
{
    int pad = zd > D - 1 - zd ? zd : D - 1 - zd;
    if (pad < 0) pad = 0;
    size_t smem = (size_t)(4 * (CI_TX + 2 * pad) + 768 + 72) * 4;
    ProfScope p("cost_init");
    if (cost_l.quad)
        STM_LAUNCH(stm_k_cost_init<true>, dim3(cdiv(W, CI_TX), H), dim3(CI_TX), smem, stream(),
                           pk_l, pk_r, cen_l, cen_r, cost_l, cost_r, lut_ad, lut_census, D, zd, H, W, pad);
    else
        STM_LAUNCH(stm_k_cost_init<false>, dim3(cdiv(W, CI_TX), H), dim3(CI_TX), smem, stream(),
                           pk_l, pk_r, cen_l, cen_r, cost_l, cost_r, lut_ad, lut_census, D, zd, H, W, pad);
    STM_CHECK_LAUNCH();
}

// ref_quirks (stm_set_ref_quirks, non-default; SURVEY A-Q7): the reference's live kernels read one element past their shared
// tiles at d = 0 in two columns of every block of 160 -- the left cost of tx = 0 pairs L(x) with census_l / img_l at
// clamp(x + 160 + zd - 2) (the last element of the LEFT tile, d_ci_census.cu:240-246 with the padding of d_ci_adcensus.cu:117-120),
// the right cost of tx = 159 pairs R(x) with census_r / img_r at clamp(x - 158 - zd) (the first element of the right tile);
// the AD term strays only when D - zd <= zd (d_ci_adcensus.cu:57-59, d_ci_ad.cu:133-144).  This kernel overwrites those
// entries of plane 0 after stm_k_cost_init wrote the clean costs.  One thread per (row, block of 160, side).
__global__ __launch_bounds__(256) void stm_k_cost_quirks(const uint32_t *__restrict__ pk_l, const uint32_t *__restrict__ pk_r,
                                                         const uint32_t *__restrict__ cen_l, const uint32_t *__restrict__ cen_r,
                                                         Vol cost_l, Vol cost_r, const float *__restrict__ lut_ad,
                                                         const float *__restrict__ lut_c, int D, int zd, int H, int W, int nblk)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * nblk * 2) return;
    const int side = t & 1, b = (t >> 1) % nblk, y = (t >> 1) / nblk;
    const int x = b * 160 + (side ? 159 : 0);
    if (x >= W) return;
    const size_t row = (size_t)y * W;
    const bool ad_stray = (D - zd) <= zd;
    const uint32_t *pk_own = side ? pk_r : pk_l, *cen_own = side ? cen_r : cen_l;
    const int xs = min(max(side ? x - 158 - zd : x + 160 + zd - 2, 0), W - 1); // the stray element, in the OWN image
    const int xc = min(max(side ? x + zd : x - zd, 0), W - 1);                   // the clean partner at d = 0, in the other image
    const uint32_t p_other = ad_stray ? pk_own[row + xs] : (side ? pk_l : pk_r)[row + xc];
    const int ad = (int)__builtin_amdgcn_sad_u8(pk_own[row + x], p_other, 0u);
    const int hd = hamdist_ref(cen_own[row + x], cen_own[row + xs]);
    const float c = lut_ad[ad] + lut_c[hd];
    const Vol &dst = side ? cost_r : cost_l;
    if (dst.quad) ((float *)dst.base)[((size_t)0 * dst.plane_stride + row + x) * 4] = c; // hypothesis 0 of quad 0
    else dst.plane(0)[row + x] = c;
}

void launch_cost_quirks(const uint32_t *pk_l, const uint32_t *pk_r, const uint32_t *cen_l, const uint32_t *cen_r, Vol cost_l,
                        Vol cost_r, const float *lut_ad, const float *lut_census, int D, int zd, int H, int W)
{
    const int nblk = cdiv(W, 160), n = H * nblk * 2;
    STM_LAUNCH(stm_k_cost_quirks, dim3(cdiv(n, 256)), dim3(256), 0, stream(), pk_l, pk_r, cen_l, cen_r, cost_l, cost_r, lut_ad,
               lut_census, D, zd, H, W, nblk);
    STM_CHECK_LAUNCH();
}

} // namespace stm
