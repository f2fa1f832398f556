// stm_kernels_refine.hip -- disparity refinement for gfx950: L/R consistency check, iterative region
// voting, bilateral filter on the disparity map, grow-only gaussian.
//
// Reference stages replaced:
//   dr_dcc_kernel / dr_ddc_kernel / dr_merge_errors_kernel   d_dr_dcc.cu:57-82 / :35-54 / :18-33
//   dr_irv_pre_kernel / dr_irv_kernel_3                      d_dr_irv.cu:134-220 / :17-43
//   filter_bilateral_1_kernel_6                              d_filter_bilateral.cu:222-304
//   filter_gaussian_1_kernel_1 (+ op_invertnormf_kernel)     d_filter_gaussian.cu:9-88 (d_op.cu:7-16)
// All of them are image-sized (<= 17 B/px of HBM traffic) and LDS/latency bound; accumulation orders
// follow the reference loops exactly (row-major taps, sequential float adds, no contraction).
#include "stm_common.h"

namespace stm {

// ------------------------------------------------------------------ L/R check
// phase 1: outlier flags + hit-map scatter (both scatters only ever write the value 0 -> benign races)
__global__ __launch_bounds__(256) void stm_k_dcc_mark(u8 *__restrict__ out_l, u8 *__restrict__ out_r,
                                                      const float *__restrict__ disp_l, const float *__restrict__ disp_r,
                                                      u8 *__restrict__ hit_l, u8 *__restrict__ hit_r, int H, int W)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    size_t row = (size_t)y * W;
    const float thresh = 1.0f; // d_dr_dcc.cu:117
    float dl = disp_l[row + x];
    int c = min(max(x + (int)dl, 0), W - 1);
    if (fabsf(dl - disp_r[row + c]) > thresh) out_l[row + x] = 1;
    hit_r[row + c] = 0; // dr_ddc_kernel: same coordinate (d_dr_dcc.cu:45-48)
    float dr = disp_r[row + x];
    c = min(max(x - (int)dr, 0), W - 1);
    if (fabsf(dr - disp_l[row + c]) > thresh) out_r[row + x] = 1;
    hit_l[row + c] = 0;
}
// phase 2: outlier and never hit -> class 2 (occlusion)
__global__ __launch_bounds__(256) void stm_k_dcc_merge(u8 *__restrict__ out_l, u8 *__restrict__ out_r,
                                                       const u8 *__restrict__ hit_l, const u8 *__restrict__ hit_r, size_t HW)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    if (out_l[p] == 1 && hit_l[p] == 1) out_l[p] = 2;
    if (out_r[p] == 1 && hit_r[p] == 1) out_r[p] = 2;
}

void launch_dcc(u8 *out_l, u8 *out_r, const float *disp_l, const float *disp_r, u8 *hit_l, u8 *hit_r, int H, int W)
{
    size_t HW = (size_t)H * W;
    STM_CHECK(hipMemsetAsync(hit_l, 1, HW, stream())); // d_dr_dcc.cu:107,111
    STM_CHECK(hipMemsetAsync(hit_r, 1, HW, stream()));
    hipLaunchKernelGGL(stm_k_dcc_mark, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), out_l, out_r, disp_l, disp_r, hit_l,
                       hit_r, H, W);
    STM_CHECK_LAUNCH();
    hipLaunchKernelGGL(stm_k_dcc_merge, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), out_l, out_r, hit_l,
                       hit_r, HW);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ iterative region voting
// Outliers are few (2 % of a synthetic frame, ~15 % of a real one, fewer every iteration) but each one
// walks a cross region of ~600 pixels, so a thread per pixel leaves most lanes idle and a thread per
// outlier leaves most of the CHIP idle.  Here: (1) a compaction kernel lists the outlier pixels;
// (2) a persistent grid of waves pulls outliers off the list, one WAVE per outlier: each half-wave takes
// one row of the cross region (32 pixels per step, coalesced), equal bins are merged with ballots and
// counted in a per-wave LDS histogram, and the winner is a wave-wide max over (count, -bin).
// Every outlier's result is independent of the list order, so the atomic compaction is deterministic
// where it matters.  Histogram: max(D,65) bins (the reference's int[65] overflows for D > 65, A-Q17 ii).
// four pixels per thread (one dword of the u8 outlier map); one global atomic per WAVE that holds any outlier
__global__ __launch_bounds__(256) void stm_k_irv_compact(const u8 *__restrict__ outl, uint32_t *__restrict__ list,
                                                         int *__restrict__ count, uint32_t HW)
{
    const uint32_t p = (blockIdx.x * 256u + threadIdx.x) * 4u;
    const int lane = threadIdx.x & 63;
    uint32_t w = 0;
    if (p < HW) {
        if (p + 4 <= HW && (((uintptr_t)outl) & 3) == 0) w = *(const uint32_t *)(outl + p);
        else
            for (uint32_t j = 0; j < 4 && p + j < HW; ++j) w |= (uint32_t)outl[p + j] << (8 * j);
    }
    int c = ((w & 0xff) != 0) + ((w & 0xff00) != 0) + ((w & 0xff0000) != 0) + ((w & 0xff000000u) != 0);
    if (__ballot(c != 0) == 0) return; // wave-uniform
    int incl = c; // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const int wave_total = __shfl(incl, 63);
    int base = 0;
    if (lane == 0) base = atomicAdd(count, wave_total);
    base = __shfl(base, 0);
    int k = base + incl - c;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if ((w >> (8 * j)) & 0xff) list[k++] = p + j;
}

constexpr int IV_WAVES = 4;     // waves per block
constexpr int IV_BLOCKS = 2048; // persistent grid
constexpr int IV_U = 4;         // row pairs whose loads are in flight together

// value held by lane (j & 63) of v0 (j < 64) or v1 (j >= 64), for a wave-uniform j
__device__ __forceinline__ int irv_row_value(int v0, int v1, int j)
{
    return j < 64 ? __builtin_amdgcn_readlane(v0, j) : __builtin_amdgcn_readlane(v1, j - 64);
}

// one LDS atomic per voting lane: equal bins serialise inside the LDS atomic unit (<= 64 cycles), which beats a
// ballot-merge loop whenever a step sees more than a couple of distinct disparities -- and outliers sit exactly
// where the disparity map is noisy
__device__ __forceinline__ void irv_tally(int code, int lane, uint32_t *hist, int &total)
{
    total += __popcll(__ballot(code != -1));
    if (code >= 0) atomicAdd(&hist[code], 1u);
}

__global__ __launch_bounds__(64 * IV_WAVES) void stm_k_irv_vote(const float *__restrict__ disp, const u8 *__restrict__ outl,
                                                                const u8 *__restrict__ aU, const u8 *__restrict__ aD,
                                                                const u8 *__restrict__ aL, const u8 *__restrict__ aR,
                                                                const uint32_t *__restrict__ list, const int *__restrict__ count,
                                                                int *__restrict__ max_disp, int *__restrict__ reliable,
                                                                int H, int W, int nb, int zd, int usd)
{
    extern __shared__ uint32_t hist_all[]; // [IV_WAVES][nb]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, l = lane & 31;
    uint32_t *hist = hist_all + wave * nb;
    const int n = *count;
    for (int i = blockIdx.x * IV_WAVES + wave; i < n; i += gridDim.x * IV_WAVES) {
        const uint32_t p = list[i];
        const int gy = (int)(p / (uint32_t)W), gx = (int)(p - (uint32_t)gy * (uint32_t)W);
        for (int b = lane; b < nb; b += 64) hist[b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int cu = aU[p], cd = aD[p];
        if (cu > usd) cu = usd;   // d_dr_irv.cu:179-180
        cu = min(cu, gy);         // arms built by ca_cross never leave the image; these two clamps only keep a
        cd = min(cd, H - 1 - gy); // caller who passes inconsistent arms from reading outside the planes
        const int nrows = cu + cd + 1; // rows gy-cu .. gy+cd inclusive (SURVEY A-Q17 iii), at most 2*255+1
        const int y_top = gy - cu;
        int total = 0;
        for (int jb = 0; jb < nrows; jb += 128) { // 128 rows per outer step covers every usd <= 63 in one go
            // horizontal arms of the region's rows, fetched once: lane <-> rows jb+lane and jb+64+lane
            int cl0 = 0, w0 = 0, cl1 = 0, w1 = 0;
            if (jb + lane < nrows) {
                const size_t q = (size_t)(y_top + jb + lane) * W + gx;
                cl0 = aL[q];
                w0 = cl0 + (int)aR[q] + 1; // x-armL .. x+armR inclusive
            }
            if (jb + 64 + lane < nrows) {
                const size_t q = (size_t)(y_top + jb + 64 + lane) * W + gx;
                cl1 = aL[q];
                w1 = cl1 + (int)aR[q] + 1;
            }
            const int jend = min(nrows - jb, 128);
            for (int j0 = 0; j0 < jend; j0 += 2 * IV_U) {
                // first 32 pixels of IV_U row pairs: all loads issued before any is consumed
                u8 o[IV_U];
                float dv[IV_U];
                int wd[IV_U];
#pragma unroll
                for (int u = 0; u < IV_U; ++u) {
                    const int ja = min(j0 + 2 * u, 127), jb2 = min(j0 + 2 * u + 1, 127); // rows of the two half-waves
                    const int cl = half ? irv_row_value(cl0, cl1, jb2) : irv_row_value(cl0, cl1, ja);
                    int w = half ? irv_row_value(w0, w1, jb2) : irv_row_value(w0, w1, ja);
                    const int j = j0 + 2 * u + half; // this half-wave's row
                    if (j >= jend) w = 0;
                    wd[u] = w;
                    const int sx = gx - cl + l;
                    o[u] = 1;
                    dv[u] = 0.f;
                    if (l < w && sx >= 0 && sx < W) {
                        const size_t s = (size_t)(y_top + jb + j) * W + sx;
                        o[u] = outl[s];
                        dv[u] = disp[s];
                    }
                }
#pragma unroll
                for (int u = 0; u < IV_U; ++u) {
                    int code = -1; // -1: no vote (outside the row segment, or an outlier itself)
                    if (o[u] == 0) {
                        const int b = (int)dv[u] + zd;      // d_dr_irv.cu:200-201
                        code = (b >= 0 && b < nb) ? b : -2; // -2: reliable, but its bin is out of range
                    }
                    irv_tally(code, lane, hist, total);
                }
                // rows wider than 32 pixels: remaining chunks
#pragma unroll
                for (int u = 0; u < IV_U; ++u) {
                    const int wmax = max(__builtin_amdgcn_readlane(wd[u], 0), __builtin_amdgcn_readlane(wd[u], 32));
                    if (wmax <= 32) continue;
                    const int j = j0 + 2 * u + half;
                    const int ja = min(j0 + 2 * u, 127), jb2 = min(j0 + 2 * u + 1, 127);
                    const int cl = half ? irv_row_value(cl0, cl1, jb2) : irv_row_value(cl0, cl1, ja);
                    for (int c0 = 32; c0 < wmax; c0 += 32) {
                        const int xo = c0 + l, sx = gx - cl + xo;
                        int code = -1;
                        if (xo < wd[u] && sx >= 0 && sx < W) {
                            const size_t s = (size_t)(y_top + jb + j) * W + sx;
                            if (outl[s] == 0) {
                                const int b = (int)disp[s] + zd;
                                code = (b >= 0 && b < nb) ? b : -2;
                            }
                        }
                        irv_tally(code, lane, hist, total);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // first bin with the strictly largest count (d_dr_irv.cu:206-215): max over (count, -bin)
        uint32_t key = 0;
        for (int b = lane; b < nb; b += 64) {
            uint32_t c = hist[b];
            uint32_t k = (c << 16) | (uint32_t)(0xFFFF - b);
            if (c != 0 && k > key) key = k;
        }
        for (int o2 = 32; o2 >= 1; o2 >>= 1) {
            uint32_t other = (uint32_t)__shfl_xor((int)key, o2);
            if (other > key) key = other;
        }
        if (lane == 0) {
            int max_d = (int)disp[p]; // default: own disparity (d_dr_irv.cu:182)
            if (key != 0) max_d = (0xFFFF - (int)(key & 0xFFFF)) - zd;
            max_disp[p] = max_d;
            reliable[p] = total;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void stm_k_irv_apply(float *__restrict__ disp, u8 *__restrict__ outl,
                                                       const int *__restrict__ max_disp, int *__restrict__ reliable,
                                                       int thresh_s, float thresh_h, int zd, size_t HW)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    if (outl[p] != 0) {
        int tr = reliable[p], md = max_disp[p];
        // ratio uses the winning BIN INDEX, not its count (d_dr_irv.cu:36, SURVEY A-Q17 iv)
        if (tr > thresh_s && (float)(md + zd) / (float)tr > thresh_h) {
            outl[p] = 0;
            reliable[p] = tr + 1;
            disp[p] = (float)md;
        }
    }
}

void launch_irv(float *disp, u8 *outl, const u8 *up, const u8 *down, const u8 *left, const u8 *right, int *max_disp,
                int *reliable, uint32_t *list, int *counter, int thresh_s, float thresh_h, int H, int W, int D, int zd,
                int usd, int iterations, bool device_flavour)
{
    size_t HW = (size_t)H * W;
    int nb = D > 65 ? D : 65;
    size_t smem = (size_t)nb * IV_WAVES * 4;
    auto vote = [&]() {
        ProfScope p("irv_vote");
        STM_CHECK(hipMemsetAsync(counter, 0, sizeof(int), stream()));
        hipLaunchKernelGGL(stm_k_irv_compact, dim3((unsigned)((HW + 1023) / 1024)), dim3(256), 0, stream(), outl, list, counter,
                           (uint32_t)HW);
        STM_CHECK_LAUNCH();
        hipLaunchKernelGGL(stm_k_irv_vote, dim3(IV_BLOCKS), dim3(64 * IV_WAVES), smem, stream(), disp, outl, up, down, left,
                           right, list, counter, max_disp, reliable, H, W, nb, zd, usd);
        STM_CHECK_LAUNCH();
    };
    auto apply = [&]() {
        hipLaunchKernelGGL(stm_k_irv_apply, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), disp, outl, max_disp,
                           reliable, thresh_s, thresh_h, zd, HW);
        STM_CHECK_LAUNCH();
    };
    if (device_flavour) { // d_dr_irv.cu:259-265
        for (int i = 0; i < iterations; ++i) { vote(); apply(); }
    } else { // d_dr_irv.cu:344-353
        vote();
        for (int i = 0; i < iterations; ++i) apply();
    }
}

// ------------------------------------------------------------------ stencils
constexpr int ST_TX = 64, ST_TY = 4;

// bilateral: w = Gs[dx,dy] * Gc[(int)|v0 - v|], out = sum(w v) / sum(w)   (d_filter_bilateral.cu:278-303)
__global__ __launch_bounds__(ST_TX *ST_TY) void stm_k_bilateral(const float *__restrict__ in, float *__restrict__ out,
                                                               const float *__restrict__ spatial,
                                                               const float *__restrict__ color, int radius, int H, int W,
                                                               int ncolor)
{
    extern __shared__ float sm[];
    const int tw = ST_TX + 2 * radius, th = ST_TY + 2 * radius, kw = 2 * radius + 1;
    float *tile = sm, *sk = sm + tw * th, *ck = sk + kw * kw;
    int tid = threadIdx.y * ST_TX + threadIdx.x;
    int x0 = blockIdx.x * ST_TX, y0 = blockIdx.y * ST_TY;
    for (int i = tid; i < tw * th; i += ST_TX * ST_TY) {
        int ty = i / tw, tx = i - ty * tw;
        int gx = min(max(x0 + tx - radius, 0), W - 1), gy = min(max(y0 + ty - radius, 0), H - 1);
        tile[i] = in[(size_t)gy * W + gx];
    }
    for (int i = tid; i < kw * kw; i += ST_TX * ST_TY) sk[i] = spatial[i];
    for (int i = tid; i < ncolor; i += ST_TX * ST_TY) ck[i] = color[i];
    __syncthreads();
    int gx = x0 + threadIdx.x, gy = y0 + threadIdx.y;
    if (gx >= W || gy >= H) return;
    float va = tile[(threadIdx.y + radius) * tw + threadIdx.x + radius];
    float norm = 0.0f, res = 0.0f;
    for (int y = 0; y < kw; ++y) {
        const float *trow = tile + (threadIdx.y + y) * tw + threadIdx.x;
        const float *krow = sk + y * kw;
        for (int x = 0; x < kw; ++x) {
            float vs = trow[x];
            int ci = (int)fabsf(va - vs);
            ci = min(ci, ncolor - 1);
            float w = krow[x] * ck[ci];
            norm = norm + w;
            float t = vs * w;
            res = res + t;
        }
    }
    out[(size_t)gy * W + gx] = res / norm;
}

void launch_bilateral(const float *in, float *out, const float *spatial, const float *color, int radius, int H, int W,
                      int D)
{
    int tw = ST_TX + 2 * radius, th = ST_TY + 2 * radius, kw = 2 * radius + 1;
    size_t smem = (size_t)(tw * th + kw * kw + D) * 4;
    ProfScope p("bilateral");
    hipLaunchKernelGGL(stm_k_bilateral, dim3(cdiv(W, ST_TX), cdiv(H, ST_TY)), dim3(ST_TX, ST_TY), smem, stream(), in, out,
                       spatial, color, radius, H, W, D);
    STM_CHECK_LAUNCH();
}

// grow-only gaussian: out = max(in, blur(in)); INVERT fuses op_invertnormf (x -> 1 - x) into the tile load
__global__ __launch_bounds__(ST_TX *ST_TY) void stm_k_gaussian_max(const float *__restrict__ in, float *__restrict__ out,
                                                                  const float *__restrict__ spatial, int radius, int H,
                                                                  int W, int invert)
{
    extern __shared__ float sm[];
    const int tw = ST_TX + 2 * radius, th = ST_TY + 2 * radius, kw = 2 * radius + 1;
    float *tile = sm, *sk = sm + tw * th;
    int tid = threadIdx.y * ST_TX + threadIdx.x;
    int x0 = blockIdx.x * ST_TX, y0 = blockIdx.y * ST_TY;
    for (int i = tid; i < tw * th; i += ST_TX * ST_TY) {
        int ty = i / tw, tx = i - ty * tw;
        int gx = min(max(x0 + tx - radius, 0), W - 1), gy = min(max(y0 + ty - radius, 0), H - 1);
        float v = in[(size_t)gy * W + gx];
        tile[i] = invert ? 1.0f - v : v;
    }
    for (int i = tid; i < kw * kw; i += ST_TX * ST_TY) sk[i] = spatial[i];
    __syncthreads();
    int gx = x0 + threadIdx.x, gy = y0 + threadIdx.y;
    if (gx >= W || gy >= H) return;
    float va = tile[(threadIdx.y + radius) * tw + threadIdx.x + radius];
    float norm = 0.0f, res = 0.0f;
    for (int y = 0; y < kw; ++y) {
        const float *trow = tile + (threadIdx.y + y) * tw + threadIdx.x;
        const float *krow = sk + y * kw;
        for (int x = 0; x < kw; ++x) {
            float w = krow[x];
            norm = norm + w;
            float t = trow[x] * w;
            res = res + t;
        }
    }
    float q = res / norm;
    out[(size_t)gy * W + gx] = (va < q) ? q : va; // d_filter_gaussian.cu:84-87
}

void launch_gaussian_max(const float *in, float *out, const float *spatial, int radius, int H, int W, bool invert_input)
{
    int tw = ST_TX + 2 * radius, th = ST_TY + 2 * radius, kw = 2 * radius + 1;
    size_t smem = (size_t)(tw * th + kw * kw) * 4;
    ProfScope p("gaussian_max");
    hipLaunchKernelGGL(stm_k_gaussian_max, dim3(cdiv(W, ST_TX), cdiv(H, ST_TY)), dim3(ST_TX, ST_TY), smem, stream(), in, out,
                       spatial, radius, H, W, invert_input ? 1 : 0);
    STM_CHECK_LAUNCH();
}

} // namespace stm
