// stm_kernels_dibr.hip -- depth-image-based rendering and view multiplexing for gfx950.
//
// Reference stages replaced:
//   demux_sbs                      d_demux_common.cu:8-33
//   dibr_find_occlusion_kernel     d_dibr_occl.cu:114-128   (hit-map scatter)
//   filter_bleed_1_kernel          d_filter.cu:105-139
//   dibr_occl_to_mask_kernel       d_dibr_occl.cu:17-31
//   dibr_backward_warp_kernel x2 + mux_merge_AB_kernel   d_dibr_bwarp.cu:5-22, d_mux_common.cu:23-46
//   dibr_forward_warp_kernel       d_dibr_fwarp.cu:9-25     (deterministic here)
//   mux_multiview_kernel_2 / mux_multiview_kernel        d_mux_multiview.cu:38-84 / :86-124
// The reference synthesises one view with two warp launches, two temporaries and a merge launch;
// here one kernel gathers both sources and blends in registers (same arithmetic, same truncations).
#include "stm_common.h"

namespace stm {

// ------------------------------------------------------------------ side-by-side splitter
__global__ __launch_bounds__(256) void stm_k_demux_sbs(u8 *__restrict__ l, u8 *__restrict__ r, const u8 *__restrict__ sbs,
                                                       int H, int Wsbs, int W, int elem_sz)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= Wsbs) return;
    const u8 *s = sbs + ((size_t)y * Wsbs + x) * elem_sz;
    u8 *d;
    if (x < W) d = l + ((size_t)y * W + x) * elem_sz;
    else if (x - W < W) d = r + ((size_t)y * W + (x - W)) * elem_sz;
    else return;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}
// frame pipeline: the split also emits the two derived pixel formats the disparity stages read, so the frame is
// touched once (requires Wsbs >= 2 W: every pixel of both halves exists)
__global__ __launch_bounds__(256) void stm_k_demux_sbs_packed(u8 *__restrict__ l, u8 *__restrict__ r, uint32_t *__restrict__ pk_l,
                                                              uint32_t *__restrict__ pk_r, uint32_t *__restrict__ wide_l,
                                                              uint32_t *__restrict__ wide_r, const u8 *__restrict__ sbs, int Wsbs,
                                                              int W, int elem_sz)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= 2 * W) return;
    const u8 *s = sbs + ((size_t)y * Wsbs + x) * elem_sz;
    const bool right = x >= W;
    const size_t p = (size_t)y * W + (right ? x - W : x);
    const uint32_t b = s[0], g = s[1], rr = s[2];
    u8 *d = (right ? r : l) + p * elem_sz;
    d[0] = (u8)b; d[1] = (u8)g; d[2] = (u8)rr;
    (right ? pk_r : pk_l)[p] = b | (g << 8) | (rr << 16);
    (right ? wide_r : wide_l)[p] = b | (g << 10) | (rr << 20);
}
void launch_demux_sbs_packed(u8 *l, u8 *r, uint32_t *pk_l, uint32_t *pk_r, uint32_t *wide_l, uint32_t *wide_r, const u8 *sbs, int H,
                             int Wsbs, int W, int elem_sz)
{
    STM_LAUNCH(stm_k_demux_sbs_packed, dim3(cdiv(2 * W, 256), H), dim3(256), 0, stream(), l, r, pk_l, pk_r, wide_l, wide_r,
                       sbs, Wsbs, W, elem_sz);
    STM_CHECK_LAUNCH();
}
void launch_demux_sbs(u8 *l, u8 *r, const u8 *sbs, int H, int Wsbs, int W, int elem_sz)
{
    STM_LAUNCH(stm_k_demux_sbs, dim3(cdiv(Wsbs, 256), H), dim3(256), 0, stream(), l, r, sbs, H, Wsbs, W, elem_sz);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ hit maps ("occlusion")
// occl_r[clamp(x + (int)(dL * 1))] = 1 ; occl_l[clamp(x + (int)(dR * -1))] = 1  (d_dibr_occl.cu:124-127, :156-157)
__global__ __launch_bounds__(256) void stm_k_occl(u8 *__restrict__ occl_l, u8 *__restrict__ occl_r,
                                                  const float *__restrict__ disp_l, const float *__restrict__ disp_r, int H, int W)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    size_t row = (size_t)y * W;
    int sd = (int)(disp_l[row + x] * 1.0f);
    occl_r[row + min(max(x + sd, 0), W - 1)] = 1;
    sd = (int)(disp_r[row + x] * -1.0f);
    occl_l[row + min(max(x + sd, 0), W - 1)] = 1;
}
void launch_occl(u8 *occl_l, u8 *occl_r, const float *disp_l, const float *disp_r, int H, int W)
{
    size_t HW = (size_t)H * W;
    STM_CHECK(hipMemsetAsync(occl_l, 0, HW, stream())); // d_dibr_occl.cu:149-150
    STM_CHECK(hipMemsetAsync(occl_r, 0, HW, stream()));
    STM_LAUNCH(stm_k_occl, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), occl_l, occl_r, disp_l, disp_r, H, W);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ majority dilate ("bleed")
__global__ __launch_bounds__(256) void stm_k_bleed(const u8 *__restrict__ in, u8 *__restrict__ out, int radius, int ksz, int H, int W)
{
    int tx = blockIdx.x * 256 + threadIdx.x, ty = blockIdx.y;
    if (tx >= W) return;
    u8 va = in[(size_t)ty * W + tx];
    int cnt = 0;
    for (int y = -radius; y <= radius; ++y)
        for (int x = -radius; x <= radius; ++x) {
            int sx = tx + x, sy = ty + y;
            if (sx < 0) sx = -sx; // the reference's odd border rule, d_filter.cu:124-127
            if (sy < 0) sy = -sy;
            if (sx > W - 1) sx = W - 1 - x;
            if (sy > H - 1) sy = H - 1 - y;
            sx = min(max(sx, 0), W - 1); sy = min(max(sy, 0), H - 1);
            if (in[(size_t)sy * W + sx] > 0) cnt = cnt + 1;
        }
    out[(size_t)ty * W + tx] = ((double)cnt > (ksz - 1) * 0.30) ? (u8)1 : va;
}
void launch_bleed(const u8 *in, u8 *out, int radius, int H, int W)
{
    int ksz = (2 * radius + 1) * (2 * radius + 1);
    STM_LAUNCH(stm_k_bleed, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), in, out, radius, ksz, H, W);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ 3x3 "median" (d_filter.cu:7-45)
// Samples by flat index without border handling (a step off the row lands in the neighbouring row; an index
// outside the buffer -- an out-of-bounds read in the reference -- is clamped), selection sort on the values
// truncated to int with the truncated values written back by every swap, slot 4 is the result.
__global__ __launch_bounds__(256) void stm_k_median3(const float *__restrict__ in, float *__restrict__ out, int H, int W)
{
    const int tx = blockIdx.x * 256 + threadIdx.x, ty = blockIdx.y;
    if (tx >= W) return;
    const long HW = (long)H * W;
    float v[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        long q = (long)(tx + n % 3 - 1) + (long)(ty + n / 3 - 1) * W;
        q = min(max(q, 0l), HW - 1);
        v[n] = in[q];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int cur = (int)v[i]; // v_cvt_i32_f32: truncates, saturates, NaN -> 0
#pragma unroll
        for (int j = i; j < 9; ++j) {
            const int comp = (int)v[j];
            if (comp < cur) {
                v[j] = (float)cur;
                v[i] = (float)comp;
                cur = comp;
            }
        }
    }
    out[(size_t)ty * W + tx] = v[4];
}
void launch_median3(const float *in, float *out, int H, int W)
{
    STM_LAUNCH(stm_k_median3, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), in, out, H, W);
    STM_CHECK_LAUNCH();
}

__global__ __launch_bounds__(256) void stm_k_occl_to_mask(float *__restrict__ ml, float *__restrict__ mr,
                                                          const u8 *__restrict__ ol, const u8 *__restrict__ orr, size_t HW)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    ml[p] = ol[p] == 1 ? 1.0f : 0.0f;
    mr[p] = orr[p] == 1 ? 1.0f : 0.0f;
}
void launch_occl_to_mask(float *mask_l, float *mask_r, const u8 *occl_l, const u8 *occl_r, int H, int W)
{
    size_t HW = (size_t)H * W;
    STM_LAUNCH(stm_k_occl_to_mask, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), mask_l, mask_r, occl_l, occl_r, HW);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ hit maps -> bleed(1) -> masks, fused (frame pipeline)
// d_io.cu:165-176 runs dibr_occl, filter_bleed_1(radius 1) on both maps and dibr_occl_to_mask: two zero-fills, the
// scatter, two stencils with a copy back each, the mask kernel.  The scatter never leaves its image row and the
// stencil is 3x3, so one block builds the hit maps of the (at most) three rows its output row reads in LDS, applies
// the majority rule with the reference's border rule and writes both float masks: one launch, no byte planes.
__global__ __launch_bounds__(256) void stm_k_hitmask_rows(float *__restrict__ mask_l, float *__restrict__ mask_r,
                                                          const float *__restrict__ disp_l, const float *__restrict__ disp_r,
                                                          int H, int W)
{
    extern __shared__ u8 hm_lds[]; // [3 rows][left W | right W]
    const int ty = blockIdx.x;
    int src[3]; // image rows behind window rows -1, 0, +1 (d_filter.cu:124-127: sy < 0 -> -sy, sy > H-1 -> H-1-y, then in range)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int y = j - 1;
        int sy = ty + y;
        if (sy < 0) sy = -sy;
        if (sy > H - 1) sy = H - 1 - y;
        src[j] = min(max(sy, 0), H - 1);
    }
    for (int i = threadIdx.x; i < 6 * W; i += 256) hm_lds[i] = 0; // d_dibr_occl.cu:149-150
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        u8 *ol = hm_lds + (size_t)j * 2 * W, *orr = ol + W;
        const size_t row = (size_t)src[j] * W;
        for (int x = threadIdx.x; x < W; x += 256) {
            int sd = (int)(disp_l[row + x] * 1.0f);
            orr[min(max(x + sd, 0), W - 1)] = 1; // d_dibr_occl.cu:124-127
            sd = (int)(disp_r[row + x] * -1.0f);
            ol[min(max(x + sd, 0), W - 1)] = 1;
        }
    }
    __syncthreads();
    const size_t orow = (size_t)ty * W;
    for (int tx = threadIdx.x; tx < W; tx += 256) {
        int cnt_l = 0, cnt_r = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const u8 *ol = hm_lds + (size_t)j * 2 * W, *orr = ol + W;
#pragma unroll
            for (int x = -1; x <= 1; ++x) {
                int sx = tx + x;
                if (sx < 0) sx = -sx;
                if (sx > W - 1) sx = W - 1 - x;
                sx = min(max(sx, 0), W - 1);
                cnt_l += ol[sx] > 0;
                cnt_r += orr[sx] > 0;
            }
        }
        // bleed: count > (9 - 1) * 0.30 -> 1, else the centre value (d_filter.cu:131-137); mask = (value == 1) (d_dibr_occl.cu:17-31)
        const u8 vl = ((double)cnt_l > 8 * 0.30) ? (u8)1 : hm_lds[2 * W + tx];
        const u8 vr = ((double)cnt_r > 8 * 0.30) ? (u8)1 : hm_lds[3 * W + tx];
        mask_l[orow + tx] = vl == 1 ? 1.0f : 0.0f;
        mask_r[orow + tx] = vr == 1 ? 1.0f : 0.0f;
    }
}
void launch_hitmask_rows(float *mask_l, float *mask_r, const float *disp_l, const float *disp_r, int H, int W)
{
    const size_t smem = 6 * (size_t)W;
    if (smem > 64 * 1024) STM_CHECK(hipFuncSetAttribute((const void *)stm_k_hitmask_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    STM_LAUNCH(stm_k_hitmask_rows, dim3(H), dim3(256), smem, stream(), mask_l, mask_r, disp_l, disp_r, H, W);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ one synthesised view
// outL = (u8)(L[sxL] * maskR), sxL = (int)clamp(x + dR * (-shift));  outR = (u8)(R[sxR] * maskL),
// sxR = (int)clamp(x + dL * (1 - shift))   -- the truncation makes alu_bilinear_interp a nearest fetch
// (d_dibr_bwarp.cu:16-21, SURVEY A-Q20);  out = (u8)((1-m) * outL) + (u8)(m * outR), u8 wrap (A-Q22).
__global__ __launch_bounds__(256) void stm_k_view_synth(u8 *__restrict__ out, const u8 *__restrict__ img_l,
                                                        const u8 *__restrict__ img_r, const float *__restrict__ disp_l,
                                                        const float *__restrict__ disp_r, const float *__restrict__ mask_l,
                                                        const float *__restrict__ mask_r, const float *__restrict__ blend,
                                                        float shift_l, float shift_r, int H, int W, int elem_sz)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    size_t row = (size_t)y * W, p = row + x;
    float wmax = (float)(W - 1);
    float sd = disp_r[p] * shift_l;
    float fx = (float)x + sd;
    int sxl = (int)fminf(fmaxf(fx, 0.0f), wmax);
    sd = disp_l[p] * shift_r;
    fx = (float)x + sd;
    int sxr = (int)fminf(fmaxf(fx, 0.0f), wmax);
    float vmr = mask_r[p], vml = mask_l[p], m = blend[p];
    float one_m = 1.0f - m;
    const u8 *sl = img_l + (row + sxl) * elem_sz, *sr = img_r + (row + sxr) * elem_sz;
    u8 *o = out + p * elem_sz;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        u8 a = (u8)((float)sl[c] * vmr); // left-sourced pixel
        u8 b = (u8)((float)sr[c] * vml); // right-sourced pixel
        float cb = one_m * (float)a;
        float ca = m * (float)b;
        o[c] = (u8)((u8)cb + (u8)ca);
    }
}
void launch_view_synth(u8 *out, const u8 *img_l, const u8 *img_r, const float *disp_l, const float *disp_r,
                       const float *mask_l, const float *mask_r, const float *blend, float shift, int H, int W, int elem_sz)
{
    float shift_l = -shift;                               // d_dibr_bwarp.cu:56
    float shift_r = (float)(1.0 - (double)shift);         // :57
    ProfScope p("view_synth");
    STM_LAUNCH(stm_k_view_synth, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), out, img_l, img_r, disp_l, disp_r,
                       mask_l, mask_r, blend, shift_l, shift_r, H, W, elem_sz);
    STM_CHECK_LAUNCH();
}

// All N-2 synthesised views of a frame in one launch (d_io.cu:186-201 loops over d_dibr_dbm): a thread owns one
// pixel, reads its two disparities, two masks and blend weight once and produces that pixel of every view v = 1..N-2
// with shift = 1 - v / (N - 1) evaluated as the reference does (:189, in double, narrowed).
__global__ __launch_bounds__(256) void stm_k_view_synth_all(u8 *__restrict__ views, size_t view_stride, int N,
                                                            const u8 *__restrict__ img_l, const u8 *__restrict__ img_r,
                                                            const float *__restrict__ disp_l, const float *__restrict__ disp_r,
                                                            const float *__restrict__ mask_l, const float *__restrict__ mask_r,
                                                            const float *__restrict__ blend, int H, int W, int elem_sz)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t row = (size_t)y * W, p = row + x;
    const float wmax = (float)(W - 1);
    const float dr = disp_r[p], dl = disp_l[p];
    const float vmr = mask_r[p], vml = mask_l[p], m = blend[p];
    const float one_m = 1.0f - m;
    for (int v = 1; v < N - 1; ++v) {
        const float shift = (float)(1.0 - ((1.0 * (double)(float)v) / ((double)(float)N - 1.0)));
        const float shift_l = -shift;                       // d_dibr_bwarp.cu:56
        const float shift_r = (float)(1.0 - (double)shift); // :57
        float sd = dr * shift_l;
        float fx = (float)x + sd;
        const int sxl = (int)fminf(fmaxf(fx, 0.0f), wmax);
        sd = dl * shift_r;
        fx = (float)x + sd;
        const int sxr = (int)fminf(fmaxf(fx, 0.0f), wmax);
        const u8 *sl = img_l + (row + sxl) * elem_sz, *sr = img_r + (row + sxr) * elem_sz;
        u8 *o = views + (size_t)v * view_stride + p * elem_sz;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const u8 a = (u8)((float)sl[c] * vmr);
            const u8 b = (u8)((float)sr[c] * vml);
            const float cb = one_m * (float)a;
            const float ca = m * (float)b;
            o[c] = (u8)((u8)cb + (u8)ca);
        }
    }
}
// views = base of N view slots of view_stride bytes; slots 1..N-2 are written
void launch_view_synth_all(u8 *views, size_t view_stride, int N, const u8 *img_l, const u8 *img_r, const float *disp_l,
                           const float *disp_r, const float *mask_l, const float *mask_r, const float *blend, int H, int W,
                           int elem_sz)
{
    if (N < 3) return;
    ProfScope p("view_synth");
    STM_LAUNCH(stm_k_view_synth_all, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), views, view_stride, N, img_l, img_r,
                       disp_l, disp_r, mask_l, mask_r, blend, H, W, elem_sz);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ forward warp (deterministic)
// The reference scatter is a data race (SURVEY A-Q23).  Rule here: of all sources landing on one target
// the LARGEST source x wins, which is what a serial ascending-x loop produces.  Pass 1 resolves the
// winner with atomicMax on a (source x + 1) key per target, pass 2 copies the winner's pixel.
__global__ __launch_bounds__(256) void stm_k_fwarp_vote(const float *__restrict__ disp, float shift,
                                                        unsigned long long *__restrict__ keys, int H, int W)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    size_t row = (size_t)y * W;
    int sd = (int)(disp[row + x] * shift);
    int sx = min(max(x + sd, 0), W - 1);
    atomicMax(&keys[row + sx], (unsigned long long)(x + 1));
}
__global__ __launch_bounds__(256) void stm_k_fwarp_copy(u8 *__restrict__ out, const u8 *__restrict__ img,
                                                        const unsigned long long *__restrict__ keys, int H, int W, int elem_sz)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    size_t row = (size_t)y * W;
    unsigned long long k = keys[row + x];
    u8 *o = out + (row + x) * elem_sz;
    if (k == 0) { o[0] = 0; o[1] = 0; o[2] = 0; return; } // holes stay 0 (cudaMemset, d_dibr_fwarp.cu:52)
    const u8 *s = img + (row + (size_t)(k - 1)) * elem_sz;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
}
void launch_fwarp(u8 *out, const u8 *img, const float *disp, float shift, unsigned long long *keys, int H, int W, int elem_sz)
{
    STM_CHECK(hipMemsetAsync(keys, 0, (size_t)H * W * 8, stream()));
    STM_LAUNCH(stm_k_fwarp_vote, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), disp, shift, keys, H, W);
    STM_CHECK_LAUNCH();
    STM_LAUNCH(stm_k_fwarp_copy, dim3(cdiv(W, 256), H), dim3(256), 0, stream(), out, img, keys, H, W, elem_sz);
    STM_CHECK_LAUNCH();
}

// table of view pointers for the interlacer: [0] = right image, [N-1] = left image, rest = synthesised
__global__ void stm_k_view_table(u8 **tab, u8 *first, u8 *last, u8 *mem, size_t stride, int N)
{
    int v = threadIdx.x;
    if (v >= N) return;
    tab[v] = v == 0 ? first : (v == N - 1 ? last : mem + (size_t)v * stride);
}
void launch_view_table(u8 **tab, u8 *first, u8 *last, u8 *mem, size_t stride, int N)
{
    STM_LAUNCH(stm_k_view_table, dim3(1), dim3(64 * ((N + 63) / 64)), 0, stream(), tab, first, last, mem, stride, N);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ bilinear resampling (reduced-resolution mode)
__device__ __forceinline__ u8 bilinear_u8(const u8 *__restrict__ data, int elem_sz, int off, float cx, float cy, int width, int height);

// tx_scale_bilinear_kernel, d_tx_scale.cu:30-52
__global__ __launch_bounds__(256) void stm_k_scale_bilinear(const u8 *__restrict__ in, u8 *__restrict__ out, int in_rows,
                                                            int in_cols, int out_rows, int out_cols, int elem_sz)
{
    int gx = blockIdx.x * 256 + threadIdx.x, gy = blockIdx.y;
    if (gx >= out_cols) return;
    float xs = ((float)gx / (float)out_cols) * (float)in_cols;
    xs = fminf(fmaxf(xs, 0.0f), (float)(in_cols - 1));
    float ys = ((float)gy / (float)out_rows) * (float)in_rows;
    ys = fminf(fmaxf(ys, 0.0f), (float)(in_rows - 1));
    size_t o = ((size_t)gx + (size_t)gy * out_cols) * elem_sz;
    out[o + 0] = bilinear_u8(in, elem_sz, 0, xs, ys, in_cols, in_rows);
    out[o + 1] = bilinear_u8(in, elem_sz, 1, xs, ys, in_cols, in_rows);
    out[o + 2] = bilinear_u8(in, elem_sz, 2, xs, ys, in_cols, in_rows);
}
void launch_scale_bilinear(const u8 *in, u8 *out, int in_rows, int in_cols, int out_rows, int out_cols, int elem_sz)
{
    STM_LAUNCH(stm_k_scale_bilinear, dim3(cdiv(out_cols, 256), out_rows), dim3(256), 0, stream(), in, out, in_rows,
                       in_cols, out_rows, out_cols, elem_sz);
    STM_CHECK_LAUNCH();
}

// tx_disp_scale_kernel + alu_bilinear_interp_f, d_tx_scale.cu:8-28, d_alu.cu:17-43
__global__ __launch_bounds__(256) void stm_k_disp_scale(float *__restrict__ out, const float *__restrict__ in, int out_rows,
                                                        int out_cols, int in_rows, int in_cols, float disp_scale)
{
    int tx = blockIdx.x * 256 + threadIdx.x, ty = blockIdx.y;
    if (tx >= out_cols) return;
    float xs = ((float)tx / (float)out_cols) * (float)in_cols;
    xs = fminf(fmaxf(xs, 0.0f), (float)(in_cols - 1));
    float ys = ((float)ty / (float)out_rows) * (float)in_rows;
    ys = fminf(fmaxf(ys, 0.0f), (float)(in_rows - 1));
    int x0 = (int)floorf(xs), y0 = (int)floorf(ys);
    int x1 = min(x0 + 1, in_cols - 1), y1 = min(y0 + 1, in_rows - 1);
    float wx = xs - (float)x0, wy = ys - (float)y0;
    float v00 = in[(size_t)x0 + (size_t)y0 * in_cols], v01 = in[(size_t)x1 + (size_t)y0 * in_cols];
    float v10 = in[(size_t)x0 + (size_t)y1 * in_cols], v11 = in[(size_t)x1 + (size_t)y1 * in_cols];
    float a = v00 * (1.0f - wx);
    float b = v01 * wx;
    float top = a + b;
    a = v10 * (1.0f - wx);
    b = v11 * wx;
    float bot = a + b;
    a = top * (1.0f - wy);
    b = bot * wy;
    float r = a + b;
    out[(size_t)tx + (size_t)ty * out_cols] = r * disp_scale;
}
void launch_disp_scale(float *out, const float *in, int out_rows, int out_cols, int in_rows, int in_cols, float disp_scale)
{
    STM_LAUNCH(stm_k_disp_scale, dim3(cdiv(out_cols, 256), out_rows), dim3(256), 0, stream(), out, in, out_rows, out_cols,
                       in_rows, in_cols, disp_scale);
    STM_CHECK_LAUNCH();
}

// ------------------------------------------------------------------ multiview interlacer
// fast_bilinear_interp, d_mux_multiview.cu:10-36 (floor, +1 neighbour clamped, u8 truncation)
__device__ __forceinline__ u8 bilinear_u8(const u8 *__restrict__ data, int elem_sz, int off, float cx, float cy, int width, int height)
{
    int x0 = (int)floorf(cx), y0 = (int)floorf(cy);
    int x1 = min(x0 + 1, width - 1), y1 = min(y0 + 1, height - 1);
    float wx = cx - (float)x0, wy = cy - (float)y0;
    float v00 = (float)data[((size_t)x0 + (size_t)y0 * width) * elem_sz + off];
    float v01 = (float)data[((size_t)x1 + (size_t)y0 * width) * elem_sz + off];
    float v10 = (float)data[((size_t)x0 + (size_t)y1 * width) * elem_sz + off];
    float v11 = (float)data[((size_t)x1 + (size_t)y1 * width) * elem_sz + off];
    float a = v00 * (1.0f - wx);
    float b = v01 * wx;
    float top = a + b;
    a = v10 * (1.0f - wx);
    b = v11 * wx;
    float bot = a + b;
    a = top * (1.0f - wy);
    b = bot * wy;
    return (u8)(a + b);
}

__global__ __launch_bounds__(256) void stm_k_mux(const u8 *const *__restrict__ views, u8 *__restrict__ out, int N,
                                                 float y_interval, float inv_y, int ymod, int Hin, int Win, int Hout,
                                                 int Wout, int elem_sz, int variant)
{
    int tx = blockIdx.x * 256 + threadIdx.x, ty = blockIdx.y;
    if (tx >= Wout) return;
    float xs = ((float)tx / (float)Wout) * (float)Win;
    xs = fminf(fmaxf(xs, 0.0f), (float)(Win - 1));
    float ys = ((float)ty / (float)Hout) * (float)Hin;
    ys = fminf(fmaxf(ys, 0.0f), (float)(Hin - 1));
    float x_interval = (float)N;
    float y_view = (float)(ty % ymod) + 1.0f;
    y_view = y_view * x_interval;
    y_view = (variant == 2) ? y_view * inv_y : y_view / y_interval; // :62-63 vs :105-106
    int x_view = (tx * 3 + (int)y_view) % N;
    int r_view = x_view;
    if (r_view < 0) r_view += N;
    int g_view = r_view + 1, b_view = r_view + 2;
    if (g_view >= N) g_view -= N;
    if (b_view >= N) b_view -= N;
    size_t o = ((size_t)tx + (size_t)ty * Wout) * elem_sz;
    out[o + 0] = bilinear_u8(views[b_view], elem_sz, 0, xs, ys, Win, Hin);
    out[o + 1] = bilinear_u8(views[g_view], elem_sz, 1, xs, ys, Win, Hin);
    out[o + 2] = bilinear_u8(views[r_view], elem_sz, 2, xs, ys, Win, Hin);
}
// ------------------------------------------------------------------ view synthesis + interlacing in one pass (frame pipeline)
// The frame pipeline used to write the N - 2 synthesised views (stm_k_view_synth_all: 37 MB at 1080p, 8 views) only for the
// interlacer to pick ONE channel of THREE views per output pixel out of them (stm_k_mux).  Here an output pixel synthesises
// exactly the samples it interlaces: per channel the view v = r, r + 1, r + 2 (mod N) of mux_multiview_kernel_2
// (d_mux_multiview.cu:60-73) at the up to four neighbours of fast_bilinear_interp (:10-36) -- a neighbour whose weight is
// exactly 0 is not evaluated (v * 0 = +0 and a + 0 = a for the finite, non-negative values involved: same result) -- each one
// computed as d_dibr_dbm does (backward warp of both images, masks, blend: d_dibr_bwarp.cu:5-70, d_mux_common.cu:23-46), with
// view 0 = the right image and view N - 1 = the left image (d_io.cu:182-183).  Same arithmetic, no view buffers.
struct SynthArgs {
    const u8 *img_l, *img_r;
    const float *disp_l, *disp_r, *mask_l, *mask_r, *blend;
};
__device__ __forceinline__ u8 synth_sample(const SynthArgs &a, int N, int v, int c, int x, int y, int W, int elem_sz)
{
    const size_t row = (size_t)y * W, p = row + x;
    if (v == 0) return a.img_r[p * elem_sz + c];
    if (v == N - 1) return a.img_l[p * elem_sz + c];
    const float wmax = (float)(W - 1);
    const float shift = (float)(1.0 - ((1.0 * (double)(float)v) / ((double)(float)N - 1.0))); // d_io.cu:189
    const float shift_l = -shift;                       // d_dibr_bwarp.cu:56
    const float shift_r = (float)(1.0 - (double)shift); // :57
    float sd = a.disp_r[p] * shift_l;
    float fx = (float)x + sd;
    const int sxl = (int)fminf(fmaxf(fx, 0.0f), wmax);
    sd = a.disp_l[p] * shift_r;
    fx = (float)x + sd;
    const int sxr = (int)fminf(fmaxf(fx, 0.0f), wmax);
    const float m = a.blend[p], one_m = 1.0f - m;
    const u8 pa = (u8)((float)a.img_l[(row + sxl) * elem_sz + c] * a.mask_r[p]); // left-sourced pixel
    const u8 pb = (u8)((float)a.img_r[(row + sxr) * elem_sz + c] * a.mask_l[p]); // right-sourced pixel
    const float cb = one_m * (float)pa;
    const float ca = m * (float)pb;
    return (u8)((u8)cb + (u8)ca);
}
__device__ __forceinline__ u8 synth_bilinear(const SynthArgs &a, int N, int v, int c, float cx, float cy, int W, int H, int elem_sz)
{
    const int x0 = (int)floorf(cx), y0 = (int)floorf(cy);
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float wx = cx - (float)x0, wy = cy - (float)y0;
    const float v00 = (float)synth_sample(a, N, v, c, x0, y0, W, elem_sz);
    const float v01 = wx != 0.0f ? (float)synth_sample(a, N, v, c, x1, y0, W, elem_sz) : 0.0f;
    float ta = v00 * (1.0f - wx);
    float tb = v01 * wx;
    const float top = ta + tb;
    float bot = 0.0f;
    if (wy != 0.0f) {
        const float v10 = (float)synth_sample(a, N, v, c, x0, y1, W, elem_sz);
        const float v11 = wx != 0.0f ? (float)synth_sample(a, N, v, c, x1, y1, W, elem_sz) : 0.0f;
        ta = v10 * (1.0f - wx);
        tb = v11 * wx;
        bot = ta + tb;
    }
    ta = top * (1.0f - wy);
    tb = bot * wy;
    return (u8)(ta + tb);
}
__global__ __launch_bounds__(256) void stm_k_synth_mux(SynthArgs a, u8 *__restrict__ out, int N, float y_interval, float inv_y, int ymod,
                                                       int Hin, int Win, int Hout, int Wout, int elem_sz, int variant)
{
    const int tx = blockIdx.x * 256 + threadIdx.x, ty = blockIdx.y;
    if (tx >= Wout) return;
    float xs = ((float)tx / (float)Wout) * (float)Win;
    xs = fminf(fmaxf(xs, 0.0f), (float)(Win - 1));
    float ys = ((float)ty / (float)Hout) * (float)Hin;
    ys = fminf(fmaxf(ys, 0.0f), (float)(Hin - 1));
    const float x_interval = (float)N;
    float y_view = (float)(ty % ymod) + 1.0f;
    y_view = y_view * x_interval;
    y_view = (variant == 2) ? y_view * inv_y : y_view / y_interval; // d_mux_multiview.cu:62-63 vs :105-106
    const int x_view = (tx * 3 + (int)y_view) % N;
    int r_view = x_view;
    if (r_view < 0) r_view += N;
    int g_view = r_view + 1, b_view = r_view + 2;
    if (g_view >= N) g_view -= N;
    if (b_view >= N) b_view -= N;
    const size_t o = ((size_t)tx + (size_t)ty * Wout) * elem_sz;
    out[o + 0] = synth_bilinear(a, N, b_view, 0, xs, ys, Win, Hin, elem_sz);
    out[o + 1] = synth_bilinear(a, N, g_view, 1, xs, ys, Win, Hin, elem_sz);
    out[o + 2] = synth_bilinear(a, N, r_view, 2, xs, ys, Win, Hin, elem_sz);
}
void launch_synth_mux(const u8 *img_l, const u8 *img_r, const float *disp_l, const float *disp_r, const float *mask_l, const float *mask_r,
                      const float *blend, u8 *out, int N, float y_interval, float inv_y_interval, int ymod, int Hin, int Win, int Hout,
                      int Wout, int elem_sz, int variant)
{
    SynthArgs a{img_l, img_r, disp_l, disp_r, mask_l, mask_r, blend};
    ProfScope p("synth_mux");
    STM_LAUNCH(stm_k_synth_mux, dim3(cdiv(Wout, 256), Hout), dim3(256), 0, stream(), a, out, N, y_interval, inv_y_interval, ymod, Hin,
               Win, Hout, Wout, elem_sz, variant);
    STM_CHECK_LAUNCH();
}

void launch_mux(const u8 *const *d_views, u8 *out, int N, float y_interval, float inv_y_interval, int ymod, int Hin,
                int Win, int Hout, int Wout, int elem_sz, int variant)
{
    ProfScope p("mux");
    STM_LAUNCH(stm_k_mux, dim3(cdiv(Wout, 256), Hout), dim3(256), 0, stream(), d_views, out, N, y_interval,
                       inv_y_interval, ymod, Hin, Win, Hout, Wout, elem_sz, variant);
    STM_CHECK_LAUNCH();
}

} // namespace stm
