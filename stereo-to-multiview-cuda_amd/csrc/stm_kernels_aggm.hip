// stm_kernels_aggm.hip -- cross-based cost aggregation of the frame pipeline on the gfx950 MATRIX pipe.
//
// Reference stages replaced, same arithmetic as stm_kernels_agg.hip:
//   ci_ad_kernel_5 / ci_census_kernel_6 / ci_adcensus_kernel   d_ci_ad.cu:73-159, d_ci_census.cu:197-254, d_ci_adcensus.cu:10-36
//   ca_cross_hsum_kernel_3        d_ca_cross_sum.cu:243-293   (horizontal window sum)
//   ca_cross_vhsum_kernel_2       d_ca_cross_sum.cu:148-198   (vertical window sum; the two transposes are deleted)
//   dc_wta_kernel                 d_dc_wta.cu:9-35
//
// Why MFMA instructions, when the path has no dense contraction: the reference sums every window element by element in
// float32, ascending (d_ca_cross_sum.cu:284-289), and WTA indices must be bit-exact, so the summation ORDER is fixed and
// a prefix-sum formulation is ruled out (hard part 1).  v_mfma_f32_16x16x1_4B_f32 computes, for each of
// 4 blocks, D[m][n] = A[m] * B[n] + C[m][n] with ONE product per output (K = 1): with A in {0, 1} that is exactly
// "acc = acc + b" (or "acc = acc", 0 * b = +0 for finite b), one float32 rounding per step -- the reference's chain, for 16
// pixels x 16 hypotheses x 4 blocks per instruction, fed by ONE LDS value per hypothesis instead of one per pixel.
// tools/mfma_probe*.hip verify on hardware: register layouts, bit-exactness of a 96-step masked chain, the CBSZ/ABID
// broadcast, and that f32 MFMAs occupy the SIMD's vector ALU (their time adds to the VALU time: 43 cycles per instruction in
// these dependent chains) -- what is bought is not a second pipe but 1024 exact adds per issue slot, 16x less LDS traffic
// than stm_k_agg_h / stm_k_agg_v and no lane idling behind a neighbour's longer window.
//
// Horizontal kernels: block b of the instruction = chunk of 16 hypotheses; A[m] = window mask of pixel m (CBSZ = 2: the A
// values of block ABID serve all four blocks, so ONE mask register holds the masks of four consecutive steps and ABID
// selects the step: 3 VALU instructions per 4 steps); B[n] = cost of hypothesis n (lane 16 b + n).
// Vertical kernel: block b = image column; A[m] = window mask of (row m, column b); B[n] = cost of hypothesis n at the
// current window row.  D: register 4 b + i of lane 16 q + n = out[pixel or row 4 q + i][block b][hypothesis n].
//
// Volume layout inside the frame pipeline ("PQ"): float4 [chunk = d / 16][y][g = x / 4][dd = d % 16], the float4 = the four
// pixels 4g..4g+3 of one hypothesis.  Horizontal pass: a lane reads one float4 = four consecutive window steps of its
// hypothesis; vertical pass: the four columns of a float4 are the four blocks (the LDS rings hold a row as float
// [4 columns][16 hypotheses], which lane 16 b + n reads linearly).  Every global access of a wave is 16 B per lane, 256 B
// contiguous per 16 lanes, 1 KB contiguous per wave in the horizontal kernels.
#include "stm_hwin.h"

namespace stm {

typedef float f4 __attribute__((ext_vector_type(4)));

#define STM_MFMA16(m, b, acc, abid) __builtin_amdgcn_mfma_f32_16x16x1f32(m, b, acc, 2, abid, 0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#define STM_MASKF(m) (__builtin_amdgcn_inverse_ballot_w64(m) ? 1.0f : 0.0f)
// 16-byte buffer load, streaming (nt).  The builtin's vector is converted as a whole: indexing its elements directly
// returned element 0 for every index with this compiler (ROCm 7.2).
#define STM_BLOAD(rsrc, voff) __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 2))


// rotate within each row of 16 lanes (DPP row_ror:n)
template <int N> __device__ __forceinline__ float row_ror_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
template <int N> __device__ __forceinline__ int row_ror_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, false); }
// wave-wide min / max as a scalar: butterfly inside the rows of 16 on the DPP path, the four rows through readlane
__device__ __forceinline__ int wave_min_i(int v)
{
    v = min(v, row_ror_i<8>(v)); v = min(v, row_ror_i<4>(v)); v = min(v, row_ror_i<2>(v)); v = min(v, row_ror_i<1>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v)
{
    v = max(v, row_ror_i<8>(v)); v = max(v, row_ror_i<4>(v)); v = max(v, row_ror_i<2>(v)); v = max(v, row_ror_i<1>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ f4 nt_load4(const f4 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void nt_store4(f4 *p, f4 v) { __builtin_nontemporal_store(v, p); }

static void allow_lds_m(const void *func, size_t bytes)
{
    if (bytes > 64 * 1024) STM_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// ------------------------------------------------------------------ initial costs into the PQ layout
// C(d, x) = rho_ad(|own(x) - other(x')|_1) + rho_c(ham(cen_own(x), cen_other(x'))), x' = clamp(x + sgn (d - zd)); sgn = +1 for
// the left view, -1 for the right view (SURVEY A-Q6).  Hypotheses d >= D and pixels x >= W of the padded layout are 0.
constexpr int PC_TX = 256;
__device__ __forceinline__ int hamdist_q1(uint32_t a, uint32_t b)
{
    const uint32_t x = a ^ b; // popc(x & 0x7fffffff) + 33 * (x >> 31) == popc(x) + 32 * (x >> 31)   (d_alu.cu:7-15, SURVEY A-Q1)
    return __popc(x) + (int)((x >> 26) & 32u);
}

__global__ __launch_bounds__(PC_TX) void stm_k_pq_cost(PQViews v, const float *__restrict__ lut_g, int D, int zd, int H, int W, int G,
                                                       int NC, int pad)
{
    extern __shared__ uint32_t sm[];
    const int view = blockIdx.z;
    const uint32_t *__restrict__ pk_own = view ? v.pk[1] : v.pk[0], *__restrict__ pk_oth = view ? v.pk[0] : v.pk[1];
    const uint32_t *__restrict__ cen_own = view ? v.cen[1] : v.cen[0], *__restrict__ cen_oth = view ? v.cen[0] : v.cen[1];
    f4 *__restrict__ out = (f4 *)(view ? v.a[1] : v.a[0]);
    const int sgn = view ? -1 : 1;
    const int span = PC_TX + 2 * pad;
    uint32_t *s_po = sm, *s_co = sm + PC_TX, *s_px = sm + 2 * PC_TX, *s_cx = sm + 2 * PC_TX + span;
    float *s_lut_ad = (float *)(sm + 2 * PC_TX + 2 * span), *s_lut_c = s_lut_ad + 768;
    const int y = blockIdx.y, x0 = blockIdx.x * PC_TX, tid = threadIdx.x;
    const size_t row = (size_t)y * W;
    {
        const int gx = min(x0 + tid, W - 1);
        s_po[tid] = pk_own[row + gx];
        s_co[tid] = cen_own[row + gx];
    }
    for (int i = tid; i < span; i += PC_TX) {
        const int gx = min(max(x0 + i - pad, 0), W - 1); // clamp-to-edge in image coordinates
        s_px[i] = pk_oth[row + gx];
        s_cx[i] = cen_oth[row + gx];
    }
    for (int i = tid; i < 768 + 65; i += PC_TX) s_lut_ad[i] = lut_g[i];
    __syncthreads();
    const int g0 = x0 >> 2;
    for (int c = 0; c < NC; ++c) {
        f4 *__restrict__ orow = out + (((size_t)c * H + y) * G) * 16;
        for (int idx = tid; idx < (PC_TX / 4) * 16; idx += PC_TX) {
            const int gi = idx >> 4, dd = idx & 15, d = c * 16 + dd;
            if (g0 + gi >= G) continue;
            f4 r = {0.f, 0.f, 0.f, 0.f};
            if (d < D) {
                const int o = sgn * (d - zd) + pad;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xl = gi * 4 + j;
                    if (x0 + xl < W) {
                        const int ad = (int)__builtin_amdgcn_sad_u8(s_po[xl], s_px[xl + o], 0u);
                        const int hd = hamdist_q1(s_co[xl], s_cx[xl + o]);
                        r[j] = s_lut_ad[ad] + s_lut_c[hd];
                    }
                }
            }
            orow[(size_t)(g0 + gi) * 16 + dd] = r;
        }
    }
}

// ------------------------------------------------------------------ horizontal pass (+ WTA)
// One block = one image row x a segment of 16 * NW pixels, all hypotheses (chunk sets of 64 in turn); one wave = 16 pixels
// (four pixel tiles of 4) x 64 hypotheses = 4 accumulation chains (one per chunk) that share the mask register.
// LDS: float4 tile[4 chunks][NG groups][16] | u32 sn[16 NW] (window start relative to the tile | length << 16).
// Lane l: pt = l / 16 (pixel tile), dq = (l / 4) % 4 (quad of the chunk = step slot of the mask), i = l % 4.
// COST: the first pass of the frame.  The tile is not read from a volume but computed in place from the two images (BGRX
// dwords + census words of this row, staged in LDS with clamped borders) and the two rho tables, as stm_k_pq_cost would have
// written it: the 2 V of initial costs are never written or read (8 V per frame with the fused vertical kernel, SURVEY 8d's
// plan), at the price of computing the 2 HG halo groups of every segment twice.
template <int NW, bool WTA, bool COST>
__global__ __launch_bounds__(64 * NW) void stm_k_pq_h(PQViews v, int D, int zd, int H, int W, int G, int NC, int HG, int nseg, int dbg,
                                                      const float *__restrict__ lut_g, int pad, int nvw)
{
    constexpr int NT = 64 * NW, TX = 16 * NW;
    extern __shared__ f4 lds4[];
    const int NG = 4 * NW + 2 * HG;
    f4 *tile = lds4;
    uint32_t *sn = (uint32_t *)(tile + 4 * NG * 16);
    const int tid = threadIdx.x;
    // block -> (segment, row, view).  Neighbouring segments of a row share 2 HG groups of input; workgroups are dealt to the 8
    // XCDs round-robin, so XCD x takes the x-th eighth of the (segment-fastest) list: neighbours run back to back on ONE
    // XCD and the shared halo is served by that XCD's L2 instead of being fetched from HBM twice.  Placement only affects speed.
    int view, y, X0seg;
    {
        const int per_xcd = (nseg * H * nvw + 7) >> 3;
        const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (logical >= nseg * H * nvw) return;
        X0seg = (logical % nseg) * TX;
        const int rest = logical / nseg;
        y = rest % H;
        view = rest / H;
    }
    const f4 *__restrict__ in = (const f4 *)(view ? v.a[1] : v.a[0]);
    f4 *__restrict__ out = (f4 *)(view ? v.b[1] : v.b[0]);
    const u8 *__restrict__ armL = view ? v.armL[1] : v.armL[0], *__restrict__ armR = view ? v.armR[1] : v.armR[0];
    float *__restrict__ disp = view ? v.disp[1] : v.disp[0];
    const size_t row = (size_t)y * W;
    const int gbase = (X0seg >> 2) - HG;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // COST staging behind the window table: (pixel, census) pairs of the own image (NG * 4) and of the other image (NG * 4 +
    // 2 pad, clamped to the row), rho tables (768 + 72)
    uint2 *s_own = (uint2 *)(sn + TX), *s_oth = s_own + NG * 4;
    float *s_lut_ad = (float *)(s_oth + NG * 4 + 2 * pad), *s_lut_c = s_lut_ad + 768;
    const int sgn = view ? -1 : 1;
    if (COST) {
        const uint32_t *__restrict__ pk_own = view ? v.pk[1] : v.pk[0], *__restrict__ pk_oth = view ? v.pk[0] : v.pk[1];
        const uint32_t *__restrict__ cen_own = view ? v.cen[1] : v.cen[0], *__restrict__ cen_oth = view ? v.cen[0] : v.cen[1];
        const int px0 = 4 * gbase; // image column of tile pixel 0
        for (int i = tid; i < NG * 4; i += NT) {
            const int gx = min(max(px0 + i, 0), W - 1);
            s_own[i] = make_uint2(pk_own[row + gx], cen_own[row + gx]);
        }
        for (int i = tid; i < NG * 4 + 2 * pad; i += NT) {
            const int gx = min(max(px0 + i - pad, 0), W - 1); // clamp-to-edge in image coordinates
            s_oth[i] = make_uint2(pk_oth[row + gx], cen_oth[row + gx]);
        }
        for (int i = tid; i < 768 + 65; i += NT) s_lut_ad[i] = lut_g[i];
    }

    if (tid < TX) {
        const int x = X0seg + tid;
        uint32_t e = (uint32_t)(x - 4 * gbase); // empty window
        if (x < W) {
            const int aL = armL[row + x], aR = armR[row + x];
            e = (uint32_t)(x - aL - 4 * gbase) | ((uint32_t)(aL + aR) << 16); // window [x - armL, x + armR), d_ca_cross_sum.cu:277-289
        }
        sn[tid] = e;
    }

    // Wave = 16 pixels x 4 chunks, ONE v_mfma_f32_16x16x1_4B_f32 per window step: block b = chunk, A[m] = mask of pixel m
    // (CBSZ = 2: the A values of block ABID serve all four blocks, so lanes 16a..16a+15 hold the masks of step a of the current
    // group), B[n] = cost of hypothesis n of the chunk (lane 16 b + n), D: register 4b + i of lane 16q + n = out[pixel 4q + i]
    // [chunk b][hypothesis n].  1024 adds per instruction at 8 issue cycles (the 4x4x1 form spends 32 on them), so the 3 mask
    // instructions and the LDS read of a group hide behind the matrix pipe.
    const int l = tid & 63, w = tid >> 6;
    const int lb = l >> 4, dd = l & 15; // B operand: chunk / D: pixel quad / A: step of the group;  hypothesis / pixel
    const int X0 = X0seg + 16 * w;
    float bc[4];
    int bd[4];
    if (WTA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { bc[i] = 3.402823466e+38f; bd[i] = 0; }
    }
    int G0r = 0, n_it = 0, t0 = 0, nn = 0;
    const int ncs = (NC + 3) >> 2;
    for (int cs = 0; cs < ncs; ++cs) {
        const int c0 = cs * 4;
        if (cs) __syncthreads(); // the previous chunk set's readers are done with the tile
        // tile fill: every load of a batch is issued before the first LDS write (a load-wait-write loop would expose the full
        // HBM latency once per element); addresses are clamped so that all loads are unconditional, zeros selected afterwards
        if (COST) {
            if (cs == 0) __syncthreads(); // staging complete
            // C(d, x) = rho_ad(|own(x) - other(x')|_1) + rho_c(ham(cen_own(x), cen_other(x'))), x' = clamp(x + sgn (d - zd))
            // (SURVEY A-Q6); hypotheses d >= D and pixels outside the row are 0
            // one thread = one (group, hypothesis-in-chunk) pair for all four chunks: the group's own pixels are read once
            for (int rr = tid; rr < NG * 16; rr += NT) {
                const int gi = rr >> 4, dd = rr & 15, x0g = 4 * (gbase + gi);
                const uint2 *own = s_own + gi * 4;
                const uint2 o0 = own[0], o1 = own[1], o2 = own[2], o3 = own[3];
                const bool in0 = x0g >= 0 && x0g < W, in1 = x0g + 1 >= 0 && x0g + 1 < W, in2 = x0g + 2 >= 0 && x0g + 2 < W,
                           in3 = x0g + 3 >= 0 && x0g + 3 < W;
#pragma unroll
                for (int cl = 0; cl < 4; ++cl) {
                    const int d = (c0 + cl) * 16 + dd;
                    f4 val = zero4;
                    if (c0 + cl < NC) { // uniform; d < 16 NC <= D + 15: inside the staged range (pad).  Branch-free per lane: the table
                                        // reads of a chunk are issued together, pixels outside the row / hypotheses >= D are zeroed by selects
                        const uint2 *oth = s_oth + gi * 4 + sgn * (d - zd) + pad;
                        const uint2 q0 = oth[0], q1 = oth[1], q2 = oth[2], q3 = oth[3];
                        const float a0 = s_lut_ad[__builtin_amdgcn_sad_u8(o0.x, q0.x, 0u)], e0 = s_lut_c[hamdist_q1(o0.y, q0.y)];
                        const float a1 = s_lut_ad[__builtin_amdgcn_sad_u8(o1.x, q1.x, 0u)], e1 = s_lut_c[hamdist_q1(o1.y, q1.y)];
                        const float a2 = s_lut_ad[__builtin_amdgcn_sad_u8(o2.x, q2.x, 0u)], e2 = s_lut_c[hamdist_q1(o2.y, q2.y)];
                        const float a3 = s_lut_ad[__builtin_amdgcn_sad_u8(o3.x, q3.x, 0u)], e3 = s_lut_c[hamdist_q1(o3.y, q3.y)];
                        const bool dok = d < D;
                        val.x = (in0 && dok) ? a0 + e0 : 0.f;
                        val.y = (in1 && dok) ? a1 + e1 : 0.f;
                        val.z = (in2 && dok) ? a2 + e2 : 0.f;
                        val.w = (in3 && dok) ? a3 + e3 : 0.f;
                    }
                    tile[cl * NG * 16 + rr] = val;
                }
            }
        } else {
            // LDS-DMA (global_load_lds_dwordx4): one wave instruction moves four consecutive groups of one chunk, 1 KB that is
            // contiguous in the volume's row and in the tile, without passing through registers.  Groups outside the row
            // and chunks past the last one are read from the nearest valid address instead of being zeroed: no window
            // reaches them (arms stop at the image border), so their masks are 0 and any finite value will do.
            const int NJ = NG >> 2; // NG is a multiple of 4 (the launcher rounds the halo)
            const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
            for (int idx = wu; idx < 4 * NJ; idx += NW) {
                const int cl = idx / NJ, j = idx - cl * NJ;
                const int g = min(max(gbase + 4 * j + ((tid & 63) >> 4), 0), G - 1);
                const f4 *src = in + ((size_t)min(c0 + cl, NC - 1) * H + y) * G * 16 + g * 16 + (tid & 15);
                __builtin_amdgcn_global_load_lds((const void *)src, (__attribute__((address_space(3))) void *)(tile + (cl * NG + 4 * j) * 16), 16, 0, 0);
            }
        }
        __syncthreads();
        if (X0 >= W) continue; // whole wave out of the image (uniform per wave); it still takes part in the barriers
        if (cs == 0) {
            const uint32_t e = sn[16 * w + dd]; // mask lanes: pixel dd of the wave
            const int srel = (int)(e & 0xffffu);
            nn = (int)(e >> 16);
            G0r = wave_min_i(nn ? (srel >> 2) : 0x7fffffff);
            const int Gend = wave_max_i(nn ? ((srel + nn + 3) >> 2) : -0x7fffffff);
            n_it = Gend - G0r; // <= 0 when every window of the wave is empty
            t0 = 4 * G0r + lb - srel;
        }
        f16v acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        {
            const f4 *p = tile + (lb * NG + G0r) * 16 + dd;
            int t = t0;
            for (int it = 0; it < n_it; ++it) {
                const f4 c4 = *p;
                p += 16;
                const float m = ((unsigned)t < (unsigned)nn) ? 1.0f : 0.0f;
                t += 4;
                acc = STM_MFMA16(m, c4.x, acc, 0);
                acc = STM_MFMA16(m, c4.y, acc, 1);
                acc = STM_MFMA16(m, c4.z, acc, 2);
                acc = STM_MFMA16(m, c4.w, acc, 3);
            }
        }
        // registers 4b..4b+3 of lane 16q + n = out[pixels X0 + 4q .. +3][hypothesis 16 (c0 + b) + n]
        if (!WTA) {
            if ((X0 >> 2) + lb < G) {
#pragma unroll
                for (int cl = 0; cl < 4; ++cl)
                    if (c0 + cl < NC) {
                        const f4 o = {acc[4 * cl], acc[4 * cl + 1], acc[4 * cl + 2], acc[4 * cl + 3]};
                        nt_store4(out + (((size_t)(c0 + cl) * H + y) * G + (X0 >> 2)) * 16 + l, o);
                    }
            }
        } else {
            // first strictly-lowest cost wins, ascending d (d_dc_wta.cu:19-34): per lane the chunks come in ascending d
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int d = (c0 + cl) * 16 + dd;
                if (d < D) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (bc[i] > acc[4 * cl + i]) { bc[i] = acc[4 * cl + i]; bd[i] = d; }
                }
            }
        }
    }
    if (WTA && X0 < W) {
        // across the 16 lanes (hypotheses n) of a pixel quad: lowest cost, ties to the lowest d
        float res[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float m = bc[i];
            m = fminf(m, row_ror_f<8>(m));
            m = fminf(m, row_ror_f<4>(m));
            m = fminf(m, row_ror_f<2>(m));
            m = fminf(m, row_ror_f<1>(m));
            int cand = (bc[i] == m) ? bd[i] : 0x7fffffff;
            cand = min(cand, row_ror_i<8>(cand));
            cand = min(cand, row_ror_i<4>(cand));
            cand = min(cand, row_ror_i<2>(cand));
            cand = min(cand, row_ror_i<1>(cand));
            res[i] = (float)cand - (float)zd;
        }
        if (dd == 0) {
            const int x = X0 + 4 * lb;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x + i < W) disp[row + x + i] = res[i];
        }
    }
}

// ------------------------------------------------------------------ horizontal pass over a volume, streaming (round 3)
// stm_k_pq_h<.., COST = false> spends most of a block's life waiting: fill the tile (HBM latency), barrier, sweep, exit; three
// blocks per CU overlap only by chance (round-3 counters: matrix pipe busy 43 % of the time, HBM at 3 TB/s; taking the
// fill's vector instructions away with LDS-DMA changed nothing).  Here a block owns a whole image row (or a part of one)
// and walks its segments left to right over a tile that is a RING of groups: a segment's tile shares its 2 HG halo groups
// with the next one, so only the 4 NW new groups are fetched per segment (every byte of the volume crosses the CU once), and
// they are already on their way from HBM into REGISTERS (KP float4 per lane) while the waves sweep the current segment; they
// are written to LDS between the two barriers that end the segment.  Loads and stores are buffer instructions with
// statically known counts, so the wait for the staged groups leaves the segment's stores in flight.  D <= 64 (one chunk
// set); masks, MFMA chains and WTA as in stm_k_pq_h.
// Ring: shifted group index a = group + HG >= 0; piece = 4 groups = 1 KB per chunk; piece a / 4 lives in slot (a / 4) % NJ,
// NJ = NG / 4.  Segment s uses pieces [8 s', 8 s' + NJ), s' = s NW / 8... (4 NW groups = NW pieces per segment).
#define STM_BSTORE(rsrc, voff, val) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, val), rsrc, voff, 0, 2)
template <int NW, bool WTA>
__global__ __launch_bounds__(64 * NW, (3 * NW) / 4) void stm_k_pq_hs(PQViews v, int D, int zd, int H, int W, int G, int NC, int HG, int nseg, int spl, int dbg)
{
    constexpr int TX = 16 * NW, KP = 4; // pixels per segment; new pieces per wave and segment (4 chunks x NW pieces / NW waves)
    extern __shared__ f4 lds4[];
    const int NG = 4 * NW + 2 * HG, NJ = NG >> 2;
    f4 *tile = lds4; // [4 chunks][NJ slots][4 groups][16 hypotheses]
    uint32_t *sn = (uint32_t *)(tile + 4 * NG * 16);
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lb = l >> 4, dd = l & 15;
    // block -> (view, row, part of the row)
    const int part = blockIdx.x % spl, rest = blockIdx.x / spl, y = rest % H, view = rest / H;
    const int seg_per = (nseg + spl - 1) / spl, seg0 = part * seg_per, seg1 = min(nseg, seg0 + seg_per);
    if (seg0 >= seg1) return;
    const f4 *__restrict__ in = (const f4 *)(view ? v.a[1] : v.a[0]);
    f4 *__restrict__ out = (f4 *)(view ? v.b[1] : v.b[0]);
    const u8 *__restrict__ armL = view ? v.armL[1] : v.armL[0], *__restrict__ armR = view ? v.armR[1] : v.armR[0];
    float *__restrict__ disp = view ? v.disp[1] : v.disp[0];
    const size_t row = (size_t)y * W;
    const uint32_t rowbytes = (uint32_t)G * 256u;
    // D > 64: the row is walked once per chunk SET of 64 hypotheses; the WTA pass keeps each pixel's best (cost bits, d) of the
    // sets before in `wb` (written and read back by the same lane, so no barrier is involved)
    const int ncs = (NC + 3) >> 2;
    uint2 *wb = (uint2 *)(sn + TX); // [pixels of this block's part of the row], only when WTA && ncs > 1
    f4 st[KP];
    int aLn = 0, aRn = 0;
  for (int cs = 0; cs < ncs; ++cs) {
    const int c0 = 4 * cs;
    // one row of each chunk; groups outside the row are out of range and read as zeros, chunks past the last are empty buffers
    __amdgpu_buffer_rsrc_t rin[4], rout[4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
        rin[cl] = __builtin_amdgcn_make_buffer_rsrc((void *)(in + ((size_t)min(c0 + cl, NC - 1) * H + y) * G * 16), 0, c0 + cl < NC ? rowbytes : 0u, 0x00020000);
        rout[cl] = __builtin_amdgcn_make_buffer_rsrc((void *)(out + ((size_t)min(c0 + cl, NC - 1) * H + y) * G * 16), 0, c0 + cl < NC ? rowbytes : 0u, 0x00020000);
    }
    // piece (chunk cl, shifted piece index pa): groups 4 pa - HG .. + 3 of the row
#define STM_HS_LOAD(K, CL, PA) st[K] = STM_DBG(dbg, 4) ? f4{0.f, 0.f, 0.f, 0.f} : __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin[CL], (4 * (PA) - HG) * 256 + l * 16, 0, 0));
#define STM_HS_ARMS(S)                                                          \
    if (tid < TX) {                                                             \
        const int x = min((S) * TX + tid, W - 1);                               \
        aLn = armL[row + x];                                                    \
        aRn = armR[row + x];                                                    \
    }
#define STM_HS_SN(S)                                                                                                       \
    if (tid < TX) {                                                                                                        \
        const int x = (S) * TX + tid, org = (S) * TX - 4 * HG;                                                             \
        /* window [x - armL, x + armR) relative to the segment's first tile pixel (d_ca_cross_sum.cu:277-289); past the row: empty */ \
        const int srel_ = x < W ? x - aLn - org : x - org, nn_ = x < W ? aLn + aRn : 0;                                    \
        /* the sweep of the wave that owns these 16 pixels (a DPP row): first group, number of groups -- computed once here */ \
        /* instead of by every wave, and carried in the spare bits of the row's first two entries (a separate array would  */ \
        /* cost the third block per CU: LDS is granted in 1280-byte granules and the tile + table fill 42 of them exactly) */ \
        int lo_ = nn_ ? (srel_ >> 2) : 0x7fffffff, hi_ = nn_ ? ((srel_ + nn_ + 3) >> 2) : -0x7fffffff;                     \
        lo_ = min(lo_, row_ror_i<8>(lo_)); lo_ = min(lo_, row_ror_i<4>(lo_)); lo_ = min(lo_, row_ror_i<2>(lo_)); lo_ = min(lo_, row_ror_i<1>(lo_)); \
        hi_ = max(hi_, row_ror_i<8>(hi_)); hi_ = max(hi_, row_ror_i<4>(hi_)); hi_ = max(hi_, row_ror_i<2>(hi_)); hi_ = max(hi_, row_ror_i<1>(hi_)); \
        const int nit_ = max(hi_ - lo_, 0);                                                                                \
        const int ex_ = (tid & 15) == 0 ? (nit_ ? lo_ : 0) : (tid & 15) == 1 ? nit_ : 0;                                   \
        sn[tid] = (uint32_t)srel_ | ((uint32_t)nn_ << 12) | ((uint32_t)ex_ << 22); /* 12 + 10 + 10 bits */                 \
    }
    // first segment: the whole tile, pieces NW seg0 .. NW seg0 + NJ - 1 of every chunk, KP per wave and round
    {
        const int np = 4 * NJ;
        for (int i0 = 0; i0 < np; i0 += NW * KP) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int idx = min(i0 + w + NW * k, np - 1), cl = idx / NJ, j = idx - cl * NJ; // uniform
                switch (cl) { case 0: STM_HS_LOAD(k, 0, NW * seg0 + j) break; case 1: STM_HS_LOAD(k, 1, NW * seg0 + j) break;
                              case 2: STM_HS_LOAD(k, 2, NW * seg0 + j) break; default: STM_HS_LOAD(k, 3, NW * seg0 + j) break; }
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int idx = i0 + w + NW * k, cl = idx / NJ, j = idx - cl * NJ;
                if (idx < np) tile[(cl * NJ + (NW * seg0 + j) % NJ) * 64 + l] = st[k];
            }
        }
        STM_HS_ARMS(seg0)
        STM_HS_SN(seg0)
    }
    __syncthreads();
    int slot0 = (NW * seg0) % NJ; // ring slot of the segment's first piece
    for (int sq = seg0; sq < seg1; ++sq) {
        const int X0 = sq * TX + 16 * w;
        const bool more = sq + 1 < seg1;
        // the NW new pieces per chunk of the next segment: piece index NW (sq + 1) + NJ - NW + jn; wave w takes (chunk k, jn = w)
        if (more) {
            const int pa = NW * (sq + 1) + NJ - NW + w;
            STM_HS_LOAD(0, 0, pa) STM_HS_LOAD(1, 1, pa) STM_HS_LOAD(2, 2, pa) STM_HS_LOAD(3, 3, pa)
            STM_HS_ARMS(sq + 1)
        }
        if (X0 < W) { // uniform per wave
            const uint32_t e = sn[16 * w + dd]; // mask lanes: pixel dd of the wave
            const int srel = (int)(e & 0xfffu), nn = (int)((e >> 12) & 0x3ffu);
            // first group of the wave's sweep, its length in groups: the spare bits of the wave's first two entries (STM_HS_SN)
            const int G0r = (int)((uint32_t)__builtin_amdgcn_readlane((int)e, 0) >> 22);
            const int n_it = STM_DBG(dbg, 1) ? 0 : STM_DBG(dbg, 2) ? 11 : (int)((uint32_t)__builtin_amdgcn_readlane((int)e, 1) >> 22); // 0 when every window of the wave is empty
            f16v acc;
            if (n_it > 0) {
                int gs = 4 * slot0 + G0r; // ring position (in groups) of the first group of the sweep
                if (gs >= NG) gs -= NG;
                const f4 *p = tile + (lb * NG + gs) * 16 + dd;
                int t = 4 * G0r + lb - srel;
                // two register sets: the LDS read of a group is issued before the MFMAs of the group in front of it
#define STM_HS_NEXT(C)                                        \
    {                                                         \
        C = *p;                                               \
        p += 16;                                              \
        if (++gs == NG) { gs = 0; p -= NG * 16; } /* uniform */ \
    }
#define STM_HS_MFMA(C)                                                          \
    {                                                                           \
        const float m = ((unsigned)t < (unsigned)nn) ? 1.0f : 0.0f;             \
        t += 4;                                                                 \
        acc = STM_MFMA16(m, C.x, acc, 0);                                       \
        acc = STM_MFMA16(m, C.y, acc, 1);                                       \
        acc = STM_MFMA16(m, C.z, acc, 2);                                       \
        acc = STM_MFMA16(m, C.w, acc, 3);                                       \
    }
                f4 ca, cb;
                STM_HS_NEXT(ca)
                STM_HS_NEXT(cb) // a sweep of one group: one group past it (any ring slot, unused)
                {   // the first MFMA takes the constant 0 as its accumulator input: no register is cleared
                    const float m = ((unsigned)t < (unsigned)nn) ? 1.0f : 0.0f;
                    t += 4;
                    const f16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc = STM_MFMA16(m, ca.x, z, 0);
                    acc = STM_MFMA16(m, ca.y, acc, 1);
                    acc = STM_MFMA16(m, ca.z, acc, 2);
                    acc = STM_MFMA16(m, ca.w, acc, 3);
                }
                int it = 1;
                for (; it + 2 <= n_it; it += 2) {
                    STM_HS_NEXT(ca)
                    STM_HS_MFMA(cb)
                    STM_HS_NEXT(cb) // on the last trip one group past the sweep: any ring slot, unused
                    STM_HS_MFMA(ca)
                }
                if (it < n_it) STM_HS_MFMA(cb)
#undef STM_HS_NEXT
#undef STM_HS_MFMA
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
            // registers 4b..4b+3 of lane 16q + n = out[pixels X0 + 4q .. +3][hypothesis 16 b + n]
            if (!WTA) {
#pragma unroll
                for (int cl = 0; cl < 4; ++cl) { // groups past the row, chunks past the last: out of range, dropped
                    const f4 o = {acc[4 * cl], acc[4 * cl + 1], acc[4 * cl + 2], acc[4 * cl + 3]};
                    STM_BSTORE(rout[cl], (X0 >> 2) * 256 + l * 16, o);
                }
            } else {
                // first strictly-lowest cost wins, ascending d (d_dc_wta.cu:19-34): per lane the chunks come in ascending d
                float bc[4];
                int bd[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { bc[i] = 3.402823466e+38f; bd[i] = 0; }
#pragma unroll
                for (int cl = 0; cl < 4; ++cl) {
                    const int d = (c0 + cl) * 16 + dd;
                    if (d < D) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (bc[i] > acc[4 * cl + i]) { bc[i] = acc[4 * cl + i]; bd[i] = d; }
                    }
                }
                // across the 16 lanes (hypotheses n) of a pixel quad: lowest cost, ties to the lowest d
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(disp + row), 0, (uint32_t)W * 4u, 0x00020000);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // aggregated costs are sums of non-negative terms (rho >= 0, masked steps add +0): never negative, never NaN,
                    // so the order of the floats is the order of their bit patterns and the minimum is four v_min_i32 with a DPP operand
                    const int bi = __builtin_bit_cast(int, bc[i]);
                    int m = bi;
                    m = min(m, row_ror_i<8>(m));
                    m = min(m, row_ror_i<4>(m));
                    m = min(m, row_ror_i<2>(m));
                    m = min(m, row_ror_i<1>(m));
                    int cand = (bi == m) ? bd[i] : 0x7fffffff;
                    cand = min(cand, row_ror_i<8>(cand));
                    cand = min(cand, row_ror_i<4>(cand));
                    cand = min(cand, row_ror_i<2>(cand));
                    cand = min(cand, row_ror_i<1>(cand));
                    if (ncs > 1) { // uniform: combine with the chunk sets before (a later set wins only when strictly lower: ties to the lowest d)
                        uint2 *slot = wb + (X0 - seg0 * TX + 4 * lb + i);
                        if (dd == 0) {
                            if (cs > 0) {
                                const uint2 pv = *slot;
                                if (!(m < (int)pv.x)) { m = (int)pv.x; cand = (int)pv.y; }
                            }
                            if (cs + 1 < ncs) *slot = make_uint2((uint32_t)m, (uint32_t)cand);
                        }
                        if (cs + 1 < ncs) continue; // uniform: the last set stores the disparity
                    }
                    // lane dd == 0 of each pixel quad stores; the other lanes and pixels past the row are out of range
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)cand - (float)zd), rs,
                                                          dd == 0 ? (X0 + 4 * lb + i) * 4 : (int)0x7ffffff0, 0, 0);
                }
            }
        }
        __syncthreads(); // every wave is done reading the tile
        if (more) {
            // the new pieces replace the NW oldest: slots slot0 .. slot0 + NW - 1 (mod NJ); wave w writes slot slot0 + w of every chunk
            int sl = slot0 + w;
            if (sl >= NJ) sl -= NJ;
#pragma unroll
            for (int k = 0; k < KP; ++k) tile[(k * NJ + sl) * 64 + l] = st[k];
            STM_HS_SN(sq + 1)
            slot0 += NW;
            if (slot0 >= NJ) slot0 -= NJ;
        }
        __syncthreads();
    }
  } // chunk sets
#undef STM_HS_LOAD
#undef STM_HS_ARMS
#undef STM_HS_SN
}

// ------------------------------------------------------------------ first horizontal pass with the costs computed in place, streaming
// The row walk of stm_k_pq_hs for the cost-computing pass (stm_k_pq_h<.., COST = true> computes the 2 HG halo groups of every
// segment twice and fills its tile in two uneven rounds): the tile is the same ring of groups, and per segment only the 4 NW
// NEW groups are computed -- exactly one float4 per lane and chunk, wave w owning new piece w.  The pixels those costs need
// (own image: the new pixels; other image: the new pixels +- pad) are staged in two alternating LDS buffers, fetched two
// segments ahead; the costs of segment s + 1 are computed into registers after the sweep of segment s and written to the ring
// between the two barriers that end the segment.  D <= 64 (one chunk set).
// C(d, x) = rho_ad(|own(x) - other(x')|_1) + rho_c(ham(cen_own(x), cen_other(x'))), x' = clamp(x + sgn (d - zd)) (SURVEY A-Q6);
// hypotheses d >= D and pixels outside the row are 0.
template <int NW>
__global__ __launch_bounds__(64 * NW, (2 * NW) / 4) void stm_k_pq_hc(PQViews v, int D, int zd, int H, int W, int G, int NC, int HG, int nseg,
                                                                     int spl, const float *__restrict__ lut_g, int pad, int dbg)
{
    constexpr int NT = 64 * NW, TX = 16 * NW, NEWPX = 16 * NW; // threads; pixels per segment = new pixels per segment
    extern __shared__ f4 lds4[];
    const int NG = 4 * NW + 2 * HG, NJ = NG >> 2;
    f4 *tile = lds4; // [4 chunks][NJ slots][4 groups][16 hypotheses]
    uint32_t *sn = (uint32_t *)(tile + 4 * NG * 16);
    int2 *sg = (int2 *)(sn + TX); // per wave tile: first group and length of its sweep
    const int SO = NEWPX + 2 * pad; // other-image pixels staged per segment
    uint2 *s_own = (uint2 *)(sg + NW), *s_oth = s_own + 2 * NEWPX; // two buffers each: [2][NEWPX], [2][SO]
    float *s_lut_ad = (float *)(s_oth + 2 * SO), *s_lut_c = s_lut_ad + 768;
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lb = l >> 4, dd = l & 15;
    const int part = blockIdx.x % spl, rest = blockIdx.x / spl, y = rest % H, view = rest / H;
    const int seg_per = (nseg + spl - 1) / spl, seg0 = part * seg_per, seg1 = min(nseg, seg0 + seg_per);
    if (seg0 >= seg1) return;
    f4 *__restrict__ out = (f4 *)(view ? v.b[1] : v.b[0]);
    const u8 *__restrict__ armL = view ? v.armL[1] : v.armL[0], *__restrict__ armR = view ? v.armR[1] : v.armR[0];
    const uint32_t *__restrict__ pk_own = view ? v.pk[1] : v.pk[0], *__restrict__ pk_oth = view ? v.pk[0] : v.pk[1];
    const uint32_t *__restrict__ cen_own = view ? v.cen[1] : v.cen[0], *__restrict__ cen_oth = view ? v.cen[0] : v.cen[1];
    const int sgn = view ? -1 : 1;
    const size_t row = (size_t)y * W;
    const uint32_t rowbytes = (uint32_t)G * 256u;
    for (int i = tid; i < 768 + 65; i += NT) s_lut_ad[i] = lut_g[i];
    uint2 so = make_uint2(0, 0), sx = make_uint2(0, 0);
    int aLn = 0, aRn = 0;
    f4 cst[4];
  const int ncs = (NC + 3) >> 2; // D > 64: the row is walked once per chunk set of 64 hypotheses
  for (int cs = 0; cs < ncs; ++cs) {
    const int c0 = 4 * cs;
    __amdgpu_buffer_rsrc_t rout[4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
        rout[cl] = __builtin_amdgcn_make_buffer_rsrc((void *)(out + ((size_t)min(c0 + cl, NC - 1) * H + y) * G * 16), 0, c0 + cl < NC ? rowbytes : 0u, 0x00020000);
    // pixels for the costs of the groups that start at shifted group A0 (group = A0 - HG): own image NEWPX pixels from
    // x0 = 4 (A0 - HG), other image the same +- pad, both clamped to the row (clamp-to-edge, d_ci_ad.cu:102)
#define STM_HC_FETCH(A0)                                                                           \
    {                                                                                              \
        const int x0_ = 4 * ((A0) - HG);                                                           \
        if (tid < NEWPX) {                                                                         \
            const int gx = min(max(x0_ + tid, 0), W - 1);                                          \
            so = make_uint2(pk_own[row + gx], cen_own[row + gx]);                                  \
        }                                                                                          \
        if (tid < SO) {                                                                            \
            const int gx = min(max(x0_ - pad + tid, 0), W - 1);                                    \
            sx = make_uint2(pk_oth[row + gx], cen_oth[row + gx]);                                  \
        }                                                                                          \
    }
#define STM_HC_STAGE(B)                                            \
    {                                                              \
        if (tid < NEWPX) s_own[(B) * NEWPX + tid] = so;            \
        if (tid < SO) s_oth[(B) * SO + tid] = sx;                  \
    }
    // costs of new piece w (groups 4 w .. 4 w + 3 from shifted group A0), lane = (group, hypothesis-in-chunk), four chunks
#define STM_HC_COSTS(B, A0)                                                                                               \
    {                                                                                                                     \
        const int gi = 4 * w + lb, x0g = 4 * ((A0) - HG + gi);                                                            \
        const uint2 *own = s_own + (B) * NEWPX + gi * 4;                                                                  \
        const uint2 o0 = own[0], o1 = own[1], o2 = own[2], o3 = own[3];                                                   \
        const bool in0 = x0g >= 0 && x0g < W, in1 = x0g + 1 >= 0 && x0g + 1 < W, in2 = x0g + 2 >= 0 && x0g + 2 < W,       \
                   in3 = x0g + 3 >= 0 && x0g + 3 < W;                                                                     \
        /* branch-free per lane: every staged pixel and table entry exists (the staging clamps to the row), so all table */ \
        /* reads of a chunk are issued together and pixels outside the row / hypotheses >= D are zeroed by selects       */ \
        _Pragma("unroll") for (int cl = 0; cl < 4; ++cl) {                                                                \
            f4 val = {0.f, 0.f, 0.f, 0.f};                                                                                \
            if (c0 + cl < NC) { /* uniform */                                                                             \
                const int d = (c0 + cl) * 16 + dd; /* < 16 NC <= D + 15: inside the staged range (pad) */                  \
                const uint2 *oth = s_oth + (B) * SO + gi * 4 + sgn * (d - zd) + pad;                                      \
                const uint2 q0 = oth[0], q1 = oth[1], q2 = oth[2], q3 = oth[3];                                           \
                const float a0 = s_lut_ad[__builtin_amdgcn_sad_u8(o0.x, q0.x, 0u)], c0 = s_lut_c[hamdist_q1(o0.y, q0.y)]; \
                const float a1 = s_lut_ad[__builtin_amdgcn_sad_u8(o1.x, q1.x, 0u)], c1 = s_lut_c[hamdist_q1(o1.y, q1.y)]; \
                const float a2 = s_lut_ad[__builtin_amdgcn_sad_u8(o2.x, q2.x, 0u)], c2 = s_lut_c[hamdist_q1(o2.y, q2.y)]; \
                const float a3 = s_lut_ad[__builtin_amdgcn_sad_u8(o3.x, q3.x, 0u)], c3 = s_lut_c[hamdist_q1(o3.y, q3.y)]; \
                const bool dok = d < D;                                                                                   \
                val.x = (in0 && dok) ? a0 + c0 : 0.f;                                                                     \
                val.y = (in1 && dok) ? a1 + c1 : 0.f;                                                                     \
                val.z = (in2 && dok) ? a2 + c2 : 0.f;                                                                     \
                val.w = (in3 && dok) ? a3 + c3 : 0.f;                                                                     \
            }                                                                                                             \
            cst[cl] = val;                                                                                                \
        }                                                                                                                 \
    }
#define STM_HC_ARMS(S)                                                          \
    if (tid < TX) {                                                             \
        const int x = min((S) * TX + tid, W - 1);                               \
        aLn = armL[row + x];                                                    \
        aRn = armR[row + x];                                                    \
    }
#define STM_HC_SN(S)                                                                                                       \
    if (tid < TX) {                                                                                                        \
        const int x = (S) * TX + tid, org = (S) * TX - 4 * HG;                                                             \
        /* window [x - armL, x + armR) relative to the segment's first tile pixel (d_ca_cross_sum.cu:277-289); past the row: empty */ \
        const int srel_ = x < W ? x - aLn - org : x - org, nn_ = x < W ? aLn + aRn : 0;                                    \
        sn[tid] = (uint32_t)srel_ | ((uint32_t)nn_ << 16);                                                                 \
        /* the sweep of the wave that owns these 16 pixels (a DPP row): first group, number of groups */                   \
        int lo_ = nn_ ? (srel_ >> 2) : 0x7fffffff, hi_ = nn_ ? ((srel_ + nn_ + 3) >> 2) : -0x7fffffff;                     \
        lo_ = min(lo_, row_ror_i<8>(lo_)); lo_ = min(lo_, row_ror_i<4>(lo_)); lo_ = min(lo_, row_ror_i<2>(lo_)); lo_ = min(lo_, row_ror_i<1>(lo_)); \
        hi_ = max(hi_, row_ror_i<8>(hi_)); hi_ = max(hi_, row_ror_i<4>(hi_)); hi_ = max(hi_, row_ror_i<2>(hi_)); hi_ = max(hi_, row_ror_i<1>(hi_)); \
        if ((tid & 15) == 0) sg[tid >> 4] = make_int2(lo_, hi_ - lo_);                                                     \
    }
    // first segment: the whole tile, NW pieces per round (shifted groups 4 NW seg0 + 4 NW r ..)
    for (int r0 = 0; r0 < NJ; r0 += NW) {
        const int a0 = 4 * NW * seg0 + 4 * r0;
        STM_HC_FETCH(a0)
        __syncthreads(); // the previous round's readers are done with buffer 0 (first round: the tables are being written)
        STM_HC_STAGE(0)
        __syncthreads();
        if (r0 + w < NJ) {
            STM_HC_COSTS(0, a0)
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[(k * NJ + (NW * seg0 + r0 + w) % NJ) * 64 + l] = cst[k];
        }
    }
    __syncthreads();
    // staging of segment seg0 + 1's new groups into buffer (seg0 + 1) & 1, arms of seg0
    {
        const int a1 = 4 * NW * (seg0 + 1) + NG - 4 * NW;
        STM_HC_FETCH(a1)
        STM_HC_STAGE((seg0 + 1) & 1)
        STM_HC_ARMS(seg0)
        STM_HC_SN(seg0)
    }
    __syncthreads();
    int slot0 = (NW * seg0) % NJ; // ring slot of the segment's first piece
    for (int sq = seg0; sq < seg1; ++sq) {
        const int X0 = sq * TX + 16 * w;
        const bool more = sq + 1 < seg1;
        // pixels for the new groups of segment sq + 2 (consumed after the next segment's sweep), arms of segment sq + 1
        if (more) {
            STM_HC_FETCH(4 * NW * (sq + 2) + NG - 4 * NW)
            STM_HC_ARMS(sq + 1)
        }
        if (X0 < W) { // uniform per wave
            const uint32_t e = sn[16 * w + dd]; // mask lanes: pixel dd of the wave
            const int srel = (int)(e & 0xffffu), nn = (int)(e >> 16);
            const int2 sgw = sg[w]; // first group of the wave's sweep, its length in groups (STM_HS_SN)
            const int G0r = __builtin_amdgcn_readfirstlane(sgw.x);
            const int n_it = STM_DBG(dbg, 1) ? 0 : __builtin_amdgcn_readfirstlane(sgw.y); // <= 0 when every window of the wave is empty
            f16v acc;
            if (n_it > 0) {
                int gs = 4 * slot0 + G0r; // ring position (in groups) of the first group of the sweep
                if (gs >= NG) gs -= NG;
                const f4 *p = tile + (lb * NG + gs) * 16 + dd;
                int t = 4 * G0r + lb - srel;
                // two register sets: the LDS read of a group is issued before the MFMAs of the group in front of it
#define STM_HS_NEXT(C)                                        \
    {                                                         \
        C = *p;                                               \
        p += 16;                                              \
        if (++gs == NG) { gs = 0; p -= NG * 16; } /* uniform */ \
    }
#define STM_HS_MFMA(C)                                                          \
    {                                                                           \
        const float m = ((unsigned)t < (unsigned)nn) ? 1.0f : 0.0f;             \
        t += 4;                                                                 \
        acc = STM_MFMA16(m, C.x, acc, 0);                                       \
        acc = STM_MFMA16(m, C.y, acc, 1);                                       \
        acc = STM_MFMA16(m, C.z, acc, 2);                                       \
        acc = STM_MFMA16(m, C.w, acc, 3);                                       \
    }
                f4 ca, cb;
                STM_HS_NEXT(ca)
                STM_HS_NEXT(cb) // a sweep of one group: one group past it (any ring slot, unused)
                {   // the first MFMA takes the constant 0 as its accumulator input: no register is cleared
                    const float m = ((unsigned)t < (unsigned)nn) ? 1.0f : 0.0f;
                    t += 4;
                    const f16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc = STM_MFMA16(m, ca.x, z, 0);
                    acc = STM_MFMA16(m, ca.y, acc, 1);
                    acc = STM_MFMA16(m, ca.z, acc, 2);
                    acc = STM_MFMA16(m, ca.w, acc, 3);
                }
                int it = 1;
                for (; it + 2 <= n_it; it += 2) {
                    STM_HS_NEXT(ca)
                    STM_HS_MFMA(cb)
                    STM_HS_NEXT(cb) // on the last trip one group past the sweep: any ring slot, unused
                    STM_HS_MFMA(ca)
                }
                if (it < n_it) STM_HS_MFMA(cb)
#undef STM_HS_NEXT
#undef STM_HS_MFMA
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
            // registers 4b..4b+3 of lane 16q + n = out[pixels X0 + 4q .. +3][hypothesis 16 b + n]
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) { // groups past the row, chunks past the last: out of range, dropped
                const f4 o = {acc[4 * cl], acc[4 * cl + 1], acc[4 * cl + 2], acc[4 * cl + 3]};
                if (!STM_DBG(dbg, 4)) STM_BSTORE(rout[cl], (X0 >> 2) * 256 + l * 16, o);
            }
        }
        if (more) STM_HC_COSTS((sq + 1) & 1, 4 * NW * (sq + 1) + NG - 4 * NW)
        __syncthreads(); // every wave is done reading the tile and the staging buffers
        if (more) {
            int sl = slot0 + w; // the new pieces replace the NW oldest
            if (sl >= NJ) sl -= NJ;
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[(k * NJ + sl) * 64 + l] = cst[k];
            STM_HC_STAGE(sq & 1) // = (sq + 2) & 1
            STM_HC_SN(sq + 1)
            slot0 += NW;
            if (slot0 >= NJ) slot0 -= NJ;
        }
        __syncthreads();
    }
  } // chunk sets
#undef STM_HC_FETCH
#undef STM_HC_STAGE
#undef STM_HC_COSTS
#undef STM_HC_ARMS
#undef STM_HC_SN
}

// ------------------------------------------------------------------ both vertical passes, fused, table-driven
// One block = one strip of 4 columns (one group) x one chunk of 16 hypotheses, 2 NTP waves: waves 0..NTP-1 run the first
// vertical pass, 16 output rows each per step, from LDS ring 1 (rows of the input volume) into LDS ring 2; waves NTP..2NTP-1
// run the second pass LAG steps behind, from ring 2 to HBM.  The intermediate volume never leaves the CU (8 V per frame
// instead of 12 V).  Two barriers per step of 16 NTP rows; the two passes run in DIFFERENT code paths: the first-pass waves do
// every global LOAD of the block and no store, the second-pass waves every STORE and no load (s_waitcnt counts loads and
// stores together, in issue order).  Wave tile = 16 rows x 4 columns x 16 hypotheses = ONE v_mfma_f32_16x16x1_4B_f32 per
// window row: block b = column, A[m] = mask of pixel (row m, column b) (lane 16 b + m), B[n] = cost of hypothesis n at
// (window row, column b) (lane 16 b + n), D: register 4b + i of lane 16q + n = out[row 4q + i][column b][hypothesis n].
// Window of pixel (y, x): rows [y - armU, y + armD).
// Round 3 took everything that is not the window sweep out of the vector ALU (round 2's kernel computed masks, window
// bounds and addresses there: 8.1 VALU instructions per MFMA, profiles/r03_pmc_sq_aggm_round2_build.txt) -- on gfx950
// the f32 MFMAs run on the SIMD's vector ALU, so every other VALU instruction is time the sweep does not get (round-2
// now 2.5 per MFMA, profiles/r03_pmc_sq_aggm.txt):
//  * the window masks of a tile (16 rows x 4 columns) are the same for the 4 chunk blocks of a strip and for both passes, so
//    stm_k_vwin_table ballots them ONCE per frame into 64-bit lane masks: the sweep fetches four of them with one
//    s_load_dwordx8 and forms the A operand with ONE v_cndmask_b32 per MFMA (was add + compare + select per column); the
//    same record carries the sweep's first row and length (were four DPP min/max reductions per tile);
//  * rings hold QUADS of rows, float4 [quad slot][column b][hypothesis n] = rows 4q..4q+3: the B operands of four window rows
//    are one ds_read_b128 (were four ds_read_b32), a first-pass tile leaves as four ds_write_b128 straight from its
//    accumulator registers (were sixteen ds_write_b32), sweeps start on a quad boundary as before;
//  * global loads and stores are buffer instructions whose row offset is one v_add of a scalar: rows past the image are
//    dropped by the buffer's range check, no 64-bit address arithmetic, no clamps.
// Table record of tile (view, u = y / 16, g), `rec` dwords: [0] K0 (first row of the sweep, multiple of 4), [1] n_it
// (sweep length in quads), [8 + 8 it ..] the four 64-bit masks of quad it: bit 16 b + i = row 16 u + i of column 4 g + b has
// row K0 + 4 it + j in its window [y - armU, y + armD)  (d_ca_cross_sum.cu:172-173,189-194).
// top >= 0: the STATIC layout of stm_k_pq_v12r (stm_kernels_aggv.hip) -- [0] q0 = (K0 - (16 u - top)) / 4, the sweep's first quad
// inside the tile's range [16 u - top, ..), [1] n_it, and the masks of range quad J at [8 + 8 J ..] (only the sweep's own quads
// are written; nothing reads the others' contents for a result).
// Round 4: the masks of a tile are no longer balloted step by step (one compare and a 64-bit select per step and lane: ~250
// instructions per tile, 0.048 ms).  A window toggles its pixel's bit twice -- on at its first row, off at the row after its last
// -- so every lane XORs its bit into two slots of an event array in LDS, and the masks are the running XOR of the events: one
// 64-lane prefix scan per 64 steps, each lane then holding (and storing, coalesced) the finished mask of one step.
constexpr int VT_EV = 2 * 255 + 32; // steps of the longest sweep (usd <= 255) + the end slot
__global__ __launch_bounds__(256) void stm_k_vwin_table(PQViews v, uint32_t *__restrict__ tab, int rec, int H, int W, int G, int nT, int top)
{
    __shared__ unsigned long long ev_all[4][VT_EV];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, g = blockIdx.x * 4 + wv, u = blockIdx.y, view = blockIdx.z;
    if (g >= G) return; // uniform per wave; no block-wide barrier below
    unsigned long long *ev = ev_all[wv];
    const u8 *__restrict__ armU = view ? v.armU[1] : v.armU[0], *__restrict__ armD = view ? v.armD[1] : v.armD[0];
    const int b = l >> 4, i = l & 15, y = u * 16 + i, x = 4 * g + b;
    int s0 = 0, nn = 0;
    if (y < H && x < W) {
        const int aU = armU[(size_t)y * W + x], aD = armD[(size_t)y * W + x];
        s0 = y - aU;
        nn = aU + aD;
    }
    vwin_build(tab + ((size_t)(view * nT + u) * G + g) * rec, ev, u, top, s0, nn);
}

// Sweep of one tile: quads K0 / 4 .. K0 / 4 + n_it - 1 of a ring of RQ quad slots.  Two register sets (A, B) alternate: the
// loads of a quad (LDS float4, four masks) are issued while the MFMAs of the quad before run.  LDS and scalar loads share one
// counter and scalar loads return out of order, so a set is waited for (STM_V12_ARRIVED) BEFORE the other set's loads are
// issued; the sched_barriers keep the compiler from re-rolling that order.
// the window table is read through the CONSTANT address space: uniform loads from it are always scalar loads (it is
// written by stm_k_vwin_table, an earlier launch of the same stream, and only read here)
typedef __attribute__((address_space(4))) const uint32_t ctab32;
typedef __attribute__((address_space(4))) const unsigned long long ctab64;
struct V12Quad { f4 c; unsigned long long m0, m1, m2, m3; };
struct V12Sweep { // a tile's sweep whose first quad is already on its way (issued before the step's barrier)
    const f4 *p;
    ctab64 *mk;
    int qs, n_it;
    V12Quad A;
};
#define STM_V12_LOAD(S, Q, RQ)                                                      \
    {                                                                               \
        Q.c = *S.p;                                                                 \
        Q.m0 = S.mk[0]; Q.m1 = S.mk[1]; Q.m2 = S.mk[2]; Q.m3 = S.mk[3];             \
        S.p += 64; S.mk += 4;                                                       \
        if (++S.qs == RQ) { S.qs = 0; S.p -= RQ * 64; } /* uniform */               \
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);           \
    }
#define STM_V12_ARRIVED(Q) __builtin_amdgcn_sched_barrier(0); asm volatile("" : : "v"(Q.c), "s"(Q.m0), "s"(Q.m1), "s"(Q.m2), "s"(Q.m3) : "memory");
#define STM_V12_MFMA(Q)                                                                                            \
    {                                                                                                              \
        float a0 = STM_MASKF(Q.m0), a1 = STM_MASKF(Q.m1), a2 = STM_MASKF(Q.m2), a3 = STM_MASKF(Q.m3);              \
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); /* all four selects in front of the MFMAs: no hazard padding between them */ \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a0, Q.c.x, acc, 0, 0, 0);                                       \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, Q.c.y, acc, 0, 0, 0);                                       \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a2, Q.c.z, acc, 0, 0, 0);                                       \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a3, Q.c.w, acc, 0, 0, 0);                                       \
    }
// hdr = {K0, n_it} of the tile's record `r`; slot0 = ring slot of the tile's first own quad (row 16 u); u4 = 4 u
__device__ __forceinline__ void v12t_begin(V12Sweep &S, const f4 *ring_l, int RQ, int slot0, int u4, ctab32 *r, int K0, int n_it)
{
    int qs = slot0 + ((K0 >> 2) - u4); // (K0 >> 2) - 4 u in [-UQ / 4, 3]
    if (qs < 0) qs += RQ;
    if (qs >= RQ) qs -= RQ;
    S.qs = qs;
    S.n_it = n_it;
    S.p = ring_l + qs * 64;
    S.mk = (ctab64 *)(r + 8);
    STM_V12_LOAD(S, S.A, RQ) // n_it == 0: any slot, any masks, unused
}
// the first MFMA of a sweep takes the constant 0 as its accumulator input (an inline operand: no register is cleared)
#define STM_V12_MFMA0(Q)                                                                                           \
    {                                                                                                              \
        float a0 = STM_MASKF(Q.m0), a1 = STM_MASKF(Q.m1), a2 = STM_MASKF(Q.m2), a3 = STM_MASKF(Q.m3);              \
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));                                                 \
        const f16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};           \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a0, Q.c.x, z, 0, 0, 0);                                         \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, Q.c.y, acc, 0, 0, 0);                                       \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a2, Q.c.z, acc, 0, 0, 0);                                       \
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a3, Q.c.w, acc, 0, 0, 0);                                       \
    }
__device__ __forceinline__ f16v v12t_run(V12Sweep &S, int RQ)
{
    f16v acc;
    const int n_it = S.n_it;
    if (n_it == 0) { // a tile without a single window row (or a timing knob)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        return acc;
    }
    V12Quad B;
    STM_V12_ARRIVED(S.A)
    STM_V12_LOAD(S, B, RQ)
    STM_V12_MFMA0(S.A)
    int it = 1;
    for (; it + 2 <= n_it; it += 2) {
        STM_V12_ARRIVED(B)
        STM_V12_LOAD(S, S.A, RQ)
        STM_V12_MFMA(B)
        STM_V12_ARRIVED(S.A)
        STM_V12_LOAD(S, B, RQ) // on the last trip: one quad past the sweep (any ring slot; the record is a quad longer than the longest sweep)
        STM_V12_MFMA(S.A)
    }
    if (it < n_it) {
        STM_V12_ARRIVED(B)
        STM_V12_MFMA(B)
    }
    return acc;
}

template <int NTP>
__global__ __launch_bounds__(128 * NTP) void stm_k_pq_v12t(PQViews v, const uint32_t *__restrict__ wtab, int rec, int H, int W, int G,
                                                           int NC, int UQ, int RQ1, int RQ2, int LAG, int dbg)
{
    constexpr int TS = 16 * NTP, TQ = 4 * NTP, NTH = 128 * NTP; // rows, quads per step; threads
    extern __shared__ f4 lds4[];
    f4 *ring1 = lds4, *ring2 = lds4 + RQ1 * 64; // a quad slot = float4 [4 columns][16 hypotheses] = rows 4q..4q+3 (1 KB)
    const int view = blockIdx.z, c = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int l = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool second = wid >= NTP;
    const int ti = wid - (second ? NTP : 0);
    const int nT = (H + 15) >> 4, nS = (nT + NTP - 1) / NTP;
    const int rsb = G * 256; // bytes between consecutive rows of the strip
    const size_t strip = ((size_t)c * H * G + g) * 16; // float4 index of (chunk c, row 0, group g, hypothesis 0)
    const uint32_t range = (uint32_t)(H - 1) * (uint32_t)rsb + 256u;
    for (int i = tid; i < (RQ1 + RQ2) * 64; i += NTH) lds4[i] = f4{0.f, 0.f, 0.f, 0.f}; // masked steps multiply ring contents by 0
    ctab32 *trow = (ctab32 *)(uintptr_t)(wtab + (STM_DBG(dbg, 8) ? (size_t)0 : ((size_t)view * nT * G + g) * rec)); // + u * G * rec: record of tile u
    const size_t tstep = STM_DBG(dbg, 8) ? (size_t)0 : (size_t)G * rec; // timing knob 8: every tile reads one record (scalar-cache hits only)
    // records of the tiles this wave sweeps: headers are fetched one step ahead (tiles past the image: the last tile's record, unused)
#define STM_V12_HDR(U, K0_, NIT_)                                                \
    {                                                                            \
        ctab32 *r_ = trow + (size_t)min(max(U, 0), nT - 1) * tstep;                       \
        K0_ = (int)r_[0];                                                        \
        NIT_ = STM_DBG(dbg, 1) ? 0 : STM_DBG(dbg, 2) ? 12 : STM_DBG(dbg, 8) ? 12 : (int)r_[1];          \
    }
    __syncthreads();
    if (!second) {
        // ---------------------------------------------------------------- first pass: ring 1 -> ring 2, all global loads
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)((const f4 *)(view ? v.b[1] : v.b[0]) + strip), 0, range, 0x00020000);
        const int qq = tid >> 4, n = tid & 15; // loader role: quad qq of a step's TQ quads, hypothesis n
        int vo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) vo[k] = (4 * qq + k) * rsb + n * 16;
        // rows [0, TS + UQ) before step 0 (UQ = usd rounded up to a quad: the windows of step t end before row TS (t + 1) + usd)
        const int nq0 = (TS + UQ) >> 2;
        for (int q0 = 0; q0 < nq0; q0 += TQ) {
            f4 tmp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) tmp[k] = STM_BLOAD(rin, vo[k] + q0 * 4 * rsb);
            if (q0 + qq < nq0) {
                f4 *q = ring1 + (q0 + qq) * 64 + n; // nq0 <= RQ1: no wrap yet
#pragma unroll
                for (int b = 0; b < 4; ++b) q[16 * b] = f4{tmp[0][b], tmp[1][b], tmp[2][b], tmp[3][b]};
            }
        }
        int ld_q = nq0;                       // first quad not yet in the ring (uniform)
        int ld_slot = nq0 == RQ1 ? 0 : nq0;   // its ring slot
        // input rows travel two steps ahead of their use (HBM latency under load exceeds one step): two register sets take turns,
        // X = the rows the NEXT step adds (written to the ring at the end of this step), Y = those of the step after
        f4 X[4], Y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = STM_BLOAD(rin, vo[k] + ld_q * 4 * rsb);
        int u = ti;
        int s1 = 4 * ti, s2 = 4 * ti; // ring slots of the tile's first quad (both rings hold more than one step)
        int K0, n_it, K0n, n_itn;
        STM_V12_HDR(u, K0, n_it)
        V12Sweep S;
        const bool early = UQ > 16 * (NTP - 1);
        __syncthreads(); // initial rows visible
        if (early) v12t_begin(S, ring1 + l, RQ1, s1, 4 * u, trow + (size_t)min(u, nT - 1) * tstep, K0, n_it);
#define STM_V12_STEP1(PRE, FAR)                                                                                              \
    {                                                                                                                        \
        __syncthreads(); /* ring 1 holds the rows of this step */                                                            \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) FAR[k] = STM_DBG(dbg, 4) ? PRE[k] : STM_BLOAD(rin, vo[k] + (ld_q + TQ) * 4 * rsb); \
        if (!early) v12t_begin(S, ring1 + l, RQ1, s1, 4 * u, trow + (size_t)min(u, nT - 1) * tstep, K0, n_it);               \
        STM_V12_HDR(u + NTP, K0n, n_itn)                                                                                     \
        if (u < nT) {                                                                                                        \
            const f16v acc = v12t_run(S, RQ1);                                                                               \
            /* registers 4b..4b+3 of lane 16q + n = out[rows 16u + 4q .. +3][column b][hypothesis n]: one float4 per column */ \
            int so = s2 + (l >> 4);                                                                                          \
            if (so >= RQ2) so -= RQ2;                                                                                        \
            f4 *q = ring2 + so * 64 + (l & 15);                                                                              \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) q[16 * b] = f4{acc[4 * b], acc[4 * b + 1], acc[4 * b + 2], acc[4 * b + 3]}; \
        }                                                                                                                    \
        s1 += TQ;                                                                                                            \
        if (s1 >= RQ1) s1 -= RQ1;                                                                                            \
        s2 += TQ;                                                                                                            \
        if (s2 >= RQ2) s2 -= RQ2;                                                                                            \
        __syncthreads(); /* everyone is done reading ring 1: the oldest TQ quads can be replaced */                           \
        {                                                                                                                    \
            int sw = ld_slot + qq;                                                                                           \
            if (sw >= RQ1) sw -= RQ1;                                                                                        \
            f4 *q = ring1 + sw * 64 + n; /* rows past the image: the buffer returned zeros */                                \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) q[16 * b] = f4{PRE[0][b], PRE[1][b], PRE[2][b], PRE[3][b]};        \
        }                                                                                                                    \
        ld_q += TQ;                                                                                                          \
        ld_slot += TQ;                                                                                                       \
        if (ld_slot >= RQ1) ld_slot -= RQ1;                                                                                  \
        u += NTP;                                                                                                            \
        K0 = K0n;                                                                                                            \
        n_it = n_itn;                                                                                                        \
        /* the next tile's first quad (rows < 16 u + 16 (NTP - 1)) was written at least a step ago when the rows being written */ \
        /* now start later (16 (NTP - 1) < UQ): fetch it in front of the barrier */                                          \
        if (early) v12t_begin(S, ring1 + l, RQ1, s1, 4 * u, trow + (size_t)min(u, nT - 1) * tstep, K0, n_it);                \
    }
        const int nst = nS + LAG;
        for (int t = 0; t < nst; t += 2) {
            STM_V12_STEP1(X, Y)
            if (t + 1 < nst) STM_V12_STEP1(Y, X)
        }
#undef STM_V12_STEP1
    } else {
        // ---------------------------------------------------------------- second pass: ring 2 -> HBM, all stores, LAG steps behind
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)((f4 *)(view ? v.a[1] : v.a[0]) + strip), 0, range, 0x00020000);
        int vo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vo[i] = (4 * (l >> 4) + i) * rsb + (l & 15) * 16;
        int u = -LAG * NTP + ti;
        int s2 = 4 * ti; // ring slot of the first quad of tile max(u, 0)
        int K0, n_it, K0n, n_itn;
        STM_V12_HDR(u, K0, n_it)
        V12Sweep S;
        __syncthreads();
        v12t_begin(S, ring2 + l, RQ2, s2, 4 * max(u, 0), trow + (size_t)min(max(u, 0), nT - 1) * tstep, K0, n_it);
        const int nst = nS + LAG;
        for (int t = 0; t < nst; ++t) {
            __syncthreads(); // ring 2 holds the first-pass rows of the steps before
            STM_V12_HDR(u + NTP, K0n, n_itn)
            if (u >= 0 && u < nT) {
                const f16v acc = v12t_run(S, RQ2);
                const int ro = 16 * u * rsb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f4 o = {acc[i], acc[4 + i], acc[8 + i], acc[12 + i]};
                    if (!STM_DBG(dbg, 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, o), rout, vo[i] + ro, 0, 2); // rows >= H: out of range, dropped
                }
                s2 += TQ;
                if (s2 >= RQ2) s2 -= RQ2;
            }
            __syncthreads();
            u += NTP;
            v12t_begin(S, ring2 + l, RQ2, s2, 4 * max(u, 0), trow + (size_t)min(max(u, 0), nT - 1) * tstep, K0n, n_itn);
        }
    }
#undef STM_V12_HDR
}

// ------------------------------------------------------------------ launchers
size_t pq_volume_floats(int D, int H, int W) { return (size_t)((D + 15) / 16) * H * ((W + 3) / 4) * 64; }

// LDS of the table-driven fused vertical kernel: rings of (48 + 2 UQ) and (48 (LAG + 1) + UQ) rows of 256 B, UQ = usd rounded up to 4
static size_t v12t_smem(int usd)
{
    const int TS = 48, UQ = (usd + 3) & ~3, LAG = (UQ + TS - 1) / TS + 1;
    return (size_t)(TS + 2 * UQ + TS * (LAG + 1) + UQ) * 256;
}
// arms longer than this do not fit the CU's 160 KB, and the kernels address a strip of the volume with 32-bit byte offsets:
// the caller falls back to the vector-ALU kernels (stm_kernels_agg.hip)
bool aggm_supports(int usd, int H, int W)
{
    if (usd > 255) usd = 255;
    return usd >= 1 && v12t_smem(usd) <= 160 * 1024 && (unsigned long long)H * ((W + 3) / 4) * 256ull < (1ull << 31);
}

// PQ -> the caller's volume (a table of plane pointers, a slab, or quads): thread = (group g, quad of the chunk); the inverse
// of stm_k_to_pq (stm_kernels_hslo.hip).  Per plane and quarter-wave a 256-byte row piece.
template <bool QUAD> __global__ __launch_bounds__(256) void stm_k_from_pq(const f4 *__restrict__ in, Vol out, int D, int H, int W, int G)
{
    const int t = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    const int g = t >> 2, q = 4 * c + (t & 3);
    if (g >= G || 4 * q >= D) return;
    const f4 *src = in + (((size_t)c * H + y) * G + g) * 16 + 4 * (t & 3);
    const f4 d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3]; // hypotheses 4q..4q+3, four pixels each
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = 4 * g + k;
        if (x < W) store_quad<QUAD, false>(out, q, D, (size_t)y * W + x, make_float4(d0[k], d1[k], d2[k], d3[k]));
    }
}
void launch_from_pq(const float *pq, Vol out, int D, int H, int W)
{
    const int G = (W + 3) / 4, NC = (D + 15) / 16;
    if (out.quad) STM_LAUNCH(stm_k_from_pq<true>, dim3(cdiv(4 * G, 256), H, NC), dim3(256), 0, stream(), (const f4 *)pq, out, D, H, W, G);
    else STM_LAUNCH(stm_k_from_pq<false>, dim3(cdiv(4 * G, 256), H, NC), dim3(256), 0, stream(), (const f4 *)pq, out, D, H, W, G);
    STM_CHECK_LAUNCH();
}

// The aggregation chain on PQ volumes for `nviews` views (1 or 2).
//   from_costs: the first horizontal pass computes the initial costs itself (images -> vol_b), else it reads vol_a;
//   then both vertical passes (vol_b -> vol_a); then the last horizontal pass, vol_a -> disparities (wta) or -> vol_b.
static void aggm_chain(PQViews &v, int nviews, bool from_costs, bool wta, const float *lut, int D, int zd, int H, int W, int usd,
                       uint32_t *htab_ready = nullptr, uint32_t *vtab_ready = nullptr)
{
    const int G = (W + 3) / 4, NC = (D + 15) / 16;
    if (usd > 255) usd = 255;
    constexpr int NW = 8;
    const int HG = ((usd + 3) / 4 + 2) & ~1, NG = 4 * NW + 2 * HG; // halo groups: ceil(usd / 4) on either side of a segment + 1 for the read-ahead, rounded to an even number (the tile is filled four groups at a time)
    const int nseg = cdiv(W, 16 * NW), nblk = ((nseg * H * nviews + 7) / 8) * 8;
    const size_t smem_h = (size_t)4 * NG * 256 + 16 * NW * 4; // tile, window table
    const int dbgh = timing_knobs();
    int pad = zd > D - 1 - zd ? zd : D - 1 - zd;
    pad = (pad < 0 ? 0 : pad) + 15; // + the padded hypotheses of the last chunk
    const size_t smem_cost = smem_h + (size_t)(4 * NG * 4 + 4 * pad + 768 + 72) * 4;
    const bool fuse_cost = from_costs && (agg_variant() / 1000000) % 10 != 1; // 1: separate stm_k_pq_cost + volume-reading first pass
    const int spl = nseg > 24 ? cdiv(nseg, 16) : 1; // blocks per image row of the streaming passes
    const bool streaming = NG / 4 >= NW && (agg_variant() / 10) % 10 != 1 && (NC <= 4 || (agg_variant() / 10) % 10 != 2); // the ring is at least one segment long; 20: D > 64 on the block-per-segment kernels as before
    // The window tables of the vertical passes and of the last horizontal pass depend on the arms only; they are built once
    // per call for all views (the horizontal one already by stm_k_cross_arms when the caller passes htab_ready)
    constexpr int NTP = 3, TS = 16 * NTP; // stm_k_pq_v12t: 2 and 4 tiles per pass and step: 0.729 ms each against 0.679
    const int UQ = (usd + 3) & ~3, nT = (H + 15) / 16;
    const bool regs = aggv_supports(usd) && (agg_variant() / 10000000) % 10 != 1; // round 4: a strip's rows in registers (stm_kernels_aggv.hip); 10000000: the LDS-ring kernel
    const int rec = regs ? aggv_table_rec() : 8 + 8 * ((2 * usd + 21) / 4 + 2); // header + the longest sweep + one quad of read-ahead
    const int LAG = (UQ + TS - 1) / TS + 1;
    const int RQ1 = (TS + 2 * UQ) / 4, RQ2 = (TS * (LAG + 1) + UQ) / 4;
    uint32_t *vtab = regs && vtab_ready ? vtab_ready : Workspace::get<uint32_t>((size_t)nviews * nT * G * rec);
    // round 4: the row's window range in registers (stm_kernels_aggh.hip); 100000000: the LDS row walk stm_k_pq_hs
    const bool hregs = wta && aggh_supports(usd, D) && (agg_variant() / 100000000) % 10 != 1;
    uint32_t *htab = !hregs ? nullptr : htab_ready ? htab_ready : Workspace::get<uint32_t>(aggh_table_dwords(nviews, H, W));
    {
        if (!(regs && vtab_ready)) {
            ProfScope p("pq_vtab");
            STM_LAUNCH(stm_k_vwin_table, dim3(cdiv(G, 4), nT, nviews), dim3(256), 0, stream(), v, vtab, rec, H, W, G, nT, regs ? aggv_table_top() : -1);
            STM_CHECK_LAUNCH();
        }
        if (hregs && !htab_ready) {
            ProfScope p("pq_htab");
            launch_hwin_table(v, nviews, htab, H, W);
        }
    }
    if (from_costs && !fuse_cost) {
        ProfScope p("pq_cost");
        const size_t smem = (size_t)(2 * PC_TX + 2 * (PC_TX + 2 * pad) + 768 + 72) * 4;
        allow_lds_m((const void *)stm_k_pq_cost, smem);
        STM_LAUNCH(stm_k_pq_cost, dim3(cdiv(W, PC_TX), H, nviews), dim3(PC_TX), smem, stream(), v, lut, D, zd, H, W, G, NC, pad);
        STM_CHECK_LAUNCH();
    }
    {
        ProfScope p("pq_h");
        // the cost-computing pass works on 192-pixel segments when two such blocks still fit a CU's LDS (24 waves per CU either way)
        constexpr int NWC = 12;
        const int NGc = 4 * NWC + 2 * HG;
        const size_t smem_c12 = (size_t)4 * NGc * 256 + 16 * NWC * 4 + (size_t)(4 * NGc * 4 + 4 * pad + 768 + 72) * 4;
        const size_t smem_hc = (size_t)4 * NGc * 256 + 16 * NWC * 4 + 8 * NWC + (size_t)(2 * 16 * NWC + 2 * (16 * NWC + 2 * pad)) * 8 + (768 + 72) * 4;
        const int nsegc = cdiv(W, 16 * NWC);
        const size_t smem_hc8 = (size_t)4 * NG * 256 + 16 * NW * 4 + 8 * NW + (size_t)(2 * 16 * NW + 2 * (16 * NW + 2 * pad)) * 8 + (768 + 72) * 4;
        const bool hc_ok = fuse_cost && (agg_variant() / 1000) % 10 == 0 && (NC <= 4 || (agg_variant() / 10) % 10 != 2);
        if (hc_ok && smem_hc <= 80 * 1024 && 16 * NWC + 2 * pad <= 64 * NWC) {
            // streaming row walk (the staged pixels fit one per thread); 2000: one block per segment as in round 2
            const int splc = nsegc > 24 ? cdiv(nsegc, 16) : 1;
            allow_lds_m((const void *)stm_k_pq_hc<NWC>, smem_hc);
            STM_LAUNCH((stm_k_pq_hc<NWC>), dim3(nviews * H * splc), dim3(64 * NWC), smem_hc, stream(), v, D, zd, H, W, G, NC, HG, nsegc, splc, lut, pad, dbgh);
        } else if (hc_ok && smem_hc8 <= 80 * 1024 && 16 * NW + 2 * pad <= 64 * NW) {
            // large D (more staged pixels): 128-pixel segments keep two blocks per CU
            allow_lds_m((const void *)stm_k_pq_hc<NW>, smem_hc8);
            STM_LAUNCH((stm_k_pq_hc<NW>), dim3(nviews * H * spl), dim3(64 * NW), smem_hc8, stream(), v, D, zd, H, W, G, NC, HG, nseg, spl, lut, pad, dbgh);
        } else if (fuse_cost && smem_c12 <= 80 * 1024 && (agg_variant() / 1000) % 10 != 1) { // 1000: 128-pixel segments as in the other passes
            const int nblkc = ((nsegc * H * nviews + 7) / 8) * 8;
            allow_lds_m((const void *)stm_k_pq_h<NWC, false, true>, smem_c12);
            STM_LAUNCH((stm_k_pq_h<NWC, false, true>), dim3(nblkc), dim3(64 * NWC), smem_c12, stream(), v, D, zd, H, W, G, NC, HG, nsegc, dbgh, lut, pad, nviews);
        } else if (fuse_cost) {
            allow_lds_m((const void *)stm_k_pq_h<NW, false, true>, smem_cost);
            STM_LAUNCH((stm_k_pq_h<NW, false, true>), dim3(nblk), dim3(64 * NW), smem_cost, stream(), v, D, zd, H, W, G, NC, HG, nseg, dbgh, lut, pad, nviews);
        } else if (streaming) { // vol_a -> vol_b
            allow_lds_m((const void *)stm_k_pq_hs<NW, false>, smem_h);
            STM_LAUNCH((stm_k_pq_hs<NW, false>), dim3(nviews * H * spl), dim3(64 * NW), smem_h, stream(), v, D, zd, H, W, G, NC, HG, nseg, spl, dbgh);
        } else {
            allow_lds_m((const void *)stm_k_pq_h<NW, false, false>, smem_h);
            STM_LAUNCH((stm_k_pq_h<NW, false, false>), dim3(nblk), dim3(64 * NW), smem_h, stream(), v, D, zd, H, W, G, NC, HG, nseg, dbgh, lut, 0, nviews);
        }
        STM_CHECK_LAUNCH();
    }
    {
        // fused vertical kernel
        ProfScope p("pq_v12");
        if (regs) {
            launch_pq_v12r(v, nviews, vtab, rec, H, W, G, NC);
        } else {
            const size_t smem = (size_t)(RQ1 + RQ2) * 1024;
            allow_lds_m((const void *)stm_k_pq_v12t<NTP>, smem);
            STM_LAUNCH(stm_k_pq_v12t<NTP>, dim3(G, NC, nviews), dim3(128 * NTP), smem, stream(), v, vtab, rec, H, W, G, NC, UQ, RQ1, RQ2, LAG, dbgh);
            STM_CHECK_LAUNCH();
        }
    }
    {
        ProfScope p("pq_hw");
        if (hregs) {
            launch_pq_hsr(v, nviews, htab, D, zd, H, W);
        } else if (streaming && !wta) {
            allow_lds_m((const void *)stm_k_pq_hs<NW, false>, smem_h);
            STM_LAUNCH((stm_k_pq_hs<NW, false>), dim3(nviews * H * spl), dim3(64 * NW), smem_h, stream(), v, D, zd, H, W, G, NC, HG, nseg, spl, dbgh);
        } else if (streaming) {
            // D > 64: a pixel's best (cost, d) of the chunk sets before waits in LDS, 8 bytes per pixel of the block's part of the
            // row: parts of at most 1920 pixels (15 KB; two blocks per CU)
            const int splw = NC > 4 ? std::max(spl, cdiv(nseg * 16 * NW, 1920)) : spl;
            const size_t smem_w = smem_h + (NC > 4 ? (size_t)cdiv(nseg, splw) * 16 * NW * 8 : 0);
            allow_lds_m((const void *)stm_k_pq_hs<NW, true>, smem_w);
            STM_LAUNCH((stm_k_pq_hs<NW, true>), dim3(nviews * H * splw), dim3(64 * NW), smem_w, stream(), v, D, zd, H, W, G, NC, HG, nseg, splw, dbgh);
        } else if (!wta) {
            allow_lds_m((const void *)stm_k_pq_h<NW, false, false>, smem_h);
            STM_LAUNCH((stm_k_pq_h<NW, false, false>), dim3(nblk), dim3(64 * NW), smem_h, stream(), v, D, zd, H, W, G, NC, HG, nseg, dbgh, lut, 0, nviews);
        } else {
            allow_lds_m((const void *)stm_k_pq_h<NW, true, false>, smem_h);
            STM_LAUNCH((stm_k_pq_h<NW, true, false>), dim3(nblk), dim3(64 * NW), smem_h, stream(), v, D, zd, H, W, G, NC, HG, nseg, dbgh, lut, 0, nviews);
        }
        STM_CHECK_LAUNCH();
    }
}

// cost -> H -> V, V -> H + WTA for both views of a frame.  vol_a / vol_b: two PQ volumes per view (pq_volume_floats each).
// keep_volume: the last pass writes the aggregated costs to vol_b instead of doing WTA (the HSLO stage follows; disp unused).
// htab_ready: the horizontal window table of both views (aggm_frame_htab_dwords) already built by launch_cross_arms2, or nullptr.
// aggm_frame_htab_dwords: its size when this frame's last pass will use it, else 0.
size_t aggm_frame_vtab_dwords(int H, int W, int usd, int *rec, int *top)
{
    if (usd > 255) usd = 255;
    const bool regs = aggv_supports(usd) && (agg_variant() / 10000000) % 10 != 1;
    *rec = regs ? aggv_table_rec() : 0;
    *top = regs ? aggv_table_top() : -1;
    if (!regs || (agg_variant() / 1000000000) % 10 == 1) return 0; // 1000000000: the stand-alone table kernels
    return (size_t)2 * ((H + 15) / 16) * ((W + 3) / 4) * *rec;
}
size_t aggm_frame_htab_dwords(int D, int H, int W, int usd, bool keep_volume)
{
    if (usd > 255) usd = 255;
    const bool hregs = !keep_volume && aggh_supports(usd, D) && (agg_variant() / 100000000) % 10 != 1;
    return hregs && (agg_variant() / 1000000000) % 10 != 1 ? aggh_table_dwords(2, H, W) : 0; // 1000000000: the stand-alone table kernel
}
void launch_aggm_frame(const uint32_t *const *pk, const uint32_t *const *cen, const float *lut, float *const *vol_a, float *const *vol_b,
                       const u8 *const *armU, const u8 *const *armD, const u8 *const *armL, const u8 *const *armR, float *const *disp,
                       int D, int zd, int H, int W, int usd, bool keep_volume, uint32_t *htab_ready, uint32_t *vtab_ready)
{
    PQViews v;
    for (int i = 0; i < 2; ++i) {
        v.pk[i] = pk[i]; v.cen[i] = cen[i]; v.a[i] = vol_a[i]; v.b[i] = vol_b[i];
        v.armU[i] = armU[i]; v.armD[i] = armD[i]; v.armL[i] = armL[i]; v.armR[i] = armR[i]; v.disp[i] = disp[i];
    }
    aggm_chain(v, 2, true, !keep_volume, lut, D, zd, H, W, usd, htab_ready, vtab_ready);
}

// The per-stage aggregation (ca_cross / d_ca_cross, d_ca_cross.cu:255-270) of ONE volume in the caller's layout on the
// matrix-pipe kernels: volume -> PQ, H, V V, H, PQ -> `out` (which may be `in`: the device flavour returns the result in its
// input volume, SURVEY A-Q11).  Two PQ volumes and the window table come from the current Workspace scope.
static int vtab_rec(int usd)
{
    if (usd > 255) usd = 255;
    const bool regs = aggv_supports(usd) && (agg_variant() / 10000000) % 10 != 1;
    return regs ? aggv_table_rec() : 8 + 8 * ((2 * usd + 21) / 4 + 2);
}
size_t aggm_stage_bytes(int D, int H, int W, int usd)
{
    const int G = (W + 3) / 4, nT = (H + 15) / 16;
    return 2 * pq_volume_floats(D, H, W) * sizeof(float) + (size_t)nT * G * vtab_rec(usd) * 4 + 4096;
}
bool launch_aggm_stage(Vol in, Vol out, const u8 *armU, const u8 *armD, const u8 *armL, const u8 *armR, int D, int H, int W, int usd)
{
    const size_t VP = pq_volume_floats(D, H, W);
    float *m = Workspace::get<float>(2 * VP);
    uint32_t *odd = Workspace::get<uint32_t>(64);
    if (failed()) return true; // (error mode 1: nothing was launched, nothing to fall back to)
    STM_CHECK(hipMemsetAsync(odd, 0, 4, stream()));
    PQViews v;
    for (int i = 0; i < 2; ++i) {
        v.pk[i] = nullptr; v.cen[i] = nullptr; v.a[i] = m; v.b[i] = m + VP;
        v.armU[i] = armU; v.armD[i] = armD; v.armL[i] = armL; v.armR[i] = armR; v.disp[i] = nullptr;
    }
    launch_to_pq(in, m, D, H, W, odd);
    // An element that is not an ordinary number cannot go through masked multiply-adds (0 * inf = NaN would reach every pixel
    // of the tile whose sweep passes it, where the reference only touches the windows that contain it, d_ca_cross_sum.cu:284-289)
    uint32_t h_odd = 0;
    STM_CHECK(hipMemcpyAsync(&h_odd, odd, 4, hipMemcpyDeviceToHost, stream()));
    STM_CHECK(hipStreamSynchronize(stream()));
    if (h_odd) return false;
    aggm_chain(v, 1, false, false, nullptr, D, 0, H, W, usd);
    launch_from_pq(m + VP, out, D, H, W);
    return true;
}

} // namespace stm
