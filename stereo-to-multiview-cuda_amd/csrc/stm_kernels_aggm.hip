// stm_kernels_aggm.hip -- cross-based cost aggregation of the frame pipeline on the gfx950 MATRIX pipe.
//
// Reference stages replaced, same arithmetic as stm_kernels_agg.hip:
//   ci_ad_kernel_5 / ci_census_kernel_6 / ci_adcensus_kernel   d_ci_ad.cu:73-159, d_ci_census.cu:197-254, d_ci_adcensus.cu:10-36
//   ca_cross_hsum_kernel_3        d_ca_cross_sum.cu:243-293   (horizontal window sum)
//   ca_cross_vhsum_kernel_2       d_ca_cross_sum.cu:148-198   (vertical window sum; the two transposes are deleted)
//   dc_wta_kernel                 d_dc_wta.cu:9-35
//
// Why the matrix pipe, when the path has no dense contraction: the reference sums every window element by element in
// float32, ascending (d_ca_cross_sum.cu:284-289), and WTA indices must be bit-exact, so the summation ORDER is fixed and
// a prefix-sum formulation is ruled out (hard part 1).  v_mfma_f32_4x4x1_16B_f32 computes, for each of
// 16 blocks, D[i][j] = A[i] * B[j] + C[i][j] with ONE product per output (K = 1): with A in {0, 1} that is exactly
// "acc = acc + b" (or "acc = acc", 0 * b = +0 for finite b), one float32 rounding per step -- the reference's chain.
// tools/mfma_probe.hip verifies on hardware: register layout, bit-exactness of a 96-step masked chain, the CBSZ/ABID
// broadcast.  One instruction adds one window element to 4 pixels x 64 hypotheses (256 adds per 8 cycles per SIMD, the
// full fp32 rate), and each LDS value feeds 4 pixels instead of 1: the LDS traffic that bounds stm_k_agg_h / stm_k_agg_v
//  drops 4x and no lane idles while a neighbour's longer window finishes.
//
// Roles.  B operand = costs: lane l of block b = l / 4 supplies B[j = l % 4].  A operand = window masks: with CBSZ = 2
// the A values of block (4 * (b / 4) + ABID) are broadcast to the 4 blocks of each group, so ONE mask register per lane
// holds the masks of four consecutive steps (block b % 4 = step) and ABID = 0..3 selects the step: 3 VALU instructions
// per 4 steps.  Result: register i of lane 4b + j = out[pixel i][hypothesis j of block b].
//
// Volume layout inside the frame pipeline ("PQ"): float4 [chunk = d / 16][y][g = x / 4][dd = d % 16], the float4 = the four
// pixels 4g..4g+3 of one hypothesis.  Horizontal pass: a lane reads one float4 = four consecutive window steps of its
// hypothesis; vertical pass: a lane reads one float4 per row = the same step of four column chains.  Every global access
// of a wave is 16 B per lane, 256 B contiguous per 16 lanes, 1 KB contiguous per wave in the horizontal kernels.
#include "stm_common.h"

namespace stm {

typedef float f4 __attribute__((ext_vector_type(4)));

#define STM_MFMA(m, b, acc, abid) __builtin_amdgcn_mfma_f32_4x4x1f32(m, b, acc, 2, abid, 0)

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
template <int NW, bool WTA>
__global__ __launch_bounds__(64 * NW) void stm_k_pq_h(PQViews v, int D, int zd, int H, int W, int G, int NC, int HG, int nseg)
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
        const int per_xcd = (nseg * H * 2 + 7) >> 3;
        const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (logical >= nseg * H * 2) return;
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

    if (tid < TX) {
        const int x = X0seg + tid;
        uint32_t e = (uint32_t)(x - 4 * gbase); // empty window
        if (x < W) {
            const int aL = armL[row + x], aR = armR[row + x];
            e = (uint32_t)(x - aL - 4 * gbase) | ((uint32_t)(aL + aR) << 16); // window [x - armL, x + armR), d_ca_cross_sum.cu:277-289
        }
        sn[tid] = e;
    }

    const int l = tid & 63, w = tid >> 6;
    const int pt = l >> 4, dq = (l >> 2) & 3, dd = l & 15;
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
        for (int r0 = 0; r0 < NG * 16; r0 += 2 * NT) {
            f4 tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cl = k & 3, r = r0 + (k >> 2) * NT + tid;
                const f4 *__restrict__ rowp = in + ((size_t)min(c0 + cl, NC - 1) * H + y) * G * 16; // uniform: scalar base
                const int g = min(max(gbase + (r >> 4), 0), G - 1);
                tmp[k] = nt_load4(rowp + (g * 16 + (r & 15)));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cl = k & 3, r = r0 + (k >> 2) * NT + tid;
                const int g = gbase + (r >> 4);
                if (r < NG * 16) tile[cl * NG * 16 + r] = (c0 + cl < NC && g >= 0 && g < G) ? tmp[k] : zero4;
            }
        }
        __syncthreads();
        if (X0 >= W) continue; // whole wave out of the image (uniform per wave); it still takes part in the barriers
        if (cs == 0) {
            const uint32_t e = sn[16 * w + 4 * pt + (l & 3)];
            const int srel = (int)(e & 0xffffu);
            nn = (int)(e >> 16);
            const int lo = nn ? ((srel - 4 * pt) >> 2) : 0x7fffffff;
            const int hi = nn ? ((srel + nn - 4 * pt + 3) >> 2) : -0x7fffffff;
            G0r = wave_min_i(lo);
            const int Gend = wave_max_i(hi);
            n_it = Gend - G0r; // <= 0 when every window of the wave is empty
            t0 = 4 * (G0r + pt) + dq - srel;
        }
        f4 acc[4] = {zero4, zero4, zero4, zero4};
        {
            const f4 *p = tile + (G0r + pt) * 16 + dd;
            int t = t0;
            for (int it = 0; it < n_it; ++it) {
                const f4 v0 = p[0], v1 = p[NG * 16], v2 = p[2 * NG * 16], v3 = p[3 * NG * 16];
                p += 16;
                const float m = ((unsigned)t < (unsigned)nn) ? 1.0f : 0.0f;
                t += 4;
                acc[0] = STM_MFMA(m, v0.x, acc[0], 0); acc[1] = STM_MFMA(m, v1.x, acc[1], 0);
                acc[2] = STM_MFMA(m, v2.x, acc[2], 0); acc[3] = STM_MFMA(m, v3.x, acc[3], 0);
                acc[0] = STM_MFMA(m, v0.y, acc[0], 1); acc[1] = STM_MFMA(m, v1.y, acc[1], 1);
                acc[2] = STM_MFMA(m, v2.y, acc[2], 1); acc[3] = STM_MFMA(m, v3.y, acc[3], 1);
                acc[0] = STM_MFMA(m, v0.z, acc[0], 2); acc[1] = STM_MFMA(m, v1.z, acc[1], 2);
                acc[2] = STM_MFMA(m, v2.z, acc[2], 2); acc[3] = STM_MFMA(m, v3.z, acc[3], 2);
                acc[0] = STM_MFMA(m, v0.w, acc[0], 3); acc[1] = STM_MFMA(m, v1.w, acc[1], 3);
                acc[2] = STM_MFMA(m, v2.w, acc[2], 3); acc[3] = STM_MFMA(m, v3.w, acc[3], 3);
            }
        }
        // register i of lane l = out[pixel X0 + 4 pt + i][hypothesis 16 (c0 + cl) + dd]
        if (!WTA) {
            if ((X0 >> 2) + pt < G) {
#pragma unroll
                for (int cl = 0; cl < 4; ++cl)
                    if (c0 + cl < NC) nt_store4(out + (((size_t)(c0 + cl) * H + y) * G + (X0 >> 2)) * 16 + l, acc[cl]);
            }
        } else {
            // first strictly-lowest cost wins, ascending d (d_dc_wta.cu:19-34): per lane the chunks come in ascending d
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int d = (c0 + cl) * 16 + dd;
                if (d < D) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (bc[i] > acc[cl][i]) { bc[i] = acc[cl][i]; bd[i] = d; }
                }
            }
        }
    }
    if (WTA && X0 < W) {
        // across the 16 lanes (dd) of a pixel tile: lowest cost, ties to the lowest d
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
            const int x = X0 + 4 * pt;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x + i < W) disp[row + x + i] = res[i];
        }
    }
}

// ------------------------------------------------------------------ both vertical passes, fused
// One block = one strip of 4 columns (one group) x one chunk of 16 hypotheses, 2 NTP waves: waves 0..NTP-1 run the first
// vertical pass, 16 output rows each per step, from LDS ring 1 (rows of the input volume) into LDS ring 2; waves NTP..2NTP-1
// run the second pass LAG steps behind, from ring 2 to HBM.  The intermediate volume never leaves the CU (8 V per frame
// instead of 12 V).  NTP = 3: a step is 48 rows, so the second ring needs a lag of only 2 steps and two blocks = 12 waves
// fit a CU's LDS; with one tile per pass and step it was 6 waves and the dependent MFMA chains ran at a third of their rate.
// Wave tile = 16 rows x 4 columns x 16 hypotheses: lane l: rt = l / 16 (row tile of 4), dq = (l / 4) % 4, i = l % 4; the four
// column chains take the four components of each ring float4.  Window of pixel (y, x): rows [y - armU, y + armD).
// Every global access of a step is issued one step ahead (input rows and the next tile's arm bytes) and the LDS reads of
// the window loop one iteration ahead.
struct ArmWords { uint32_t u, d; }; // armU / armD bytes of the lane's row, columns 4g..4g+3
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
// One dword load per plane, always issued (row clamped; when W % 4 != 0 the address is not dword-aligned and the last group's
// bytes beyond column W - 1 belong to the next row -- they are masked by the caller, and the workspace slab extends past
// the last plane): the loads of a step must be unconditional for the compiler to count them in its s_waitcnt.
__device__ __forceinline__ ArmWords load_arm_words(const u8 *__restrict__ armU, const u8 *__restrict__ armD, int yy, int g, int H, int W)
{
    const size_t p = (size_t)min(yy, H - 1) * W + 4 * g;
    ArmWords a;
    a.u = *(const u32_unaligned *)(armU + p);
    a.d = *(const u32_unaligned *)(armD + p);
    return a;
}

template <int NTP>
__global__ __launch_bounds__(128 * NTP) void stm_k_pq_v12(PQViews v, int H, int W, int G, int NC, int usd, int R1, int R2, int LAG, int dbg)
{
    constexpr int PV_TS = 16 * NTP, NTH = 128 * NTP, LB = 8 * NTP; // rows per step, threads, rows per load batch
    extern __shared__ f4 lds4[];
    f4 *ring1 = lds4, *ring2 = lds4 + R1 * 16;
    const int view = blockIdx.z, c = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const f4 *__restrict__ in = (const f4 *)(view ? v.b[1] : v.b[0]) + ((size_t)c * H * G + g) * 16;
    f4 *__restrict__ out = (f4 *)(view ? v.a[1] : v.a[0]) + ((size_t)c * H * G + g) * 16;
    const u8 *__restrict__ armU = view ? v.armU[1] : v.armU[0], *__restrict__ armD = view ? v.armD[1] : v.armD[0];
    const size_t rstride = (size_t)G * 16; // float4 elements between consecutive rows of the strip
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < (R1 + R2) * 16; i += NTH) lds4[i] = zero4; // masked steps multiply ring contents by 0: keep them finite

    const int l = tid & 63, wv = (tid >> 6) >= NTP, ti = (tid >> 6) - (wv ? NTP : 0); // pass, tile of the step
    const int rt = l >> 4, dq = (l >> 2) & 3, dd = l & 15;
    const int nT = (H + 15) >> 4, nS = (nT + NTP - 1) / NTP; // tiles of 16 rows, steps
    const f4 *ring = wv ? ring2 : ring1;
    const int R = wv ? R2 : R1;
    // rows of the input volume: the strip's row r is one 256-B piece; thread t loads piece (t / 16) of a LB-row batch
    const int lrow = tid >> 4;
    __syncthreads();
    // rows needed by step 0: [0, TS + usd - 1); four loads in flight per thread
    int loaded = min(PV_TS + usd - 1, H);
    for (int r0 = 0; r0 < loaded; r0 += 4 * LB) {
        f4 tmp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) tmp[k] = nt_load4(in + (size_t)min(r0 + LB * k + lrow, H - 1) * rstride + dd);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + LB * k + lrow;
            if (r < loaded) ring1[(r % R1) * 16 + dd] = tmp[k];
        }
    }
    int slot_ld = loaded % R1; // ring-1 slot of row `loaded`
    // this wave's tile of step t is u = NTP t + ti (first pass) or NTP (t - LAG) + ti (second pass); its arm bytes are fetched
    // one step ahead
    int u = (wv ? -LAG * NTP : 0) + ti;
    int slot_rd = (16 * ti) % R, slot_wr = (16 * ti) % R2; // slots of the tile's first row in the ring this wave reads / in ring 2
    ArmWords nxt = load_arm_words(armU, armD, max(u, 0) * 16 + 4 * rt + (l & 3), g, H, W);
    for (int t = 0; t < nS + LAG; ++t, u += NTP) {
        __syncthreads(); // ring 1 holds the rows of this step; ring 2 the first-pass rows of the steps before
        // decode this step's arm bytes (loaded a step ago) BEFORE the loads below are issued: no wait on anything recent
        const int y0 = u * 16;
        const int yy = y0 + 4 * rt + (l & 3);
        int s[4], n[4];
        int lo = 0x7fffffff, hi = -0x7fffffff;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int aU = (int)((nxt.u >> (8 * cc)) & 0xffu), aD = (int)((nxt.d >> (8 * cc)) & 0xffu);
            s[cc] = yy - aU;
            n[cc] = (u >= 0 && yy < H && 4 * g + cc < W) ? aU + aD : 0;
            if (n[cc]) {
                lo = min(lo, s[cc]);
                hi = max(hi, s[cc] + n[cc]);
            }
        }
        // rows the NEXT step adds: [loaded, loaded + TS), and the next tile's arm bytes; all unconditional (clamped rows)
        f4 pre[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) pre[k] = nt_load4(in + (size_t)min(loaded + LB * k + lrow, H - 1) * rstride + dd);
        nxt = load_arm_words(armU, armD, max(u + NTP, 0) * 16 + 4 * rt + (l & 3), g, H, W);
        if (u >= 0 && u < nT) {
            lo = wave_min_i(lo == 0x7fffffff ? lo : lo - 4 * rt);
            hi = wave_max_i(hi == -0x7fffffff ? hi : hi - 4 * rt);
            f4 acc[4] = {zero4, zero4, zero4, zero4};
            if (hi > lo && !(dbg & 1)) {
                const int K0 = lo & ~3; // multiple of 4 (two's complement floor), as R is: a 4-row read never wraps
                const int n_it = (hi - K0 + 3) >> 2;
                int tt[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) tt[cc] = K0 + 4 * rt + dq - s[cc];
                int sl = slot_rd + (K0 - y0) + 4 * rt; // K0 - y0 in [-usd - 15, 12]
                if (sl < 0) sl += R;
                if (sl >= R) sl -= R;
                const f4 *p = ring + sl * 16 + dd;
                const f4 *const pend = ring + R * 16 + dd; // this lane's address one ring length on
                f4 v0 = p[0], v1 = p[16], v2 = p[32], v3 = p[48];
                f4 w0, w1, w2, w3;
                // one iteration = 4 window rows: prefetch the next four rows into (N0..N3), mask VALU, 16 MFMAs on (C0..C3)
#define STM_V_ITER(C0, C1, C2, C3, N0, N1, N2, N3)                                                          \
    {                                                                                                       \
        p += 64;                                                                                            \
        if (p >= pend) p -= R * 16;                                                                         \
        N0 = p[0]; N1 = p[16]; N2 = p[32]; N3 = p[48];                                                      \
        const float m0 = ((unsigned)tt[0] < (unsigned)n[0]) ? 1.0f : 0.0f;                                  \
        const float m1 = ((unsigned)tt[1] < (unsigned)n[1]) ? 1.0f : 0.0f;                                  \
        const float m2 = ((unsigned)tt[2] < (unsigned)n[2]) ? 1.0f : 0.0f;                                  \
        const float m3 = ((unsigned)tt[3] < (unsigned)n[3]) ? 1.0f : 0.0f;                                  \
        tt[0] += 4; tt[1] += 4; tt[2] += 4; tt[3] += 4;                                                     \
        acc[0] = STM_MFMA(m0, C0.x, acc[0], 0); acc[1] = STM_MFMA(m1, C0.y, acc[1], 0);                     \
        acc[2] = STM_MFMA(m2, C0.z, acc[2], 0); acc[3] = STM_MFMA(m3, C0.w, acc[3], 0);                     \
        acc[0] = STM_MFMA(m0, C1.x, acc[0], 1); acc[1] = STM_MFMA(m1, C1.y, acc[1], 1);                     \
        acc[2] = STM_MFMA(m2, C1.z, acc[2], 1); acc[3] = STM_MFMA(m3, C1.w, acc[3], 1);                     \
        acc[0] = STM_MFMA(m0, C2.x, acc[0], 2); acc[1] = STM_MFMA(m1, C2.y, acc[1], 2);                     \
        acc[2] = STM_MFMA(m2, C2.z, acc[2], 2); acc[3] = STM_MFMA(m3, C2.w, acc[3], 2);                     \
        acc[0] = STM_MFMA(m0, C3.x, acc[0], 3); acc[1] = STM_MFMA(m1, C3.y, acc[1], 3);                     \
        acc[2] = STM_MFMA(m2, C3.z, acc[2], 3); acc[3] = STM_MFMA(m3, C3.w, acc[3], 3);                     \
    }
                for (int it = 0; it + 1 < n_it; it += 2) {
                    STM_V_ITER(v0, v1, v2, v3, w0, w1, w2, w3)
                    STM_V_ITER(w0, w1, w2, w3, v0, v1, v2, v3)
                }
                if (n_it & 1) STM_V_ITER(v0, v1, v2, v3, w0, w1, w2, w3)
#undef STM_V_ITER
            }
            // register i of chain cc = out[row y0 + 4 rt + i][column 4 g + cc][hypothesis 16 c + dd]
            int so = slot_wr + 4 * rt; // ring 2's slot of the output rows (first pass only)
            if (so >= R2) so -= R2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = y0 + 4 * rt + i;
                const f4 o = {acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
                if (r < H) {
                    if (wv) nt_store4(out + (size_t)r * rstride + dd, o);
                    else ring2[(so + i) * 16 + dd] = o;
                }
            }
            slot_rd += PV_TS;
            if (slot_rd >= R) slot_rd -= R;
            slot_wr += PV_TS;
            if (slot_wr >= R2) slot_wr -= R2;
        }
        __syncthreads(); // everyone is done reading ring 1: the oldest TS rows can be replaced
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = loaded + LB * k + lrow;
            int sw = slot_ld + LB * k + lrow;
            if (sw >= R1) sw -= R1;
            if (r < H) ring1[sw * 16 + dd] = pre[k];
        }
        loaded += PV_TS; // keeps advancing past H so that later steps load nothing
        slot_ld += PV_TS;
        if (slot_ld >= R1) slot_ld -= R1;
    }
}

// ------------------------------------------------------------------ launchers
size_t pq_volume_floats(int D, int H, int W) { return (size_t)((D + 15) / 16) * H * ((W + 3) / 4) * 64; }

// cost -> H -> V, V -> H + WTA for both views of a frame.  vol_a / vol_b: two PQ volumes per view (pq_volume_floats each).
void launch_aggm_frame(const uint32_t *const *pk, const uint32_t *const *cen, const float *lut, float *const *vol_a, float *const *vol_b,
                       const u8 *const *armU, const u8 *const *armD, const u8 *const *armL, const u8 *const *armR, float *const *disp,
                       int D, int zd, int H, int W, int usd)
{
    PQViews v;
    for (int i = 0; i < 2; ++i) {
        v.pk[i] = pk[i]; v.cen[i] = cen[i]; v.a[i] = vol_a[i]; v.b[i] = vol_b[i];
        v.armU[i] = armU[i]; v.armD[i] = armD[i]; v.armL[i] = armL[i]; v.armR[i] = armR[i]; v.disp[i] = disp[i];
    }
    const int G = (W + 3) / 4, NC = (D + 15) / 16;
    if (usd > 255) usd = 255;
    {
        ProfScope p("pq_cost");
        int pad = zd > D - 1 - zd ? zd : D - 1 - zd;
        if (pad < 0) pad = 0;
        const size_t smem = (size_t)(2 * PC_TX + 2 * (PC_TX + 2 * pad) + 768 + 72) * 4;
        allow_lds_m((const void *)stm_k_pq_cost, smem);
        hipLaunchKernelGGL(stm_k_pq_cost, dim3(cdiv(W, PC_TX), H, 2), dim3(PC_TX), smem, stream(), v, lut, D, zd, H, W, G, NC, pad);
        STM_CHECK_LAUNCH();
    }
    constexpr int NW = 8;
    const int HG = (usd + 3) / 4, NG = 4 * NW + 2 * HG; // halo groups: ceil(usd / 4) on either side of a segment
    const int nseg = cdiv(W, 16 * NW), nblk = ((nseg * H * 2 + 7) / 8) * 8;
    const size_t smem_h = (size_t)4 * NG * 256 + 16 * NW * 4;
    {
        ProfScope p("pq_h");
        allow_lds_m((const void *)stm_k_pq_h<NW, false>, smem_h);
        hipLaunchKernelGGL((stm_k_pq_h<NW, false>), dim3(nblk), dim3(64 * NW), smem_h, stream(), v, D, zd, H, W, G, NC, HG, nseg);
        STM_CHECK_LAUNCH();
    }
    {
        ProfScope p("pq_v12");
        constexpr int NTP = 3, TS = 16 * NTP;
        const int R1 = (TS + 2 * usd + 3) & ~3;
        const int LAG = (usd > 1 ? (usd - 1 + TS - 1) / TS : 0) + 1; // the second pass may use first-pass rows of EARLIER steps only
        const int R2 = (TS * (LAG + 1) + usd + 3) & ~3;
        const size_t smem = (size_t)(R1 + R2) * 256;
        allow_lds_m((const void *)stm_k_pq_v12<NTP>, smem);
        hipLaunchKernelGGL(stm_k_pq_v12<NTP>, dim3(G, NC, 2), dim3(128 * NTP), smem, stream(), v, H, W, G, NC, usd, R1, R2, LAG,
                           (agg_variant() / 100000) % 10);
        STM_CHECK_LAUNCH();
    }
    {
        ProfScope p("pq_hw");
        allow_lds_m((const void *)stm_k_pq_h<NW, true>, smem_h);
        hipLaunchKernelGGL((stm_k_pq_h<NW, true>), dim3(nblk), dim3(64 * NW), smem_h, stream(), v, D, zd, H, W, G, NC, HG, nseg);
        STM_CHECK_LAUNCH();
    }
}

} // namespace stm
