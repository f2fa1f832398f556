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
#include "stm_hwin.h"

namespace stm {

// ------------------------------------------------------------------ cross arms
// Colour tests on "wide" pixels: B | G << 10 | R << 20, i.e. three 10-bit fields of which bit 9 is a guard bit.
// For a threshold t in [-1, 255], maxdiff(c, a) <= t  <=>  for every channel c + t >= a and a + t >= c.  With
// TG = (512 + t) in every field, the fields of (c + TG) - a and of (a + TG) - c are c_i + 512 + t - a_i and
// a_i + 512 + t - c_i: always inside [256, 1022], so fields never borrow from or carry into each other, and a
// field's guard bit is set exactly when its inequality holds.  One add, two subtracts, two ANDs and a compare test
// all three channels in both directions; the per-channel version took three masks, three v_sad_u8 and a v_max3.
constexpr uint32_t W10_ONE = 1u | (1u << 10) | (1u << 20);
constexpr uint32_t W10_GUARD = 512u * W10_ONE;

__global__ __launch_bounds__(256) void stm_k_widen_px(const uint32_t *__restrict__ bgrx, uint32_t *__restrict__ wide, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = bgrx[i];
    wide[i] = (p & 0xffu) | ((p & 0xff00u) << 2) | ((p & 0xff0000u) << 4);
}

// The arms of a pixel: walk k = 1..kmax pixels from the anchor in each direction (kmax = min(usd, distance to the border): the
// reference's border test, d_ca_cross.cu:44-45, hoisted out of the loop).  The arm value is recorded BEFORE the colour
// test (SURVEY A-Q9, :47-64): near tier (k <= lsd) stops when anchor-vs-current or previous-vs-current exceeds
// lcd, far tier when anchor-vs-current exceeds ucd.  `(float)int > float` is evaluated as int > floor(float),
// which is the same predicate for every integer left-hand side.
//   tg_near / tg_far = (512 + threshold) in every field, anchor = the wide anchor pixel.
// Round 3: no per-lane early exit, four arms at once.  The round-2 loop left each lane at its first failing pixel, one arm
// after the other, and the exec-mask bookkeeping of that divergent exit cost 16 scalar instructions per step -- the kernel ran
// at 0.72 scalar instructions per clock and CU, the scalar unit's limit (profiles/r03_pmc_sq_aggm.txt).  Here
// arm = min(kmax, first failing k) is tracked with vector min / select only: k is wave-uniform, a lane that has failed (or left
// the image: its reads are clamped to its last pixel) keeps computing tests whose outcome cannot lower its arm any more, the
// four directions issue their loads together, and the whole wave leaves once every arm of every lane is final (one ballot
// per two steps).  0.184 -> 0.137 ms per frame.

struct ArmsArgs {
    const uint32_t *img[2]; // wide pixels
    u8 *up[2], *down[2], *left[2], *right[2];
    uint32_t *htab; // MODE >= 1: the horizontal window table of stm_k_pq_hsr, records of view 0 then view 1 (stm_hwin.h)
    uint32_t *vtab; // MODE 2: the vertical window table of stm_k_pq_v12r (static layout), `vrec` dwords per record, `vtop` = its range's reach
    int vrec, vtop;
};

// MODE 0: the arms.  MODE 1: + the horizontal window table (a wave = 64 pixels of a row = four of its tiles, arms in registers).
// MODE 2 (round 4; the review's item 1c): + the VERTICAL window table as well.  A block is 16 rows x 64 columns, wave w = row w of
// the tile row; the arms leave through LDS, and after one barrier wave w builds the record of the block's w-th group of four columns
// (vwin_build: lane = (column of the group, row of the tile)) -- stm_k_vwin_table's launch and its read of the arm planes are gone.
template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 1024 : 256) void stm_k_cross_arms(ArmsArgs a, uint32_t tg_far, uint32_t tg_near, int usd, int lsd, int H, int W)
{
    constexpr bool HTAB = MODE >= 1;
    const int v = blockIdx.z;
    const int wv_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xr = MODE == 2 ? blockIdx.x * 64 + (threadIdx.x & 63) : blockIdx.x * 256 + threadIdx.x;
    const int yr = MODE == 2 ? blockIdx.y * 16 + wv_ : blockIdx.y;
    if (MODE == 1 ? (xr & ~63) >= W : MODE == 0 ? xr >= W : false) return; // MODE 1: whole waves only (no block-wide barrier below); MODE 2: everybody stays for the barrier
    const int x = min(xr, W - 1), y = min(yr, H - 1); // a lane past the row / a wave past the image repeats the last pixel / row and stores nothing
    const int p = y * W + x;
    const uint32_t *__restrict__ img = a.img[v];
    const uint32_t anchor = img[p];
    // the four arms of a pixel walk together: four independent loads per step, one exit test for all of them.
    // Addresses: a block is one image row, so the row an up / down step reads is wave-uniform -- a scalar base + the lane's
    // constant byte offset, no vector arithmetic; left / right steps clamp the lane's offset into the row (two instructions).
    // Steps past an arm's border re-read the border pixel; their verdict cannot shorten the arm (it starts at kmax).
    const char *rowbase = (const char *)img + (size_t)y * W * 4;
    const uint32_t x4 = 4u * (uint32_t)x;
    const int xmax4 = 4 * (W - 1);
    const size_t rowb = (size_t)W * 4;
    const int kmax[4] = {min(usd, y), min(usd, H - 1 - y), min(usd, x), min(usd, W - 1 - x)};
#define STM_ARM_LOADS(K)                                                                                         \
    {                                                                                                            \
        uint32_t xo = x4;                                                                                        \
        asm volatile("" : "+v"(xo)); /* keeps (row base + x) from being folded into one 64-bit vector address: the row base stays scalar */ \
        c[0] = *(const uint32_t *)(rowbase - (size_t)min(K, kmax[0]) * rowb + (size_t)xo);                        \
        c[1] = *(const uint32_t *)(rowbase + (size_t)min(K, kmax[1]) * rowb + (size_t)xo);                        \
        c[2] = *(const uint32_t *)(rowbase + (size_t)(uint32_t)max((int)x4 - 4 * (K), 0));                        \
        c[3] = *(const uint32_t *)(rowbase + (size_t)(uint32_t)min((int)x4 + 4 * (K), xmax4));                    \
    }
    int arm[4] = {kmax[0], kmax[1], kmax[2], kmax[3]};
    const uint32_t anchor_n = anchor + tg_near;
    uint32_t prev[4] = {anchor, anchor, anchor, anchor}, prev_t[4] = {anchor_n, anchor_n, anchor_n, anchor_n};
    int k = 1;
    const int knear = min(usd, lsd);
    while (k <= knear) {
        if (__ballot(k <= max(max(arm[0], arm[1]), max(arm[2], arm[3]))) == 0) break;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (k <= knear) { // uniform
                uint32_t c[4];
                STM_ARM_LOADS(k)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t c_t = c[d] + tg_near;
                    const uint32_t ok = (c_t - anchor) & (anchor_n - c[d]) & (c_t - prev[d]) & (prev_t[d] - c[d]) & W10_GUARD;
                    arm[d] = min(arm[d], ok != W10_GUARD ? k : 0x7fffffff);
                    prev[d] = c[d];
                    prev_t[d] = c_t;
                }
                ++k;
            }
        }
    }
    const uint32_t far_lo = tg_far - anchor, far_hi = anchor + tg_far;
    k = max(k, knear + 1); // an early exit of the near tier means every arm is final: the far loop leaves at once
    while (k <= usd) {
        if (__ballot(k <= max(max(arm[0], arm[1]), max(arm[2], arm[3]))) == 0) break;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (k <= usd) {
                uint32_t c[4];
                STM_ARM_LOADS(k)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t ok = (c[d] + far_lo) & (far_hi - c[d]) & W10_GUARD;
                    arm[d] = min(arm[d], ok != W10_GUARD ? k : 0x7fffffff);
                }
                ++k;
            }
        }
    }
#undef STM_ARM_LOADS
    const bool inside = xr < W && yr < H;
    if (inside) {
        a.up[v][p] = (u8)arm[0];
        a.down[v][p] = (u8)arm[1];
        a.left[v][p] = (u8)arm[2];
        a.right[v][p] = (u8)arm[3];
    }
    if (HTAB) {
        __shared__ uint32_t ev_all[MODE == 2 ? 16 : 4][4 * 96];
        const int nTx = (W + 15) >> 4;
        if (yr < H) // (wave-uniform)
            hwin_build(a.htab + (size_t)v * H * nTx * HR_REC, ev_all[wv_], yr, xr & ~63, arm[2], arm[3], W, nTx);
    }
    if (MODE == 2) {
        __shared__ u8 s_u[16][64], s_d[16][64];
        __shared__ unsigned long long ev_v[16][96];
        const int l = threadIdx.x & 63;
        s_u[wv_][l] = inside ? (u8)arm[0] : (u8)0; // no window outside the image
        s_d[wv_][l] = inside ? (u8)arm[1] : (u8)0;
        __syncthreads();
        const int G = (W + 3) >> 2, nT = (H + 15) >> 4, gg = blockIdx.x * 16 + wv_, u = blockIdx.y;
        if (gg < G) { // (wave-uniform)
            const int b = l >> 4, i = l & 15;
            const int aU = s_u[i][4 * wv_ + b], aD = s_d[i][4 * wv_ + b];
            vwin_build(a.vtab + ((size_t)(v * nT + u) * G + gg) * a.vrec, ev_v[wv_], u, a.vtop, blockIdx.y * 16 + i - aU, aU + aD);
        }
    }
}

// threshold as the integer t with (int diff > threshold) <=> (diff > t), clamped to [-1, 255], times the field pattern
static uint32_t wide_threshold(float t)
{
    int ti;
    if (t != t) ti = 255;         // NaN: '>' is never true
    else if (t >= 255.f) ti = 255; // no 8-bit difference exceeds it
    else if (t < 0.f) ti = -1;     // every difference (>= 0) exceeds it
    else ti = (int)floorf(t);
    return (uint32_t)(512 + ti) * W10_ONE;
}

// nviews = 1 or 2: both views of a frame share the launch.  packed[] = BGRX planes (launch_pack_bgrx).
void launch_cross_arms2(int nviews, const uint32_t *const *packed, u8 *const *up, u8 *const *down, u8 *const *left,
                        u8 *const *right, float ucd, float lcd, int usd, int lsd, int H, int W, const uint32_t *const *wide_ready,
                        uint32_t *htab, uint32_t *vtab, int vrec, int vtop)
{
    ArmsArgs a;
    const int n = H * W;
    ProfScope p("cross_arms");
    const uint32_t *wide[2];
    for (int v = 0; v < nviews; ++v) {
        if (wide_ready) { // the caller already holds the wide planes (launch_demux_sbs_packed)
            wide[v] = wide_ready[v];
            continue;
        }
        uint32_t *w = Workspace::get<uint32_t>((size_t)n);
        STM_LAUNCH(stm_k_widen_px, dim3(cdiv(n, 256)), dim3(256), 0, stream(), packed[v], w, n);
        STM_CHECK_LAUNCH();
        wide[v] = w;
    }
    for (int v = 0; v < 2; ++v) {
        const int s = v < nviews ? v : 0;
        a.img[v] = wide[s]; a.up[v] = up[s]; a.down[v] = down[s]; a.left[v] = left[s]; a.right[v] = right[s];
    }
    if (usd > 255) usd = 255; // arms are stored as u8 (reference T2)
    a.htab = htab;
    a.vtab = vtab;
    a.vrec = vrec;
    a.vtop = vtop;
    if (htab && vtab && vtop >= 0 && usd <= HR_TOP && usd <= vtop) // both window tables (the caller asks for them when the register-ring kernels will run)
        STM_LAUNCH(stm_k_cross_arms<2>, dim3(cdiv(W, 64), cdiv(H, 16), nviews), dim3(1024), 0, stream(), a, wide_threshold(ucd),
                           wide_threshold(lcd), usd, lsd, H, W);
    else if (htab && usd <= HR_TOP) // (longer arms than the table's range: the caller does not ask for it, aggh_supports)
        STM_LAUNCH(stm_k_cross_arms<1>, dim3(cdiv(W, 256), H, nviews), dim3(256), 0, stream(), a, wide_threshold(ucd),
                           wide_threshold(lcd), usd, lsd, H, W);
    else
        STM_LAUNCH(stm_k_cross_arms<0>, dim3(cdiv(W, 256), H, nviews), dim3(256), 0, stream(), a, wide_threshold(ucd),
                           wide_threshold(lcd), usd, lsd, H, W);
    STM_CHECK_LAUNCH();
}

void launch_cross_arms(const uint32_t *packed, u8 *up, u8 *down, u8 *left, u8 *right, float ucd, float lcd, int usd,
                       int lsd, int H, int W)
{
    launch_cross_arms2(1, &packed, &up, &down, &left, &right, ucd, lcd, usd, lsd, H, W, nullptr);
}

#define STM_ACC(s, v) { s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w; }

// s += p[0] + p[S] + ... + p[(n-1)S], added strictly left to right (the reference's order,
// d_ca_cross_sum.cu:284-289).  The four LDS reads of a group are issued back to back so their latency
// overlaps; the adds stay a single dependent chain per hypothesis, which is what bit-exactness requires.
template <int S> __device__ __forceinline__ float4 window_sum(const float4 *__restrict__ p, int n, float4 s)
{
    while (n >= 4) {
        float4 v0 = p[0], v1 = p[S], v2 = p[2 * S], v3 = p[3 * S];
        p += 4 * S;
        n -= 4;
        STM_ACC(s, v0) STM_ACC(s, v1) STM_ACC(s, v2) STM_ACC(s, v3)
    }
    if (n & 2) {
        float4 v0 = p[0], v1 = p[S];
        p += 2 * S;
        STM_ACC(s, v0) STM_ACC(s, v1)
    }
    if (n & 1) {
        float4 v0 = p[0];
        STM_ACC(s, v0)
    }
    return s;
}

// dynamic LDS above the 64 KiB default needs an explicit opt-in (gfx950 has 160 KiB per CU)
static void allow_lds(const void *func, size_t bytes)
{
    if (bytes > 64 * 1024) STM_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// ------------------------------------------------------------------ horizontal pass
constexpr int AH_QPB = 4; // disparity quads per block (un-fused pass)

// One block (T threads) = one image row x `qpb` quads; a thread owns the pixels tid, tid+T, ... (at most PPT).
// LDS: float4 tile[W] | u16 arms[W] (armL | armR << 8).  While quad q is being summed out of LDS, quad q+1
// is already in flight from HBM into registers.  WTA: the running (best cost, best index) of the thread's
// pixels stay in registers across all quads; the aggregated volume is never written.
// `second`: the other view of the frame (blockIdx.z = 1); its blocks share the launch so that the tail of one view's
// blocks overlaps the other view's (the fused WTA pass is one block per image row: 1080 blocks are 1.05 waves of
// what the chip holds, two launches waste almost a whole wave each).
struct AggHView {
    Vol in, out;
    const u8 *armL, *armR;
    float *disp;
    // COST mode (first pass of the frame pipeline): the input volume is not read but computed on the fly, as
    // stm_k_cost_init would have written it -- C(d, x) = rho_ad(|own(x) - other(x')|_1) + rho_c(ham(cen_own(x), cen_other(x')))
    // with x' = clamp(x + sgn (d - zd)); sgn = +1 for the left view, -1 for the right view (SURVEY A-Q6)
    const uint32_t *pk_own, *cen_own, *pk_oth, *cen_oth;
    int sgn;
};

// popc(x & 0x7fffffff) + 33 * (x >> 31)  ==  the 64-iteration loop of d_alu.cu:7-15 (SURVEY A-Q1)
__device__ __forceinline__ int hamdist_low32(uint32_t a, uint32_t b)
{
    const uint32_t x = a ^ b;
    return __popc(x & 0x7fffffffu) + 33 * (int)(x >> 31);
}

template <bool QUAD, bool WTA, int T, int PPT, bool COST = false>
__global__ __launch_bounds__(T) void stm_k_agg_h(Vol in, Vol out, const u8 *__restrict__ armL,
                                                 const u8 *__restrict__ armR, float *__restrict__ disp,
                                                 int D, int zd, int H, int W, int qpb, AggHView first, AggHView second,
                                                 const float *__restrict__ lut_g)
{
    const AggHView &vw = blockIdx.z ? second : first;
    if (blockIdx.z) {
        in = second.in;
        out = second.out;
        armL = second.armL;
        armR = second.armR;
        disp = second.disp;
    }
    extern __shared__ float4 smem4[];
    float4 *tile = smem4;
    uint16_t *arms = (uint16_t *)(tile + W);
    const int y = blockIdx.x, tid = threadIdx.x;
    const size_t row = (size_t)y * W;
    const int nq = (D + 3) >> 2;
    const int q0 = blockIdx.y * qpb, q1 = min(q0 + qpb, nq);

    for (int x = tid; x < W; x += T) arms[x] = (uint16_t)armL[row + x] | ((uint16_t)armR[row + x] << 8);

    // COST: rho tables (766 + 65 floats, host-built: bit-identical to the CPU) behind the arms; this thread's own pixels
    float *lut_ad = (float *)(arms + ((W + 1) & ~1));
    float *lut_c = lut_ad + 768;
    uint32_t pk0[COST ? PPT : 1], cen0[COST ? PPT : 1];
    if (COST) {
        for (int i = tid; i < 768 + 65; i += T) lut_ad[i] = lut_g[i];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int x = min(tid + i * T, W - 1);
            pk0[i] = vw.pk_own[row + x];
            cen0[i] = vw.cen_own[row + x];
        }
        __syncthreads();
    }
    const uint32_t *__restrict__ pk_oth = vw.pk_oth + row, *__restrict__ cen_oth = vw.cen_oth + row;
    const int sgn = vw.sgn;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
    auto cost_quad = [&](int q, int x, uint32_t p0, uint32_t c0) {
        float v[4];
        // fast path, taken by every wave that is not at an image border: the four matched pixels are four consecutive
        // dwords, fetched with one 16-byte load per plane instead of four clamped gathers
        const int xb = x + sgn * (q * 4 - zd), xe = xb + 3 * sgn;
        const int lo = min(xb, xe);
        const bool edge = lo < 0 || lo + 3 > W - 1 || q * 4 + 3 >= D;
        if (__builtin_amdgcn_ballot_w64(edge) == 0) {
            const u32x4 pp = *(const u32x4 *)(pk_oth + lo), cc = *(const u32x4 *)(cen_oth + lo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t p1 = sgn > 0 ? pp[j] : pp[3 - j], c1 = sgn > 0 ? cc[j] : cc[3 - j];
                const int ad = (int)__builtin_amdgcn_sad_u8(p0, p1, 0u);
                const int hd = hamdist_low32(c0, c1);
                v[j] = lut_ad[ad] + lut_c[hd];
            }
            return make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = q * 4 + j;
            v[j] = 0.f;
            if (d < D) {
                const int xo = min(max(x + sgn * (d - zd), 0), W - 1); // clamp-to-edge in image coordinates
                const int ad = (int)__builtin_amdgcn_sad_u8(p0, pk_oth[xo], 0u);
                const int hd = hamdist_low32(c0, cen_oth[xo]);
                v[j] = lut_ad[ad] + lut_c[hd];
            }
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };

    float4 pre[PPT];
    float best_c[WTA ? PPT : 1];
    int best_d[WTA ? PPT : 1];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int x = tid + i * T;
        if (x < W) pre[i] = COST ? cost_quad(q0, x, pk0[COST ? i : 0], cen0[COST ? i : 0]) : load_quad<QUAD>(in, q0, D, row + x);
        if (WTA) { best_c[i] = 3.402823466e+38f; best_d[i] = 0; }
    }
    for (int q = q0; q < q1; ++q) {
        __syncthreads(); // previous quad's readers are done with the tile (first time: arms visible)
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int x = tid + i * T;
            if (x < W) tile[x] = pre[i];
        }
        __syncthreads();
        if (q + 1 < q1) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int x = tid + i * T;
                if (x < W) pre[i] = COST ? cost_quad(q + 1, x, pk0[COST ? i : 0], cen0[COST ? i : 0]) : load_quad<QUAD>(in, q + 1, D, row + x);
            }
        }
        const int d0 = q * 4;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int x = tid + i * T;
            if (x < W) {
                const uint32_t ar = arms[x];
                const int aL = (int)(ar & 0xff), n = aL + (int)(ar >> 8); // window [x - armL, x + armR)
                const float4 s = window_sum<1>(tile + (x - aL), n, make_float4(0.f, 0.f, 0.f, 0.f));
                if (WTA) {
                    // first strictly-lowest cost wins, ascending d (d_dc_wta.cu:19-34)
                    float bc = best_c[i];
                    int bd = best_d[i];
                    if (bc > s.x) { bc = s.x; bd = d0; }
                    if (d0 + 1 < D && bc > s.y) { bc = s.y; bd = d0 + 1; }
                    if (d0 + 2 < D && bc > s.z) { bc = s.z; bd = d0 + 2; }
                    if (d0 + 3 < D && bc > s.w) { bc = s.w; bd = d0 + 3; }
                    best_c[i] = bc;
                    best_d[i] = bd;
                } else {
                    store_quad<QUAD>(out, q, D, row + x, s);
                }
            }
        }
    }
    if (WTA) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int x = tid + i * T;
            if (x < W) disp[row + x] = (float)best_d[i] - (float)zd;
        }
    }
}

static size_t agg_h_smem(int W, bool cost = false) { return (size_t)W * 16 + (size_t)((W + 1) & ~1) * 2 + (cost ? (768 + 72) * 4 : 0); }

template <bool QUAD, bool WTA, int T, int PPT, bool COST = false>
static void launch_agg_h_tt(Vol in, Vol out, const u8 *armL, const u8 *armR, float *disp, int D, int zd, int H, int W, int qpb,
                            const AggHView *first, const AggHView *second, const float *lut)
{
    int nq = (D + 3) / 4;
    size_t smem = agg_h_smem(W, COST);
    allow_lds((const void *)stm_k_agg_h<QUAD, WTA, T, PPT, COST>, smem);
    AggHView none{in, out, armL, armR, disp, nullptr, nullptr, nullptr, nullptr, 0};
    STM_LAUNCH((stm_k_agg_h<QUAD, WTA, T, PPT, COST>), dim3(H, cdiv(nq, qpb), second ? 2 : 1), dim3(T), smem, stream(), in,
                       out, armL, armR, disp, D, zd, H, W, qpb, first ? *first : none, second ? *second : none, lut);
    STM_CHECK_LAUNCH();
}

// picks (threads, pixels per thread) so that T * PPT >= W with the smallest register footprint
template <bool QUAD, bool WTA, bool COST = false>
static void launch_agg_h_t(Vol in, Vol out, const u8 *armL, const u8 *armR, float *disp, int D, int zd, int H, int W, int qpb,
                           const AggHView *second = nullptr, const AggHView *first = nullptr, const float *lut = nullptr)
{
    const int hv = (agg_variant() / 100) % 10;
    if (W > 8192) {
        fail("aggregation: num_cols > 8192 is not supported by the row-tile kernel", "W", __FILE__, __LINE__);
        return; // only reached in error mode 1 (record and return)
    }
    if (hv == 1 && W <= 2048) launch_agg_h_tt<QUAD, WTA, 256, 8, COST>(in, out, armL, armR, disp, D, zd, H, W, qpb, first, second, lut);
    else if (hv == 2 && W <= 2048) launch_agg_h_tt<QUAD, WTA, 1024, 2, COST>(in, out, armL, armR, disp, D, zd, H, W, qpb, first, second, lut);
    else if (W <= 1024) launch_agg_h_tt<QUAD, WTA, 256, 4, COST>(in, out, armL, armR, disp, D, zd, H, W, qpb, first, second, lut);
    else if (W <= 2048) launch_agg_h_tt<QUAD, WTA, 512, 4, COST>(in, out, armL, armR, disp, D, zd, H, W, qpb, first, second, lut);
    else if (W <= 4096) launch_agg_h_tt<QUAD, WTA, 1024, 4, COST>(in, out, armL, armR, disp, D, zd, H, W, qpb, first, second, lut);
    else launch_agg_h_tt<QUAD, WTA, 1024, 8, COST>(in, out, armL, armR, disp, D, zd, H, W, qpb, first, second, lut);
}

void launch_agg_h(Vol in, Vol out, const u8 *armL, const u8 *armR, int D, int H, int W)
{
    ProfScope p("agg_h");
    if (in.quad) launch_agg_h_t<true, false>(in, out, armL, armR, nullptr, D, 0, H, W, AH_QPB);
    else launch_agg_h_t<false, false>(in, out, armL, armR, nullptr, D, 0, H, W, AH_QPB);
}

// both views of a frame in one launch
void launch_agg_h2(Vol in_a, Vol out_a, const u8 *armL_a, const u8 *armR_a, Vol in_b, Vol out_b, const u8 *armL_b, const u8 *armR_b,
                   int D, int H, int W)
{
    ProfScope p("agg_h");
    const AggHView second{in_b, out_b, armL_b, armR_b, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (in_a.quad) launch_agg_h_t<true, false>(in_a, out_a, armL_a, armR_a, nullptr, D, 0, H, W, AH_QPB, &second);
    else launch_agg_h_t<false, false>(in_a, out_a, armL_a, armR_a, nullptr, D, 0, H, W, AH_QPB, &second);
}

// First pass of the frame pipeline, both views in one launch, with the cost volume computed on the fly instead of read
// (the frame never writes or re-reads the 2 V of initial costs).  lut = rho tables: [0..765] AD, [768..832] census.
void launch_agg_h2_cost(const uint32_t *pk_l, const uint32_t *cen_l, const uint32_t *pk_r, const uint32_t *cen_r, const float *lut,
                        Vol out_l, const u8 *armL_l, const u8 *armR_l, Vol out_r, const u8 *armL_r, const u8 *armR_r, int D, int zd,
                        int H, int W)
{
    ProfScope p("agg_h");
    const AggHView first{out_l, out_l, armL_l, armR_l, nullptr, pk_l, cen_l, pk_r, cen_r, 1};
    const AggHView second{out_r, out_r, armL_r, armR_r, nullptr, pk_r, cen_r, pk_l, cen_l, -1};
    launch_agg_h_t<true, false, true>(out_l, out_l, armL_l, armR_l, nullptr, D, zd, H, W, AH_QPB, &second, &first, lut);
}

// last horizontal pass fused with WTA: the aggregated volume is consumed in LDS and never written
// both views of a frame in one launch (same layout for both volumes)
void launch_agg_h_wta2(Vol in_a, const u8 *armL_a, const u8 *armR_a, float *disp_a, Vol in_b, const u8 *armL_b, const u8 *armR_b,
                       float *disp_b, int D, int zd, int H, int W)
{
    int nq = (D + 3) / 4;
    Vol none = vol_slab(nullptr, 0);
    const AggHView second{in_b, none, armL_b, armR_b, disp_b, nullptr, nullptr, nullptr, nullptr, 0};
    ProfScope p("agg_hw");
    if (in_a.quad) launch_agg_h_t<true, true>(in_a, none, armL_a, armR_a, disp_a, D, zd, H, W, nq, &second);
    else launch_agg_h_t<false, true>(in_a, none, armL_a, armR_a, disp_a, D, zd, H, W, nq, &second);
}

// ------------------------------------------------------------------ vertical pass
// One block = TX columns x a band of rows x one quad; TX x TY threads, every thread owns OPT rows of each
// step (CH = TY * OPT output rows per step).  Rows stream top to bottom through an LDS ring of
// R = roundup(2 usd, CH) + CH rows (row r lives in slot r % R; R is a multiple of CH and every band starts
// at a multiple of CH, so slots advance without any division).  A window wraps around the ring at most
// once and is summed as two linear segments.  The CH rows of the next step are fetched from HBM into
// registers while the current step is summed; a step costs two barriers, amortised over CH rows.
template <bool QUAD, int TX, int TY, int OPT>
__global__ __launch_bounds__(TX *TY) void stm_k_agg_v(Vol in, Vol out, const u8 *__restrict__ armU,
                                                      const u8 *__restrict__ armD, int D, int H, int W, int usd, int R,
                                                      int band, int nstrips, int nbands, int nq, int xcd_map)
{
    constexpr int CH = TY * OPT;
    extern __shared__ float4 ring[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    // block -> (strip, band, quad).  XCD-aware order: blocks are dealt round-robin over the 8 XCDs (b % 8 labels
    // the XCD group), so the blocks one XCD sees, b = xcd, xcd+8, ..., walk the quads of ONE (strip, band) tile
    // before moving to the next tile: the tile's arm bytes are fetched into that XCD's L2 once and reused by the
    // other nq-1 quads.  Placement only affects speed, never results.
    int strip, bandi, q;
    {
        const int b = blockIdx.x;
        int tile;
        if (xcd_map) {
            const int xcd = b & 7, i = b >> 3;
            q = i % nq;
            tile = (i / nq) * 8 + xcd;
        } else {
            tile = b % (nstrips * nbands);
            q = b / (nstrips * nbands);
        }
        if (tile >= nstrips * nbands) return; // padding blocks of the last group of 8 tiles
        strip = tile % nstrips;
        bandi = tile / nstrips;
    }
    const int x = strip * TX + tx;
    const int yb0 = bandi * band, yb1 = min(yb0 + band, H);
    const bool xin = x < W;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // initial fill: rows [first, yb0 + usd - 1); every step then adds the CH rows it prefetched
    const int first = max(yb0 - usd, 0);
    const int slot_first = first % R;
    int loaded = min(yb0 + usd - 1, H);
    if (loaded < first) loaded = first;
    for (int r = first + ty; r < loaded; r += TY) {
        int s = slot_first + (r - first);
        if (s >= R) s -= R;
        ring[s * TX + tx] = xin ? load_quad<QUAD>(in, q, D, (size_t)r * W + x) : zero4;
    }
    int slot_loaded = slot_first + (loaded - first);
    if (slot_loaded >= R) slot_loaded -= R;
    int slot_y0 = yb0 % R;

    float4 pre[OPT];
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        const int r = loaded + ty + i * TY;
        pre[i] = (xin && r < H) ? load_quad<QUAD>(in, q, D, (size_t)r * W + x) : zero4;
    }
    for (int y0 = yb0; y0 < yb1; y0 += CH) {
        __syncthreads(); // everyone finished the previous step before its oldest rows are overwritten
#pragma unroll
        for (int i = 0; i < OPT; ++i) {
            const int r = loaded + ty + i * TY;
            int s = slot_loaded + ty + i * TY;
            if (s >= R) s -= R;
            if (r < H) ring[s * TX + tx] = pre[i];
        }
        loaded += CH;
        slot_loaded += CH;
        if (slot_loaded >= R) slot_loaded -= R;
        __syncthreads();
        if (y0 + CH < yb1) { // next step's rows, in flight during this step's sums
#pragma unroll
            for (int i = 0; i < OPT; ++i) {
                const int r = loaded + ty + i * TY;
                if (xin && r < H) pre[i] = load_quad<QUAD>(in, q, D, (size_t)r * W + x);
            }
        }
#pragma unroll
        for (int i = 0; i < OPT; ++i) {
            const int y = y0 + ty + i * TY;
            if (xin && y < yb1) {
                const size_t p = (size_t)y * W + x;
                const int aU = (int)armU[p], n = aU + (int)armD[p]; // window [y - armU, y + armD)
                int sa = slot_y0 + ty + i * TY - aU;
                if (sa < 0) sa += R;
                const int n1 = min(n, R - sa);
                float4 s = window_sum<TX>(ring + sa * TX + tx, n1, zero4);
                s = window_sum<TX>(ring + tx, n - n1, s);
                store_quad<QUAD>(out, q, D, p, s);
            }
        }
        slot_y0 += CH;
        if (slot_y0 >= R) slot_y0 -= R;
    }
}

template <bool QUAD, int TX, int TY, int OPT>
static void launch_agg_v_t(Vol in, Vol out, const u8 *armU, const u8 *armD, int D, int H, int W, int usd, int nbands)
{
    constexpr int CH = TY * OPT;
    int nq = (D + 3) / 4;
    int R = (2 * usd + CH - 1) / CH * CH + CH;
    size_t smem = (size_t)R * TX * 16;
    int band = (cdiv(H, nbands) + CH - 1) / CH * CH; // a few bands: +usd rows of halo each, more blocks in flight
    if (band < CH) band = CH;
    allow_lds((const void *)stm_k_agg_v<QUAD, TX, TY, OPT>, smem);
    const int nstrips = cdiv(W, TX), nb = cdiv(H, band);
    const int xcd_map = (agg_variant() / 1000) % 10 == 1 ? 0 : 1;
    const int ntiles8 = cdiv(nstrips * nb, 8) * 8;
    STM_LAUNCH((stm_k_agg_v<QUAD, TX, TY, OPT>), dim3((unsigned)ntiles8 * nq), dim3(TX, TY), smem, stream(), in, out,
                       armU, armD, D, H, W, usd, R, band, nstrips, nb, nq, xcd_map);
    STM_CHECK_LAUNCH();
}

void launch_agg_v(Vol in, Vol out, const u8 *armU, const u8 *armD, int D, int H, int W, int usd)
{
    ProfScope p("agg_v");
    const int v = agg_variant() % 100;
    const int nbands = (v / 10) ? (v / 10) : 3;
    if (in.quad) {
        switch (v % 10) {
        default: launch_agg_v_t<true, 16, 32, 1>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 1: launch_agg_v_t<true, 32, 8, 2>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 2: launch_agg_v_t<true, 32, 8, 4>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 3: launch_agg_v_t<true, 16, 16, 2>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 4: launch_agg_v_t<true, 16, 16, 4>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 5: launch_agg_v_t<true, 64, 4, 4>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 6: launch_agg_v_t<true, 16, 16, 1>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 7: launch_agg_v_t<true, 16, 8, 1>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 8: launch_agg_v_t<true, 16, 32, 1>(in, out, armU, armD, D, H, W, usd, nbands); break;
        case 9: launch_agg_v_t<true, 8, 32, 1>(in, out, armU, armD, D, H, W, usd, nbands); break;
        }
    } else {
        launch_agg_v_t<false, 32, 16, 1>(in, out, armU, armD, D, H, W, usd, nbands); // 128-B row segments per plane
    }
}

// ------------------------------------------------------------------ WTA (un-fused, per-stage API)
__global__ __launch_bounds__(256) void stm_k_wta(Vol cost, float *__restrict__ disp, int D, int zd, size_t HW)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    float lowest = 3.402823466e+38f;
    int best = 0;
    if (cost.quad) {
        for (int q = 0; q * 4 < D; ++q) {
            float4 c = ((const float4 *)cost.base)[(size_t)q * cost.plane_stride + p];
            if (lowest > c.x) { lowest = c.x; best = q * 4; }
            if (q * 4 + 1 < D && lowest > c.y) { lowest = c.y; best = q * 4 + 1; }
            if (q * 4 + 2 < D && lowest > c.z) { lowest = c.z; best = q * 4 + 2; }
            if (q * 4 + 3 < D && lowest > c.w) { lowest = c.w; best = q * 4 + 3; }
        }
    } else {
        for (int d = 0; d < D; ++d) {
            float c = cost.plane(d)[p];
            if (lowest > c) { lowest = c; best = d; }
        }
    }
    disp[p] = (float)best - (float)zd;
}

void launch_wta(Vol cost, float *disp, int D, int zd, int H, int W)
{
    size_t HW = (size_t)H * W;
    ProfScope p("wta");
    STM_LAUNCH(stm_k_wta, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream(), cost, disp, D, zd, HW);
    STM_CHECK_LAUNCH();
}

} // namespace stm
