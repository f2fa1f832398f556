// stm_hwin.h -- the horizontal window table of stm_k_pq_hsr (stm_kernels_aggh.hip): record layout and the routine that builds
// the records of the four 16-pixel tiles a wave's 64 lanes cover.  Shared by the stand-alone table kernel and stm_k_cross_arms,
// which has the arms in registers when it finishes (one pass over the arm planes and one launch less per frame).
#pragma once
#include "stm_common.h"

namespace stm {

constexpr int HR_TOP = 36;  // a tile's sweep range starts this many pixels left of the tile: pixels [16 t - 36, 16 t + 52)
constexpr int HR_NG = 22;   // groups of four pixels in that range
constexpr int HR_REC = 64;  // dwords per table record: [0] first group of the sweep inside the range, [1] its groups n, [8 + 2 p ..] mask of the group that block p of
                            // the kernel's 22 sweep blocks handles (the sweep is blocks 22 - n .. 21)

template <int N> __device__ __forceinline__ int hr_row_ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, false); }

template <int N> __device__ __forceinline__ uint32_t hr_row_shr(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, false); }

// lane l of the wave = pixel x0 + l of image row y (x0 a multiple of 64), aL / aR its horizontal arms (any value for x >= W).
// A DPP row of 16 lanes per tile.  A window toggles its pixel's bit at its first step and at the step after its last; the
// running XOR over the steps gives every step's set of pixels, and four consecutive steps are one group's 64-bit mask
// (d_ca_cross_sum.cu:277-289).  ev: 4 x 96 dwords of LDS owned by this wave.  `tab`: records of this view.
static __device__ __forceinline__ void hwin_build(uint32_t *__restrict__ tab, uint32_t *ev4, int y, int x0, int aL, int aR, int W, int nTx)
{
    const int l = threadIdx.x & 63, r = l >> 4, m = l & 15;
    const int x = x0 + l, t = x >> 4;
    uint32_t *ev = ev4 + r * 96;
    int s0 = 0, nn = 0;
    if (x < W) { // (a tile past the row: no windows, nothing is stored)
        s0 = x - aL;
        nn = aL + aR;
    }
    int lo = nn ? s0 : 0x7fffffff, hi = nn ? s0 + nn : -0x7fffffff; // over the tile's 16 lanes; every lane gets the result
    lo = min(lo, hr_row_ror<8>(lo)); lo = min(lo, hr_row_ror<4>(lo)); lo = min(lo, hr_row_ror<2>(lo)); lo = min(lo, hr_row_ror<1>(lo));
    hi = max(hi, hr_row_ror<8>(hi)); hi = max(hi, hr_row_ror<4>(hi)); hi = max(hi, hr_row_ror<2>(hi)); hi = max(hi, hr_row_ror<1>(hi));
    const int R0 = 16 * t - HR_TOP; // a multiple of 4
    const int K0 = hi > lo ? (lo & ~3) : 0, n = hi > lo ? (hi - K0 + 3) >> 2 : 0; // n <= 22
    const int q0 = n ? (K0 - R0) >> 2 : 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) ev[6 * m + i] = 0u;
    __builtin_amdgcn_wave_barrier();
    if (nn) {
        atomicXor(&ev[s0 - K0], 1u << m);
        atomicXor(&ev[s0 + nn - K0], 1u << m);
    }
    __builtin_amdgcn_wave_barrier();
    // lane m scans steps 6 m .. 6 m + 5, then the lanes' totals are scanned along the row
    uint32_t e[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = ev[6 * m + i];
#pragma unroll
    for (int i = 1; i < 6; ++i) e[i] ^= e[i - 1];
    uint32_t tot = e[5];
    tot ^= hr_row_shr<1>(tot); tot ^= hr_row_shr<2>(tot); tot ^= hr_row_shr<4>(tot); tot ^= hr_row_shr<8>(tot);
    const uint32_t before = hr_row_shr<1>(tot); // XOR of all steps in front of this lane's six
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 6; ++i) ev[6 * m + i] = e[i] ^ before; // the step's set of pixels
    __builtin_amdgcn_wave_barrier();
    if (t >= nTx) return;
    uint32_t *dst = tab + ((size_t)y * nTx + t) * HR_REC;
    if (m < 2) dst[m] = m == 0 ? (uint32_t)q0 : (uint32_t)n;
#pragma unroll
    for (int j = m; j < HR_NG; j += 16)
        if (j < n) {
            const unsigned long long mk = (unsigned long long)ev[4 * j] | ((unsigned long long)ev[4 * j + 1] << 16) |
                                          ((unsigned long long)ev[4 * j + 2] << 32) | ((unsigned long long)ev[4 * j + 3] << 48);
            *(unsigned long long *)(dst + 8 + 2 * (HR_NG - n + j)) = mk; // the sweep's last group in slot 21
        }
}

// ---- the vertical window table (stm_k_vwin_table, stm_kernels_aggm.hip; layout: DESIGN.md section 3) -- one record, one wave.
// Lane l = (column b = l / 16 of the group, row i = l % 16 of the tile); s0 / nn = first row and length of the lane's window
// [y - armU, y + armD) (nn = 0: no window).  ev: VT_EV (or, top >= 0: 96) 64-bit slots of LDS owned by this wave.
// top >= 0: the static layout of stm_k_pq_v12r ([0] first quad of the sweep inside the tile's range, masks of range quad J at 8 + 8 J).
__device__ __forceinline__ int hw_wave_min_i(int v)
{
    v = min(v, hr_row_ror<8>(v)); v = min(v, hr_row_ror<4>(v)); v = min(v, hr_row_ror<2>(v)); v = min(v, hr_row_ror<1>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int hw_wave_max_i(int v)
{
    v = max(v, hr_row_ror<8>(v)); v = max(v, hr_row_ror<4>(v)); v = max(v, hr_row_ror<2>(v)); v = max(v, hr_row_ror<1>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
static __device__ __forceinline__ void vwin_build(uint32_t *__restrict__ dst, unsigned long long *ev, int u, int top, int s0, int nn)
{
    const int l = threadIdx.x & 63;
    const int lo = hw_wave_min_i(nn ? s0 : 0x7fffffff);
    const int hi = hw_wave_max_i(nn ? s0 + nn : -0x7fffffff);
    const int K0 = hi > lo ? (lo & ~3) : 0, n_it = hi > lo ? (hi - K0 + 3) >> 2 : 0;
    const int q0 = top >= 0 && n_it ? (K0 - (16 * u - top)) >> 2 : 0;
    if (l < 8) dst[l] = l == 0 ? (uint32_t)(top >= 0 ? q0 : K0) : l == 1 ? (uint32_t)n_it : 0u;
    unsigned long long *mk = (unsigned long long *)(dst + 8) + 4 * q0;
    const int steps = 4 * n_it;
    for (int j = l; j <= steps; j += 64) ev[j] = 0ull;
    __builtin_amdgcn_wave_barrier();
    if (nn) { // the window [s0, s0 + nn) lies inside [K0, K0 + steps]
        atomicXor(&ev[s0 - K0], 1ull << l);
        atomicXor(&ev[s0 + nn - K0], 1ull << l);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long carry = 0ull;
    for (int base = 0; base < steps; base += 64) {
        unsigned long long e = base + l < steps ? ev[base + l] : 0ull;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(e, o);
            if (l >= o) e ^= t;
        }
        e ^= carry;
        if (base + l < steps) mk[base + l] = e;
        carry = __shfl(e, 63);
    }
}

} // namespace stm
