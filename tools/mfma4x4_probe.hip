// v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4, K = 1; 2 passes): layout with CBSZ / ABID, and issue rate of dependent chains.
// The masked-add sweeps use v_mfma_f32_16x16x1_4b_f32 (a tile of 16 pixels per instruction, 8 passes).  The 4 x 4 form does a quarter
// of the work per instruction at the same MACs per pass, so a tile of FOUR pixels could sweep only ITS windows' union.
//   F = 0: one dependent chain; 1: two independent accumulators alternating; 2: four; 3: one chain, per 16 MFMAs one v_cndmask
//   (the A operand of the next 16), per 4 MFMAs the scalar ring-index update, B through the VGPR index mode
// hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o /tmp/m4 && /tmp/m4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_layout(float *out, int abid_sel)
{
    const int l = threadIdx.x;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    const float a = (float)(l + 1), b = 1000.f * (float)(l + 1);
    if (abid_sel == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:0\n\ts_nop 7" : "+v"(acc) : "v"(a), "v"(b));
    else if (abid_sel == 5) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:5\n\ts_nop 7" : "+v"(acc) : "v"(a), "v"(b));
    else if (abid_sel == 100) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:2 abid:1\n\ts_nop 7" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\ts_nop 7" : "+v"(acc) : "v"(a), "v"(b));
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}

template <int F> __global__ __launch_bounds__(256) void k_rate(float *out, int iters, float b, unsigned long long m)
{
    f4v c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    float a0 = 1.f, a1 = 1.f;
    const float b0 = b;
    int ridx = 0;
    if (F >= 3) asm volatile("s_set_gpr_idx_on %0, 0x2" : : "s"(0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { // 4 x 16 MFMAs per iteration
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (F == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c0) : "v"(a0), "v"(b0), "n"(k));
                if (F == 1) {
                    if (k & 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c1) : "v"(a0), "v"(b0), "n"(k));
                    else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c0) : "v"(a0), "v"(b0), "n"(k));
                }
                if (F == 2) {
                    if ((k & 3) == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c0) : "v"(a0), "v"(b0), "n"(k));
                    if ((k & 3) == 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c1) : "v"(a0), "v"(b0), "n"(k));
                    if ((k & 3) == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c2) : "v"(a0), "v"(b0), "n"(k));
                    if ((k & 3) == 3) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c3) : "v"(a0), "v"(b0), "n"(k));
                }
                if (F >= 4) { // blocks of 4 (F = 4, 6) or 8 (F = 5) MFMAs, A shared by four MFMAs (cbsz:2), one v_cndmask per four
                    const int blk = F == 5 ? 8 : 4;
                    if (F == 6 && (k % blk) == 0) asm volatile("s_set_gpr_idx_idx %0" : : "s"(0));
                    if ((k >> 2) & 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:2 abid:%3" : "+v"(c0) : "v"(a1), "v"(b0), "n"(k & 3));
                    else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:2 abid:%3" : "+v"(c0) : "v"(a0), "v"(b0), "n"(k & 3));
                    if ((k & 3) == 0) {
                        if ((k >> 2) & 1) asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(a0) : "s"(m));
                        else asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(a1) : "s"(m));
                    }
                    if ((k % blk) == 1) {
                        if (F == 6) asm volatile("s_add_u32 %0, %0, 4\n\ts_cmp_eq_u32 %0, 88\n\ts_cselect_b32 %0, 0, %0" : "+s"(ridx) : : "scc");
                        else asm volatile("s_add_u32 m0, m0, 0\n\ts_cmp_eq_u32 m0, 88\n\ts_cselect_b32 m0, 0x2000, m0" : : : "scc");
                    }
                }
                if (F == 3) {
                    if ((k & 3) == 0) asm volatile("s_set_gpr_idx_idx %0" : : "s"(0));
                    if (q & 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c0) : "v"(a1), "v"(b0), "n"(k));
                    else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3" : "+v"(c0) : "v"(a0), "v"(b0), "n"(k));
                    if (k == 1) {
                        if (q & 1) asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(a0) : "s"(m));
                        else asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(a1) : "s"(m));
                    }
                    if ((k & 3) == 2) asm volatile("s_add_u32 %0, %0, 4\n\ts_cmp_eq_u32 %0, 88\n\ts_cselect_b32 %0, 0, %0" : "+s"(ridx) : : "scc");
                }
            }
        }
    }
    if (F >= 3) asm volatile("s_set_gpr_idx_off");
    asm volatile("s_nop 15" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    float s = (float)ridx;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int F> static void run(float *dout, const char *what)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int wps = 1; wps <= 4; ++wps) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_rate<F>, dim3(256 * wps), dim3(256), 0, 0, dout, iters, 0.5f, 0x5555555555555555ull);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-70s waves/SIMD %d: %.2f cycles per MFMA per SIMD at 2.4 GHz\n", what, wps, best * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
    }
}

int main()
{
    float *dout;
    CK(hipMalloc(&dout, 1024 * 256 * 4));
    float h[256];
    const int sels[4] = {-1, 0, 5, 100};
    for (int s = 0; s < 4; ++s) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dout, sels[s]);
        CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
        printf("layout, %s: A lane L = L + 1, B lane L = 1000 (L + 1); acc[i] of lanes 0, 1, 5, 22, 63:\n", s == 0 ? "no broadcast" : s == 1 ? "cbsz:4 abid:0" : s == 2 ? "cbsz:4 abid:5" : "cbsz:2 abid:1");
        const int ls[5] = {0, 1, 5, 22, 63};
        for (int k = 0; k < 5; ++k) printf("   lane %2d: %9.0f %9.0f %9.0f %9.0f\n", ls[k], h[ls[k] * 4], h[ls[k] * 4 + 1], h[ls[k] * 4 + 2], h[ls[k] * 4 + 3]);
    }
    run<0>(dout, "one dependent chain");
    run<1>(dout, "two independent accumulators, alternating");
    run<2>(dout, "four independent accumulators");
    run<3>(dout, "one chain + cndmask per 16 + ring index update per 4, index mode");
    run<6>(dout, "cbsz:2, per 4 MFMAs: cndmask, s_set_gpr_idx_idx + 3 scalar");
    run<4>(dout, "cbsz:2, per 4 MFMAs: cndmask, 3 scalar on M0 directly");
    run<5>(dout, "cbsz:2, per 8 MFMAs: 2 cndmask, 3 scalar on M0 directly");
    return 0;
}
