// Do the MFMA sweep of one wave and the scalar / vector bookkeeping of ANOTHER wave of the same SIMD overlap?
// Blocks of 8 waves fill a CU (launch_bounds 512, 250 VGPRs are not needed here: 2 waves per SIMD by block shape: one block per CU
// via 128 KB of LDS).  Waves 0-3 run `ma` iterations of the sweep block (4 MFMAs each), waves 4-7 run `sb` iterations of a
// bookkeeping loop (SALU-heavy, VALU-heavy or mixed).  Times: A alone, B alone, both.
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o /tmp/op && /tmp/op
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND> __global__ __launch_bounds__(512) void k_probe(float *out, int ma, int mb, float b)
{
    extern __shared__ float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    if (w < 4) {
        f16v acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float a0 = 1.f, b0 = b;
        for (int it = 0; it < ma; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a0), "v"(b0));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else {
        int x = mb + 7, y = ma + 3;
        float v0 = b, v1 = b + 1.f;
        for (int it = 0; it < mb; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (KIND == 0) // 16 dependent scalar instructions
                    asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\t"
                                 "s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\t"
                                 "s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\t"
                                 "s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1" : "+s"(x) : "s"(y) : "scc");
                else if (KIND == 1) // 16 vector moves
                    asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\t"
                                 "v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0" : "+v"(v0), "+v"(v1));
                else // 12 scalar + 4 vector
                    asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\tv_mov_b32 %2, %3\n\t"
                                 "s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\tv_mov_b32 %3, %2\n\t"
                                 "s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\tv_mov_b32 %2, %3\n\t"
                                 "s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\tv_mov_b32 %3, %2" : "+s"(x), "+s"(y), "+v"(v0), "+v"(v1) : : "scc");
            }
        }
        s = (float)x + v0 + v1;
    }
    if (ma < 0) lds[threadIdx.x] = s;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND> static void run(float *dout, const char *what)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k_probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const int ma = 1500, mb = 1500;
    const int cfg[3][2] = {{ma, 0}, {0, mb}, {ma, mb}};
    float t[3];
    for (int c = 0; c < 3; ++c) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_probe<KIND>), dim3(256), dim3(512), 128 * 1024, 0, dout, cfg[c][0], cfg[c][1], 0.5f);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        t[c] = best;
    }
    printf("%-28s MFMA wave alone %.3f ms (%.1f cycles per MFMA), bookkeeping wave alone %.3f ms (%.1f cycles per instruction), both on one SIMD %.3f ms (sum %.3f, max %.3f)\n",
           what, t[0], t[0] * 1e-3 * 2.4e9 / (ma * 32.0), t[1], t[1] * 1e-3 * 2.4e9 / (mb * 128.0), t[2], t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
}

int main()
{
    float *dout;
    CK(hipMalloc(&dout, 256 * 512 * 4));
    run<0>(dout, "scalar bookkeeping");
    run<1>(dout, "vector moves");
    run<2>(dout, "12 scalar + 4 vector");
    return 0;
}
