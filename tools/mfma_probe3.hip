// probe3: what slows the realistic loop -- cbsz broadcast, AGPR accumulators, LDS operands, or chain spacing?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE bit0: use cbsz=2 + abid cycling; bit1: operands from LDS (b128 per 4 steps); bit2: mask VALU per 4 steps
template <int NCH, int MODE> __global__ __launch_bounds__(256) void k_var(float *out, int iters, float a, float b, int ngroups)
{
    extern __shared__ f4 tile[];
    const int tid = threadIdx.x, l = tid & 63;
    if (MODE & 2) {
        for (int i = tid; i < NCH * ngroups * 16; i += 256) tile[i] = (f4){1.f, 0.5f, 0.25f, 2.f};
        __syncthreads();
    }
    f4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    int t = -(l & 7);
    const int n = 24 + (l & 3);
    int g = (l >> 4);
    float m = a;
    for (int it = 0; it < iters; ++it) {
        f4 v[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = (MODE & 2) ? tile[(c * ngroups + g) * 16 + (l & 15)] : (f4){b, b, b, b};
        if (MODE & 2) { g += 1; if (g >= ngroups) g = 0; }
        if (MODE & 4) { m = ((unsigned)t < (unsigned)n) ? 1.0f : 0.0f; t += 4; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = (MODE & 1) ? __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].x, acc[c], 2, 0, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].x, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = (MODE & 1) ? __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].y, acc[c], 2, 1, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].y, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = (MODE & 1) ? __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].z, acc[c], 2, 2, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].z, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = (MODE & 1) ? __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].w, acc[c], 2, 3, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].w, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NCH, int MODE> static void run(float *dout, const char *name)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2048, ngroups = 24;
    const size_t smem = (MODE & 2) ? (size_t)NCH * ngroups * 16 * 16 : 0;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        const int nb = 256 * bpc;
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_var<NCH, MODE>), dim3(nb), dim3(256), smem, 0, dout, iters, 1.0f, 0.5f, ngroups);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-34s chains %d waves/SIMD %d: %.2f cycles/MFMA/SIMD\n", name, NCH, bpc, best * 1e-3 * 2.4e9 / ((double)iters * 4 * NCH * bpc));
    }
}

int main()
{
    float *dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    run<4, 0>(dout, "plain (reg operands)");
    run<4, 1>(dout, "cbsz2+abid");
    run<4, 2>(dout, "LDS b128 operands");
    run<4, 3>(dout, "LDS + cbsz");
    run<4, 7>(dout, "LDS + cbsz + mask VALU");
    run<8, 0>(dout, "plain 8 chains");
    run<8, 7>(dout, "LDS + cbsz + mask, 8 chains");
    run<2, 7>(dout, "LDS + cbsz + mask, 2 chains");
    return 0;
}
