// probe4: the inner loop of the vertical kernel in isolation (16x16x1_4B chain fed from an LDS ring, masks from VALU).
// VAR bit0: masks computed per step (else constant 1.0);  bit1: LDS operands (else register constants);
//     bit2: operands of iteration i+1 read before the MFMAs of iteration i (software pipelining);  bit3: two independent chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int VAR> __global__ __launch_bounds__(384) void k_loop(float *out, int iters, int R, int n_it, float cst)
{
    extern __shared__ float ring[];
    const int tid = threadIdx.x, l = tid & 63;
    for (int i = tid; i < R * 64; i += 384) ring[i] = 1.0f + (float)(i & 7);
    __syncthreads();
    f16v acc, acc2;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    const int nn = 20 + (l & 15), s0 = (l & 3);
    for (int rep = 0; rep < iters; ++rep) {
        int tt = -s0;
        const float *p = ring + ((rep * 4) % R) * 64 + l;
        const float *const pend = ring + R * 64 + l;
        float c0 = cst, c1 = cst, c2 = cst, c3 = cst;
        if (VAR & 4) { c0 = p[0]; c1 = p[64]; c2 = p[128]; c3 = p[192]; }
        for (int it = 0; it < n_it; ++it) {
            float d0 = c0, d1 = c1, d2 = c2, d3 = c3;
            if ((VAR & 2) && !(VAR & 4)) { d0 = p[0]; d1 = p[64]; d2 = p[128]; d3 = p[192]; }
            p += 256;
            if (p >= pend) p -= R * 64;
            if (VAR & 4) { c0 = p[0]; c1 = p[64]; c2 = p[128]; c3 = p[192]; }
            float m0 = 1.f, m1 = 1.f, m2 = 1.f, m3 = 1.f;
            if (VAR & 1) {
                m0 = ((unsigned)tt < (unsigned)nn) ? 1.0f : 0.0f;
                m1 = ((unsigned)(tt + 1) < (unsigned)nn) ? 1.0f : 0.0f;
                m2 = ((unsigned)(tt + 2) < (unsigned)nn) ? 1.0f : 0.0f;
                m3 = ((unsigned)(tt + 3) < (unsigned)nn) ? 1.0f : 0.0f;
                tt += 4;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x1f32(m0, d0, acc, 0, 0, 0);
            if (VAR & 8) acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(m0, d1, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x1f32(m1, d1, acc, 0, 0, 0);
            if (VAR & 8) acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(m1, d2, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x1f32(m2, d2, acc, 0, 0, 0);
            if (VAR & 8) acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(m2, d3, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x1f32(m3, d3, acc, 0, 0, 0);
            if (VAR & 8) acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(m3, d0, acc2, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    out[blockIdx.x * 384 + tid] = s;
}

template <int VAR> static void run(float *dout, const char *name)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 200, R = 116, n_it = 12;
    const size_t smem = (size_t)R * 256;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int nb = 256 * bpc;
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_loop<VAR>), dim3(nb), dim3(384), smem, 0, dout, iters, R, n_it, 0.5f);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double nm = (double)iters * n_it * 4 * ((VAR & 8) ? 2 : 1); // MFMAs per wave
        // waves per SIMD = 6 * bpc / 4
        printf("%-44s blocks/CU %d (%.1f waves/SIMD): %.3f ms, %.1f cycles per MFMA per SIMD at 2.1 GHz, %.0f cycles per wave-iteration\n", name, bpc,
               1.5 * bpc, best, best * 1e-3 * 2.1e9 / (nm * 1.5 * bpc), best * 1e-3 * 2.1e9 / ((double)iters * n_it));
    }
}

int main()
{
    float *dout;
    CK(hipMalloc(&dout, 512 * 384 * 4));
    run<0>(dout, "const operands, const masks");
    run<1>(dout, "masks");
    run<2>(dout, "LDS operands");
    run<3>(dout, "LDS operands + masks (the kernel's loop)");
    run<7>(dout, "LDS (read ahead) + masks");
    run<11>(dout, "LDS + masks, two chains");
    run<15>(dout, "LDS (read ahead) + masks, two chains");
    return 0;
}
