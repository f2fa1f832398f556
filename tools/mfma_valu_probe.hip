// Does a K = 1 f32 MFMA share its issue time with ordinary vector instructions?  One dependent chain of v_mfma_f32_16x16x1_4B_f32
// per wave, NV independent v_add_f32 between consecutive MFMAs, W waves per SIMD.  If the adds were free (a second pipe) the time
// per MFMA would stay at the chain's own rate; if they serialise it grows by 4 cycles per add.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/mvp && /tmp/mvp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NV> __global__ __launch_bounds__(256) void k_probe(float *out, int iters, float m, float b)
{
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float t0 = b, t1 = b + 1.f, t2 = b + 2.f, t3 = b + 3.f, t4 = b + 4.f, t5 = b + 5.f, t6 = b + 6.f, t7 = b + 7.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc = __builtin_amdgcn_mfma_f32_16x16x1f32(m, b, acc, 0, 0, 0);
            if (NV > 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t0) : "v"(m));
            if (NV > 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t1) : "v"(m));
            if (NV > 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t2) : "v"(m));
            if (NV > 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t3) : "v"(m));
            if (NV > 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t4) : "v"(m));
            if (NV > 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t5) : "v"(m));
            if (NV > 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t6) : "v"(m));
            if (NV > 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t7) : "v"(m));
        }
    }
    float s = t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV> static void run(float *dout)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int wps = 1; wps <= 4; ++wps) { // block = 4 waves = one per SIMD; wps blocks per CU
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_probe<NV>), dim3(256 * wps), dim3(256), 0, 0, dout, iters, 1.0f, 0.5f);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("v_add per MFMA %d, waves/SIMD %d: %.1f cycles per MFMA per SIMD at 2.4 GHz\n", NV, wps, best * 1e-3 * 2.4e9 / ((double)iters * 8 * wps));
    }
}

int main()
{
    float *dout;
    CK(hipMalloc(&dout, 1024 * 256 * 4));
    run<0>(dout);
    run<1>(dout);
    run<2>(dout);
    run<4>(dout);
    run<8>(dout);
    return 0;
}
