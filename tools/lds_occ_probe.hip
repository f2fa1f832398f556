// How many workgroups with ~50 KB of dynamic LDS are resident per CU on gfx950?  (occupancy API + an empirical timing test)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_spin(float *out, int iters)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a = lds[(threadIdx.x + 1) % blockDim.x];
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out[0] = a;
}
int main()
{
    float *d;
    CK(hipMalloc(&d, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int threads[] = {128, 512};
    const size_t lds[] = {16 * 1024, 32 * 1024, 40 * 1024, 51200, 53760, 60 * 1024, 64 * 1024, 70 * 1024, 80 * 1024};
    for (int ti = 0; ti < 2; ++ti)
        for (size_t s : lds) {
            if (s > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s));
            int nb = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spin, threads[ti], s));
            float t[3];
            for (int k = 1; k <= 3; ++k) { // k blocks per CU: if all are resident the time stays flat
                hipLaunchKernelGGL(k_spin, dim3(256 * k), dim3(threads[ti]), s, 0, d, 20000);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_spin, dim3(256 * k), dim3(threads[ti]), s, 0, d, 20000);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t[k - 1], e0, e1));
            }
            printf("threads %d LDS %6zu B: occupancy API %d blocks/CU; time for 1,2,3 blocks per CU: %.3f %.3f %.3f ms\n", threads[ti], s, nb, t[0], t[1], t[2]);
        }
    return 0;
}
