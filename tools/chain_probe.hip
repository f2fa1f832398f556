// chain_probe.hip -- cost model check for the "shared-start chains" formulation of the window sums:
// lanes = the 64 hypotheses of one pixel, a wave walks ONE running sum per distinct window start, so every add is useful
// (12.9 adds per pixel and pass on the benchmark frame instead of the 44-47 masked steps of a 16-pixel MFMA tile), but the
// control flow is per wave: every group of four adds needs its own scalar bookkeeping (window table, loop, "is a result
// due here") and every add its operand from LDS.  This probe runs the inner structure -- one ds_read_b128 (1 KB per wave),
// four DEPENDENT v_add_f32, S scalar instructions, one branch per group -- with 4 or 8 waves per SIMD on every CU and
// reports cycles per wave-add per CU.  hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int S> __global__ __launch_bounds__(512) void k(float *out, int groups, int seed)
{
    __shared__ f4 tile[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 512) tile[i] = f4{1.f + i, 2.f, 3.f, 4.f};
    __syncthreads();
    const int l = threadIdx.x & 63;
    float acc = 0.f;
    int g = __builtin_amdgcn_readfirstlane(seed + (threadIdx.x >> 6)), ctl = seed, lim = seed ^ 0x55;
    for (int it = 0; it < groups; ++it) {
        const f4 c = tile[(g & 63) * 64 + l];
        g += 1;
        // S scalar instructions of bookkeeping per group (opaque to the compiler)
        if (S >= 2) asm volatile("s_add_u32 %0, %0, 3\n\ts_lshr_b32 %1, %1, 1" : "+s"(ctl), "+s"(lim));
        if (S >= 4) asm volatile("s_xor_b32 %0, %0, %1\n\ts_and_b32 %1, %1, 0xffff" : "+s"(ctl), "+s"(lim));
        if (S >= 6) asm volatile("s_sub_u32 %0, %0, 7\n\ts_or_b32 %1, %1, 0x100" : "+s"(ctl), "+s"(lim));
        if (S >= 8) asm volatile("s_add_u32 %0, %0, 5\n\ts_bfe_u32 %1, %1, 0x100008" : "+s"(ctl), "+s"(lim));
        acc = acc + c.x;
        acc = acc + c.y;
        acc = acc + c.z;
        acc = acc + c.w;
        if (S > 0 && ctl == 0x7fffffff) acc += 1.f; // a result-due branch that is never taken
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc + (float)(ctl + lim);
}
template <int S> static void run(float *d, int blocks_per_cu, int cus, double mhz)
{
    const int groups = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<S><<<cus * blocks_per_cu, 512>>>(d, 64, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<S><<<cus * blocks_per_cu, 512>>>(d, groups, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_adds_per_cu = (double)blocks_per_cu * 8 * groups * 4;
    printf("S = %d scalar instr / group, %d waves / SIMD: %.3f ms -> %.2f cycles per wave-add per CU (%.1f adds / clk / CU)\n", S, 2 * blocks_per_cu,
           ms, ms * 1e-3 * mhz * 1e6 / wave_adds_per_cu, 64.0 * wave_adds_per_cu / (ms * 1e-3 * mhz * 1e6));
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz\n", p.name, cus, mhz);
    float *d;
    hipMalloc(&d, (size_t)cus * 4 * 512 * 4);
    for (int bpc = 2; bpc <= 4; bpc += 2) {
        run<0>(d, bpc, cus, mhz); run<2>(d, bpc, cus, mhz); run<4>(d, bpc, cus, mhz); run<6>(d, bpc, cus, mhz); run<8>(d, bpc, cus, mhz);
    }
    return 0;
}
