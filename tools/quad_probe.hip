// What does the quad sequence of stm_k_pq_v12r cost with nothing around it?  One dependent chain of v_mfma_f32_16x16x1_4B_f32
// per wave (accumulators in VGPRs), per quad: MFMA, four v_cndmask (A operands of the next quad), MFMA, MFMA, MFMA.  Features
// are added one at a time: F & 1 = a never-taken s_bitcmp / s_cbranch per quad, F & 2 = four s_waitcnt vmcnt(16) per quad,
// F & 4 = s_waitcnt lgkmcnt(0) + s_load_dwordx16 per two quads (scalar-cache hits), F & 8 = B operands through the VGPR index
// mode (s_set_gpr_idx_on ... src1 relative).   hipcc --offload-arch=gfx950 -O3 tools/quad_probe.hip -o /tmp/qp && /tmp/qp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int F> __global__ __launch_bounds__(256) void k_probe(float *out, const uint32_t *tab, int iters, unsigned act, float b)
{
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a0 = 1.f, a1 = 1.f, a2 = 1.f, a3 = 1.f, n0, n1, n2, n3;
    float b0 = b, b1 = b + 1.f, b2 = b + 2.f, b3 = b + 3.f;
    u16v S;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(S) : "s"(tab));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (F & 1) { if (!(act & (1u << q))) continue; }
            if (F & 8) asm volatile("s_set_gpr_idx_on %0, 0x2" : : "s"(0));
            asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a0), "v"(b0));
            if (F & 8) asm volatile("s_set_gpr_idx_off");
            if ((F & 4) && (q & 1)) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(S));
            asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %4\n\tv_cndmask_b32_e64 %1, 0, 1.0, %5\n\tv_cndmask_b32_e64 %2, 0, 1.0, %6\n\tv_cndmask_b32_e64 %3, 0, 1.0, %7"
                         : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
                         : "s"((unsigned long long)S[0] | ((unsigned long long)S[1] << 32)), "s"((unsigned long long)S[2] | ((unsigned long long)S[3] << 32)),
                           "s"((unsigned long long)S[4] | ((unsigned long long)S[5] << 32)), "s"((unsigned long long)S[6] | ((unsigned long long)S[7] << 32)));
            if (F & 2) asm volatile("s_waitcnt vmcnt(16)");
            if (F & 8) asm volatile("s_set_gpr_idx_on %0, 0x2" : : "s"(0));
            asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a1), "v"(b1));
            if ((F & 4) && (q & 1)) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(S) : "s"(tab));
            if (F & 2) asm volatile("s_waitcnt vmcnt(16)");
            asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a2), "v"(b2));
            if (F & 2) asm volatile("s_waitcnt vmcnt(16)");
            asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a3), "v"(b3));
            if (F & 8) asm volatile("s_set_gpr_idx_off");
            a0 = n0; a1 = n1; a2 = n2; a3 = n3;
        }
    }
    if (F & 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(S));
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    float s = (float)S[0];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the planned branch-free block: index mode on for the whole sweep, per block s_set_gpr_idx_idx + three scalar instructions that
// advance the ring index, odd blocks wait for / reload a mask set
// V & 1: three s_nop after the even blocks (the padding that makes all blocks of the kernel 92 bytes); V & 2: the ring index really
// changes from block to block (B operands = v[64 + index ..], an array of 64 registers kept live by the clobbers)
template <int V> __global__ __launch_bounds__(256) void k_planned(float *out, const uint32_t *tab, int iters, float b)
{
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a0 = 1.f, a1 = 1.f, a2 = 1.f, a3 = 1.f, n0, n1, n2, n3;
    float b0 = b, b1 = b + 1.f, b2 = b + 2.f, b3 = b + 3.f;
    u16v S;
    int ridx = 0, rn = 60;
    if (V & 2) asm volatile("v_mov_b32 v64, 1.0\n\tv_mov_b32 v100, 2.0" ::: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131");
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(S) : "s"(tab));
    asm volatile("s_set_gpr_idx_on %0, 0x2" : : "s"(0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (V & 2) asm volatile("s_set_gpr_idx_idx %0" : : "s"(ridx));
            else asm volatile("s_set_gpr_idx_idx %0" : : "s"(0));
            if (V & 2) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, v64, %0" : "+v"(acc) : "v"(a0));
            else asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a0), "v"(b0));
            if (q & 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(S));
            asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %4\n\tv_cndmask_b32_e64 %1, 0, 1.0, %5\n\tv_cndmask_b32_e64 %2, 0, 1.0, %6\n\tv_cndmask_b32_e64 %3, 0, 1.0, %7"
                         : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
                         : "s"((unsigned long long)S[0] | ((unsigned long long)S[1] << 32)), "s"((unsigned long long)S[2] | ((unsigned long long)S[3] << 32)),
                           "s"((unsigned long long)S[4] | ((unsigned long long)S[5] << 32)), "s"((unsigned long long)S[6] | ((unsigned long long)S[7] << 32)));
            if (V & 2) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, v65, %0" : "+v"(acc) : "v"(a1));
            else asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a1), "v"(b1));
            if (q & 1) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(S) : "s"(tab));
            asm volatile("s_add_u32 %0, %0, 4\n\ts_cmp_eq_u32 %0, %1\n\ts_cselect_b32 %0, 0, %0" : "+s"(ridx) : "s"(rn) : "scc");
            if (V & 2) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, v66, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, v67, %0" : "+v"(acc) : "v"(a2), "v"(a3));
            else {
                asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a2), "v"(b2));
                asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a3), "v"(b3));
            }
            if ((V & 1) && !(q & 1)) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0");
            a0 = n0; a1 = n1; a2 = n2; a3 = n3;
        }
    }
    asm volatile("s_set_gpr_idx_off");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(S));
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    float s = (float)S[0] + (float)ridx;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int F> static void run(float *dout, const uint32_t *tab, const char *what)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 1000;
    for (int wps = 1; wps <= 3; ++wps) { // block = 4 waves = one per SIMD; wps blocks per CU
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_probe<F>), dim3(256 * wps), dim3(256), 0, 0, dout, tab, iters, 0xffffffffu, 0.5f);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-64s waves/SIMD %d: %.1f cycles per MFMA per SIMD at 2.4 GHz\n", what, wps, best * 1e-3 * 2.4e9 / ((double)iters * 32 * wps));
    }
}

int main()
{
    float *dout;
    uint32_t *tab;
    CK(hipMalloc(&dout, 1024 * 256 * 4));
    CK(hipMalloc(&tab, 4096));
    CK(hipMemset(tab, 0xff, 4096));
    run<0>(dout, tab, "quad = MFMA, 4 cndmask, 3 MFMA");
    run<1>(dout, tab, "+ s_bitcmp / s_cbranch per quad");
    run<2>(dout, tab, "+ 3 s_waitcnt vmcnt(16) per quad");
    run<4>(dout, tab, "+ lgkmcnt(0) wait and s_load_dwordx16 per two quads");
    run<7>(dout, tab, "all three");
    run<8>(dout, tab, "B through the VGPR index mode (gpr_idx on/off around MFMAs)");
    run<15>(dout, tab, "everything");
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int iters = 1000;
        for (int variant = 0; variant < 4; ++variant)
        for (int wps = 1; wps <= 3; ++wps) {
            float best = 1e30f;
            for (int r = 0; r < 4; ++r) {
                CK(hipEventRecord(e0, 0));
                if (variant == 0) hipLaunchKernelGGL(k_planned<0>, dim3(256 * wps), dim3(256), 0, 0, dout, tab, iters, 0.5f);
                else if (variant == 1) hipLaunchKernelGGL(k_planned<1>, dim3(256 * wps), dim3(256), 0, 0, dout, tab, iters, 0.5f);
                else if (variant == 2) hipLaunchKernelGGL(k_planned<2>, dim3(256 * wps), dim3(256), 0, 0, dout, tab, iters, 0.5f);
                else hipLaunchKernelGGL(k_planned<3>, dim3(256 * wps), dim3(256), 0, 0, dout, tab, iters, 0.5f);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%-64s variant %d (1: + s_nop x3, 2: moving index) waves/SIMD %d: %.1f cycles per MFMA per SIMD at 2.4 GHz\n", "planned block: index mode, ring index update, waits + loads, no branch", variant, wps, best * 1e-3 * 2.4e9 / ((double)iters * 32 * wps));
        }
    }
    return 0;
}
