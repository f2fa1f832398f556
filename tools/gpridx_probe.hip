// Does the VGPR index mode (s_set_gpr_idx_on) apply to the operands of v_mfma_f32_16x16x1_4b_f32 on gfx950?
// If SRC1 of an MFMA can be M0-relative, a ring of rows held in registers can be swept by a ROLLED loop (no unrolling over ring
// positions).   hipcc --offload-arch=gfx950 -O3 tools/gpridx_probe.hip -o /tmp/gip && /tmp/gip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_probe(float *out, int idx)
{
    float r_mfma_src1, r_mfma_src0, r_mov, r_mfma_dst;
    asm volatile(
        "v_mov_b32 v40, 1.0\n\t"      // A candidates: v40 = 1, v41 = 2
        "v_mov_b32 v41, 2.0\n\t"
        "v_mov_b32 v50, 3.0\n\t"      // B candidates: v50 = 3, v51 = 5
        "v_mov_b32 v51, 5.0\n\t"
        "s_nop 4\n\t"
        // 1: SRC1 relative
        "s_set_gpr_idx_on %4, 0x2\n\t"
        "v_mfma_f32_16x16x1_4b_f32 v[0:15], v40, v50, 0\n\t"
        "s_set_gpr_idx_off\n\t"
        "s_nop 15\n\ts_nop 7\n\t"
        "v_mov_b32 %0, v0\n\t"
        // 2: SRC0 relative
        "s_set_gpr_idx_on %4, 0x1\n\t"
        "v_mfma_f32_16x16x1_4b_f32 v[0:15], v40, v50, 0\n\t"
        "s_set_gpr_idx_off\n\t"
        "s_nop 15\n\ts_nop 7\n\t"
        "v_mov_b32 %1, v0\n\t"
        // 3: v_mov with SRC0 relative (the documented use)
        "s_set_gpr_idx_on %4, 0x1\n\t"
        "v_mov_b32 v60, v50\n\t"
        "s_set_gpr_idx_off\n\t"
        "v_mov_b32 %2, v60\n\t"
        // 4: DST relative on the MFMA (acc tuple v[0:15] vs v[1:16])
        "v_mov_b32 v16, 0\n\t"
        "s_nop 2\n\t"
        "s_set_gpr_idx_on %4, 0x8\n\t"
        "v_mfma_f32_16x16x1_4b_f32 v[0:15], v40, v50, 0\n\t"
        "s_set_gpr_idx_off\n\t"
        "s_nop 15\n\ts_nop 7\n\t"
        "v_mov_b32 %3, v16\n\t"
        : "=v"(r_mfma_src1), "=v"(r_mfma_src0), "=v"(r_mov), "=v"(r_mfma_dst)
        : "s"(idx)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v40", "v41", "v50", "v51", "v60", "m0");
    if (threadIdx.x == 0) { out[0] = r_mfma_src1; out[1] = r_mfma_src0; out[2] = r_mov; out[3] = r_mfma_dst; }
}

int main()
{
    float *d, h[4];
    CK(hipMalloc(&d, 16));
    for (int idx = 0; idx < 2; ++idx) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, idx);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("idx %d: MFMA src1-relative -> %.1f (3 = v50, 5 = v51: relative works) | MFMA src0-relative -> %.1f (3 = A from v40 = 1.0, 6 = A from v41) | v_mov src0-relative -> %.1f (3 / 5) | MFMA dst-relative: v16 = %.1f (0 = not applied)\n",
               idx, h[0], h[1], h[2], h[3]);
    }
    return 0;
}
