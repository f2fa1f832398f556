// mfma_probe2.hip -- CBSZ/ABID broadcast semantics of v_mfma_f32_4x4x1_16B_f32 on gfx950, and the cost of a
// realistic window-sum iteration (ds_read_b128 + mask VALU + 4 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int CBSZ, int ABID> __global__ void k_bcast(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

template <int CBSZ, int ABID> static void run_bcast(const float *da, const float *db, float *dd, const std::vector<float> &a)
{
    std::vector<float> d(256);
    hipLaunchKernelGGL((k_bcast<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    // report, for every block b, which lane's A value row i came from (B = 1 so D[i][j] = A_i)
    printf("CBSZ=%d ABID=%d: source block of A per destination block:", CBSZ, ABID);
    int consistent = 1;
    for (int b = 0; b < 16; ++b) {
        const int src_lane = (int)d[(4 * b) * 4 + 0] - 1; // row 0, col 0 -> A lane
        printf(" %d", src_lane / 4);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if ((int)d[(4 * b + j) * 4 + i] - 1 != (src_lane / 4) * 4 + i) consistent = 0;
    }
    printf("  (%s)\n", consistent ? "rows consistent" : "INCONSISTENT");
}

// realistic H-pass iteration: LDS tile [groups][16 d][4 px] per chunk; wave = 16 px x 64 d = 4 chunk chains sharing masks.
// lane = 4*b + i, b = 4*pt + dq.  B operand = cost (lane -> d = 4*dq + i of chunk c at step), A = mask via cbsz=2/abid.
__global__ __launch_bounds__(256) void k_hloop(float *out, int iters, int ngroups)
{
    extern __shared__ f4 tile[]; // [4 chunks][ngroups][16]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    for (int i = tid; i < 4 * ngroups * 16; i += 256) tile[i] = (f4){1.f, 0.5f, 0.25f, 2.f};
    __syncthreads();
    const int pt = l >> 4, d = l & 15;
    f4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    int t = -(l & 7) - w;
    const int n = 24 + (l & 3);
    int g = pt + w;
    for (int it = 0; it < iters; ++it) {
        f4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = tile[(c * ngroups + (g % ngroups)) * 16 + d];
        const float m = ((unsigned)t < (unsigned)n) ? 1.0f : 0.0f;
        t += 4;
        g += 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].x, acc[c], 2, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].y, acc[c], 2, 1, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].z, acc[c], 2, 2, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(m, v[c].w, acc[c], 2, 3, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + tid] = s;
}

int main()
{
    float *da, *db, *dd;
    CK(hipMalloc(&da, 256));
    CK(hipMalloc(&db, 256));
    CK(hipMalloc(&dd, 4096));
    std::vector<float> a(64), b(64, 1.0f);
    for (int l = 0; l < 64; ++l) a[l] = (float)(l + 1);
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    run_bcast<0, 0>(da, db, dd, a);
    run_bcast<2, 0>(da, db, dd, a);
    run_bcast<2, 1>(da, db, dd, a);
    run_bcast<2, 3>(da, db, dd, a);
    run_bcast<4, 0>(da, db, dd, a);
    run_bcast<4, 5>(da, db, dd, a);
    run_bcast<4, 15>(da, db, dd, a);
    run_bcast<3, 2>(da, db, dd, a);
    run_bcast<1, 1>(da, db, dd, a);

    float *dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2048, ngroups = 48;
    const size_t smem = (size_t)4 * ngroups * 16 * 16; // 48 KB
    for (int bpc = 1; bpc <= 3; ++bpc) {
        const int nb = 256 * bpc;
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_hloop, dim3(nb), dim3(256), smem, 0, dout, iters, ngroups);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // per SIMD: bpc waves, each iters * 16 MFMAs
        const double cyc = best * 1e-3 * 2.4e9 / ((double)iters * 16 * bpc);
        printf("HLOOP blocks/CU %d (waves/SIMD %d): %.3f ms -> %.2f cycles per MFMA per SIMD at 2.4 GHz\n", bpc, bpc, best, cyc);
    }
    return 0;
}
