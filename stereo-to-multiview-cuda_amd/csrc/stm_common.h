// stm_common.h -- internal runtime shared by the kernel files and the C-ABI layer.
// gfx950 only.  Not installed; the public surface is include/stm_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef unsigned char u8;

namespace stm {

// ---- error handling: reference semantics (cuda_utils.h:12-21) = message + exit(1) ----
void fail(const char *what, const char *expr, const char *file, int line);
// Error mode 1 (record and return, stm_set_error_mode): a failure is STICKY for the rest of the API call on this thread --
// every later kernel launch of the call is skipped (STM_LAUNCH) and the workspace hands out no memory, so a failed
// allocation can never turn into a kernel running on a null or stale pointer.  Each API entry point clears the flag.
bool failed();
void clear_failed();
struct ApiNest { ApiNest(); ~ApiNest(); }; // held by an entry point around the entry points it calls
bool api_outermost();
void release_host_frame_bufs(); // stm_api.hip: the calling thread's staging buffers of the host-flavour frame calls
#define STM_CHECK(expr)                                                                  \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) ::stm::fail(hipGetErrorString(_e), #expr, __FILE__, __LINE__); \
    } while (0)
// the reference never checks launches; we do
#define STM_CHECK_LAUNCH() STM_CHECK(hipGetLastError())
#define STM_LAUNCH(...)                                     \
    do {                                                    \
        if (!::stm::failed()) hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

hipStream_t stream();
// One word per device, allocated on first use and never freed (so a captured graph can keep its address), zero unless a kernel
// clamped something that cannot happen (see IrvArgs::diag).  stm_last_error() reads and clears it.
uint32_t *device_diag();

// ---- grow-only device workspace, bump-allocated per top-level call -----------------
// The reference hipMalloc/hipFree's ~35 buffers per frame (d_io.cu:43-235); here one
// cached slab is carved up, so a steady-state frame performs no allocation at all.
struct Workspace {
    static void begin(size_t bytes_hint = 0); // start a carve scope (resets the bump pointer)
    static void *alloc(size_t bytes);         // 256-B aligned; grows (sync + realloc) if needed
    template <class T> static T *get(size_t n) { return (T *)alloc(n * sizeof(T)); }
};

void *ws_private_create();
void ws_private_bind(void *ws); // nullptr: back to the thread's shared workspace
void ws_private_destroy(void *ws);
void ws_identity(void **base, size_t *cap);
bool prof_enabled();

// ---- profiling of named kernels with HIP events on the launch stream ----------------
struct ProfScope {
    ProfScope(const char *name);
    ~ProfScope();
    int slot;
    void *ev_b; // end event (kept here: the record vector may be resized by another thread)
};

int agg_variant();
int irv_paper_ratio(); // stm_set_irv_paper_ratio: accept on count / S instead of the reference's bin index / S (SURVEY A-Q17 iv)
// Timing experiments (skip loads / sweeps / stores; results NOT valid) exist only in the separate libstm_hip_timing.so
// (make timing, -DSTM_TIMING): in the product library every STM_DBG test is the constant false and stm_set_agg_variant
// accepts result-preserving variants only.
#ifdef STM_TIMING
#define STM_DBG(dbg, bit) (((dbg) & (bit)) != 0)
static inline int timing_knobs() { return (agg_variant() / 100000) % 10; }
#else
#define STM_DBG(dbg, bit) false
static inline int timing_knobs() { return 0; }
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// A cost volume as the kernels see it: either the reference's table of plane pointers
// (T1 in SURVEY 8a; `tab` is a DEVICE array of D device pointers) or a dense slab
// [D][H][W] (`base` + d * plane_stride).  One branch per plane access, uniform per wave.
// Third form, internal to the frame pipeline: QUADS = float4 [ceil(D/4)][H][W], the four hypotheses
// 4q..4q+3 of a pixel interleaved in one 16-byte element (`plane_stride` counts float4 elements).
struct Vol {
    float *const *tab;
    float *base;
    size_t plane_stride;
    int quad;
    __device__ __forceinline__ float *plane(int d) const { return tab ? tab[d] : base + (size_t)d * plane_stride; }
};
static inline Vol vol_table(float **d_tab) { Vol v; v.tab = d_tab; v.base = nullptr; v.plane_stride = 0; v.quad = 0; return v; }
static inline Vol vol_slab(float *base, size_t stride) { Vol v; v.tab = nullptr; v.base = base; v.plane_stride = stride; v.quad = 0; return v; }
static inline Vol vol_quads(float *base, size_t stride) { Vol v; v.tab = nullptr; v.base = base; v.plane_stride = stride; v.quad = 1; return v; }

#ifdef __HIPCC__
// A "quad" = the four hypotheses d0..d0+3 of one pixel.  PLANES layout: four dword accesses, one per
// plane (each wave instruction is a contiguous 256-B row segment).  QUADS layout (internal to the frame
// pipeline, Vol::quad): one 16-byte access.
template <bool QUAD> __device__ __forceinline__ float4 load_quad(const Vol &v, int q, int D, size_t idx)
{
    if (QUAD) { // the volume is streamed: each element is read once per pass, so do not let it displace the arm planes in L2
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = __builtin_nontemporal_load((const v4f *)v.base + (size_t)q * v.plane_stride + idx);
        return make_float4(t.x, t.y, t.z, t.w);
    }
    const int d0 = q * 4;
    float4 r;
    r.x = v.plane(d0)[idx];
    r.y = d0 + 1 < D ? v.plane(d0 + 1)[idx] : 0.f;
    r.z = d0 + 2 < D ? v.plane(d0 + 2)[idx] : 0.f;
    r.w = d0 + 3 < D ? v.plane(d0 + 3)[idx] : 0.f;
    return r;
}
// STREAM: non-temporal store (the aggregation passes: measured 3-5 % faster; the cost-init kernel, which only writes, is
// 10 % slower with it and keeps ordinary stores)
template <bool QUAD, bool STREAM = true> __device__ __forceinline__ void store_quad(const Vol &v, int q, int D, size_t idx, float4 s)
{
    if (QUAD && !STREAM) { ((float4 *)v.base)[(size_t)q * v.plane_stride + idx] = s; return; }
    if (QUAD) { // written once, read by the next pass from HBM (530 MB >> L2): streaming store
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = {s.x, s.y, s.z, s.w};
        __builtin_nontemporal_store(t, (v4f *)v.base + (size_t)q * v.plane_stride + idx);
        return;
    }
    const int d0 = q * 4;
    v.plane(d0)[idx] = s.x;
    if (d0 + 1 < D) v.plane(d0 + 1)[idx] = s.y;
    if (d0 + 2 < D) v.plane(d0 + 2)[idx] = s.z;
    if (d0 + 3 < D) v.plane(d0 + 3)[idx] = s.w;
}

#endif

// ---- launchers (one per kernel family; definitions next to the kernels) -------------
// cost init (stm_kernels_cost.hip)
void launch_pack_bgrx(const u8 *bgr, uint32_t *packed, int H, int W, int elem_sz);
void launch_census32_pair(const uint32_t *packed_l, uint32_t *census_l, const uint32_t *packed_r, uint32_t *census_r, int H, int W);
void launch_cost_init(const uint32_t *pk_l, const uint32_t *pk_r, const uint32_t *cen_l, const uint32_t *cen_r,
                      Vol cost_l, Vol cost_r, const float *lut_ad, const float *lut_census,
                      int D, int zd, int H, int W);
void launch_cost_quirks(const uint32_t *pk_l, const uint32_t *pk_r, const uint32_t *cen_l, const uint32_t *cen_r, Vol cost_l,
                        Vol cost_r, const float *lut_ad, const float *lut_census, int D, int zd, int H, int W);
int ref_quirks(); // stm_set_ref_quirks
// aggregation (stm_kernels_agg.hip)
void launch_cross_arms(const uint32_t *packed, u8 *up, u8 *down, u8 *left, u8 *right,
                       float ucd, float lcd, int usd, int lsd, int H, int W);
void launch_cross_arms2(int nviews, const uint32_t *const *packed, u8 *const *up, u8 *const *down, u8 *const *left,
                        u8 *const *right, float ucd, float lcd, int usd, int lsd, int H, int W, const uint32_t *const *wide_ready = nullptr,
                        uint32_t *htab = nullptr, // htab: also build stm_k_pq_hsr's horizontal window table (aggh_table_dwords(nviews, H, W))
                        uint32_t *vtab = nullptr, int vrec = 0, int vtop = -1); // + vtab: and stm_k_pq_v12r's vertical one (aggm_frame_vtab_dwords)
void launch_agg_h(Vol in, Vol out, const u8 *armL, const u8 *armR, int D, int H, int W);
void launch_agg_v(Vol in, Vol out, const u8 *armU, const u8 *armD, int D, int H, int W, int usd);
void launch_agg_h2(Vol in_a, Vol out_a, const u8 *armL_a, const u8 *armR_a, Vol in_b, Vol out_b, const u8 *armL_b, const u8 *armR_b,
                   int D, int H, int W);
void launch_agg_h2_cost(const uint32_t *pk_l, const uint32_t *cen_l, const uint32_t *pk_r, const uint32_t *cen_r, const float *lut,
                        Vol out_l, const u8 *armL_l, const u8 *armR_l, Vol out_r, const u8 *armL_r, const u8 *armR_r, int D, int zd,
                        int H, int W);
void launch_agg_h_wta2(Vol in_a, const u8 *armL_a, const u8 *armR_a, float *disp_a, Vol in_b, const u8 *armL_b, const u8 *armR_b,
                       float *disp_b, int D, int zd, int H, int W);
void launch_wta(Vol cost, float *disp, int D, int zd, int H, int W);
// refinement (stm_kernels_refine.hip)
void launch_dcc(u8 *out_l, u8 *out_r, const float *disp_l, const float *disp_r, u8 *hit_l, u8 *hit_r, int H, int W);
// nviews = 1 or 2 (both views of a frame share every launch); scratch comes from the current Workspace scope
// (16 bytes per pixel per view)
void launch_irv(int nviews, float *const *disp, u8 *const *outl, const u8 *const *up, const u8 *const *down,
                const u8 *const *left, const u8 *const *right, int thresh_s, float thresh_h,
                int H, int W, int D, int zd, int usd, int iterations, bool device_flavour);
void launch_bilateral(const float *in, float *out, const float *spatial, const float *color,
                      int radius, int H, int W, int D);
// integer_maps: both maps hold integer-valued disparities any two of which differ by less than D (the frame pipeline's own
// WTA / region-voting output)
void launch_bilateral2(const float *in_a, float *out_a, const float *in_b, float *out_b, const float *spatial, const float *color,
                       int radius, int H, int W, int D, bool integer_maps = false, const float *one_value = nullptr, int zd = 0);
void launch_gaussian_max(const float *in, float *out, const float *spatial, int radius, float sigma, int H, int W,
                         bool invert_input);
// DIBR + mux (stm_kernels_dibr.hip)
void launch_demux_sbs(u8 *l, u8 *r, const u8 *sbs, int H, int Wsbs, int W, int elem_sz);
// same, and the BGRX dwords (launch_pack_bgrx) + wide pixels (launch_cross_arms2) of both halves in the same pass
void launch_demux_sbs_packed(u8 *l, u8 *r, uint32_t *pk_l, uint32_t *pk_r, uint32_t *wide_l, uint32_t *wide_r, const u8 *sbs, int H,
                             int Wsbs, int W, int elem_sz);
void launch_occl(u8 *occl_l, u8 *occl_r, const float *disp_l, const float *disp_r, int H, int W);
void launch_bleed(const u8 *in, u8 *out, int radius, int H, int W);
void launch_median3(const float *in, float *out, int H, int W);
void launch_dcc_rows(u8 *out_l, u8 *out_r, const float *disp_l, const float *disp_r, int H, int W);
void launch_hitmask_rows(float *mask_l, float *mask_r, const float *disp_l, const float *disp_r, int H, int W);
void launch_occl_to_mask(float *mask_l, float *mask_r, const u8 *occl_l, const u8 *occl_r, int H, int W);
void launch_view_synth(u8 *out, const u8 *img_l, const u8 *img_r, const float *disp_l, const float *disp_r,
                       const float *mask_l, const float *mask_r, const float *blend, float shift, int H, int W, int elem_sz);
void launch_view_synth_all(u8 *views, size_t view_stride, int N, const u8 *img_l, const u8 *img_r, const float *disp_l,
                           const float *disp_r, const float *mask_l, const float *mask_r, const float *blend, int H, int W,
                           int elem_sz);
void launch_fwarp(u8 *out, const u8 *img, const float *disp, float shift, unsigned long long *keys, int H, int W, int elem_sz);
void launch_scale_bilinear(const u8 *in, u8 *out, int in_rows, int in_cols, int out_rows, int out_cols, int elem_sz);
void launch_disp_scale(float *out, const float *in, int out_rows, int out_cols, int in_rows, int in_cols, float disp_scale);
void launch_view_table(u8 **tab, u8 *first, u8 *last, u8 *mem, size_t stride, int N);
// frame pipeline: the N - 2 views are synthesised inside the interlacer, sample by sample (no view buffers)
void launch_synth_mux(const u8 *img_l, const u8 *img_r, const float *disp_l, const float *disp_r, const float *mask_l, const float *mask_r,
                      const float *blend, u8 *out, int N, float y_interval, float inv_y_interval, int ymod, int Hin, int Win, int Hout,
                      int Wout, int elem_sz, int variant);
void launch_mux(const u8 *const *d_views, u8 *out, int N, float y_interval, float inv_y_interval, int ymod,
                int Hin, int Win, int Hout, int Wout, int elem_sz, int variant);
// aggregation on the matrix pipe (stm_kernels_aggm.hip): the frame pipeline's cost -> H -> V, V -> H + WTA
struct PQViews { // both views of a frame; a / b = the two PQ volumes of a view
    const uint32_t *pk[2], *cen[2];
    float *a[2], *b[2];
    const u8 *armU[2], *armD[2], *armL[2], *armR[2];
    float *disp[2];
};
size_t pq_volume_floats(int D, int H, int W);
bool aggm_supports(int usd, int H, int W);
void launch_aggm_frame(const uint32_t *const *pk, const uint32_t *const *cen, const float *lut, float *const *vol_a, float *const *vol_b,
                       const u8 *const *armU, const u8 *const *armD, const u8 *const *armL, const u8 *const *armR, float *const *disp,
                       int D, int zd, int H, int W, int usd, bool keep_volume = false, uint32_t *htab_ready = nullptr, uint32_t *vtab_ready = nullptr);
size_t aggm_frame_htab_dwords(int D, int H, int W, int usd, bool keep_volume);
size_t aggm_frame_vtab_dwords(int H, int W, int usd, int *rec, int *top); // the vertical table in the register-ring kernel's static layout, or 0
// ca_cross / d_ca_cross of one volume in the caller's layout on the matrix-pipe kernels; `out` may be `in`.  Returns false, with
// `out` untouched, when the volume holds an infinite, NaN or denormal element (one host read-back of a flag): the caller runs
// the vector-ALU kernels instead.  aggm_stage_bytes: what it carves from the current Workspace scope.
bool launch_aggm_stage(Vol in, Vol out, const u8 *armU, const u8 *armD, const u8 *armL, const u8 *armR, int D, int H, int W, int usd);
size_t aggm_stage_bytes(int D, int H, int W, int usd);
// both vertical passes with a strip's rows in registers (stm_kernels_aggv.hip); `tab` / `rec`: the table of stm_k_vwin_table
bool aggh_supports(int usd, int D); // stm_kernels_aggh.hip: last horizontal pass + WTA with the row's window range in registers
size_t aggh_table_dwords(int nviews, int H, int W);
void launch_hwin_table(PQViews &v, int nviews, uint32_t *tab, int H, int W); // (the frame path builds the table inside stm_k_cross_arms: launch_cross_arms2's htab)
void launch_pq_hsr(PQViews &v, int nviews, const uint32_t *tab, int D, int zd, int H, int W);
bool aggv_supports(int usd);
int aggv_table_top();
int aggv_table_rec();
void launch_pq_v12r(PQViews &v, int nviews, const uint32_t *tab, int rec, int H, int W, int G, int NC);
void launch_to_pq(Vol in, float *pq, int D, int H, int W, uint32_t *odd = nullptr); // stm_kernels_hslo.hip; odd: see stm_k_to_pq
void launch_from_pq(const float *pq, Vol out, int D, int H, int W); // stm_kernels_aggm.hip
// HSLO (stm_kernels_hslo.hip)
void launch_hslo_wta(int nviews, const Vol *cost, const u8 *const *img_a, const u8 *const *img_b, const int *osign,
                     float *const *disp, float T, float H1, float H2, int D, int zd, int H, int W, int elem_sz);
// the same on PQ volumes (the frame pipeline): cost_pq read only, acc_pq scratch of the same size
void launch_hslo_wta_pq(int nviews, float *const *cost_pq, float *const *acc_pq, const u8 *const *img_a, const u8 *const *img_b,
                        const int *osign, float *const *disp, float T, float H1, float H2, int D, int zd, int H, int W, int elem_sz);

// host-built tables (same formulas as the reference's host code; see stm_tables.cpp)
void rho_luts(float ad_coeff, float census_coeff, float *lut_ad /*766*/, float *lut_census /*65*/);
void gaussian_kernel_2d(float *kernel, int radius, float sigma);
void gaussian_kernel_1d(float *kernel, int size, float sigma);
float mux_y_interval(int num_views, float angle, int elem_sz);

} // namespace stm
