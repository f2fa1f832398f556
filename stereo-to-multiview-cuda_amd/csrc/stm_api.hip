// stm_api.hip -- the C ABI (include/stm_hip.h): host-flavour and device-flavour stage entry points
// and the device-resident frame pipeline.  Mirrors the reference's per-stage host API
// (SURVEY.md section 8b); each function cites the reference wrapper it replaces.
#include "stm_common.h"
#include <initializer_list>
#include <mutex>
#include "../../include/stm_hip.h"

#include <map>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tuple>
#include <vector>

using namespace stm;

namespace {

// ---------------------------------------------------------------- small persistent device tables
struct DevTable {
    float *d = nullptr;
    size_t n = 0;
};
std::map<std::tuple<int, int, float, float>, DevTable> g_tables; // (dev*8+kind, size/radius, p0, p1)

int cur_dev()
{
    int dev = 0;
    STM_CHECK(hipGetDevice(&dev));
    return dev;
}

const float *dev_table(int kind, int size, float p0, float p1, size_t n, void (*fill)(float *, int, float, float))
{
    static std::mutex mu; // tables are shared by all host threads of the process
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(cur_dev() * 8 + kind, size, p0, p1);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return it->second.d;
    std::vector<float> h(n);
    fill(h.data(), size, p0, p1);
    DevTable t;
    t.n = n;
    STM_CHECK(hipMalloc((void **)&t.d, n * sizeof(float)));
    STM_CHECK(hipMemcpy(t.d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    g_tables[key] = t;
    return t.d;
}
void fill_rho(float *h, int, float ad, float ce) { rho_luts(ad, ce, h, h + 768); }
void fill_g2(float *h, int r, float s, float) { gaussian_kernel_2d(h, r, s); }
void fill_g1(float *h, int n, float s, float) { gaussian_kernel_1d(h, n, s); }

const float *rho_table(float ad, float ce) { return dev_table(0, 0, ad, ce, 768 + 72, fill_rho); } // [0..765] ad, [768..832] census
const float *gauss2d_table(int r, float s) { return dev_table(1, r, s, 0.f, (size_t)(2 * r + 1) * (2 * r + 1), fill_g2); }
const float *gauss1d_table(int n, float s) { return dev_table(2, n, s, 0.f, (size_t)(n > 0 ? n : 1), fill_g1); }
// The radius-7 bilateral filter of a map that holds ONE whole number c in a pixel's whole 15 x 15 neighbourhood: every tap has the
// weight spatial x colour[0], and the pixel's result is the same sequence of float operations whatever the pixel
// (d_filter_bilateral.cu:284-300: weight = spatial * colour, norm += weight, res += value * weight, res / norm) -- a function of c
// alone.  Entry i = the result for c = i - zd (the values a disparity map of this frame can hold), computed here with the kernel's
// own operations in the kernel's order (this file is compiled -ffp-contract=off like the kernels).
void fill_bil1(float *h, int size, float sigma_spatial, float sigma_color)
{
    const int D = size >> 12, zd = size & 4095;
    std::vector<float> g2(15 * 15), g1((size_t)(D > 0 ? D : 1));
    gaussian_kernel_2d(g2.data(), 7, sigma_spatial);
    gaussian_kernel_1d(g1.data(), D, sigma_color);
    const float gc = g1[0];
    for (int i = 0; i < D; ++i) {
        const float c = (float)(i - zd);
        volatile float norm = 0.0f, res = 0.0f;
        for (int t = 0; t < 15 * 15; ++t) {
            volatile float w = g2[t] * gc;
            norm = norm + w;
            volatile float cw = c * w;
            res = res + cw;
        }
        h[i] = res / norm;
    }
}
const float *bilateral_one_value_table(int D, int zd, float sigma_spatial, float sigma_color)
{
    if (D < 1 || D >= (1 << 19) || zd < 0 || zd >= 4096) return nullptr;
    return dev_table(3, D * 4096 + zd, sigma_spatial, sigma_color, (size_t)D, fill_bil1);
}

void sync() { STM_CHECK(hipStreamSynchronize(stream())); }

template <class T> T *up(const T *h, size_t n)
{
    T *d = Workspace::get<T>(n);
    STM_CHECK(hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, stream()));
    return d;
}
template <class T> void down(T *h, const T *d, size_t n)
{
    STM_CHECK(hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, stream()));
}
float *up_planes(float **planes, int D, size_t HW)
{
    float *slab = Workspace::get<float>((size_t)D * HW);
    for (int d = 0; d < D; ++d)
        STM_CHECK(hipMemcpyAsync(slab + (size_t)d * HW, planes[d], HW * sizeof(float), hipMemcpyHostToDevice, stream()));
    return slab;
}
void down_planes(float **planes, const float *slab, int D, size_t HW)
{
    for (int d = 0; d < D; ++d)
        STM_CHECK(hipMemcpyAsync(planes[d], slab + (size_t)d * HW, HW * sizeof(float), hipMemcpyDeviceToHost, stream()));
}

struct Arms {
    u8 *up, *down, *left, *right;
};

// Argument screen shared by every entry point.  The reference checks nothing (a zero-sized launch or an
// out-of-range view index is undefined behaviour there); here such calls fail like any other error (stm_hip.h).
struct Dim { const char *name; int v, lo; };
bool args_ok(const char *fn, std::initializer_list<Dim> dims)
{
    if (api_outermost()) clear_failed(); // every entry point starts here: a failure is sticky for one (outermost) API call (stm_common.h)
    for (const Dim &d : dims)
        if (d.v < d.lo) {
            char msg[160];
            snprintf(msg, sizeof msg, "%s: %s = %d, must be >= %d", fn, d.name, d.v, d.lo);
            fail(msg, d.name, __FILE__, __LINE__);
            return false;
        }
    return true;
}


// ---------------------------------------------------------------- device cores
// cost init: pack -> census -> fused AD + census + robust combine
// packed_ready: pk_l / pk_r already hold the BGRX dwords of the two images (launch_demux_sbs_packed)
// census_out != nullptr: stop after the census planes (returned there) -- the caller computes the costs on the fly
void core_ci(const u8 *d_img_l, const u8 *d_img_r, Vol cl, Vol cr, uint32_t *pk_l, uint32_t *pk_r, float ad_coeff,
             float census_coeff, int D, int zd, int H, int W, int elem_sz, bool packed_ready = false,
             uint32_t **census_out = nullptr)
{
    size_t HW = (size_t)H * W;
    uint32_t *cen_l = Workspace::get<uint32_t>(HW), *cen_r = Workspace::get<uint32_t>(HW);
    if (!packed_ready) {
        launch_pack_bgrx(d_img_l, pk_l, H, W, elem_sz);
        launch_pack_bgrx(d_img_r, pk_r, H, W, elem_sz);
    }
    launch_census32_pair(pk_l, cen_l, pk_r, cen_r, H, W);
    if (census_out) {
        census_out[0] = cen_l;
        census_out[1] = cen_r;
        return;
    }
    const float *lut = rho_table(ad_coeff, census_coeff);
    launch_cost_init(pk_l, pk_r, cen_l, cen_r, cl, cr, lut, lut + 768, D, zd, H, W);
    if (ref_quirks()) launch_cost_quirks(pk_l, pk_r, cen_l, cen_r, cl, cr, lut, lut + 768, D, zd, H, W); // per-stage ci_adcensus only (stm_hip.h)
}

// aggregation H, V, V, H (d_ca_cross.cu:255-270 minus the transposes); result ends in `cost`
// scratch.base == nullptr: carved here when the vector-ALU kernels run (a plane slab of D * H * W floats)
static bool agg_on_matrix_pipe(int usd, int H, int W) { return (agg_variant() / 10000) % 10 != 1 && aggm_supports(usd, H, W); }
void core_agg(Vol cost, Vol scratch, const Arms &a, int D, int H, int W, int usd)
{
    // the frame pipeline's matrix-pipe kernels (round 3) -- unless the caller's volume holds infinities, NaNs or denormals
    if (agg_on_matrix_pipe(usd, H, W) && launch_aggm_stage(cost, cost, a.up, a.down, a.left, a.right, D, H, W, usd)) return;
    if (!scratch.base && !scratch.tab) scratch = vol_slab(Workspace::get<float>((size_t)D * H * W), (size_t)H * W);
    launch_agg_h(cost, scratch, a.left, a.right, D, H, W);
    launch_agg_v(scratch, cost, a.up, a.down, D, H, W, usd);
    launch_agg_v(cost, scratch, a.up, a.down, D, H, W, usd);
    launch_agg_h(scratch, cost, a.left, a.right, D, H, W);
}
Arms carve_arms(size_t HW)
{
    u8 *m = Workspace::get<u8>(4 * HW);
    return Arms{m, m + HW, m + 2 * HW, m + 3 * HW};
}
Arms arms_from_table(unsigned char **d_cross)
{
    // the reference hands the four plane pointers over as a DEVICE table (d_io.cu:94-101); fetch them
    u8 *h[4];
    STM_CHECK(hipMemcpyAsync(h, d_cross, sizeof h, hipMemcpyDeviceToHost, stream()));
    sync();
    return Arms{h[0], h[1], h[2], h[3]};
}

void core_bilateral(float *d_img, int radius, float sigma_color, float sigma_spatial, int H, int W, int D)
{
    size_t HW = (size_t)H * W;
    float *tmp = Workspace::get<float>(HW);
    launch_bilateral(d_img, tmp, gauss2d_table(radius, sigma_spatial), gauss1d_table(D, sigma_color), radius, H, W, D);
    STM_CHECK(hipMemcpyAsync(d_img, tmp, HW * sizeof(float), hipMemcpyDeviceToDevice, stream())); // d_filter_bilateral.cu:560
}

void core_dbm(u8 *d_out, const u8 *d_l, const u8 *d_r, const float *disp_l, const float *disp_r, const float *mask_l,
              const float *mask_r, float shift, int H, int W, int elem_sz, int g_radius, float g_sigma)
{
    size_t HW = (size_t)H * W;
    float *blend = Workspace::get<float>(HW);
    launch_gaussian_max(mask_r, blend, gauss2d_table(g_radius, g_sigma), g_radius, g_sigma, H, W, true); // G(1 - maskR)
    launch_view_synth(d_out, d_l, d_r, disp_l, disp_r, mask_l, mask_r, blend, shift, H, W, elem_sz);
}

void core_mux(const u8 *const *d_views, u8 *d_out, int N, float angle, int Hin, int Win, int Hout, int Wout, int elem_sz,
              int variant)
{
    float yi = mux_y_interval(N, angle, elem_sz);
    // tan(angle) = 0 divides by zero in the reference (d_mux_multiview.cu:146, SURVEY A-Q24); a period beyond the int
    // range would make the (int) conversion undefined
    if (!(fabsf(yi) < 1.0e9f)) {
        fail("mux_multiview: y_interval is not finite (tan(angle) == 0 or angle is not a number)", "angle", __FILE__, __LINE__);
        return;
    }
    int ymod = (int)roundf(yi);
    if (ymod == 0) {
        fail("mux_multiview: round(y_interval) == 0 (angle too steep)", "ymod", __FILE__, __LINE__);
        return;
    }
    launch_mux(d_views, d_out, N, yi, 1.0f / yi, ymod, Hin, Win, Hout, Wout, elem_sz, variant);
}

} // namespace

extern "C" {

// =============================================================== cost init
void stm_d_ci_adcensus(unsigned char *d_img_l, unsigned char *d_img_r, float **d_adcensus_cost_l,
                       float **d_adcensus_cost_r, float **h_adcensus_cost_l, float **h_adcensus_cost_r,
                       float *d_adcensus_cost_memory, float ad_coeff, float census_coeff, int num_disp, int zero_disp,
                       int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("d_ci_adcensus", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                                   {"elem_sz", elem_sz, 3}}))
        return;
    size_t HW = (size_t)num_rows * num_cols, V = HW * num_disp;
    Workspace::begin(4 * HW * 4 + 4096);
    for (int d = 0; d < num_disp; ++d) { // d_ci_adcensus.cu:150-157
        h_adcensus_cost_l[d] = d_adcensus_cost_memory + (size_t)d * HW;
        h_adcensus_cost_r[d] = d_adcensus_cost_memory + (size_t)d * HW + V;
    }
    STM_CHECK(hipMemcpyAsync(d_adcensus_cost_l, h_adcensus_cost_l, sizeof(float *) * num_disp, hipMemcpyHostToDevice, stream()));
    STM_CHECK(hipMemcpyAsync(d_adcensus_cost_r, h_adcensus_cost_r, sizeof(float *) * num_disp, hipMemcpyHostToDevice, stream()));
    uint32_t *pk_l = Workspace::get<uint32_t>(HW), *pk_r = Workspace::get<uint32_t>(HW);
    core_ci(d_img_l, d_img_r, vol_slab(d_adcensus_cost_memory, HW), vol_slab(d_adcensus_cost_memory + V, HW), pk_l, pk_r,
            ad_coeff, census_coeff, num_disp, zero_disp, num_rows, num_cols, elem_sz);
}

void stm_ci_adcensus(unsigned char *img_l, unsigned char *img_r, float **cost_l, float **cost_r, float ad_coeff,
                     float census_coeff, int num_disp, int zero_disp, int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("ci_adcensus", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                                 {"elem_sz", elem_sz, 3}}))
        return;
    size_t HW = (size_t)num_rows * num_cols, V = HW * num_disp;
    Workspace::begin(2 * V * 4 + 2 * HW * elem_sz + 4 * HW * 4 + 8192);
    u8 *dl = up(img_l, HW * elem_sz), *dr = up(img_r, HW * elem_sz);
    float *slab = Workspace::get<float>(2 * V);
    uint32_t *pk_l = Workspace::get<uint32_t>(HW), *pk_r = Workspace::get<uint32_t>(HW);
    core_ci(dl, dr, vol_slab(slab, HW), vol_slab(slab + V, HW), pk_l, pk_r, ad_coeff, census_coeff, num_disp, zero_disp,
            num_rows, num_cols, elem_sz);
    down_planes(cost_l, slab, num_disp, HW);
    down_planes(cost_r, slab + V, num_disp, HW);
    sync();
}

// =============================================================== aggregation
void stm_d_ca_cross(unsigned char *d_img, float **d_cost, float **d_acost, float **h_acost, float *d_acost_memory,
                    unsigned char **d_cross, float ucd, float lcd, int usd, int lsd, int num_disp, int num_rows,
                    int num_cols, int elem_sz)
{
    if (!args_ok("d_ca_cross", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                                {"elem_sz", elem_sz, 3}}))
        return;
    size_t HW = (size_t)num_rows * num_cols;
    // (what launch_aggm_stage carves is part of the hint: no allocation happens inside the call once the slab has this size)
    Workspace::begin(12 * HW + 8192 + (agg_on_matrix_pipe(usd, num_rows, num_cols) ? aggm_stage_bytes(num_disp, num_rows, num_cols, usd) : 0));
    for (int d = 0; d < num_disp; ++d) h_acost[d] = d_acost_memory + (size_t)d * HW; // d_ca_cross.cu:207-210
    STM_CHECK(hipMemcpyAsync(d_acost, h_acost, sizeof(float *) * num_disp, hipMemcpyHostToDevice, stream()));
    Arms a = arms_from_table(d_cross);
    uint32_t *pk = Workspace::get<uint32_t>(HW);
    launch_pack_bgrx(d_img, pk, num_rows, num_cols, elem_sz);
    launch_cross_arms(pk, a.up, a.down, a.left, a.right, ucd, lcd, usd, lsd, num_rows, num_cols);
    core_agg(vol_table(d_cost), vol_slab(d_acost_memory, HW), a, num_disp, num_rows, num_cols, usd); // result in d_cost (A-Q11)
}

void stm_ca_cross(unsigned char *img, unsigned char **cross, float **cost, float **acost, float ucd, float lcd, int usd,
                  int lsd, int num_disp, int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("ca_cross", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                              {"elem_sz", elem_sz, 3}}))
        return;
    size_t HW = (size_t)num_rows * num_cols, V = HW * num_disp;
    const bool mp = agg_on_matrix_pipe(usd, num_rows, num_cols);
    Workspace::begin(V * 4 + HW * elem_sz + 12 * HW + 16384 + (mp ? aggm_stage_bytes(num_disp, num_rows, num_cols, usd) : V * 4));
    u8 *dimg = up(img, HW * elem_sz);
    float *c = up_planes(cost, num_disp, HW);
    float *s = mp ? nullptr : Workspace::get<float>(V); // the vector-ALU kernels' scratch volume (carved late if they run as the fallback)
    Arms a = carve_arms(HW);
    uint32_t *pk = Workspace::get<uint32_t>(HW);
    launch_pack_bgrx(dimg, pk, num_rows, num_cols, elem_sz);
    launch_cross_arms(pk, a.up, a.down, a.left, a.right, ucd, lcd, usd, lsd, num_rows, num_cols);
    core_agg(vol_slab(c, HW), vol_slab(s, HW), a, num_disp, num_rows, num_cols, usd);
    down_planes(acost, c, num_disp, HW); // d_ca_cross.cu:419-422: the "cost" device buffer goes to acost
    down(cross[0], a.up, HW); down(cross[1], a.down, HW); down(cross[2], a.left, HW); down(cross[3], a.right, HW);
    sync();
}

// =============================================================== disparity selection
void stm_d_dc_wta(float **d_cost, float *d_disp, int num_disp, int zero_disp, int num_rows, int num_cols)
{
    if (!args_ok("d_dc_wta", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    launch_wta(vol_table(d_cost), d_disp, num_disp, zero_disp, num_rows, num_cols);
}
void stm_dc_wta(float **cost, float *disp, int num_disp, int zero_disp, int num_rows, int num_cols)
{
    if (!args_ok("dc_wta", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin((size_t)num_disp * HW * 4 + HW * 4 + 4096);
    float *c = up_planes(cost, num_disp, HW);
    float *d = Workspace::get<float>(HW);
    launch_wta(vol_slab(c, HW), d, num_disp, zero_disp, num_rows, num_cols);
    down(disp, d, HW);
    sync();
}

void stm_d_dc_hslo(float **d_cost, float *d_disp, unsigned char *d_img_l, unsigned char *d_img_r, float T, float H1,
                   float H2, int num_disp, int zero_disp, int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("d_dc_hslo", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                               {"elem_sz", elem_sz, 3}}))
        return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin((size_t)((num_disp + 3) / 4) * HW * 16 * 6 + 16 * HW + 16384);
    Vol c = vol_table(d_cost);
    const u8 *ia[1] = {d_img_l}, *ib[1] = {d_img_r};
    const int os[1] = {1};
    float *dv[1] = {d_disp};
    launch_hslo_wta(1, &c, ia, ib, os, dv, T, H1, H2, num_disp, zero_disp, num_rows, num_cols, elem_sz);
}
void stm_dc_hslo(float **cost, float *disp, unsigned char *img_l, unsigned char *img_r, float T, float H1, float H2,
                 int num_disp, int zero_disp, int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("dc_hslo", {{"num_disp", num_disp, 1}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                             {"elem_sz", elem_sz, 3}}))
        return;
    size_t HW = (size_t)num_rows * num_cols, V = HW * num_disp;
    Workspace::begin(V * 4 + (size_t)((num_disp + 3) / 4) * HW * 16 * 6 + 2 * HW * elem_sz + 5 * HW * 4 + 32768);
    float *c = up_planes(cost, num_disp, HW);
    u8 *dl = up(img_l, HW * elem_sz), *dr = up(img_r, HW * elem_sz);
    float *d = Workspace::get<float>(HW);
    Vol cv = vol_slab(c, HW);
    const u8 *ia[1] = {dl}, *ib[1] = {dr};
    const int os[1] = {1};
    float *dv[1] = {d};
    launch_hslo_wta(1, &cv, ia, ib, os, dv, T, H1, H2, num_disp, zero_disp, num_rows, num_cols, elem_sz);
    down(disp, d, HW);
    sync();
}

// =============================================================== refinement
void stm_d_dr_dcc(unsigned char *d_outliers_l, unsigned char *d_outliers_r, float *d_disp_l, float *d_disp_r, int num_rows,
                  int num_cols)
{
    if (!args_ok("d_dr_dcc", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(2 * HW + 1024);
    u8 *hl = Workspace::get<u8>(HW), *hr = Workspace::get<u8>(HW);
    launch_dcc(d_outliers_l, d_outliers_r, d_disp_l, d_disp_r, hl, hr, num_rows, num_cols);
}
void stm_dr_dcc(unsigned char *outliers_l, unsigned char *outliers_r, float *disp_l, float *disp_r, int num_rows,
                int num_cols)
{
    if (!args_ok("dr_dcc", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(12 * HW + 4096);
    float *dl = up(disp_l, HW), *dr = up(disp_r, HW);
    u8 *ol = Workspace::get<u8>(HW), *orr = Workspace::get<u8>(HW), *hl = Workspace::get<u8>(HW), *hr = Workspace::get<u8>(HW);
    STM_CHECK(hipMemsetAsync(ol, 0, HW, stream())); // d_dr_dcc.cu:166-171
    STM_CHECK(hipMemsetAsync(orr, 0, HW, stream()));
    launch_dcc(ol, orr, dl, dr, hl, hr, num_rows, num_cols);
    down(outliers_l, ol, HW); down(outliers_r, orr, HW);
    sync();
}

void stm_d_dr_irv(float *d_disp, unsigned char *d_outliers, unsigned char **d_cross, int thresh_s, float thresh_h,
                  int num_rows, int num_cols, int num_disp, int zero_disp, int usd, int iterations)
{
    if (!args_ok("d_dr_irv", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}, {"num_disp", num_disp, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(30 * HW + 65536);
    Arms a = arms_from_table(d_cross);
    float *dv[1] = {d_disp};
    u8 *ov[1] = {d_outliers};
    const u8 *u[1] = {a.up}, *d[1] = {a.down}, *l[1] = {a.left}, *r[1] = {a.right};
    launch_irv(1, dv, ov, u, d, l, r, thresh_s, thresh_h, num_rows, num_cols, num_disp, zero_disp, usd, iterations, true);
}
void stm_dr_irv(float *disp, unsigned char *outliers, unsigned char **cross, int thresh_s, float thresh_h, int num_rows,
                int num_cols, int num_disp, int zero_disp, int usd, int iterations)
{
    if (!args_ok("dr_irv", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}, {"num_disp", num_disp, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(40 * HW + 65536);
    float *d = up(disp, HW);
    u8 *o = up(outliers, HW);
    Arms a{up(cross[0], HW), up(cross[1], HW), up(cross[2], HW), up(cross[3], HW)};
    float *dv[1] = {d};
    u8 *ov[1] = {o};
    const u8 *uu[1] = {a.up}, *dd[1] = {a.down}, *ll[1] = {a.left}, *rr[1] = {a.right};
    launch_irv(1, dv, ov, uu, dd, ll, rr, thresh_s, thresh_h, num_rows, num_cols, num_disp, zero_disp, usd, iterations, false);
    down(disp, d, HW); down(outliers, o, HW);
    sync();
}

void stm_d_filter_bilateral_1(float *d_img, int radius, float sigma_color, float sigma_spatial, int num_rows, int num_cols,
                              int num_disp)
{
    if (!args_ok("d_filter_bilateral_1", {{"radius", radius, 0}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                                          {"num_disp", num_disp, 1}}))
        return;
    Workspace::begin((size_t)num_rows * num_cols * 4 + 1024);
    core_bilateral(d_img, radius, sigma_color, sigma_spatial, num_rows, num_cols, num_disp);
}
void stm_filter_bilateral_1(float *img, int radius, float sigma_color, float sigma_spatial, int num_rows, int num_cols,
                            int num_disp)
{
    if (!args_ok("filter_bilateral_1", {{"radius", radius, 0}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1},
                                        {"num_disp", num_disp, 1}}))
        return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(8 * HW + 4096);
    float *d = up(img, HW);
    core_bilateral(d, radius, sigma_color, sigma_spatial, num_rows, num_cols, num_disp);
    down(img, d, HW);
    sync();
}

void stm_d_filter_gaussian_1(float *d_img, int radius, float sigma_spatial, int num_rows, int num_cols)
{
    if (!args_ok("d_filter_gaussian_1", {{"radius", radius, 0}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(HW * 4 + 1024);
    float *tmp = Workspace::get<float>(HW);
    launch_gaussian_max(d_img, tmp, gauss2d_table(radius, sigma_spatial), radius, sigma_spatial, num_rows, num_cols, false);
    STM_CHECK(hipMemcpyAsync(d_img, tmp, HW * 4, hipMemcpyDeviceToDevice, stream())); // d_filter_gaussian.cu:171
}
void stm_filter_gaussian_1(float *img, int radius, float sigma_spatial, int num_rows, int num_cols)
{
    if (!args_ok("filter_gaussian_1", {{"radius", radius, 0}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(8 * HW + 4096);
    float *d = up(img, HW), *tmp = Workspace::get<float>(HW);
    launch_gaussian_max(d, tmp, gauss2d_table(radius, sigma_spatial), radius, sigma_spatial, num_rows, num_cols, false);
    down(img, tmp, HW);
    sync();
}

void stm_d_filter_bleed_1(unsigned char *d_img, int radius, int num_rows, int num_cols)
{
    if (!args_ok("d_filter_bleed_1", {{"radius", radius, 0}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(HW + 1024);
    u8 *tmp = Workspace::get<u8>(HW);
    launch_bleed(d_img, tmp, radius, num_rows, num_cols);
    STM_CHECK(hipMemcpyAsync(d_img, tmp, HW, hipMemcpyDeviceToDevice, stream())); // d_filter.cu:164
}
void stm_filter_bleed_1(unsigned char *img, int radius, int num_rows, int num_cols)
{
    if (!args_ok("filter_bleed_1", {{"radius", radius, 0}, {"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(2 * HW + 4096);
    u8 *d = up(img, HW), *tmp = Workspace::get<u8>(HW);
    launch_bleed(d, tmp, radius, num_rows, num_cols);
    down(img, tmp, HW);
    sync();
}

// d_filter.cu:47-103
void stm_d_filter_median(float *d_img, int num_rows, int num_cols)
{
    if (!args_ok("d_filter_median", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(HW * 4 + 1024);
    float *tmp = Workspace::get<float>(HW);
    launch_median3(d_img, tmp, num_rows, num_cols);
    STM_CHECK(hipMemcpyAsync(d_img, tmp, HW * 4, hipMemcpyDeviceToDevice, stream())); // d_filter.cu:67
}
void stm_filter_median(float *img, int num_rows, int num_cols)
{
    if (!args_ok("filter_median", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(8 * HW + 4096);
    float *d = up(img, HW), *tmp = Workspace::get<float>(HW);
    launch_median3(d, tmp, num_rows, num_cols);
    down(img, tmp, HW);
    sync();
}

// =============================================================== DIBR
void stm_d_dibr_occl(unsigned char *d_occl_l, unsigned char *d_occl_r, float *d_disp_l, float *d_disp_r, int num_rows,
                     int num_cols)
{
    if (!args_ok("d_dibr_occl", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    launch_occl(d_occl_l, d_occl_r, d_disp_l, d_disp_r, num_rows, num_cols);
}
void stm_dibr_occl(unsigned char *occl_l, unsigned char *occl_r, float *disp_l, float *disp_r, int num_rows, int num_cols)
{
    if (!args_ok("dibr_occl", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(10 * HW + 4096);
    float *dl = up(disp_l, HW), *dr = up(disp_r, HW);
    u8 *ol = Workspace::get<u8>(HW), *orr = Workspace::get<u8>(HW);
    launch_occl(ol, orr, dl, dr, num_rows, num_cols);
    down(occl_l, ol, HW); down(occl_r, orr, HW);
    sync();
}

void stm_d_dibr_occl_to_mask(float *d_mask_l, float *d_mask_r, unsigned char *d_occl_l, unsigned char *d_occl_r,
                             int num_rows, int num_cols)
{
    if (!args_ok("d_dibr_occl_to_mask", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    launch_occl_to_mask(d_mask_l, d_mask_r, d_occl_l, d_occl_r, num_rows, num_cols);
}
void stm_dibr_occl_to_mask(float *mask_l, float *mask_r, unsigned char *occl_l, unsigned char *occl_r, int num_rows,
                           int num_cols)
{
    if (!args_ok("dibr_occl_to_mask", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}})) return;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(10 * HW + 4096);
    u8 *ol = up(occl_l, HW), *orr = up(occl_r, HW);
    float *ml = Workspace::get<float>(HW), *mr = Workspace::get<float>(HW);
    launch_occl_to_mask(ml, mr, ol, orr, num_rows, num_cols);
    down(mask_l, ml, HW); down(mask_r, mr, HW);
    sync();
}

void stm_d_dibr_dbm(unsigned char *d_img_out, unsigned char *d_img_in_l, unsigned char *d_img_in_r, float *d_disp_l,
                    float *d_disp_r, unsigned char *d_occl_l, unsigned char *d_occl_r, float *d_mask_l, float *d_mask_r,
                    float shift, int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("d_dibr_dbm", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}, {"elem_sz", elem_sz, 3}})) return;
    (void)d_occl_l; (void)d_occl_r; // unused by the reference too (d_dibr_bwarp.cu:24-70)
    Workspace::begin((size_t)num_rows * num_cols * 4 + 1024);
    core_dbm(d_img_out, d_img_in_l, d_img_in_r, d_disp_l, d_disp_r, d_mask_l, d_mask_r, shift, num_rows, num_cols, elem_sz,
             10, 15.0f); // d_dibr_bwarp.cu:63
}
void stm_dibr_dbm(unsigned char *img_out, unsigned char *img_in_l, unsigned char *img_in_r, float *disp_l, float *disp_r,
                  unsigned char *occl_l, unsigned char *occl_r, float *mask_l, float *mask_r, float shift, int num_rows,
                  int num_cols, int elem_sz)
{
    if (!args_ok("dibr_dbm", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}, {"elem_sz", elem_sz, 3}})) return;
    (void)occl_l; (void)occl_r;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(3 * HW * elem_sz + 20 * HW + 8192);
    u8 *l = up(img_in_l, HW * elem_sz), *r = up(img_in_r, HW * elem_sz), *o = Workspace::get<u8>(HW * elem_sz);
    float *dl = up(disp_l, HW), *dr = up(disp_r, HW), *ml = up(mask_l, HW), *mr = up(mask_r, HW);
    core_dbm(o, l, r, dl, dr, ml, mr, shift, num_rows, num_cols, elem_sz, 7, 10.0f); // d_dibr_bwarp.cu:151
    down(img_out, o, HW * elem_sz);
    sync();
}

void stm_d_dibr_dfm(unsigned char *d_img_out, unsigned char *d_img_in_l, unsigned char *d_img_in_r, float *d_disp_l,
                    float *d_disp_r, float shift, int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("d_dibr_dfm", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}, {"elem_sz", elem_sz, 3}})) return;
    (void)d_img_in_r; (void)d_disp_r; // the right warp is computed and discarded in the reference (A-Q23)
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(HW * 8 + 1024);
    unsigned long long *keys = Workspace::get<unsigned long long>(HW);
    launch_fwarp(d_img_out, d_img_in_l, d_disp_l, shift, keys, num_rows, num_cols, elem_sz);
}
void stm_dibr_dfm(unsigned char *img_out, unsigned char *img_in_l, unsigned char *img_in_r, float *disp_l, float *disp_r,
                  float shift, int num_rows, int num_cols, int elem_sz)
{
    if (!args_ok("dibr_dfm", {{"num_rows", num_rows, 1}, {"num_cols", num_cols, 1}, {"elem_sz", elem_sz, 3}})) return;
    (void)img_in_r; (void)disp_r;
    size_t HW = (size_t)num_rows * num_cols;
    Workspace::begin(2 * HW * elem_sz + 12 * HW + 8192);
    u8 *l = up(img_in_l, HW * elem_sz), *o = Workspace::get<u8>(HW * elem_sz);
    float *dl = up(disp_l, HW);
    unsigned long long *keys = Workspace::get<unsigned long long>(HW);
    launch_fwarp(o, l, dl, shift, keys, num_rows, num_cols, elem_sz);
    down(img_out, o, HW * elem_sz);
    sync();
}

// =============================================================== mux
void stm_d_mux_multiview(unsigned char **d_views, unsigned char *d_out_data, int num_views, float angle, int in_rows,
                         int in_cols, int out_rows, int out_cols, int elem_sz)
{
    if (!args_ok("d_mux_multiview", {{"num_views", num_views, 2}, {"in_rows", in_rows, 1}, {"in_cols", in_cols, 1},
                                     {"out_rows", out_rows, 1}, {"out_cols", out_cols, 1}, {"elem_sz", elem_sz, 3}}))
        return;
    core_mux((const u8 *const *)d_views, d_out_data, num_views, angle, in_rows, in_cols, out_rows, out_cols, elem_sz, 2); // :148-151
}
void stm_mux_multiview(unsigned char **views, unsigned char *out_data, int num_views, float angle, int in_rows, int in_cols,
                       int out_rows, int out_cols, int elem_sz)
{
    if (!args_ok("mux_multiview", {{"num_views", num_views, 2}, {"in_rows", in_rows, 1}, {"in_cols", in_cols, 1},
                                   {"out_rows", out_rows, 1}, {"out_cols", out_cols, 1}, {"elem_sz", elem_sz, 3}}))
        return;
    size_t in_sz = (size_t)in_rows * in_cols * elem_sz, out_sz = (size_t)out_rows * out_cols * elem_sz;
    Workspace::begin(num_views * (in_sz + 256) + out_sz + 8192);
    std::vector<u8 *> h(num_views);
    for (int v = 0; v < num_views; ++v) h[v] = up(views[v], in_sz);
    u8 **dv = Workspace::get<u8 *>(num_views);
    STM_CHECK(hipMemcpyAsync(dv, h.data(), sizeof(u8 *) * num_views, hipMemcpyHostToDevice, stream()));
    sync(); // h goes out of scope below
    u8 *o = Workspace::get<u8>(out_sz);
    STM_CHECK(hipMemsetAsync(o, 0, out_sz, stream()));
    int variant = (out_rows % num_views == 0) ? 2 : 1; // d_mux_multiview.cu:184-192
    core_mux((const u8 *const *)dv, o, num_views, angle, in_rows, in_cols, out_rows, out_cols, elem_sz, variant);
    down(out_data, o, out_sz);
    sync();
}

void stm_d_demux_sbs(unsigned char *d_img_l, unsigned char *d_img_r, unsigned char *d_img_sbs, int num_rows,
                     int num_cols_sbs, int num_cols_out, int elem_sz)
{
    if (!args_ok("d_demux_sbs", {{"num_rows", num_rows, 1}, {"num_cols_sbs", num_cols_sbs, 1},
                                 {"num_cols_out", num_cols_out, 1}, {"elem_sz", elem_sz, 3}}))
        return;
    launch_demux_sbs(d_img_l, d_img_r, d_img_sbs, num_rows, num_cols_sbs, num_cols_out, elem_sz);
}

} // extern "C"

// device staging buffers of the blocking host-flavour frame calls: grow-only, one set per host thread and device (the
// reference's adcensus_stm allocates and frees them on every frame, d_io.cu:43-235)
namespace {
struct HostFrameBufs {
    void *p[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
};
constexpr int HFB_DEVS = 64;
thread_local HostFrameBufs g_hfb[HFB_DEVS];
// returns false (error recorded, nothing usable) if an allocation fails
bool host_frame_bufs(const size_t (&bytes)[4], void *(&out)[4])
{
    const int dev = cur_dev();
    if (dev < 0 || dev >= HFB_DEVS) {
        fail("adcensus_stm: device index out of range", "cur_dev", __FILE__, __LINE__);
        return false;
    }
    HostFrameBufs &b = g_hfb[dev];
    for (int i = 0; i < 4; ++i) {
        if (bytes[i] > b.cap[i]) {
            if (b.p[i]) {
                STM_CHECK(hipStreamSynchronize(stream()));
                STM_CHECK(hipFree(b.p[i]));
                b.p[i] = nullptr;
                b.cap[i] = 0;
            }
            void *q = nullptr;
            if (hipMalloc(&q, bytes[i]) != hipSuccess || !q) {
                (void)hipGetLastError();
                fail("adcensus_stm: device buffer allocation failed", "hipMalloc", __FILE__, __LINE__);
                return false;
            }
            b.p[i] = q;
            b.cap[i] = bytes[i];
        }
        out[i] = b.p[i];
    }
    return true;
}
} // namespace

// stm_release_workspace(): a thread that ends frees these together with its workspace
namespace stm {
void release_host_frame_bufs()
{
    int keep = 0;
    if (hipGetDevice(&keep) != hipSuccess) return;
    for (int dev = 0; dev < HFB_DEVS; ++dev) {
        HostFrameBufs &b = g_hfb[dev];
        bool any = false;
        for (int i = 0; i < 4; ++i) any = any || b.p[i];
        if (!any) continue;
        STM_CHECK(hipSetDevice(dev));
        STM_CHECK(hipDeviceSynchronize());
        for (int i = 0; i < 4; ++i) {
            if (b.p[i]) STM_CHECK(hipFree(b.p[i]));
            b.p[i] = nullptr;
            b.cap[i] = 0;
        }
    }
    STM_CHECK(hipSetDevice(keep));
}
} // namespace stm

// =============================================================== whole frame
// adcensus_stm, d_io.cu:7-238: demux -> cost init -> aggregation (L, R) -> WTA -> DCC -> IRV x5 ->
// bilateral(7,5,10) -> hit maps -> bleed(1) -> masks -> N-2 synthesised views -> interlace.
// Differences in mechanics (not in results): one cached workspace instead of ~35 cudaMalloc/cudaFree,
// no host synchronisation inside the frame, quad-interleaved volumes, the last aggregation pass is fused
// with WTA, both views share the arms / IRV launches, the mask blur G(1 - maskR) is computed once per frame
// instead of once per view (it does not depend on the view).
namespace {

// cost init .. WTA (.. DCC/IRV/bilateral when `refine`) on one rectified pair already split into L / R
// pre: optional {BGRX left, BGRX right, wide left, wide right} planes produced together with the split (full-resolution path)
void frame_disparity(u8 *img_l, u8 *img_r, float *d_disp_l, float *d_disp_r, Arms &al, Arms &ar, int H, int W, int elem_sz,
                     int D, int zero_disp, float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                     int thresh_s, float thresh_h, bool refine, bool hslo = false, uint32_t *const *pre = nullptr)
{
    const size_t HW = (size_t)H * W;
    const int NQ = (D + 3) / 4;
    const size_t V = HW * NQ * 4; // volumes are kept quad-interleaved (float4 [NQ][H][W]) inside the frame
    const bool matrix_pipe = (agg_variant() / 10000) % 10 != 1 && aggm_supports(usd, H, W); // default aggregation path: stm_kernels_aggm.hip
    float *cost = matrix_pipe ? nullptr : Workspace::get<float>(2 * V), *scratch = matrix_pipe ? nullptr : Workspace::get<float>(V);
    uint32_t *pk_l = pre ? pre[0] : Workspace::get<uint32_t>(HW), *pk_r = pre ? pre[1] : Workspace::get<uint32_t>(HW);
    Vol cl = vol_quads(cost, HW), cr = vol_quads(cost ? cost + V : nullptr, HW), sc = vol_quads(scratch, HW);
    // the first aggregation pass computes the initial costs itself (COST mode) and the 2 V of initial costs are never
    // written; only HSLO on the vector-ALU path and the per-stage API materialise them with stm_k_cost_init
    const bool cost_volume = hslo && !matrix_pipe;
    uint32_t *cen[2] = {nullptr, nullptr};
    core_ci(img_l, img_r, cl, cr, pk_l, pk_r, ad_coeff, census_coeff, D, zero_disp, H, W, elem_sz, pre != nullptr,
            cost_volume ? nullptr : cen);

    al = carve_arms(HW);
    ar = carve_arms(HW);
    uint32_t *htab = nullptr, *vtab = nullptr;
    int vrec = 0, vtop = -1;
    {
        const uint32_t *pk[2] = {pk_l, pk_r};
        u8 *u[2] = {al.up, ar.up}, *d[2] = {al.down, ar.down}, *l[2] = {al.left, ar.left}, *r[2] = {al.right, ar.right};
        const uint32_t *wide[2] = {pre ? pre[2] : nullptr, pre ? pre[3] : nullptr};
        // the last aggregation pass's horizontal window table comes out of the same kernel (it has the arms in registers)
        const size_t hw_dw = matrix_pipe ? aggm_frame_htab_dwords(D, H, W, usd, hslo) : 0;
        htab = hw_dw ? Workspace::get<uint32_t>(hw_dw) : nullptr;
        const size_t vw_dw = matrix_pipe && htab ? aggm_frame_vtab_dwords(H, W, usd, &vrec, &vtop) : 0; // (both tables or the horizontal one alone)
        vtab = vw_dw ? Workspace::get<uint32_t>(vw_dw) : nullptr;
        launch_cross_arms2(2, pk, u, d, l, r, ucd, lcd, usd, lsd, H, W, pre ? wide : nullptr, htab, vtab, vrec, vtop);
    }
    // with refinement the raw WTA maps live in scratch and the bilateral filter, the last step, writes the caller's buffers
    float *wl = refine ? Workspace::get<float>(HW) : d_disp_l, *wr = refine ? Workspace::get<float>(HW) : d_disp_r;
    // HSLO = Mei et al. 3.3: scanline optimisation of the aggregated cost, then WTA.  Penalty constants: the values the
    // reference's (commented-out) test call uses, image_io.cpp:311-313.  Parity unpinned.
    const u8 *hs_a[2] = {img_l, img_r}, *hs_b[2] = {img_r, img_l}; // the right view's own image plays "left"
    const int hs_sign[2] = {1, -1};
    if (matrix_pipe) {
        // the aggregation kernels on the matrix pipe: cost -> H -> V, V -> H (+ WTA, or + the HSLO passes on the volume), two
        // PQ-layout volumes per view
        const size_t VP = pq_volume_floats(D, H, W);
        float *m = Workspace::get<float>(4 * VP);
        float *va[2] = {m, m + VP}, *vb[2] = {m + 2 * VP, m + 3 * VP};
        const uint32_t *pk[2] = {pk_l, pk_r}, *cn[2] = {cen[0], cen[1]};
        const u8 *u[2] = {al.up, ar.up}, *d[2] = {al.down, ar.down}, *l[2] = {al.left, ar.left}, *r[2] = {al.right, ar.right};
        float *dv[2] = {wl, wr};
        launch_aggm_frame(pk, cn, rho_table(ad_coeff, census_coeff), va, vb, u, d, l, r, dv, D, zero_disp, H, W, usd, hslo, htab, vtab);
        if (hslo) launch_hslo_wta_pq(2, vb, va, hs_a, hs_b, hs_sign, dv, 15.0f, 1.0f, 3.0f, D, zero_disp, H, W, elem_sz);
    } else if (hslo) {
        core_agg(cl, sc, al, D, H, W, usd);
        core_agg(cr, sc, ar, D, H, W, usd);
        const Vol cv[2] = {cl, cr};
        float *dv[2] = {wl, wr};
        launch_hslo_wta(2, cv, hs_a, hs_b, hs_sign, dv, 15.0f, 1.0f, 3.0f, D, zero_disp, H, W, elem_sz);
    } else {
        // legacy (stm_set_agg_variant(10000)): H, V, V per view on the vector ALU, then the last H pass + WTA of both views in one launch.  After three passes a view's
        // data sits in its scratch volume; the right view uses the left view's (now free) cost volume as scratch.
        float *scratch2 = Workspace::get<float>(V);
        Vol s2 = vol_quads(scratch2, HW);
        launch_agg_h2_cost(pk_l, cen[0], pk_r, cen[1], rho_table(ad_coeff, census_coeff), sc, al.left, al.right, s2, ar.left, ar.right,
                           D, zero_disp, H, W);
        launch_agg_v(sc, cl, al.up, al.down, D, H, W, usd);
        launch_agg_v(s2, cr, ar.up, ar.down, D, H, W, usd);
        launch_agg_v(cl, sc, al.up, al.down, D, H, W, usd);
        launch_agg_v(cr, s2, ar.up, ar.down, D, H, W, usd);
        launch_agg_h_wta2(sc, al.left, al.right, wl, s2, ar.left, ar.right, wr, D, zero_disp, H, W);
    }
    if (!refine) return;

    u8 *outl_l = Workspace::get<u8>(HW), *outl_r = Workspace::get<u8>(HW);
    launch_dcc_rows(outl_l, outl_r, wl, wr, H, W); // d_io.cu:138-143 (outlier maps zeroed, dr_dcc)
    {   // d_io.cu:147-148, both views per launch
        float *dv[2] = {wl, wr};
        u8 *ov[2] = {outl_l, outl_r};
        const u8 *u[2] = {al.up, ar.up}, *d[2] = {al.down, ar.down}, *l[2] = {al.left, ar.left}, *r[2] = {al.right, ar.right};
        launch_irv(2, dv, ov, u, d, l, r, thresh_s, thresh_h, H, W, D, zero_disp, usd, 5, true);
    }
    // the maps are this pipeline's own WTA / region-voting output: integer-valued, any two of them differ by at most D - 1
    launch_bilateral2(wl, d_disp_l, wr, d_disp_r, gauss2d_table(7, 10.0f), gauss1d_table(D, 5.0f), 7, H, W, D, true, // :150-151  (7, 5, 10)
                      bilateral_one_value_table(D, zero_disp, 10.0f, 5.0f), zero_disp);
}

// hit maps -> bleed -> masks -> N-2 views -> interlace (d_io.cu:160-205)
void frame_render(u8 *img_l, u8 *img_r, float *d_disp_l, float *d_disp_r, u8 *d_interlaced, int H, int W, int Hout, int Wout,
                  int elem_sz, int N, float angle)
{
    const size_t HW = (size_t)H * W, IMG = HW * elem_sz;
    float *mask_l = Workspace::get<float>(HW), *mask_r = Workspace::get<float>(HW), *blend = Workspace::get<float>(HW);
    launch_hitmask_rows(mask_l, mask_r, d_disp_l, d_disp_r, H, W); // :165-176: dibr_occl, bleed(1) x2, occl_to_mask
    launch_gaussian_max(mask_r, blend, gauss2d_table(10, 15.0f), 10, 15.0f, H, W, true); // d_dibr_bwarp.cu:60-63, once per frame

    if ((agg_variant() / 100) % 10 != 2) {
        // views + interlacing in one pass: an output pixel synthesises exactly the samples it interlaces (stm_k_synth_mux)
        const float yi = mux_y_interval(N, angle, elem_sz);
        if (!(fabsf(yi) < 1.0e9f)) {
            fail("mux_multiview: y_interval is not finite (tan(angle) == 0 or angle is not a number)", "angle", __FILE__, __LINE__);
            return;
        }
        const int ymod = (int)roundf(yi);
        if (ymod == 0) {
            fail("mux_multiview: round(y_interval) == 0 (angle too steep)", "ymod", __FILE__, __LINE__);
            return;
        }
        launch_synth_mux(img_l, img_r, d_disp_l, d_disp_r, mask_l, mask_r, blend, d_interlaced, N, yi, 1.0f / yi, ymod, H, W, Hout, Wout,
                         elem_sz, 2);
        return;
    }
    // 200: the un-fused form (every view written, then interlaced), as the reference structures it (d_io.cu:182-203)
    u8 *views_mem = Workspace::get<u8>((size_t)N * IMG);
    // views[0] = right image, views[N-1] = left image (d_io.cu:182-183)
    launch_view_synth_all(views_mem, IMG, N, img_l, img_r, d_disp_l, d_disp_r, mask_l, mask_r, blend, H, W, elem_sz); // :186-201
    // view table built on the device (no host memory involved, so nothing to keep alive or synchronise)
    u8 **dv = Workspace::get<u8 *>(N);
    launch_view_table(dv, img_r, img_l, views_mem, IMG, N);
    core_mux((const u8 *const *)dv, d_interlaced, N, angle, H, W, Hout, Wout, elem_sz, 2); // :203
}

} // namespace

extern "C" {

void stm_d_adcensus_stm(unsigned char *d_img_sbs, float *d_disp_l, float *d_disp_r, unsigned char *d_interlaced,
                        int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out, int elem_sz,
                        int num_views, float angle, int num_disp, int zero_disp, float ad_coeff, float census_coeff,
                        float ucd, float lcd, int usd, int lsd, int thresh_s, float thresh_h, int stages)
{
    if (!args_ok("d_adcensus_stm", {{"num_rows", num_rows, 1}, {"num_cols_sbs", num_cols_sbs, 1},
                                    {"num_cols", num_cols, 1}, {"num_rows_out", num_rows_out, 1},
                                    {"num_cols_out", num_cols_out, 1}, {"elem_sz", elem_sz, 3},
                                    {"num_views", num_views, 2}, {"num_disp", num_disp, 1}}))
        return;
    const int H = num_rows, W = num_cols, N = num_views;
    const size_t HW = (size_t)H * W, IMG = HW * elem_sz;
    const size_t V = pq_volume_floats(num_disp, H, W); // >= the quad-interleaved volume of the HSLO / legacy paths
    Workspace::begin(((stages & 0x100) ? 13 : 4) * V * 4 + (size_t)(N + 2) * IMG + 168 * HW + (1u << 20));
    u8 *img_l = Workspace::get<u8>(IMG), *img_r = Workspace::get<u8>(IMG);
    uint32_t *pre[4] = {nullptr, nullptr, nullptr, nullptr};
    const bool fused_split = num_cols_sbs >= 2 * W; // both halves complete: emit the derived pixel formats in the same pass
    if (fused_split) {
        for (int i = 0; i < 4; ++i) pre[i] = Workspace::get<uint32_t>(HW);
        launch_demux_sbs_packed(img_l, img_r, pre[0], pre[1], pre[2], pre[3], d_img_sbs, H, num_cols_sbs, W, elem_sz);
    } else {
        launch_demux_sbs(img_l, img_r, d_img_sbs, H, num_cols_sbs, W, elem_sz);
    }
    Arms al, ar;
    const bool hslo = (stages & 0x100) != 0; // + scanline optimisation between aggregation and WTA (BASELINE config 3)
    stages &= 0xff;
    frame_disparity(img_l, img_r, d_disp_l, d_disp_r, al, ar, H, W, elem_sz, num_disp, zero_disp, ad_coeff, census_coeff, ucd,
                    lcd, usd, lsd, thresh_s, thresh_h, stages >= 2, hslo, fused_split ? pre : nullptr);
    if (stages < 3) return;
    frame_render(img_l, img_r, d_disp_l, d_disp_r, d_interlaced, H, W, num_rows_out, num_cols_out, elem_sz, N, angle);
}

// adcensus_stm_2, d_io.cu:240-508: the disparity is computed on a bilinearly reduced pair
// (num_rows_disp x num_cols_disp, tx_scale_bilinear_kernel :302-304), scaled back up with
// tx_disp_scale_kernel(1/disp_scale) (:415-417), then the views are rendered at full resolution.
void stm_d_adcensus_stm_2(unsigned char *d_img_sbs, float *d_disp_l, float *d_disp_r, unsigned char *d_interlaced,
                          int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out,
                          int num_rows_disp, int num_cols_disp, int elem_sz, float disp_scale, int num_views, float angle,
                          int num_disp, int zero_disp, float ad_coeff, float census_coeff, float ucd, float lcd, int usd,
                          int lsd, int thresh_s, float thresh_h)
{
    if (!args_ok("d_adcensus_stm_2", {{"num_rows", num_rows, 1}, {"num_cols_sbs", num_cols_sbs, 1},
                                      {"num_cols", num_cols, 1}, {"num_rows_out", num_rows_out, 1},
                                      {"num_cols_out", num_cols_out, 1}, {"num_rows_disp", num_rows_disp, 1},
                                      {"num_cols_disp", num_cols_disp, 1}, {"elem_sz", elem_sz, 3},
                                      {"num_views", num_views, 2}, {"num_disp", num_disp, 1}}))
        return;
    const int H = num_rows, W = num_cols, h = num_rows_disp, w = num_cols_disp, N = num_views;
    const size_t HW = (size_t)H * W, IMG = HW * elem_sz, hw = (size_t)h * w;
    const size_t V = pq_volume_floats(num_disp, h, w);
    Workspace::begin(4 * V * 4 + (size_t)(N + 4) * IMG + 136 * HW + 8 * hw + (1u << 20));
    u8 *img_l = Workspace::get<u8>(IMG), *img_r = Workspace::get<u8>(IMG);
    launch_demux_sbs(img_l, img_r, d_img_sbs, H, num_cols_sbs, W, elem_sz);
    u8 *low_l = Workspace::get<u8>(hw * elem_sz), *low_r = Workspace::get<u8>(hw * elem_sz);
    launch_scale_bilinear(img_l, low_l, H, W, h, w, elem_sz);
    launch_scale_bilinear(img_r, low_r, H, W, h, w, elem_sz);
    float *low_dl = Workspace::get<float>(hw), *low_dr = Workspace::get<float>(hw);
    Arms al, ar;
    frame_disparity(low_l, low_r, low_dl, low_dr, al, ar, h, w, elem_sz, num_disp, zero_disp, ad_coeff, census_coeff, ucd, lcd,
                    usd, lsd, thresh_s, thresh_h, true);
    const float up = 1.0f / disp_scale; // :415
    launch_disp_scale(d_disp_l, low_dl, H, W, h, w, up);
    launch_disp_scale(d_disp_r, low_dr, H, W, h, w, up);
    frame_render(img_l, img_r, d_disp_l, d_disp_r, d_interlaced, H, W, num_rows_out, num_cols_out, elem_sz, N, angle);
}

void stm_adcensus_stm_2(unsigned char *img_sbs, float *disp_l, float *disp_r, unsigned char *interlaced, int num_rows,
                        int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out, int num_rows_disp,
                        int num_cols_disp, int elem_sz, float disp_scale, int num_views, float angle, int num_disp,
                        int zero_disp, float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                        int thresh_s, float thresh_h)
{
    if (!args_ok("adcensus_stm_2", {{"num_rows", num_rows, 1}, {"num_cols_sbs", num_cols_sbs, 1},
                                    {"num_cols", num_cols, 1}, {"num_rows_out", num_rows_out, 1},
                                    {"num_cols_out", num_cols_out, 1}, {"num_rows_disp", num_rows_disp, 1},
                                    {"num_cols_disp", num_cols_disp, 1}, {"elem_sz", elem_sz, 3},
                                    {"num_views", num_views, 2}, {"num_disp", num_disp, 1}}))
        return;
    size_t HW = (size_t)num_rows * num_cols, sbs_sz = (size_t)num_rows * num_cols_sbs * elem_sz;
    size_t out_sz = (size_t)num_rows_out * num_cols_out * elem_sz;
    const size_t need[4] = {sbs_sz, out_sz, HW * 4, HW * 4};
    void *buf[4];
    if (!host_frame_bufs(need, buf)) return;
    u8 *d_sbs = (u8 *)buf[0], *d_out = (u8 *)buf[1];
    float *d_dl = (float *)buf[2], *d_dr = (float *)buf[3];
    STM_CHECK(hipMemcpyAsync(d_sbs, img_sbs, sbs_sz, hipMemcpyHostToDevice, stream()));
    STM_CHECK(hipMemsetAsync(d_out, 0, out_sz, stream()));
    ApiNest nest; // the device flavour must not forget a failed upload
    stm_d_adcensus_stm_2(d_sbs, d_dl, d_dr, d_out, num_rows, num_cols_sbs, num_cols, num_rows_out, num_cols_out, num_rows_disp,
                         num_cols_disp, elem_sz, disp_scale, num_views, angle, num_disp, zero_disp, ad_coeff, census_coeff, ucd,
                         lcd, usd, lsd, thresh_s, thresh_h);
    down(disp_l, d_dl, HW); down(disp_r, d_dr, HW); down(interlaced, d_out, out_sz);
    sync();
}

// d_tx_scale.h:17-18  d_tx_scale (d_tx_scale.cu:83-121): despite the d_ prefix it takes HOST pointers
void stm_d_tx_scale(unsigned char *img_in, unsigned char *img_out, int in_rows, int in_cols, int out_rows, int out_cols,
                    int elem_sz)
{
    if (!args_ok("d_tx_scale", {{"in_rows", in_rows, 1}, {"in_cols", in_cols, 1}, {"out_rows", out_rows, 1},
                                {"out_cols", out_cols, 1}, {"elem_sz", elem_sz, 3}}))
        return;
    size_t in_sz = (size_t)in_rows * in_cols * elem_sz, out_sz = (size_t)out_rows * out_cols * elem_sz;
    Workspace::begin(in_sz + out_sz + 4096);
    u8 *di = up(img_in, in_sz), *dout = Workspace::get<u8>(out_sz);
    launch_scale_bilinear(di, dout, in_rows, in_cols, out_rows, out_cols, elem_sz);
    down(img_out, dout, out_sz);
    sync();
}

void stm_generate_gaussian_kernel(float *kernel, int radius, float sigma)
{
    if (!args_ok("generate_gaussian_kernel", {{"radius", radius, 0}})) return;
    gaussian_kernel_2d(kernel, radius, sigma);
}

void stm_adcensus_stm(unsigned char *img_sbs, float *disp_l, float *disp_r, unsigned char *interlaced, int num_rows,
                      int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out, int elem_sz, int num_views,
                      float angle, int num_disp, int zero_disp, float ad_coeff, float census_coeff, float ucd, float lcd,
                      int usd, int lsd, int thresh_s, float thresh_h)
{
    if (!args_ok("adcensus_stm", {{"num_rows", num_rows, 1}, {"num_cols_sbs", num_cols_sbs, 1}, {"num_cols", num_cols, 1},
                                  {"num_rows_out", num_rows_out, 1}, {"num_cols_out", num_cols_out, 1},
                                  {"elem_sz", elem_sz, 3}, {"num_views", num_views, 2}, {"num_disp", num_disp, 1}}))
        return;
    size_t HW = (size_t)num_rows * num_cols, sbs_sz = (size_t)num_rows * num_cols_sbs * elem_sz;
    size_t out_sz = (size_t)num_rows_out * num_cols_out * elem_sz;
    // own buffers are cached outside the workspace: the pipeline call below re-carves the workspace
    const size_t need[4] = {sbs_sz, out_sz, HW * 4, HW * 4};
    void *buf[4];
    if (!host_frame_bufs(need, buf)) return;
    u8 *d_sbs = (u8 *)buf[0], *d_out = (u8 *)buf[1];
    float *d_dl = (float *)buf[2], *d_dr = (float *)buf[3];
    STM_CHECK(hipMemcpyAsync(d_sbs, img_sbs, sbs_sz, hipMemcpyHostToDevice, stream()));
    STM_CHECK(hipMemsetAsync(d_out, 0, out_sz, stream()));
    ApiNest nest; // the device flavour must not forget a failed upload
    stm_d_adcensus_stm(d_sbs, d_dl, d_dr, d_out, num_rows, num_cols_sbs, num_cols, num_rows_out, num_cols_out, elem_sz,
                       num_views, angle, num_disp, zero_disp, ad_coeff, census_coeff, ucd, lcd, usd, lsd, thresh_s, thresh_h, 3);
    down(disp_l, d_dl, HW); down(disp_r, d_dr, HW); down(interlaced, d_out, out_sz);
    sync();
}

} // extern "C"
