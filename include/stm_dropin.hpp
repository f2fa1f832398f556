// stm_dropin.hpp -- the reference's per-stage host API under its ORIGINAL C++ names and signatures,
// implemented by libstm_hip.so.  A caller written against the reference's headers (image_io.cpp:171-292,
// d_io.cu:74-203) compiles and links unchanged against this header + libstm_hip.so: the mangled names
// are identical because the argument lists are identical.  Each function forwards to its stm_* C twin
// (include/stm_hip.h), which documents semantics and cites the reference line by line.
//
// All 33 reference names are exported with the reference's exact signatures (tests/test_abi.py holds the mangled names).
// adcensus_stm / adcensus_stm_2 keep the reference's `int angle` (d_io.h:36,48): the float its only caller passes
// (video_io.cpp:158) is truncated at the call site exactly as upstream (SURVEY A-Q24); adcensus_stm_f / adcensus_stm_2_f
// are additions that keep the fractional angle.
#ifndef STM_DROPIN_HPP
#define STM_DROPIN_HPP

#if defined(STM_BUILD) && defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

// d_ci_adcensus.h:16-25
void d_ci_adcensus(unsigned char* d_img_l, unsigned char* d_img_r,
                   float** d_adcensus_cost_l, float** d_adcensus_cost_r,
                   float** h_adcensus_cost_l, float** h_adcensus_cost_r,
                   float* d_adcensus_cost_memory,
                   float ad_coeff, float census_coeff, int num_disp, int zero_disp,
                   int num_rows, int num_cols, int elem_sz);
void ci_adcensus(unsigned char* img_l, unsigned char* img_r, float** cost_l, float** cost_r,
                 float ad_coeff, float census_coeff, int num_disp, int zero_disp,
                 int num_rows, int num_cols, int elem_sz);
// d_ca_cross.h:13-21
void d_ca_cross(unsigned char* d_img, float** d_cost,
                float** d_acost, float** h_acost, float* d_acost_memory,
                unsigned char** d_cross,
                float ucd, float lcd, int usd, int lsd,
                int num_disp, int num_rows, int num_cols, int elem_sz);
void ca_cross(unsigned char* img, unsigned char** cross, float** cost, float** acost,
              float ucd, float lcd, int usd, int lsd,
              int num_disp, int num_rows, int num_cols, int elem_sz);
// d_dc_wta.h:12-18
void d_dc_wta(float** d_cost, float* d_disp, int num_disp, int zero_disp, int num_rows, int num_cols);
void dc_wta(float** cost, float* disp, int num_disp, int zero_disp, int num_rows, int num_cols);
// d_dc_hslo.h:18-22
void dc_hslo(float** cost, float* disp, unsigned char* img_l, unsigned char* img_r,
             float T, float H1, float H2, int num_disp, int zero_disp,
             int num_rows, int num_cols, int elem_sz);
// d_dr_dcc.h:13-19
void d_dr_dcc(unsigned char* d_outliers_l, unsigned char* d_outliers_r, float* d_disp_l, float* d_disp_r,
              int num_rows, int num_cols);
void dr_dcc(unsigned char* outliers_l, unsigned char* outliers_r, float* disp_l, float* disp_r,
            int num_rows, int num_cols);
// d_dr_irv.h:8-19
void d_dr_irv(float* d_disp, unsigned char* d_outliers, unsigned char** d_cross,
              int thresh_s, float thresh_h, int num_rows, int num_cols, int num_disp, int zero_disp,
              int usd, int iterations);
void dr_irv(float* disp, unsigned char* outliers, unsigned char** cross,
            int thresh_s, float thresh_h, int num_rows, int num_cols, int num_disp, int zero_disp,
            int usd, int iterations);
// d_filter_bilateral.h:13-20
void d_filter_bilateral_1(float* d_img, int radius, float sigma_color, float sigma_spatial,
                          int num_rows, int num_cols, int num_disp);
void filter_bilateral_1(float* img, int radius, float sigma_color, float sigma_spatial,
                        int num_rows, int num_cols, int num_disp);
// d_filter_gaussian.h:20-26
void filter_gaussian_1(float* img, int radius, float sigma_spatial, int num_rows, int num_cols);
void d_filter_gaussian_1(float* d_img, int radius, float sigma_spatial, int num_rows, int num_cols);
// d_filter.h:22-28
void d_filter_bleed_1(unsigned char* d_img, int radius, int num_rows, int num_cols);
void filter_bleed_1(unsigned char* img, int radius, int num_rows, int num_cols);
// d_filter.h:11-16
void filter_median(float* img, int num_rows, int num_cols);
void d_filter_median(float* d_img_in, int num_rows, int num_cols);
// d_dibr_occl.h:14-33
void d_dibr_occl_to_mask(float* d_mask_l, float* d_mask_r, unsigned char* d_occl_l, unsigned char* d_occl_r,
                         int num_rows, int num_cols);
void dibr_occl_to_mask(float* mask_l, float* mask_r, unsigned char* occl_l, unsigned char* occl_r,
                       int num_rows, int num_cols);
void d_dibr_occl(unsigned char* d_occl_l, unsigned char* d_occl_r, float* d_disp_l, float* d_disp_r,
                 int num_rows, int num_cols);
void dibr_occl(unsigned char* occl_l, unsigned char* occl_r, float* disp_l, float* disp_r,
               int num_rows, int num_cols);
// d_dibr_fwarp.h:12-20
void d_dibr_dfm(unsigned char* d_img_out, unsigned char* d_img_in_l, unsigned char* d_img_in_r,
                float* disp_l, float* disp_r, float shift, int num_rows, int num_cols, int elem_sz);
void dibr_dfm(unsigned char* img_out, unsigned char* img_in_l, unsigned char* img_in_r,
              float* disp_l, float* disp_r, float shift, int num_rows, int num_cols, int elem_sz);
// d_dibr_bwarp.h:22-34
void d_dibr_dbm(unsigned char* d_img_out, unsigned char* d_img_in_l, unsigned char* d_img_in_r,
                float* d_disp_l, float* d_disp_r, unsigned char* d_occl_l, unsigned char* d_occl_r,
                float* d_mask_l, float* d_mask_r, float shift, int num_rows, int num_cols, int elem_sz);
void dibr_dbm(unsigned char* img_out, unsigned char* img_in_l, unsigned char* img_in_r,
              float* disp_l, float* disp_r, unsigned char* occl_l, unsigned char* occl_r,
              float* mask_l, float* mask_r, float shift, int num_rows, int num_cols, int elem_sz);
// d_mux_multiview.h:35-41
void d_mux_multiview(unsigned char** d_views, unsigned char* d_out_data, int num_views, float angle,
                     int in_rows, int in_cols, int out_rows, int out_cols, int elem_sz);
void mux_multiview(unsigned char** views, unsigned char* out_data, int num_views, float angle,
                   int in_rows, int in_cols, int out_rows, int out_cols, int elem_sz);
// d_io.h:32-40.  `int angle` exactly as the reference declares it: a caller that passes a float (video_io.cpp:158 does) has
// it truncated at the call site, as upstream (SURVEY A-Q24).  adcensus_stm_f keeps the fractional angle.
void adcensus_stm(unsigned char* img_sbs, float* disp_l, float* disp_r, unsigned char* interlaced,
                  int num_rows, int num_cols_sbs, int num_cols,
                  int num_rows_out, int num_cols_out, int elem_sz,
                  int num_views, int angle, int num_disp, int zero_disp,
                  float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                  int thresh_s, float thresh_h);
void adcensus_stm_f(unsigned char* img_sbs, float* disp_l, float* disp_r, unsigned char* interlaced,
                    int num_rows, int num_cols_sbs, int num_cols,
                    int num_rows_out, int num_cols_out, int elem_sz,
                    int num_views, float angle, int num_disp, int zero_disp,
                    float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                    int thresh_s, float thresh_h);
// d_io.h:42-52 (same `int angle`)
void adcensus_stm_2(unsigned char* img_sbs, float* disp_l, float* disp_r, unsigned char* interlaced,
                    int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out,
                    int num_rows_disp, int num_cols_disp, int elem_sz, float disp_scale,
                    int num_views, int angle, int num_disp, int zero_disp,
                    float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                    int thresh_s, float thresh_h);
void adcensus_stm_2_f(unsigned char* img_sbs, float* disp_l, float* disp_r, unsigned char* interlaced,
                      int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out,
                      int num_rows_disp, int num_cols_disp, int elem_sz, float disp_scale,
                      int num_views, float angle, int num_disp, int zero_disp,
                      float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                      int thresh_s, float thresh_h);
// d_tx_scale.h:19-20 (host pointers, despite the prefix: d_tx_scale.cu:82-127)
void d_tx_scale(unsigned char* in_data, unsigned char* out_data, int in_rows, int in_cols, int out_rows, int out_cols,
                int elem_sz);
// d_filter_gaussian.h:30
void generateGaussianKernel(float* kernel, int radius, float sigma);

#if defined(STM_BUILD) && defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif // STM_DROPIN_HPP
