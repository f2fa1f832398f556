// stm_dropin.hip -- C++-linkage forwards under the reference's own names (include/stm_dropin.hpp).
#include "../../include/stm_dropin.hpp"
#include "../../include/stm_hip.h"

void d_ci_adcensus(unsigned char* a, unsigned char* b, float** c, float** d, float** e, float** f, float* g,
                   float h, float i, int j, int k, int l, int m, int n)
{ stm_d_ci_adcensus(a, b, c, d, e, f, g, h, i, j, k, l, m, n); }
void ci_adcensus(unsigned char* a, unsigned char* b, float** c, float** d, float e, float f, int g, int h, int i, int j, int k)
{ stm_ci_adcensus(a, b, c, d, e, f, g, h, i, j, k); }
void d_ca_cross(unsigned char* a, float** b, float** c, float** d, float* e, unsigned char** f,
                float g, float h, int i, int j, int k, int l, int m, int n)
{ stm_d_ca_cross(a, b, c, d, e, f, g, h, i, j, k, l, m, n); }
void ca_cross(unsigned char* a, unsigned char** b, float** c, float** d, float e, float f, int g, int h, int i, int j, int k, int l)
{ stm_ca_cross(a, b, c, d, e, f, g, h, i, j, k, l); }
void d_dc_wta(float** a, float* b, int c, int d, int e, int f) { stm_d_dc_wta(a, b, c, d, e, f); }
void dc_wta(float** a, float* b, int c, int d, int e, int f) { stm_dc_wta(a, b, c, d, e, f); }
void dc_hslo(float** a, float* b, unsigned char* c, unsigned char* d, float e, float f, float g, int h, int i, int j, int k, int l)
{ stm_dc_hslo(a, b, c, d, e, f, g, h, i, j, k, l); }
void d_dr_dcc(unsigned char* a, unsigned char* b, float* c, float* d, int e, int f) { stm_d_dr_dcc(a, b, c, d, e, f); }
void dr_dcc(unsigned char* a, unsigned char* b, float* c, float* d, int e, int f) { stm_dr_dcc(a, b, c, d, e, f); }
void d_dr_irv(float* a, unsigned char* b, unsigned char** c, int d, float e, int f, int g, int h, int i, int j, int k)
{ stm_d_dr_irv(a, b, c, d, e, f, g, h, i, j, k); }
void dr_irv(float* a, unsigned char* b, unsigned char** c, int d, float e, int f, int g, int h, int i, int j, int k)
{ stm_dr_irv(a, b, c, d, e, f, g, h, i, j, k); }
void d_filter_bilateral_1(float* a, int b, float c, float d, int e, int f, int g) { stm_d_filter_bilateral_1(a, b, c, d, e, f, g); }
void filter_bilateral_1(float* a, int b, float c, float d, int e, int f, int g) { stm_filter_bilateral_1(a, b, c, d, e, f, g); }
void filter_gaussian_1(float* a, int b, float c, int d, int e) { stm_filter_gaussian_1(a, b, c, d, e); }
void d_filter_gaussian_1(float* a, int b, float c, int d, int e) { stm_d_filter_gaussian_1(a, b, c, d, e); }
void d_filter_bleed_1(unsigned char* a, int b, int c, int d) { stm_d_filter_bleed_1(a, b, c, d); }
void filter_bleed_1(unsigned char* a, int b, int c, int d) { stm_filter_bleed_1(a, b, c, d); }
void filter_median(float* a, int b, int c) { stm_filter_median(a, b, c); }
void d_filter_median(float* a, int b, int c) { stm_d_filter_median(a, b, c); }
void d_dibr_occl_to_mask(float* a, float* b, unsigned char* c, unsigned char* d, int e, int f) { stm_d_dibr_occl_to_mask(a, b, c, d, e, f); }
void dibr_occl_to_mask(float* a, float* b, unsigned char* c, unsigned char* d, int e, int f) { stm_dibr_occl_to_mask(a, b, c, d, e, f); }
void d_dibr_occl(unsigned char* a, unsigned char* b, float* c, float* d, int e, int f) { stm_d_dibr_occl(a, b, c, d, e, f); }
void dibr_occl(unsigned char* a, unsigned char* b, float* c, float* d, int e, int f) { stm_dibr_occl(a, b, c, d, e, f); }
void d_dibr_dfm(unsigned char* a, unsigned char* b, unsigned char* c, float* d, float* e, float f, int g, int h, int i)
{ stm_d_dibr_dfm(a, b, c, d, e, f, g, h, i); }
void dibr_dfm(unsigned char* a, unsigned char* b, unsigned char* c, float* d, float* e, float f, int g, int h, int i)
{ stm_dibr_dfm(a, b, c, d, e, f, g, h, i); }
void d_dibr_dbm(unsigned char* a, unsigned char* b, unsigned char* c, float* d, float* e, unsigned char* f, unsigned char* g,
                float* h, float* i, float j, int k, int l, int m)
{ stm_d_dibr_dbm(a, b, c, d, e, f, g, h, i, j, k, l, m); }
void dibr_dbm(unsigned char* a, unsigned char* b, unsigned char* c, float* d, float* e, unsigned char* f, unsigned char* g,
              float* h, float* i, float j, int k, int l, int m)
{ stm_dibr_dbm(a, b, c, d, e, f, g, h, i, j, k, l, m); }
void d_mux_multiview(unsigned char** a, unsigned char* b, int c, float d, int e, int f, int g, int h, int i)
{ stm_d_mux_multiview(a, b, c, d, e, f, g, h, i); }
void mux_multiview(unsigned char** a, unsigned char* b, int c, float d, int e, int f, int g, int h, int i)
{ stm_mux_multiview(a, b, c, d, e, f, g, h, i); }
void adcensus_stm(unsigned char* a, float* b, float* c, unsigned char* d, int e, int f, int g, int h, int i, int j,
                  int k, int l, int m, int n, float o, float p, float q, float r, int s, int t, int u, float v)
{ stm_adcensus_stm(a, b, c, d, e, f, g, h, i, j, k, (float)l, m, n, o, p, q, r, s, t, u, v); }
void adcensus_stm_f(unsigned char* a, float* b, float* c, unsigned char* d, int e, int f, int g, int h, int i, int j,
                    int k, float l, int m, int n, float o, float p, float q, float r, int s, int t, int u, float v)
{ stm_adcensus_stm(a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p, q, r, s, t, u, v); }
void adcensus_stm_2(unsigned char* a, float* b, float* c, unsigned char* d, int e, int f, int g, int h, int i, int j, int k,
                    int l, float m, int n, int o, int p, int q, float r, float s, float t, float u, int v, int w, int x, float y)
{ stm_adcensus_stm_2(a, b, c, d, e, f, g, h, i, j, k, l, m, n, (float)o, p, q, r, s, t, u, v, w, x, y); }
void adcensus_stm_2_f(unsigned char* a, float* b, float* c, unsigned char* d, int e, int f, int g, int h, int i, int j, int k,
                      int l, float m, int n, float o, int p, int q, float r, float s, float t, float u, int v, int w, int x, float y)
{ stm_adcensus_stm_2(a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p, q, r, s, t, u, v, w, x, y); }
void d_tx_scale(unsigned char* a, unsigned char* b, int c, int d, int e, int f, int g) { stm_d_tx_scale(a, b, c, d, e, f, g); }
void generateGaussianKernel(float* a, int b, float c) { stm_generate_gaussian_kernel(a, b, c); }
