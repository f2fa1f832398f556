/*
 * stm_hip.h -- C ABI of the MI355X-native stereo->multiview hot path (libstm_hip.so).
 *
 * Every entry point below replaces one function of the reference's per-stage host
 * API.  The reference relies on C++ name mangling and has no
 * C ABI, so each function is exported here as  stm_<reference name>  with the
 * reference's argument list unchanged (same order, same meaning, same ownership):
 *   - "host flavour"   stm_xxx   : every pointer is a HOST pointer; the call uploads,
 *                                   runs the HIP kernels, downloads and returns only
 *                                   when the outputs are visible to the host.
 *   - "device flavour" stm_d_xxx : every pointer is a DEVICE pointer (plus the host
 *                                   mirror tables the reference also passes); kernels
 *                                   are enqueued on the current stream (stm_set_stream),
 *                                   nothing is synchronised.
 * C++ callers that want the reference's exact (mangled) names -- image_io.cpp /
 * d_io.cu style call sites -- include stm_dropin.hpp instead.
 *
 * Layouts (reference: image_io.cpp:155-189, d_io.cu:71-101):
 *   images      interleaved BGR u8, row-major, no row padding, elem_sz == 3
 *   cost volume table of num_disp pointers, each a dense num_rows*num_cols float plane
 *   cross arms  table of 4 pointers to u8 planes, order UP, DOWN, LEFT, RIGHT
 *   disparity   float [H][W], signed offset (d - zero_disp)
 *
 * Threading: the reference is single-threaded on the default stream.  Here every host thread has
 * its own current stream (stm_set_stream) and its own cached workspace per device, so threads may call into the
 * library concurrently; device-flavour calls of two threads that touch the same buffers need streams ordered by
 * the caller.  A thread that ends should call stm_release_workspace() first (its slab is not freed for it).
 *
 * Errors: like the reference (cuda_utils.h:12-21) a HIP failure prints a message and
 * calls exit(1); unlike it, kernel launches are checked too.  stm_set_error_mode(1)
 * turns that into "record and return" for embedding hosts (query stm_last_error()).
 * Arguments the reference would turn into undefined behaviour are errors of the same kind:
 * a dimension < 1, elem_sz < 3 (three channels of every element are read), num_views < 2
 * for the interlacer (d_mux_multiview.cu:62-66 reads views[1]), an angle whose row period
 * round(num_views / tan(angle) / elem_sz) is 0 (ty % 0, :55) or not finite (tan(angle) == 0, :146),
 * num_cols > 8192 in ca_cross.
 *
 * All file:line citations are relative to the reference repository root.
 */
#ifndef STM_HIP_H
#define STM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif
#if defined(STM_BUILD) && defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---------------------------------------------------------------- runtime */
int         stm_version(void);
/* current HIP stream (a hipStream_t) used by every device-flavour call of this thread */
void        stm_set_stream(void *hip_stream);
void       *stm_get_stream(void);
/* 0 = print + exit(1) like cuda_utils.h:12-21 (default); 1 = record, return, keep going */
void        stm_set_error_mode(int mode);
/* the calling thread's last error text.  It also waits for the thread's stream and reads (and clears) a device-side word that
 * the region-voting kernels set if they had to clamp their outlier list -- a condition that cannot arise while the library
 * clears its counters per call, reported instead of swallowed if it ever does. */
const char *stm_last_error(void);
/* frees the cached device workspace (the reference cudaMalloc/cudaFree's per call) */
void        stm_release_workspace(void);
/* per-kernel HIP-event profiling of the named kernels inside the frame pipeline: 0 off, 1 every named kernel,
 * 2 the aggregation kernels only (three event pairs per frame: what bench.py keeps on inside its timed region) */
void        stm_prof_enable(int on);
void        stm_prof_reset(void);
/* returns number of timed launches of `kernel` ("pq_h","pq_v12","pq_hw","cross_arms","irv","hslo_lr","hslo_rl","hslo_tb",
 * "hslo_bt", ...) and their
 * summed duration in ms; synchronises the recorded events. */
int         stm_prof_read(const char *kernel, float *total_ms);
/* aggregation variant of the frame pipeline (0 = default: matrix-pipe kernels, stm_kernels_aggm.hip); decimal digits, used by
 * the benchmark and the tools to A/B result-preserving variants in one process: 10000 = vector-ALU aggregation kernels
 * (stm_kernels_agg.hip; the low digits then select their tunables), 1000000 = separate initial-cost kernel instead of
 * computing the costs inside the first pass, 1000 / 2000 = the cost-computing pass as one block per segment (128- / 192-pixel
 * segments) instead of the row walk, 10 = the volume-reading horizontal passes as one block per segment instead of the row walk
 * (20: only for num_disp > 64), 200 = view synthesis and interlacing as two kernels, 300 = region voting over the raster-ordered outlier list of round 3
 * instead of over column runs, 400 = the two horizontal scanline-optimisation passes as two launches instead of one walk from both
 * ends of a row, 500 (tests) = the per-stage filter_bilateral_1 through the frame pipeline's integer-map kernel (which checks each tile
 * of its input and takes the general form where the map is not integer-valued inside the colour table), 10000000 = the vertical passes on the LDS-ring
 * kernel of round 3 instead of the register-ring kernel (stm_kernels_aggv.hip), 100000000 = the last horizontal pass + WTA on the
 * LDS row walk instead of the register-ring kernel (stm_kernels_aggh.hip), 1000000000 = the window tables of the two register-ring
 * kernels from their own launches instead of from the cross-arm kernel.  Every accepted variant produces identical
 * results (tests/test_gpu_parity.py::test_device_frame_agg_variants).  The
 * digit N00000 (timing experiments that skip parts of kernels) is ignored here: it exists only in libstm_hip_timing.so,
 * a separate build of the same sources with -DSTM_TIMING (csrc/Makefile, `make timing`). */
void        stm_set_agg_variant(int v);
/* dr_irv / d_dr_irv / the frame calls: 0 (default) = the reference's accept rule, (winning bin index + zero_disp) / S > thresh_h
 * (d_dr_irv.cu:36 -- the bin INDEX, SURVEY A-Q17 iv); 1 = the paper's rule, (winning bin's COUNT) / S > thresh_h (Mei et al.,
 * region voting).  An addition: the reference has no such switch. */
void        stm_set_irv_paper_ratio(int on);
/* ci_adcensus / d_ci_adcensus (the per-stage calls; the frame calls always compute the clean costs): 0 (default) = clean
 * clamped indexing, the canonical form (SURVEY A-Q7); 1 = reproduce the reference's shared-tile strays at d = 0 in columns
 * 160 k (left cost) and 160 k + 159 (right cost): the census term always, the AD term when num_disp - zero_disp <= zero_disp
 * (d_ci_adcensus.cu:57-59,117-120; d_ci_census.cu:240-246; d_ci_ad.cu:133-144).  Meaningful for num_cols % 160 == 0, the
 * only widths for which the reference's own launch fills its tiles.  An addition: the reference has no such switch. */
void        stm_set_ref_quirks(int on);

/* ------------------------------------------------------- cost init (a1-a7) */
/* d_ci_adcensus.h:23-25  ci_adcensus  (d_ci_adcensus.cu:188-378) */
void stm_ci_adcensus(unsigned char *img_l, unsigned char *img_r, float **cost_l, float **cost_r,
                     float ad_coeff, float census_coeff, int num_disp, int zero_disp,
                     int num_rows, int num_cols, int elem_sz);
/* d_ci_adcensus.h:16-21  d_ci_adcensus (d_ci_adcensus.cu:38-186).  Fills the host and device plane
 * tables exactly as :150-157 does: L plane d at memory + d*H*W, R plane d at memory + (D + d)*H*W. */
void stm_d_ci_adcensus(unsigned char *d_img_l, unsigned char *d_img_r,
                       float **d_adcensus_cost_l, float **d_adcensus_cost_r,
                       float **h_adcensus_cost_l, float **h_adcensus_cost_r,
                       float *d_adcensus_cost_memory,
                       float ad_coeff, float census_coeff, int num_disp, int zero_disp,
                       int num_rows, int num_cols, int elem_sz);

/* ------------------------------------------------ cross aggregation (a8-a12) */
/* d_ca_cross.h:19-21  ca_cross (d_ca_cross.cu:275-444): cost untouched, result in acost, arms in cross */
void stm_ca_cross(unsigned char *img, unsigned char **cross, float **cost, float **acost,
                  float ucd, float lcd, int usd, int lsd,
                  int num_disp, int num_rows, int num_cols, int elem_sz);
/* d_ca_cross.h:13-17  d_ca_cross (d_ca_cross.cu:174-273): RESULT LANDS IN d_cost (input overwritten),
 * d_acost/d_acost_memory are scratch; h_acost and d_acost tables are filled as :207-210 does. */
void stm_d_ca_cross(unsigned char *d_img, float **d_cost,
                    float **d_acost, float **h_acost, float *d_acost_memory,
                    unsigned char **d_cross,
                    float ucd, float lcd, int usd, int lsd,
                    int num_disp, int num_rows, int num_cols, int elem_sz);

/* ------------------------------------------------- disparity selection (a13,a14) */
/* d_dc_wta.h:16-18 / :12-14  (d_dc_wta.cu:9-59) */
void stm_dc_wta(float **cost, float *disp, int num_disp, int zero_disp, int num_rows, int num_cols);
void stm_d_dc_wta(float **d_cost, float *d_disp, int num_disp, int zero_disp, int num_rows, int num_cols);
/* d_dc_hslo.h:18-22  dc_hslo (d_dc_hslo.cu:97-221, a stub in the reference; implemented here, parity unpinned) */
void stm_dc_hslo(float **cost, float *disp, unsigned char *img_l, unsigned char *img_r,
                 float T, float H1, float H2, int num_disp, int zero_disp,
                 int num_rows, int num_cols, int elem_sz);
/* device flavour: the reference has none; same contract as the other d_ calls */
void stm_d_dc_hslo(float **d_cost, float *d_disp, unsigned char *d_img_l, unsigned char *d_img_r,
                   float T, float H1, float H2, int num_disp, int zero_disp,
                   int num_rows, int num_cols, int elem_sz);

/* -------------------------------------------------------- refinement (a15-a17) */
/* d_dr_dcc.h:17-19 / :13-15  (d_dr_dcc.cu:84-203).  Host flavour zero-fills the outlier maps itself
 * (:166-171); the device flavour expects them zero-filled by the caller (d_io.cu:138-141). */
void stm_dr_dcc(unsigned char *outliers_l, unsigned char *outliers_r, float *disp_l, float *disp_r,
                int num_rows, int num_cols);
void stm_d_dr_dcc(unsigned char *d_outliers_l, unsigned char *d_outliers_r, float *d_disp_l, float *d_disp_r,
                  int num_rows, int num_cols);
/* d_dr_irv.h:15-19 / :8-13  (d_dr_irv.cu:222-364).  Host flavour votes once and applies `iterations`
 * times (:344-353); device flavour repeats vote+apply (:259-265). */
void stm_dr_irv(float *disp, unsigned char *outliers, unsigned char **cross, int thresh_s, float thresh_h,
                int num_rows, int num_cols, int num_disp, int zero_disp, int usd, int iterations);
void stm_d_dr_irv(float *d_disp, unsigned char *d_outliers, unsigned char **d_cross, int thresh_s, float thresh_h,
                  int num_rows, int num_cols, int num_disp, int zero_disp, int usd, int iterations);
/* d_filter_bilateral.h:17-20 / :13-15  (d_filter_bilateral.cu:517-630) */
void stm_filter_bilateral_1(float *img, int radius, float sigma_color, float sigma_spatial,
                            int num_rows, int num_cols, int num_disp);
void stm_d_filter_bilateral_1(float *d_img, int radius, float sigma_color, float sigma_spatial,
                              int num_rows, int num_cols, int num_disp);
/* d_filter_gaussian.h:20-26  (d_filter_gaussian.cu:134-234): grow-only gaussian, out = max(in, blur) */
void stm_filter_gaussian_1(float *img, int radius, float sigma_spatial, int num_rows, int num_cols);
void stm_d_filter_gaussian_1(float *d_img, int radius, float sigma_spatial, int num_rows, int num_cols);
/* d_filter.h:22-28  (d_filter.cu:105-167 and following) */
void stm_filter_bleed_1(unsigned char *img, int radius, int num_rows, int num_cols);
void stm_d_filter_bleed_1(unsigned char *d_img, int radius, int num_rows, int num_cols);
/* d_filter.h:11-16  (d_filter.cu:7-103): 3x3 "median" on int-truncated values; unused by the reference's
 * drivers (image_io.cpp:239-240 are commented out) but part of its stage API */
void stm_filter_median(float *img, int num_rows, int num_cols);
void stm_d_filter_median(float *d_img, int num_rows, int num_cols);

/* --------------------------------------------------------------- DIBR (a18-a23) */
/* d_dibr_occl.h:27-33  (d_dibr_occl.cu:130-218) */
void stm_dibr_occl(unsigned char *occl_l, unsigned char *occl_r, float *disp_l, float *disp_r,
                   int num_rows, int num_cols);
void stm_d_dibr_occl(unsigned char *d_occl_l, unsigned char *d_occl_r, float *d_disp_l, float *d_disp_r,
                     int num_rows, int num_cols);
/* d_dibr_occl.h:14-20  (d_dibr_occl.cu:17-112) */
void stm_dibr_occl_to_mask(float *mask_l, float *mask_r, unsigned char *occl_l, unsigned char *occl_r,
                           int num_rows, int num_cols);
void stm_d_dibr_occl_to_mask(float *d_mask_l, float *d_mask_r, unsigned char *d_occl_l, unsigned char *d_occl_r,
                             int num_rows, int num_cols);
/* d_dibr_bwarp.h:22-34  (d_dibr_bwarp.cu:24-180).  Host flavour blurs the mask with gaussian(7,10)
 * (:151), device flavour with gaussian(10,15) (:63).  Neither modifies the caller's masks. */
void stm_dibr_dbm(unsigned char *img_out, unsigned char *img_in_l, unsigned char *img_in_r,
                  float *disp_l, float *disp_r, unsigned char *occl_l, unsigned char *occl_r,
                  float *mask_l, float *mask_r, float shift, int num_rows, int num_cols, int elem_sz);
void stm_d_dibr_dbm(unsigned char *d_img_out, unsigned char *d_img_in_l, unsigned char *d_img_in_r,
                    float *d_disp_l, float *d_disp_r, unsigned char *d_occl_l, unsigned char *d_occl_r,
                    float *d_mask_l, float *d_mask_r, float shift, int num_rows, int num_cols, int elem_sz);
/* d_dibr_fwarp.h:12-20  (d_dibr_fwarp.cu:27-193): racy in the reference; deterministic here
 * (largest source x wins), parity unpinned */
void stm_dibr_dfm(unsigned char *img_out, unsigned char *img_in_l, unsigned char *img_in_r,
                  float *disp_l, float *disp_r, float shift, int num_rows, int num_cols, int elem_sz);
void stm_d_dibr_dfm(unsigned char *d_img_out, unsigned char *d_img_in_l, unsigned char *d_img_in_r,
                    float *d_disp_l, float *d_disp_r, float shift, int num_rows, int num_cols, int elem_sz);

/* ---------------------------------------------------------------- mux (a24, a25) */
/* d_mux_multiview.h:35-41  (d_mux_multiview.cu:126-220) */
void stm_mux_multiview(unsigned char **views, unsigned char *out_data, int num_views, float angle,
                       int in_rows, int in_cols, int out_rows, int out_cols, int elem_sz);
void stm_d_mux_multiview(unsigned char **d_views, unsigned char *d_out_data, int num_views, float angle,
                         int in_rows, int in_cols, int out_rows, int out_cols, int elem_sz);
/* d_demux_common.h:10-13  demux_sbs kernel (d_demux_common.cu:8-33) as a host-callable stage */
void stm_d_demux_sbs(unsigned char *d_img_l, unsigned char *d_img_r, unsigned char *d_img_sbs,
                     int num_rows, int num_cols_sbs, int num_cols_out, int elem_sz);

/* ------------------------------------------------------------ whole frame (a26) */
/* d_io.h:32-40  adcensus_stm (d_io.cu:7-238).  `angle` is float here (the reference's int truncates
 * the caller's float, SURVEY A-Q24). */
void stm_adcensus_stm(unsigned char *img_sbs, float *disp_l, float *disp_r, unsigned char *interlaced,
                      int num_rows, int num_cols_sbs, int num_cols,
                      int num_rows_out, int num_cols_out, int elem_sz,
                      int num_views, float angle, int num_disp, int zero_disp,
                      float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                      int thresh_s, float thresh_h);
/* Device-resident frame: same pipeline, all four buffers already in HBM, nothing synchronised.
 * stages: 1 = cost init + aggregation + WTA only (BASELINE config 2);
 *         2 = + DCC / IRV x5 / bilateral          (config 3);
 *         3 = + DIBR views + interlacing           (config 4, the full adcensus_stm).
 * OR-ing 0x100 inserts the scanline optimisation (HSLO, constants of image_io.cpp:311-313) between aggregation
 * and WTA for both views, as Mei et al. order it; the reference never wires it in (parity unpinned). */
void stm_d_adcensus_stm(unsigned char *d_img_sbs, float *d_disp_l, float *d_disp_r, unsigned char *d_interlaced,
                        int num_rows, int num_cols_sbs, int num_cols,
                        int num_rows_out, int num_cols_out, int elem_sz,
                        int num_views, float angle, int num_disp, int zero_disp,
                        float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                        int thresh_s, float thresh_h, int stages);

/* Reduced-resolution disparity: d_io.h:42-52 adcensus_stm_2 (d_io.cu:240-508).  The pair is
 * bilinearly reduced to num_rows_disp x num_cols_disp, matched there, and the disparity maps are scaled back up
 * by 1/disp_scale before the views are rendered at full resolution.  `angle` is float (A-Q24). */
void stm_adcensus_stm_2(unsigned char *img_sbs, float *disp_l, float *disp_r, unsigned char *interlaced,
                        int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out,
                        int num_rows_disp, int num_cols_disp, int elem_sz, float disp_scale,
                        int num_views, float angle, int num_disp, int zero_disp,
                        float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                        int thresh_s, float thresh_h);
void stm_d_adcensus_stm_2(unsigned char *d_img_sbs, float *d_disp_l, float *d_disp_r, unsigned char *d_interlaced,
                          int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out,
                          int num_rows_disp, int num_cols_disp, int elem_sz, float disp_scale,
                          int num_views, float angle, int num_disp, int zero_disp,
                          float ad_coeff, float census_coeff, float ucd, float lcd, int usd, int lsd,
                          int thresh_s, float thresh_h);
/* d_tx_scale.h:17-18  d_tx_scale (d_tx_scale.cu:83-121): bilinear image resize; HOST pointers despite the name */
void stm_d_tx_scale(unsigned char *img_in, unsigned char *img_out, int in_rows, int in_cols, int out_rows, int out_cols,
                    int elem_sz);

/* d_filter_gaussian.h:30 (d_filter_gaussian.cu:237-255): the (2r+1)^2 spatial kernel the two big stencils use,
 * exp(-(x^2+y^2)/(2 s^2)) / (2 pi s^2) with the reference's float/double mix, row-major.  Host-only helper. */
void stm_generate_gaussian_kernel(float *kernel, int radius, float sigma);

/* ------------------------------------------------- frame sequences */
/* The reference's video loop (video_io.cpp:144-165) calls adcensus_stm once per decoded frame, serialising upload,
 * compute and download.  A frame stream keeps the same per-frame contract (one side-by-side frame in; disp_l,
 * disp_r and the interlaced frame out, in submission order) over double-buffered pinned/device buffers and three
 * HIP streams, so frame k+1 uploads and frame k-1 downloads while frame k computes.  From its third frame on each
 * buffer slot replays the frame's kernel launches as a captured hipGraph (environment STM_STREAM_GRAPH=0 turns
 * that off).  A stream belongs to the host thread that created it.  Parameters as adcensus_stm. */
void *stm_stream_create(int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out, int elem_sz,
                        int num_views, float angle, int num_disp, int zero_disp, float ad_coeff, float census_coeff,
                        float ucd, float lcd, int usd, int lsd, int thresh_s, float thresh_h);
/* stages the next frame (the caller's buffer is reusable on return); at most two frames in flight.
 * Returns the frame index, or -1 if both slots are uncollected. */
long  stm_stream_submit(void *stream, const unsigned char *img_sbs);
/* waits for the oldest uncollected frame and copies its results out (NULL = skip).  Returns its index or -1. */
long  stm_stream_collect(void *stream, float *disp_l, float *disp_r, unsigned char *interlaced);
/* zero-copy variants (at 1080p the two host copies of submit / collect take longer than the frame does on the GPU):
 * the pinned input buffer of the slot the next submit will use (NULL while that slot is uncollected) -- write the frame
 * into it and pass the same pointer (or NULL) to stm_stream_submit; and a collect that hands out pointers to the pinned
 * result buffers, valid until the frame after the next one is submitted. */
unsigned char *stm_stream_input_buffer(void *stream);
long  stm_stream_collect_view(void *stream, const float **disp_l, const float **disp_r, const unsigned char **interlaced);
void  stm_stream_destroy(void *stream);

/* ----------------------------------------------------------------- BMP I/O */
/* image_io.cpp:95-112 reads the pair with cv::imread; these read/write the same 24-bit BMPs.
 * stm_bmp_read returns a malloc'd BGR buffer (free with stm_bmp_free) or NULL. */
unsigned char *stm_bmp_read(const char *path, int *num_rows, int *num_cols);
int            stm_bmp_write(const char *path, const unsigned char *bgr, int num_rows, int num_cols);
void           stm_bmp_free(unsigned char *p);

#if defined(STM_BUILD) && defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* STM_HIP_H */
