// stage_chain.cpp -- a host program written against the REFERENCE's C++ names (include/stm_dropin.hpp), compiled with
// plain g++ and linked to libstm_hip.so.  It walks the still-image call sequence of the reference's driver
// (image_io.cpp:171-292: ci_adcensus, ca_cross x2, dc_wta x2, dr_dcc, dr_irv x2 with one iteration,
// filter_bilateral_1 7/7/7, dibr_occl, filter_bleed_1, dibr_occl_to_mask, dibr_dbm per view, mux_multiview) with the
// buffer layouts that driver builds (tables of D plane pointers, :155-169; 4 cross planes, :177-189; views[0] = right
// image, views[N-1] = left, :268-272).  tests/test_gpu_parity.py builds and runs it and compares the files it writes
// with the oracle; INTEGRATION.md points here as the worked example of the header swap.
//
//   g++ -O2 -I include examples/stage_chain.cpp -L stereo-to-multiview-cuda_amd -lstm_hip
//       -Wl,-rpath,$PWD/stereo-to-multiview-cuda_amd -o stage_chain          (one command line)
//   ./stage_chain left.bmp right.bmp <ndisp> <zerodisp> <usd> <lsd> <views> <out dir>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "stm_dropin.hpp" // instead of the reference's d_*.h block
#include "stm_hip.h"      // BMP reader / writer only

static void write_raw(const std::string &path, const void *p, size_t bytes)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "cannot write %s\n", path.c_str());
        exit(2);
    }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc != 9) {
        fprintf(stderr, "usage: %s left.bmp right.bmp ndisp zerodisp usd lsd views outdir\n", argv[0]);
        return 1;
    }
    int H = 0, W = 0, Hr = 0, Wr = 0;
    unsigned char *img_l = stm_bmp_read(argv[1], &H, &W), *img_r = stm_bmp_read(argv[2], &Hr, &Wr);
    if (!img_l || !img_r || H != Hr || W != Wr) {
        fprintf(stderr, "bad input pair\n");
        return 1;
    }
    const int D = atoi(argv[3]), zd = atoi(argv[4]), usd = atoi(argv[5]), lsd = atoi(argv[6]), N = atoi(argv[7]);
    const std::string out = argv[8];
    const int E = 3;
    const size_t HW = (size_t)H * W;

    // cost volumes: caller-owned tables of D plane pointers
    std::vector<float> cost_mem(4 * (size_t)D * HW);
    std::vector<float *> cost_l(D), cost_r(D), acost_l(D), acost_r(D);
    for (int d = 0; d < D; ++d) {
        cost_l[d] = &cost_mem[(0 * (size_t)D + d) * HW];
        cost_r[d] = &cost_mem[(1 * (size_t)D + d) * HW];
        acost_l[d] = &cost_mem[(2 * (size_t)D + d) * HW];
        acost_r[d] = &cost_mem[(3 * (size_t)D + d) * HW];
    }
    ci_adcensus(img_l, img_r, cost_l.data(), cost_r.data(), 10.0f, 30.0f, D, zd, H, W, E);

    std::vector<unsigned char> cross_mem(8 * HW);
    unsigned char *cross_l[4], *cross_r[4]; // UP, DOWN, LEFT, RIGHT
    for (int a = 0; a < 4; ++a) {
        cross_l[a] = &cross_mem[a * HW];
        cross_r[a] = &cross_mem[(4 + a) * HW];
    }
    ca_cross(img_l, cross_l, cost_l.data(), acost_l.data(), 6.0f, 20.0f, usd, lsd, D, H, W, E);
    ca_cross(img_r, cross_r, cost_r.data(), acost_r.data(), 6.0f, 20.0f, usd, lsd, D, H, W, E);

    std::vector<float> disp_l(HW), disp_r(HW);
    dc_wta(acost_l.data(), disp_l.data(), D, zd, H, W);
    dc_wta(acost_r.data(), disp_r.data(), D, zd, H, W);
    write_raw(out + "/wta_l.f32", disp_l.data(), HW * 4);

    std::vector<unsigned char> outl_l(HW, 0), outl_r(HW, 0);
    dr_dcc(outl_l.data(), outl_r.data(), disp_l.data(), disp_r.data(), H, W);
    dr_irv(disp_l.data(), outl_l.data(), cross_l, 20, 0.4f, H, W, D, zd, usd, 1);
    dr_irv(disp_r.data(), outl_r.data(), cross_r, 20, 0.4f, H, W, D, zd, usd, 1);
    filter_bilateral_1(disp_l.data(), 7, 7.0f, 7.0f, H, W, D);
    filter_bilateral_1(disp_r.data(), 7, 7.0f, 7.0f, H, W, D);
    write_raw(out + "/disp_l.f32", disp_l.data(), HW * 4);
    write_raw(out + "/disp_r.f32", disp_r.data(), HW * 4);

    std::vector<unsigned char> occl_l(HW), occl_r(HW);
    dibr_occl(occl_l.data(), occl_r.data(), disp_l.data(), disp_r.data(), H, W);
    filter_bleed_1(occl_l.data(), 1, H, W);
    filter_bleed_1(occl_r.data(), 1, H, W);
    std::vector<float> mask_l(HW), mask_r(HW);
    dibr_occl_to_mask(mask_l.data(), mask_r.data(), occl_l.data(), occl_r.data(), H, W);

    std::vector<unsigned char> view_mem((size_t)N * HW * E);
    std::vector<unsigned char *> views(N);
    views[0] = img_r;
    views[N - 1] = img_l;
    for (int v = 1; v < N - 1; ++v) {
        views[v] = &view_mem[(size_t)v * HW * E];
        const float shift = 1.0 - ((1.0 * (float)v) / ((float)N - 1.0));
        dibr_dbm(views[v], img_l, img_r, disp_l.data(), disp_r.data(), occl_l.data(), occl_r.data(), mask_l.data(),
                 mask_r.data(), shift, H, W, E);
    }
    std::vector<unsigned char> interlaced(HW * E);
    mux_multiview(views.data(), interlaced.data(), N, 18.43f, H, W, H, W, E);
    if (stm_bmp_write((out + "/interlaced.bmp").c_str(), interlaced.data(), H, W) != 0) return 2;
    stm_bmp_free(img_l);
    stm_bmp_free(img_r);
    printf("stage_chain: %dx%d D=%d views=%d ok\n", W, H, D, N);
    return 0;
}
