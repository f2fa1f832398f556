// stm_bmp.cpp -- 24-bit BMP reader/writer behind the C ABI (stm_bmp_read / stm_bmp_write).
// Replaces the cv::imread call of the reference's still-image driver (image_io.cpp:95-112): same
// result layout (interleaved BGR u8, top row first, no padding).  BITMAPINFOHEADER, BI_RGB only;
// bottom-up and top-down files, row padding to 4 bytes, trailing bytes tolerated (img/bud_1.bmp).
#include "../../include/stm_hip.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static void wr32(unsigned char *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
static void wr16(unsigned char *p, uint16_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }

extern "C" unsigned char *stm_bmp_read(const char *path, int *num_rows, int *num_cols)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    unsigned char hdr[54];
    if (fread(hdr, 1, 54, f) != 54 || hdr[0] != 'B' || hdr[1] != 'M') { fclose(f); return NULL; }
    uint32_t off = rd32(hdr + 10), hsz = rd32(hdr + 14);
    int32_t w = (int32_t)rd32(hdr + 18), h = (int32_t)rd32(hdr + 22);
    uint16_t bpp = rd16(hdr + 28);
    uint32_t comp = rd32(hdr + 30);
    if (hsz < 40 || bpp != 24 || comp != 0 || w <= 0 || h == 0) { fclose(f); return NULL; }
    int top_down = h < 0;
    int H = h < 0 ? -h : h, W = w;
    size_t stride = ((size_t)W * 3 + 3) & ~(size_t)3;
    unsigned char *row = (unsigned char *)malloc(stride);
    unsigned char *img = (unsigned char *)malloc((size_t)H * W * 3);
    if (!row || !img || fseek(f, (long)off, SEEK_SET) != 0) { free(row); free(img); fclose(f); return NULL; }
    for (int i = 0; i < H; ++i) {
        if (fread(row, 1, stride, f) != stride) { free(row); free(img); fclose(f); return NULL; }
        int y = top_down ? i : H - 1 - i;
        memcpy(img + (size_t)y * W * 3, row, (size_t)W * 3);
    }
    free(row);
    fclose(f);
    *num_rows = H;
    *num_cols = W;
    return img;
}

extern "C" int stm_bmp_write(const char *path, const unsigned char *bgr, int num_rows, int num_cols)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    size_t stride = ((size_t)num_cols * 3 + 3) & ~(size_t)3;
    unsigned char hdr[54];
    memset(hdr, 0, sizeof hdr);
    hdr[0] = 'B'; hdr[1] = 'M';
    wr32(hdr + 2, (uint32_t)(54 + stride * num_rows));
    wr32(hdr + 10, 54);
    wr32(hdr + 14, 40);
    wr32(hdr + 18, (uint32_t)num_cols);
    wr32(hdr + 22, (uint32_t)num_rows);
    wr16(hdr + 26, 1);
    wr16(hdr + 28, 24);
    wr32(hdr + 34, (uint32_t)(stride * num_rows));
    wr32(hdr + 38, 2835);
    wr32(hdr + 42, 2835);
    int ok = fwrite(hdr, 1, 54, f) == 54;
    unsigned char *row = (unsigned char *)calloc(1, stride);
    for (int i = 0; ok && i < num_rows; ++i) {
        memcpy(row, bgr + (size_t)(num_rows - 1 - i) * num_cols * 3, (size_t)num_cols * 3);
        ok = fwrite(row, 1, stride, f) == stride;
    }
    free(row);
    fclose(f);
    return ok ? 0 : -1;
}

extern "C" void stm_bmp_free(unsigned char *p) { free(p); }
