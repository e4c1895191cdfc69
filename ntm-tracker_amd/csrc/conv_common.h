// Row <-> pixel mapping shared by the conv kernels (fp32 and bf16).
#pragma once
#include "common.h"

struct ConvRowInfo {
    int pix;   // linear pixel index (n*H + y)*W + x, or -1 past the end
    int yx;    // y << 16 | x
    int ppix;  // pooled linear pixel index (n*H/2 + y/2)*(W/2) + x/2
};

__device__ __forceinline__ ConvRowInfo conv_row_info(int m, int npatch, int H, int W) {
    ConvRowInfo r;
    const int patch = m >> 4, q = m & 15;
    if (patch >= npatch) { r.pix = -1; r.yx = 0; r.ppix = -1; return r; }
    const int PW = W >> 2, PH = H >> 2;
    const int px = patch % PW;
    const int t = patch / PW;
    const int py = t % PH;
    const int n = t / PH;
    const int y = py * 4 + (q >> 3) * 2 + ((q >> 1) & 1);
    const int x = px * 4 + ((q >> 2) & 1) * 2 + (q & 1);
    r.pix = (n * H + y) * W + x;
    r.yx = (y << 16) | x;
    r.ppix = (n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
    return r;
}

