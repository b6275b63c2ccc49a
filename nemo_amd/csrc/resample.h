// Re-centring resample of align_no_centered (nemo/models/batch_solve_pose.py:65-90, tensor_linspace general.py:11-40):
// target pixel (x, y) of a h0 x w0 map reads the h1 x w1 source bilinearly (grid_sample, align_corners = false) at
//   gx(x) = (1 - t_x) x_min + t_x x_max,  t_x = x / (w0 - 1),
//   x_min/max = 2 ps_x / w1 -/+ w0 / (w1 rate) - 2 pt_x / w0,  rate = dist_source / dist_target   (same for y).
// The four taps and their weights depend on the image and the pixel only, not on the channel.
#pragma once
#include <hip/hip_runtime.h>

namespace nemo {

struct AlignTaps {
    int i00, i01, i10, i11;        // source pixel offsets (clamped into the map)
    float w00, w01, w10, w11;      // weights; 0 for taps outside the source (zeros padding)
};

// border != 0: coordinates are clamped to the source (padding_mode = 'border'), else taps outside read 0.
__device__ inline AlignTaps align_taps(float dist_src, float dist_tgt, float psx, float psy, float ptx, float pty, int h1, int w1,
                                       int h0, int w0, int border, int x, int y) {
#pragma clang fp contract(off)      // the same rounding wherever the header is compiled
    const float rate = dist_src / dist_tgt;
    const float fw0 = (float)w0, fh0 = (float)h0, fw1 = (float)w1, fh1 = (float)h1;
    const float xmin = 2.0f * psx / fw1 - fw0 / (fw1 * rate) - ptx * 2.0f / fw0;
    const float xmax = 2.0f * psx / fw1 + fw0 / (fw1 * rate) - ptx * 2.0f / fw0;
    const float ymin = 2.0f * psy / fh1 - fh0 / (fh1 * rate) - pty * 2.0f / fh0;
    const float ymax = 2.0f * psy / fh1 + fh0 / (fh1 * rate) - pty * 2.0f / fh0;
    const float tx = w0 > 1 ? (float)x / (float)(w0 - 1) : 0.0f, ty = h0 > 1 ? (float)y / (float)(h0 - 1) : 0.0f;
    const float gx = (1.0f - tx) * xmin + tx * xmax, gy = (1.0f - ty) * ymin + ty * ymax;
    float ix = ((gx + 1.0f) * fw1 - 1.0f) / 2.0f, iy = ((gy + 1.0f) * fh1 - 1.0f) / 2.0f;
    if (border) {
        ix = fminf(fmaxf(ix, 0.0f), fw1 - 1.0f);
        iy = fminf(fmaxf(iy, 0.0f), fh1 - 1.0f);
    }
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wx0 = 1.0f - wx1, wy1 = iy - fy0, wy0 = 1.0f - wy1;
    const bool vx0 = x0 >= 0 && x0 < w1, vx1 = x1 >= 0 && x1 < w1, vy0 = y0 >= 0 && y0 < h1, vy1 = y1 >= 0 && y1 < h1;
    const int cx0 = min(max(x0, 0), w1 - 1), cx1 = min(max(x1, 0), w1 - 1), cy0 = min(max(y0, 0), h1 - 1), cy1 = min(max(y1, 0), h1 - 1);
    AlignTaps t;
    t.i00 = cy0 * w1 + cx0; t.i01 = cy0 * w1 + cx1; t.i10 = cy1 * w1 + cx0; t.i11 = cy1 * w1 + cx1;
    t.w00 = (vx0 && vy0) ? wx0 * wy0 : 0.0f; t.w01 = (vx1 && vy0) ? wx1 * wy0 : 0.0f;
    t.w10 = (vx0 && vy1) ? wx0 * wy1 : 0.0f; t.w11 = (vx1 && vy1) ? wx1 * wy1 : 0.0f;
    return t;
}

// One resampled value of a channel plane sc (h1 x w1).
__device__ inline float align_sample(const float* __restrict__ sc, const AlignTaps& t) {
#pragma clang fp contract(off)
    return sc[t.i00] * t.w00 + sc[t.i01] * t.w01 + sc[t.i10] * t.w10 + sc[t.i11] * t.w11;
}

}  // namespace nemo
