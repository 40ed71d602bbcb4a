// The reference's channel functions called directly on one image WITH ARGUMENTS -- grad_hist(image, n_bins, full,
// bias) and grad_mag(image, norm, eps), reference channels.py:30-52.  (With their default arguments, and inside
// channel_pyramid, they run in the fused kernels of wb_channels.hip.)  Plain one-thread-per-pixel kernels on the
// whole image through global memory: these calls are outside the detection path, correct rather than tuned.
// Arithmetic as SURVEY S5-S7: each convolve1d pass accumulates in fp64 in scipy's order and stores fp32, 'reflect'
// borders; the projection is fp64 with one rounding; everything after it is NumPy float32 arithmetic.
#include "wb_common.h"

namespace {

#define WB_GH_MAX_BINS 32
#define WB_GM_MAX_TAPS 127

struct GhArgs {
    const float *img;
    int H, W, n_bins, full;
    float bias;
    double bias64;                  // wide: a float64 / int64 NumPy scalar -- |chns| - bias and the rest in float64
    int wide;
    double cs[WB_GH_MAX_BINS], sn[WB_GH_MAX_BINS];
    void *out;
};

// scipy 'reflect' (d c b a | a b c d | d c b a) for any distance outside [0, n)
__device__ inline int reflect(int i, int n) {
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - 1 - i : i;
}

// correlate1d with the symmetric kernel [1,2,1] / the antisymmetric [-1,0,1] (as a convolution): fp64, one rounding
__device__ inline float hpass(float lo, float mid, float hi) { return (float)((double)mid * 2.0 + ((double)lo + (double)hi) * 1.0); }
__device__ inline float dpass(float lo, float mid, float hi) { return (float)((double)mid * 0.0 + ((double)lo - (double)hi) * 1.0); }

// reference channels.py:16-21 at pixel (y, x)
__device__ inline void gradients_at(const float *img, int H, int W, int y, int x, float &gx, float &gy) {
    const int ym = reflect(y - 1, H), yp = reflect(y + 1, H), xm = reflect(x - 1, W), xp = reflect(x + 1, W);
    auto I = [&](int r, int c) { return img[(int64_t)r * W + c]; };
    // H along axis 1 at rows ym, y, yp (column x), then D along axis 0
    const float h_m = hpass(I(ym, xm), I(ym, x), I(ym, xp));
    const float h_0 = hpass(I(y, xm), I(y, x), I(y, xp));
    const float h_p = hpass(I(yp, xm), I(yp, x), I(yp, xp));
    gy = dpass(h_m, h_0, h_p);
    // H along axis 0 at columns xm, x, xp (row y), then D along axis 1
    const float v_m = hpass(I(ym, xm), I(y, xm), I(yp, xm));
    const float v_0 = hpass(I(ym, x), I(y, x), I(yp, x));
    const float v_p = hpass(I(ym, xp), I(y, xp), I(yp, xp));
    gx = dpass(v_m, v_0, v_p);
}

__global__ __launch_bounds__(256) void grad_hist_args_kernel(GhArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)a.H * a.W) return;
    const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
    float gx, gy;
    gradients_at(a.img, a.H, a.W, y, x, gx, gy);
    const double gxd = (double)gx, gyd = (double)gy;
    float *o = reinterpret_cast<float *>(a.out) + i * a.n_bins;
    double *o64 = reinterpret_cast<double *>(a.out) + i * a.n_bins;
    for (int k = 0; k < a.n_bins; ++k) {
        const float c = (float)(gxd * a.cs[k] - gyd * a.sn[k]);
        // np.sign: -1 / 0 / +1 (NaN stays NaN)
        const float sg = c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : (c == c ? 0.0f : c));
        if (a.wide) {
            const double value = fmax(fabs((double)c) - a.bias64, 0.0);   // float32 array - float64 scalar: float64
            o64[k] = a.full ? (double)sg * value : value;
        } else {
            const float value = fmaxf(fabsf(c) - a.bias, 0.0f);           // np.fmax: a NaN operand loses
            o[k] = a.full ? sg * value : value;
        }
    }
}

__global__ __launch_bounds__(256) void grad_mag_kernel(const float *img, int H, int W, float *mag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    float gx, gy;
    gradients_at(img, H, W, y, x, gx, gy);
    mag[i] = sqrtf(gx * gx + gy * gy);
}

struct TriArgs {
    int n_taps;                      // odd
    double w[WB_GM_MAX_TAPS];        // the float32 kernel values, widened
};

// scipy correlate1d, symmetric odd kernel, along `axis` (0: rows, 1: columns): tmp = x[l]*w[c]; then for
// jj = -size1..-1: tmp += (x[l+jj] + x[l-jj]) * w[c+jj]; one rounding.  divide_by: 0 = store the filtered value,
// else out = mag / (filtered + eps) in float32 (reference channels.py:35-36).
__global__ __launch_bounds__(256) void tri_pass_kernel(const float *src, int H, int W, int axis, TriArgs t, const float *mag,
                                                       float eps, double eps64, int wide, float *dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const int size1 = t.n_taps / 2, n = axis == 0 ? H : W, l = axis == 0 ? y : x;
    auto at = [&](int p) {
        const int q = reflect(p, n);
        return (double)(axis == 0 ? src[(int64_t)q * W + x] : src[(int64_t)y * W + q]);
    };
    double tmp = at(l) * t.w[size1];
    for (int jj = -size1; jj < 0; ++jj) tmp = tmp + (at(l + jj) + at(l - jj)) * t.w[size1 + jj];
    const float f = (float)tmp;
    // (wide: `mag /= norm + eps` with a float64 eps -- the sum and the quotient are float64, stored back as float32)
    dst[i] = mag ? (wide ? (float)((double)mag[i] / ((double)f + eps64)) : mag[i] / (f + eps)) : f;
}

}  // namespace

extern "C" int wb_grad_hist_launch(void *stream, const float *img, int H, int W, int n_bins, int full, double bias, int wide,
                                   const double *cs_sn, void *out) {
    WB_REQUIRE(img && cs_sn && out, "wb_grad_hist_launch: null pointer");
    WB_REQUIRE(H >= 1 && W >= 1 && (int64_t)H * W < (1ll << 31) * 256, "wb_grad_hist_launch: bad shape %dx%d", H, W);
    WB_REQUIRE(n_bins >= 1 && n_bins <= WB_GH_MAX_BINS, "wb_grad_hist_launch: n_bins=%d (1..%d)", n_bins, WB_GH_MAX_BINS);
    GhArgs a;
    a.img = img; a.H = H; a.W = W; a.n_bins = n_bins; a.full = full != 0; a.bias = (float)bias; a.bias64 = bias; a.wide = wide != 0;
    a.out = out;
    for (int k = 0; k < n_bins; ++k) {
        a.cs[k] = cs_sn[k];
        a.sn[k] = cs_sn[n_bins + k];
    }
    hipLaunchKernelGGL(grad_hist_args_kernel, dim3((unsigned)(((int64_t)H * W + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_grad_mag_launch(void *stream, const float *img, int H, int W, int n_taps, const float *taps, double eps, int wide,
                                  float *scratch, float *out) {
    WB_REQUIRE(img && out, "wb_grad_mag_launch: null pointer");
    WB_REQUIRE(H >= 1 && W >= 1 && (int64_t)H * W < (1ll << 31) * 256, "wb_grad_mag_launch: bad shape %dx%d", H, W);
    WB_REQUIRE(n_taps == 0 || (taps && scratch && n_taps % 2 == 1 && n_taps <= WB_GM_MAX_TAPS),
               "wb_grad_mag_launch: n_taps=%d (0, or odd up to %d, with the taps and 2*H*W floats of scratch)", n_taps, WB_GM_MAX_TAPS);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(((int64_t)H * W + 255) / 256));
    float *mag = n_taps ? scratch : out;
    hipLaunchKernelGGL(grad_mag_kernel, grid, dim3(256), 0, st, img, H, W, mag);
    if (n_taps) {
        TriArgs t;
        t.n_taps = n_taps;
        for (int k = 0; k < n_taps; ++k) t.w[k] = (double)taps[k];
        float *tmp = scratch + (int64_t)H * W;
        hipLaunchKernelGGL(tri_pass_kernel, grid, dim3(256), 0, st, (const float *)mag, H, W, 0, t, (const float *)nullptr, 0.0f, 0.0, 0, tmp);
        hipLaunchKernelGGL(tri_pass_kernel, grid, dim3(256), 0, st, (const float *)tmp, H, W, 1, t, (const float *)mag, (float)eps, eps, wide, out);
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}
