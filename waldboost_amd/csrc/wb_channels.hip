// Fused channel-pyramid kernel for gfx950: one workgroup produces one TU x TV tile of one
// level's final channel image, staging through LDS
//
//   octave base (HBM, uint8/float32)
//     --bilinear, fp64, clip, truncating cast-->   R  : resized tile incl. halo   (LDS, fp32)
//     --Sobel H/D passes, fp64 projection, |.|, 2x2 shrink-->  Sh : shrunk tile  (LDS, float4)
//     --3x3 binomial smooth (fp64 accumulate), zero border-->  channels           (HBM, fp32)
//
// so the resized image, the gradients and the un-smoothed channels never touch HBM.
// Arithmetic follows SURVEY.md S3..S9 operation by operation (compile with
// -ffp-contract=off: no FMA fusion anywhere in this file).
//
// Replaces reference channels.py:127-146 (per-level body of channel_pyramid), :40-52
// (grad_hist), :16-21 (gradients), :55-64 (avg_pool_2), :78-90 (smooth).
#include "wb_common.h"

namespace {

struct ChanArgs {
    const void *img;
    const void *oct;
    int64_t img_stride, oct_stride;
    const WbLevel *levels;
    const WbTile *tiles;
    const uint32_t *minmax;
    int n_oct;
    int layout;
    float *chn;
    int64_t chn_stride;
    double cs[4], sn[4];
};

struct Tap {          // one axis of the bilinear resample (scipy NI_ZoomShift, order 1)
    int i0, i1;       // source indices (mirror-mapped)
    double w0, w1;    // w0 = 1 - frac, w1 = 1 - w0
};

__device__ inline int mirror_idx(int i, int n) {
    // scipy 'mirror' (no edge repeat); only ever reached with weight 0 when down-scaling
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i < 0 ? 0 : i;
}

__device__ inline Tap make_tap(int k, double step, int n_in) {
    double cc = (((double)k + 0.5) * step) - 0.5;
    double fl = floor(cc);
    double x = cc - fl;
    Tap t;
    t.w0 = 1.0 - x;
    t.w1 = 1.0 - t.w0;
    int i0 = (int)fl;
    t.i0 = mirror_idx(i0, n_in);
    t.i1 = mirror_idx(i0 + 1, n_in);
    return t;
}

template <typename T> struct Src;
template <> struct Src<uint8_t> {
    static __device__ double lo(uint32_t k) { return (double)k; }
    // fp64 result is clipped in fp64, then cast to uint8 by truncation
    static __device__ float finish(double t, double mn, double mx) {
        t = fmin(fmax(t, mn), mx);
        return (float)(int)t;
    }
    // [1,2,1] pass: exact in fp32 for integer pixels (|.| <= 1020)
    static __device__ float hpass(float a, float b, float c) { return b * 2.0f + (a + c); }
    static __device__ float dpass(float lo, float hi) { return lo - hi; }
};
template <> struct Src<float> {
    static __device__ double lo(uint32_t k) { return (double)wb_key_f32(k); }
    // float32 images: zoom stores fp32, then np.clip in fp32
    static __device__ float finish(double t, double mn, double mx) {
        float f = (float)t;
        return fminf(fmaxf(f, (float)mn), (float)mx);
    }
    // scipy correlate1d: fp64 accumulate, one fp32 rounding per pass
    static __device__ float hpass(float a, float b, float c) {
        return (float)((double)b * 2.0 + ((double)a + (double)c));
    }
    static __device__ float dpass(float lo, float hi) { return (float)((double)lo - (double)hi); }
};

struct F4 {
    float x, y, z, w;
};

// reference channels.py:78-83: nine-term sum in source order; numba promotes int64*float32 to
// fp64, so the sum is fp64; "/16" and one rounding to fp32 on the store
__device__ inline float smooth9(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    double s = (double)a + 2.0 * (double)b;
    s = s + (double)c;
    s = s + 2.0 * (double)d;
    s = s + 4.0 * (double)e;
    s = s + 2.0 * (double)f;
    s = s + (double)g;
    s = s + 2.0 * (double)h;
    s = s + (double)i;
    return (float)(s / 16.0);
}

template <typename T, int S, int TU, int TV, bool SMOOTH>
__global__ __launch_bounds__(256) void channels_kernel(ChanArgs a) {
    constexpr int HS = SMOOTH ? 1 : 0;
    constexpr int SU = TU + 2 * HS, SV = TV + 2 * HS;  // shrunk tile incl. smooth halo
    constexpr int RH = S * SU + 2, RW = S * SV + 2;    // resized tile incl. Sobel halo
    constexpr int P = S + 2;                           // patch side per shrunk pixel

    __shared__ Tap rowtab[RH];
    __shared__ Tap coltab[RW];
    __shared__ float R[RH * RW];
    __shared__ __attribute__((aligned(16))) F4 Sh[SU * SV];

    const WbTile tile = a.tiles[blockIdx.x];
    const WbLevel L = a.levels[tile.level];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int u0 = tile.ty * TU, v0 = tile.tx * TV;

    const T *src = (L.oct == 0) ? (const T *)a.img + (int64_t)b * a.img_stride
                                : (const T *)a.oct + (int64_t)b * a.oct_stride + L.src_off;
    const uint32_t *mm = a.minmax + ((int64_t)b * a.n_oct + L.oct) * 2;
    const double mn = Src<T>::lo(~mm[0]), mx = Src<T>::lo(mm[1]);   // mm[0] holds max(~key)

    // ---- step 0: per-row / per-column resampling taps (coordinates clamped = 'reflect'
    //      halo of convolve1d for a 1-pixel border)
    const int ry0 = S * (u0 - HS) - 1, rx0 = S * (v0 - HS) - 1;
    for (int k = tid; k < RH + RW; k += 256) {
        if (k < RH) {
            int y = ry0 + k;
            y = y < 0 ? 0 : (y > L.nh - 1 ? L.nh - 1 : y);
            rowtab[k] = make_tap(y, L.sy, L.src_h);
        } else {
            int x = rx0 + (k - RH);
            x = x < 0 ? 0 : (x > L.nw - 1 ? L.nw - 1 : x);
            coltab[k - RH] = make_tap(x, L.sx, L.src_w);
        }
    }
    __syncthreads();

    // ---- step 1: bilinear resample into R (fp64, scipy tap order), cast back to the image dtype
    for (int p = tid; p < RH * RW; p += 256) {
        int k = p / RW, q = p - k * RW;
        Tap tr = rowtab[k], tc = coltab[q];
        const T *r0 = src + (int64_t)tr.i0 * L.src_w;
        const T *r1 = src + (int64_t)tr.i1 * L.src_w;
        double v00 = (double)r0[tc.i0], v01 = (double)r0[tc.i1];
        double v10 = (double)r1[tc.i0], v11 = (double)r1[tc.i1];
        double t = (v00 * tr.w0) * tc.w0;
        t = t + (v01 * tr.w0) * tc.w1;
        t = t + (v10 * tr.w1) * tc.w0;
        t = t + (v11 * tr.w1) * tc.w1;
        R[p] = Src<T>::finish(t, mn, mx);
    }
    __syncthreads();

    // ---- step 2: gradients -> 4 oriented channels -> shrink, one shrunk pixel per iteration
    for (int p = tid; p < SU * SV; p += 256) {
        int i = p / SV, j = p - i * SV;
        float pt[P][P];
#pragma unroll
        for (int y = 0; y < P; ++y)
#pragma unroll
            for (int x = 0; x < P; ++x) pt[y][x] = R[(S * i + y) * RW + (S * j + x)];

        float hc[S][P];   // vertical [1,2,1] pass at patch rows 1..S
        float hr[P][S];   // horizontal [1,2,1] pass at patch cols 1..S
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
            for (int x = 0; x < P; ++x) hc[y][x] = Src<T>::hpass(pt[y][x], pt[y + 1][x], pt[y + 2][x]);
#pragma unroll
        for (int y = 0; y < P; ++y)
#pragma unroll
            for (int x = 0; x < S; ++x) hr[y][x] = Src<T>::hpass(pt[y][x], pt[y][x + 1], pt[y][x + 2]);

        float ch[S][S][4];
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
            for (int x = 0; x < S; ++x) {
                float gx = Src<T>::dpass(hc[y][x], hc[y][x + 2]);
                float gy = Src<T>::dpass(hr[y][x], hr[y + 2][x]);
                double gxd = (double)gx, gyd = (double)gy;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float val = (float)(gxd * a.cs[k] - gyd * a.sn[k]);
                    ch[y][x][k] = fmaxf(fabsf(val), 0.0f);
                }
            }

        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (S == 1) {
                o[k] = ch[0][0][k];
            } else if constexpr (S == 2) {
                o[k] = (((ch[0][0][k] + ch[1][0][k]) + ch[0][1][k]) + ch[1][1][k]) * 0.25f;
            } else {  // S == 4 (extension): avg_pool_2 applied twice
                float q[2][2];
#pragma unroll
                for (int A = 0; A < 2; ++A)
#pragma unroll
                    for (int B = 0; B < 2; ++B)
                        q[A][B] = (((ch[2 * A][2 * B][k] + ch[2 * A + 1][2 * B][k]) + ch[2 * A][2 * B + 1][k]) +
                                   ch[2 * A + 1][2 * B + 1][k]) * 0.25f;
                o[k] = (((q[0][0] + q[1][0]) + q[0][1]) + q[1][1]) * 0.25f;
            }
        }
        Sh[p] = F4{o[0], o[1], o[2], o[3]};
    }
    __syncthreads();

    // ---- step 3: 3x3 binomial smooth (fp64 sum in source order, /16, one rounding), border = 0
    float *out = a.chn + (int64_t)b * a.chn_stride + L.chn_off;
    const int64_t plane = (int64_t)L.u * L.vp;
    for (int p = tid; p < TU * TV; p += 256) {
        int i = p / TV, j = p - i * TV;
        int su = u0 + i, sv = v0 + j;
        if (su >= L.u || sv >= L.v) continue;
        float o[4];
        if constexpr (SMOOTH) {
            if (su == 0 || sv == 0 || su == L.u - 1 || sv == L.v - 1) {
                o[0] = o[1] = o[2] = o[3] = 0.0f;
            } else {
                const F4 *c = &Sh[(i + 1) * SV + (j + 1)];
                F4 n00 = c[-SV - 1], n01 = c[-SV], n02 = c[-SV + 1];
                F4 n10 = c[-1], n11 = c[0], n12 = c[1];
                F4 n20 = c[SV - 1], n21 = c[SV], n22 = c[SV + 1];
                o[0] = smooth9(n00.x, n01.x, n02.x, n10.x, n11.x, n12.x, n20.x, n21.x, n22.x);
                o[1] = smooth9(n00.y, n01.y, n02.y, n10.y, n11.y, n12.y, n20.y, n21.y, n22.y);
                o[2] = smooth9(n00.z, n01.z, n02.z, n10.z, n11.z, n12.z, n20.z, n21.z, n22.z);
                o[3] = smooth9(n00.w, n01.w, n02.w, n10.w, n11.w, n12.w, n20.w, n21.w, n22.w);
            }
        } else {
            F4 c = Sh[p];
            o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
        }
        if (a.layout == WB_LAYOUT_HWC) {
            float4 *dst = reinterpret_cast<float4 *>(out + ((int64_t)su * L.v + sv) * 4);
            *dst = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            float *dst = out + (int64_t)su * L.vp + sv;
            dst[0] = o[0];
            dst[plane] = o[1];
            dst[2 * plane] = o[2];
            dst[3 * plane] = o[3];
        }
    }
}

template <typename T, int S, int TU, int TV>
void launch_variant(hipStream_t st, dim3 grid, const ChanArgs &a, bool smooth) {
    if (smooth)
        hipLaunchKernelGGL((channels_kernel<T, S, TU, TV, true>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((channels_kernel<T, S, TU, TV, false>), grid, dim3(256), 0, st, a);
}

template <typename T>
int launch_dtype(hipStream_t st, dim3 grid, const ChanArgs &a, int shrink, bool smooth) {
    switch (shrink) {
        case 1: launch_variant<T, 1, 16, 64>(st, grid, a, smooth); break;
        case 2: launch_variant<T, 2, 16, 64>(st, grid, a, smooth); break;
        case 4: launch_variant<T, 4, 8, 32>(st, grid, a, smooth); break;
        default:
            wb_set_error("wb_channels_launch: shrink=%d unsupported (1, 2; 4 as an extension)", shrink);
            return WB_ERR_UNSUPPORTED;
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

}  // namespace

extern "C" int wb_channels_tile(int shrink, int *tile_u, int *tile_v) {
    WB_REQUIRE(tile_u && tile_v, "wb_channels_tile: null pointer");
    if (shrink == 1 || shrink == 2) {
        *tile_u = 16;
        *tile_v = 64;
    } else if (shrink == 4) {
        *tile_u = 8;
        *tile_v = 32;
    } else {
        wb_set_error("wb_channels_tile: shrink=%d unsupported", shrink);
        return WB_ERR_UNSUPPORTED;
    }
    return WB_OK;
}

extern "C" int wb_channels_launch(void *stream, const void *img, int64_t img_stride, const void *oct,
                                  int64_t oct_stride, int dtype, int batch, const WbLevel *levels,
                                  int n_levels, const WbTile *tiles, int n_tiles, const uint32_t *minmax,
                                  int n_oct, int shrink, int smooth, const double *cs_sn, float *chn,
                                  int64_t chn_stride, int layout) {
    WB_REQUIRE(img && levels && tiles && minmax && cs_sn && chn, "wb_channels_launch: null pointer");
    WB_REQUIRE(batch >= 1 && n_levels >= 1 && n_tiles >= 1, "wb_channels_launch: empty launch");
    WB_REQUIRE(batch <= 65535, "wb_channels_launch: batch %d exceeds grid.y limit", batch);
    WB_REQUIRE(smooth == 0 || smooth == 1, "wb_channels_launch: smooth must be 0 or 1");
    WB_REQUIRE(layout == WB_LAYOUT_PLANAR || layout == WB_LAYOUT_HWC, "wb_channels_launch: bad layout %d", layout);
    WB_REQUIRE(n_oct >= 1 && n_oct <= WB_MAX_OCTAVES, "wb_channels_launch: n_oct out of range");
    ChanArgs a;
    a.img = img;
    a.oct = oct;
    a.img_stride = img_stride;
    a.oct_stride = oct_stride;
    a.levels = levels;
    a.tiles = tiles;
    a.minmax = minmax;
    a.n_oct = n_oct;
    a.layout = layout;
    a.chn = chn;
    a.chn_stride = chn_stride;
    for (int k = 0; k < 4; ++k) {
        a.cs[k] = cs_sn[k];
        a.sn[k] = cs_sn[4 + k];
    }
    dim3 grid((unsigned)n_tiles, (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == WB_DTYPE_U8) return launch_dtype<uint8_t>(st, grid, a, shrink, smooth != 0);
    if (dtype == WB_DTYPE_F32) return launch_dtype<float>(st, grid, a, shrink, smooth != 0);
    wb_set_error("wb_channels_launch: unsupported dtype %d (uint8 and float32 images only)", dtype);
    return WB_ERR_UNSUPPORTED;
}
