// Octave pyramid of the raw image (reference channels.py:93-101 + avg_pool_2 :55-64)
// and the per-octave min/max that skimage.resize clips its output to (channels.py:132).
//
// HBM-bound integer/byte work: every octave is read once and written once.
#include "wb_common.h"

namespace {

template <typename T> struct PixKey;
template <> struct PixKey<uint8_t> {
    static __device__ uint32_t key(uint8_t v) { return v; }
};
template <> struct PixKey<float> {
    static __device__ uint32_t key(float v) { return wb_f32_key(v); }
};

__device__ inline void wave_minmax_commit(uint32_t lo, uint32_t hi, uint32_t *mm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm + 0, lo);
        atomicMax(mm + 1, hi);
    }
}

__global__ void minmax_init_kernel(uint32_t *mm, int n_pairs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pairs) {
        mm[2 * i + 0] = 0xffffffffu;
        mm[2 * i + 1] = 0u;
    }
}

// min/max of octave 0 (the image itself); grid = (blocks, batch), grid-stride over pixels.
template <typename T>
__global__ void minmax_kernel(const T *img, int64_t img_stride, int64_t n_px, uint32_t *mm, int n_oct) {
    const T *p = img + (int64_t)blockIdx.y * img_stride;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t k = PixKey<T>::key(p[i]);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    wave_minmax_commit(lo, hi, mm + ((int64_t)blockIdx.y * n_oct + 0) * 2);
}

template <typename T> __device__ inline T pool4(T a, T b, T c, T d);
// uint8: the reference's adds are uint8 ufunc adds (wrap mod 256), then /4 in fp64 and a
// truncating cast  ==  ((a+b+c+d) & 255) >> 2   (SURVEY S2).
template <> __device__ inline uint8_t pool4<uint8_t>(uint8_t a, uint8_t b, uint8_t c, uint8_t d) {
    return (uint8_t)((((uint32_t)a + b + c + d) & 255u) >> 2);
}
// float32: ((a+b)+c)+d in fp32, exact /4.
template <> __device__ inline float pool4<float>(float a, float b, float c, float d) {
    return (((a + b) + c) + d) * 0.25f;
}

// dst[i][j] = pool(src[2i][2j], src[2i+1][2j], src[2i][2j+1], src[2i+1][2j+1]); odd tail dropped.
template <typename T>
__global__ void pool2_kernel(const T *src, int64_t src_stride, int sh, int sw, T *dst, int64_t dst_stride,
                             uint32_t *mm, int n_oct, int oct_k) {
    const int dh = sh >> 1, dw = sw >> 1;
    const T *s = src + (int64_t)blockIdx.y * src_stride;
    T *d = dst + (int64_t)blockIdx.y * dst_stride;
    const int64_t n = (int64_t)dh * dw;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i / dw), c = (int)(i - (int64_t)r * dw);
        const T *p0 = s + (int64_t)(2 * r) * sw + 2 * c;
        const T *p1 = p0 + sw;
        T v = pool4<T>(p0[0], p1[0], p0[1], p1[1]);
        d[i] = v;
        uint32_t k = PixKey<T>::key(v);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    wave_minmax_commit(lo, hi, mm + ((int64_t)blockIdx.y * n_oct + oct_k) * 2);
}

template <typename T>
int launch_octaves(hipStream_t st, const T *img, int batch, int H, int W, int64_t img_stride, T *oct,
                   int64_t oct_stride, const int64_t *oct_off, int n_oct, uint32_t *minmax) {
    const int threads = 256;
    int n_pairs = batch * n_oct;
    hipLaunchKernelGGL(minmax_init_kernel, dim3((n_pairs + threads - 1) / threads), dim3(threads), 0, st, minmax, n_pairs);
    {
        int64_t n_px = (int64_t)H * W;
        int blocks = (int)((n_px + threads * 8 - 1) / (threads * 8));
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(minmax_kernel<T>, dim3(blocks, batch), dim3(threads), 0, st, img, img_stride, n_px, minmax, n_oct);
    }
    int sh = H, sw = W;
    const T *src = img;
    int64_t sstride = img_stride;
    for (int k = 1; k < n_oct; ++k) {
        T *dst = oct + oct_off[k];
        int64_t n = (int64_t)(sh >> 1) * (sw >> 1);
        int blocks = (int)((n + threads - 1) / threads);
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(pool2_kernel<T>, dim3(blocks, batch), dim3(threads), 0, st, src, sstride, sh, sw, dst,
                           oct_stride, minmax, n_oct, k);
        src = dst;
        sstride = oct_stride;
        sh >>= 1;
        sw >>= 1;
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

}  // namespace

extern "C" int wb_octaves_launch(void *stream, const void *img, int dtype, int batch, int H, int W,
                                 int64_t img_stride, void *oct, int64_t oct_stride, const int64_t *oct_off,
                                 int n_oct, uint32_t *minmax) {
    WB_REQUIRE(img && minmax, "wb_octaves_launch: null pointer");
    WB_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "wb_octaves_launch: bad shape batch=%d H=%d W=%d", batch, H, W);
    WB_REQUIRE(n_oct >= 1 && n_oct <= WB_MAX_OCTAVES, "wb_octaves_launch: n_oct=%d out of range", n_oct);
    WB_REQUIRE(n_oct == 1 || (oct && oct_off), "wb_octaves_launch: octave buffer missing");
    // the octave chain must match reference channels.py:93-101 (halve until w<8 or h<8)
    {
        int h = H, w = W, k = 0;
        while (!(w < 8 || h < 8)) { ++k; h >>= 1; w >>= 1; }
        WB_REQUIRE(k == n_oct, "wb_octaves_launch: n_oct=%d but a %dx%d image has %d octaves", n_oct, H, W, k);
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == WB_DTYPE_U8)
        return launch_octaves<uint8_t>(st, (const uint8_t *)img, batch, H, W, img_stride, (uint8_t *)oct, oct_stride, oct_off, n_oct, minmax);
    if (dtype == WB_DTYPE_F32)
        return launch_octaves<float>(st, (const float *)img, batch, H, W, img_stride, (float *)oct, oct_stride, oct_off, n_oct, minmax);
    wb_set_error("wb_octaves_launch: unsupported dtype %d (uint8 and float32 images only)", dtype);
    return WB_ERR_UNSUPPORTED;
}
