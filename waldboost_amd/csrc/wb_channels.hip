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
#include <stdlib.h>
#include <type_traits>

#include "wb_common.h"

namespace {

struct ChanArgs {
    const void *img;
    const void *oct;
    int64_t img_stride, oct_stride;
    const WbLevel *levels;
    const WbTile *tiles;
    const uint32_t *minmax;
    const WbTap *taps;
    int n_oct;
    void *chn;           // [u][v][C] per level, dtype of the channel function
    int64_t chn_stride;
    double cs[4], sn[4];
    float chi, clo;      // sin(pi/4) = chi + clo (two-float split) for the integer-gradient fast path
    float c2hi, c2lo;    // cos(pi/2) (fp64: 6.1e-17) likewise
    double tri[11];      // grad_mag: triangle_kernel(5) (float32 values, widened)
    float gm_eps;        // grad_mag: float32(1e-3)
    int src_int;         // float64-held image dtypes: how the resize result is cast back (WB_CAST_*: .astype(image dtype))
    int dbg;             // diagnostics (WB_CHAN_DBG): 1 = stop after step 1, 2 = after step 2, 4 = skip the stores
    // optional second output of channels_kernel: the pixels as threshold ranks of one model (WB_DTYPE_RANK8)
    uint8_t *rank;       // [u][v][4] bytes per level, same element offsets as chn; nullptr = none
    int64_t rank_stride;
    const WbTilePatch *patches;   // optional (uint8 images): per tile, the source patch it stages (wb_channels_tile_patches)
    const uint4 *rank_lut;   // WbModel::bin_lut_dev: float S[4][256], then uint8 base[4][WB_BIN_CELLS]
    int rank_iters;
    float rank_k[4], rank_b[4];
    int rank_wide;       // 0: WB_DTYPE_RANK8 (one dword per pixel), 1: WB_DTYPE_RANK16 (uint16 x 4 = 8 bytes per pixel; WB_BIN16_* tables)
};

// Diagnostic build only (make STAMPS=1): thread 0 of every workgroup stores s_memrealtime at the
// phase boundaries into a private slot; wb_debug_channel_stamps turns them into mean wall-clock per
// phase.  Never part of a measured build.
#ifdef WB_CASC_STAMPS
#define WB_CSTAMP_SLOTS 8
#define WB_CSTAMP_WGS (1 << 17)
__device__ unsigned long long g_chan_stamps[WB_CSTAMP_WGS * WB_CSTAMP_SLOTS];
#define WB_CSTAMP(k)                                                                                      \
    do {                                                                                                  \
        unsigned long long _wg = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;                 \
        if (threadIdx.x == 0 && _wg < WB_CSTAMP_WGS)                                                      \
            g_chan_stamps[_wg * WB_CSTAMP_SLOTS + (k)] = __builtin_amdgcn_s_memrealtime();                \
    } while (0)
#else
#define WB_CSTAMP(k) do {} while (0)
#endif

#ifndef WB_CHAN_UR
#define WB_CHAN_UR 8
#endif
// shrink 4 / uint8: R as bytes (1) or floats (0: the round-3 form, three workgroups per CU), and the waves per SIMD the
// register allocator is held to
#ifndef WB_CHAN_S4_BYTES
#define WB_CHAN_S4_BYTES 1
#endif
#ifndef WB_CHAN_S4_WAVES
#define WB_CHAN_S4_WAVES (WB_CHAN_S4_BYTES ? 5 : 3)
#endif
// shrink 4, grad_hist: output tile 8 x WB_CHAN_S4_TV.  30 (round 4): the shrunk tile with its smooth halo is then 10 x 32 = 320
// pixels = five full waves of step 2 (8 x 32 gave 340: a sixth wave ran for twenty lanes), and the resized tile 130 columns =
// two per lane + 2 left over (138: + 10)
#ifndef WB_CHAN_S4_TV
#define WB_CHAN_S4_TV 30
#endif
typedef WbTap Tap;   // one axis of the bilinear resample (scipy NI_ZoomShift, order 1), host-built table

// a double held by lane `k` (wave-uniform k), to every lane
__device__ inline double lane_f64(double v, int k) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// a byte from LDS through an address the compiler cannot relate to its neighbour's: two adjacent byte loads are
// otherwise merged into ONE 16-bit load at an arbitrary (odd) address, and unaligned LDS accesses are slow on gfx950
// (measured: -3.7 % on the whole kernel with the pairs kept apart; switching the compiler's unaligned-access mode off
// instead would also split the patch staging's unaligned global dword loads into bytes).  The pointer stays an LDS
// (address space 3) pointer through the barrier: a generic one made every such load a flat_load with 64-bit address
// arithmetic in front of it.
typedef const __attribute__((address_space(3))) unsigned char *LdsBytePtr;
__device__ inline uint8_t lds_byte_apart(const unsigned char *p) {
    LdsBytePtr q = (LdsBytePtr)p;
    asm volatile("" : "+v"(q));
    return *q;
}
// The same guarantee without the opaque pointer: a VOLATILE byte load is never merged with its neighbour, and -- unlike a
// load through a laundered pointer, which needs an address register of its own -- a constant displacement still folds into
// the instruction's offset field: the four tap bytes of a pixel (i0, i0 + 1 in two consecutive patch rows) come from ONE
// address register (round 4: one vector add per tap pair less in the row loop).
typedef const volatile __attribute__((address_space(3))) unsigned char *LdsVolBytePtr;
__device__ __forceinline__ uint8_t lds_byte_vol(const unsigned char *patch, int off) {
    return *((LdsVolBytePtr)patch + off);
}

// scipy's order-1 resample of one output pixel: fp64, taps and additions in NI_ZoomShift's order
__device__ inline double resample_f64(double v00, double v01, double v10, double v11, const Tap &tr, const Tap &tc) {
    double t = (v00 * tr.w0) * tc.w0;
    t = t + (v01 * tr.w0) * tc.w1;
    t = t + (v10 * tr.w1) * tc.w0;
    t = t + (v11 * tr.w1) * tc.w1;
    return t;
}

template <typename T> struct Src;
template <> struct Src<uint8_t> {
    static constexpr bool kFastResample = true;
    // The uint8 result is floor(clip(t)), so only the integer part of t matters.  An fp32 estimate
    // (4 bytes x weights rounded to fp32, fma chain) is within 1.1e-4 of the exact sum, and scipy's
    // fp64 value within 1e-12: unless the estimate lies within EPS of an integer both have the same
    // floor.  Lanes inside that band (flat 2x2 patches always are) redo the pixel in fp64.
    static constexpr float kEps = 2.5e-4f;
    // No clip here: outside the band the exact value lies strictly between two integers of
    // [min, max] (it is a convex combination of pixels of the octave), so its floor is in range.
    static __device__ bool fast(float v00, float v01, float v10, float v11, float wr0, float wr1, float wc0, float wc1,
                                float &out) {
        float top = __builtin_fmaf(v01, wc1, v00 * wc0), bot = __builtin_fmaf(v11, wc1, v10 * wc0);
        return fast_rows(top, bot, wr0, wr1, out);
    }
    // the same from the two rows' horizontal interpolations (a row's value is shared by the output rows that tap it)
    static __device__ bool fast_rows(float top, float bot, float wr0, float wr1, float &out) {
        // (opaque to the SLP vectoriser: paired into v_pk_mul / v_pk_fma / v_pk_add the two rows of a pass cost more issue
        // cycles than as plain fp32 instructions)
        float t = hold(__builtin_fmaf(bot, wr1, hold(top * wr0)));
        float fl = floorf(t), fr = hold(t - fl);
        out = fl;
        return fabsf(hold(fr - 0.5f)) <= 0.5f - kEps;
    }
    static __device__ __forceinline__ float hold(float v) {
        asm volatile("" : "+v"(v));
        return v;
    }
    static __device__ double lo(uint32_t k) { return (double)k; }
    // fp64 result is clipped in fp64, then cast to uint8 by truncation
    static __device__ float finish(double t, double mn, double mx, int) {
        t = fmin(fmax(t, mn), mx);
        return (float)(int)t;
    }
    static __device__ bool taps_finite(uint8_t, uint8_t, uint8_t) { return true; }
    // [1,2,1] pass: exact in fp32 for integer pixels (|.| <= 1020); 2b is exact, so the fma rounds like b*2 + (a+c)
    static __device__ float hpass(float a, float b, float c) { return __builtin_fmaf(b, 2.0f, a + c); }
    // [-1,0,1] pass: scipy multiplies the centre tap too (weight 0); integer pixels are finite, so it adds nothing
    static __device__ float dpass(float lo, float, float hi) { return lo - hi; }
};
template <> struct Src<float> {
    static constexpr bool kFastResample = false;
    static __device__ bool fast(float, float, float, float, float, float, float, float, float &) { return false; }
    static __device__ bool fast_rows(float, float, float, float, float &) { return false; }
    static __device__ double lo(uint32_t k) { return (double)wb_key_f32(k); }
    // float32 images: zoom stores fp32, then np.clip in fp32 -- np.minimum(np.maximum(x, lo), hi), which hands a NaN
    // through from x AND from a bound: an octave that holds a NaN pixel has a NaN min or max (wb_octaves.hip: the
    // keys order NaNs outside +-inf) and every pixel resized from it is NaN, as under NumPy
    static __device__ float finish(double t, double mn, double mx, int) {
        const float f = (float)t, lo = (float)mn, hi = (float)mx;
        if (lo != lo || hi != hi) return __builtin_nanf("");
        return f < lo ? lo : (f > hi ? hi : f);              // (a NaN f fails both tests and stays)
    }
    // a pixel copy stands for scipy's (v00*1)*1 + (v01*1)*0 + (v10*0)*1 + (v11*0)*0 only while the three taps of
    // weight 0 are finite (0 * inf = NaN)
    static __device__ bool taps_finite(float a, float b, float c) { return fabsf(a) < INFINITY && fabsf(b) < INFINITY && fabsf(c) < INFINITY; }
    // scipy correlate1d: fp64 accumulate, one fp32 rounding per pass
    static __device__ float hpass(float a, float b, float c) {
        return (float)((double)b * 2.0 + ((double)a + (double)c));
    }
    // the centre tap of weight 0 is part of the sum (correlate1d's antisymmetric branch): 0 * inf = NaN next to an
    // infinite value, as under scipy; for a finite centre it adds +-0
    static __device__ float dpass(float lo, float mid, float hi) { return (float)((double)mid * 0.0 + ((double)lo - (double)hi)); }
};

// float64 images, and integer images held as float64 (WB_DTYPE_F64 / WB_DTYPE_I8..U32): zoom in fp64, np.clip in
// fp64 to the octave's range, .astype(image dtype) -- truncation toward zero for the integer types -- and then the
// channel function's own astype("f") (reference channels.py:132, :41).  Gradients as for float32 images.
template <> struct Src<double> {
    static constexpr bool kFastResample = false;
    static __device__ bool fast(float, float, float, float, float, float, float, float, float &) { return false; }
    static __device__ bool fast_rows(float, float, float, float, float &) { return false; }
    static __device__ double lo(unsigned long long k) {
        const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        return __longlong_as_double((long long)b);
    }
    static __device__ float finish(double t, double mn, double mx, int src_int) {
        if (mn != mn || mx != mx) return __builtin_nanf("");     // (np.clip with a NaN bound: see Src<float>::finish)
        t = t < mn ? mn : (t > mx ? mx : t);
        switch (src_int) {
            case WB_CAST_TRUNC: t = trunc(t); break;
            case WB_CAST_BOOL: t = t != 0.0 ? 1.0 : 0.0; break;
            case WB_CAST_F16: t = wb_round_f16(t); break;
        }
        return (float)t;
    }
    static __device__ bool taps_finite(double a, double b, double c) { return fabs(a) < INFINITY && fabs(b) < INFINITY && fabs(c) < INFINITY; }
    static __device__ float hpass(float a, float b, float c) { return Src<float>::hpass(a, b, c); }
    static __device__ float dpass(float lo, float mid, float hi) { return Src<float>::dpass(lo, mid, hi); }
};

// the (min, max) an octave's resize result is clipped to, from the order-preserving keys the octave kernel left
// (32-bit keys for uint8 / float32 images, 64-bit ones for the float64-held dtypes; word 0 holds max(~key))
template <typename T> __device__ inline void clip_range(const ChanArgs &a, int b, int oct, double &mn, double &mx) {
    if constexpr (sizeof(T) == 8) {
        const unsigned long long *mm = reinterpret_cast<const unsigned long long *>(a.minmax) + ((int64_t)b * a.n_oct + oct) * 2;
        mn = Src<T>::lo(~mm[0]);
        mx = Src<T>::lo(mm[1]);
    } else {
        const uint32_t *mm = a.minmax + ((int64_t)b * a.n_oct + oct) * 2;
        mn = Src<T>::lo(~mm[0]);
        mx = Src<T>::lo(mm[1]);
    }
}

struct F4 {
    float x, y, z, w;
};

// Opaque to the optimiser: stops the SLP vectoriser from pairing neighbouring scalar fp32 operations into
// v_pk_* instructions -- on gfx950 a packed op issues in 4 cycles against 2 for each scalar op, and building
// its operand pairs costs extra v_movs (tools/valu_rate_probe.hip)
__device__ inline float scalar_only(float v) {
    asm volatile("" : "+v"(v));
    return v;
}

// reference channels.py:78-83: nine-term sum in source order; numba promotes int64*float32 to
// fp64, so the sum is fp64; "/16" and one rounding to fp32 on the store.
// 2*x and 4*x are exact, so fma(2, b, acc) rounds exactly like acc + 2*b: same bits, half the ops.
__device__ inline float smooth9(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
    double s = __builtin_fma(2.0, b, a);
    s = s + c;
    s = __builtin_fma(2.0, d, s);
    s = __builtin_fma(4.0, e, s);
    s = __builtin_fma(2.0, f, s);
    s = s + g;
    s = __builtin_fma(2.0, h, s);
    s = s + i;
    return (float)(s * 0.0625);
}

// grad_hist projection of one pixel: out[k] = | fp32( fp64(gx)*cos_k - fp64(gy)*sin_k ) |
// (reference channels.py:47-52)
__device__ inline void project_f64(float gx, float gy, const ChanArgs &a, float *out) {
    double gxd = (double)gx, gyd = (double)gy;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float val = (float)(gxd * a.cs[k] - gyd * a.sn[k]);
        out[k] = fmaxf(fabsf(val), 0.0f);
    }
}

// Integer-valued gradients (uint8 images: |g| <= 1020) with the canonical 4-bin constants:
// bit-identical to project_f64 for every (gx, gy) in [-1020, 1020]^2 -- checked exhaustively on
// the device by wb_selftest_projection -- and branch-free:
//   k=0: gx*1 - gy*0 = gx
//   k=2: gx*6.1e-17 - gy rounds to -gy unless gy == 0; then it is fp32(gx * cos(pi/2)), which a
//        two-float split of the constant reproduces
//   k=1,3: the result depends only on d = |gx -/+ gy| (the 1-ulp difference between the fp64 cos
//        and sin of pi/4 is far below fp32 resolution) and fp32(d * sin(pi/4)) equals
//        fma(d, chi, d*clo) for all d <= 2040 -- unless d == 0: then what is left is that 1-ulp
//        difference, |RN64(gx*c1) - RN64(gx*s1)| (0 or one ulp of the product), computed as such.
__device__ inline void project_int(float gx, float gy, const ChanArgs &a, float *out) {
    const float d1 = fabsf(gx - gy), d3 = fabsf(gx + gy), ax = fabsf(gx);
    const double g = (double)gx;
    const float tiny = fabsf((float)(g * a.cs[1] - g * a.sn[1]));
    const float o1 = __builtin_fmaf(d1, a.chi, d1 * a.clo), o3 = __builtin_fmaf(d3, a.chi, d3 * a.clo);
    out[0] = ax;
    out[1] = d1 == 0.0f ? tiny : o1;
    out[2] = gy == 0.0f ? __builtin_fmaf(ax, a.c2hi, ax * a.c2lo) : fabsf(gy);
    out[3] = d3 == 0.0f ? tiny : o3;
}

// project_int without the residues: the four channel values wherever they are "ordinary" (an integer or
// fp32(d*sin(pi/4)) >= 0.7071), and exactly 0 where the reference leaves a ~1e-13 residue (or a true 0).
// Under the 2x2 shrink a residue only shows in the result if every pixel of the block has one (or 0) in
// that channel: fp32 absorbs anything below 3.5e-13 into a value >= 0.7071, in any position of
// ((a+b)+c)+d.  So the shrink is first formed from these values, and only a block whose pooled value
// comes out 0 although it contains a gradient is redone with project_int (channels_kernel, step 2).
// The FAST path runs with the canonical constants only (wb_channels_launch checks), so sin(pi/4) as a two-float split is
// a compile-time constant: literal operands.  (Kernel arguments live in scalar registers, and an fp32 multiply / fmac
// with a scalar-register operand issues in 4 cycles instead of 2 on gfx950 -- tools/valu_class_probe.hip.)  The split
// product is odd in d, so it is formed from the signed difference and the |.| is left to the consumer's source modifier.
constexpr float kSinHi = 0x1.6a09e6p-1f;
constexpr float kSinLo = (float)(0x1.6a09e667f3bccp-1 - (double)kSinHi);
__device__ inline void project_ordinary(float gx, float gy, const ChanArgs &, float *out) {
    // (every result opaque to the SLP vectoriser: paired into v_pk_* the operations cost more issue cycles than plain ones,
    // and a packed instruction takes no |x| modifier -- eight v_and per shrunk pixel materialised the absolute values)
    const float d1 = gx - gy, d3 = gx + gy;
    out[0] = fabsf(gx);
    out[1] = fabsf(scalar_only(__builtin_fmaf(d1, kSinHi, scalar_only(d1 * kSinLo))));
    out[2] = fabsf(gy);
    out[3] = fabsf(scalar_only(__builtin_fmaf(d3, kSinHi, scalar_only(d3 * kSinLo))));
}

// Tile geometry shared by the channel kernels: TU x TV outputs per workgroup of NT threads, shrink S
template <int S_, int TU_, int TV_, bool SMOOTH_, int NT_ = 256> struct TileGeom {
    static constexpr int S = S_, TU = TU_, TV = TV_, NT = NT_, NW = NT_ / 64;
    static constexpr bool SMOOTH = SMOOTH_;
    static constexpr int HS = SMOOTH ? 1 : 0;
    static constexpr int SU = TU + 2 * HS, SV = TV + 2 * HS;  // shrunk tile incl. smooth halo
    static constexpr int RH = S * SU + 2, RW = S * SV + 2;    // resized tile incl. the 3x3 gradient halo
    static constexpr int P = S + 2;                           // patch side per shrunk pixel
    // LDS: R (resized tile) | one region shared by the uint8 source patch (live in step 1 only)
    // and the shrunk tile Sh (live from step 2 on).  Source patch capacity: no larger than a
    // float4 Sh, so that R + region stay under 40 KiB (4 workgroups per CU); tiles of the most
    // down-scaled levels of an octave that do not fit take the direct path
    // (shrink 2: 74 rows x 256 bytes instead of 80 x 236 -- the most down-scaled level of an octave of 8, zoom step
    // 1.834, needs 72 rows of 254 bytes and took the direct path before)
    // (32-row tiles: 70 resized rows at zoom step 1.834 tap 130 source rows)
    // (shrink 4, an extension: 42 x 138 resized pixels per 8 x 32 outputs tap up to 86 source rows of 278 bytes -- five
    // times the shrunk tile; the patch gets its own size, three workgroups per CU.  Sized after the shrunk tile, as until
    // round 3, no shrink-4 tile was ever staged: four byte gathers per resized pixel straight from memory)
    static constexpr int PROWS = S == 2 ? (TU == 16 ? 74 : 2 * RH - 8) : 2 * RH + 4;
    static constexpr int PPITCH = S == 4 ? 2 * RW + 12 : (S == 2 && TU != 16) ? 256 : ((SU * SV * 16) / PROWS) & ~3;
    static_assert(S == 4 || PROWS * PPITCH <= SU * SV * 16, "the source patch shares the shrunk tile's memory");
    static_assert(PPITCH % 4 == 0, "patch rows are written as dwords");
    static constexpr int SH_BYTES = SU * SV * 16;
    static constexpr int PATCH_BYTES = PROWS * PPITCH;
};

// ---- step 1 of every channel kernel: bilinear resample of the tile (+ 1-pixel gradient halo) into
//      R, cast back to the image dtype.
//      One tile row per wave at a time: the row's taps are wave-uniform (scalar registers,
//      scalar row base pointers), the column taps of a lane's NCS columns live in registers,
//      and the 4*NCS source loads of a row are issued before any arithmetic.  Coordinates are
//      clamped to the level = the 'reflect' halo of convolve1d for a 1-pixel border.
//      The RW % 64 right-most columns are done afterwards, one pixel per thread.
//      Ends without a barrier: the caller synchronises before reading R.
__host__ __device__ inline int reflect_index(int i, int n) {      // scipy 'reflect': (d c b a | a b c d | d c b a)
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - 1 - i : i;
}

// Coordinate i of a tile (possibly outside its level of n pixels) -> the level pixel it stands for: clamped (= the
// 'reflect' halo of a 1-pixel border: the gradient kernels) or reflected (grad_mag's 6-pixel halo).
template <bool REFLECT> __host__ __device__ __forceinline__ int tile_coord(int i, int n) {
    if constexpr (REFLECT)
        return reflect_index(i, n);
    else
        return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}

// ... and bounds [lo_out, hi_out] on the level pixels the coordinates lo..hi stand for (conservative under reflection:
// they only size the staged source patch).
template <bool REFLECT> __host__ __device__ __forceinline__ void tile_coord_range(int lo, int hi, int n, int &lo_out, int &hi_out) {
    if constexpr (!REFLECT) {
        lo_out = tile_coord<false>(lo, n);
        hi_out = tile_coord<false>(hi, n);
    } else if (lo >= 0 && hi < n) {
        lo_out = lo;
        hi_out = hi;
    } else if (lo < -n || hi >= 2 * n || (lo < 0 && hi >= n)) {
        lo_out = 0;
        hi_out = n - 1;
    } else if (lo < 0) {                                  // mirrored at the top / left edge: -1 - i
        lo_out = hi < 0 ? -1 - hi : 0;
        hi_out = hi < 0 ? -1 - lo : (hi > -1 - lo ? hi : -1 - lo);
    } else {                                              // mirrored at the bottom / right edge: 2n - 1 - i
        hi_out = lo >= n ? 2 * n - 1 - lo : n - 1;
        lo_out = lo >= n ? 2 * n - 1 - hi : (lo < 2 * n - 1 - hi ? lo : 2 * n - 1 - hi);
    }
}

// The source patch a uint8 tile stages: rows r_lo .. r_lo + nrow - 1, bytes c_lo .. c_lo + nbyte - 1 of the level's octave,
// for tile rows ry0 .. ry0 + rh - 1 and columns rx0 .. rx0 + RW - 1; false when the tile does not stage one.  Strict
// down-scale on both axes: every tap pair is (i0, i0 + 1), no mirroring (plan.axis_taps), and the extents follow from the
// first and last coordinate's i0 = floor((k + 0.5) * step - 0.5) -- the host's own fp64 expression for the tap table.
// The SAME function runs on the host (wb_channels_tile_patches: IEEE fp64 on both sides, no contraction) and, without a
// table, in every workgroup.
template <typename G, bool REFLECT>
__host__ __device__ __forceinline__ bool tile_patch_extent(const WbLevel &L, int ry0, int rx0, int rh, int &r_lo, int &c_lo, int &nrow,
                                                           int &nbyte) {
    int yf, yl, xf, xl;
    tile_coord_range<REFLECT>(ry0, ry0 + rh - 1, L.nh, yf, yl);
    tile_coord_range<REFLECT>(rx0, rx0 + G::RW - 1, L.nw, xf, xl);
    const bool ident = (L.src_h == L.nh) && (L.src_w == L.nw);
    const bool strict = !ident && L.src_h > L.nh && L.src_w > L.nw;
    auto first_tap = [](int k, double step) { return (int)floor(((double)k + 0.5) * step - 0.5); };
    r_lo = first_tap(yf, L.sy);
    c_lo = first_tap(xf, L.sx);
    const int r_hi = first_tap(yl, L.sy) + 1, c_hi = first_tap(xl, L.sx) + 1;
    nrow = r_hi - r_lo + 1;
    nbyte = c_hi - c_lo + 1;
    return strict && nrow + 1 <= G::PROWS && nbyte + 8 <= G::PPITCH;
}

// RT: how R holds a resized pixel -- float, or (uint8 images only: the pixels are integers 0..255) one byte, rows padded to
// whole dwords: a quarter of the LDS, for the price of one conversion per store here and one per read in the caller.
template <typename RT, int RW> struct RPitch { static constexpr int value = sizeof(RT) == 1 ? ((RW + 3) & ~3) : RW; };

// rows of a wave's strip of the resample: an even share of the rh tile rows, rounded up to whole passes of WB_CHAN_RB rows --
// 42 rows on four waves are then 12 + 12 + 12 + 6 (21 passes) instead of 11 + 11 + 11 + 9 (23: every wave ended on a pass of
// one row)
#ifndef WB_CHAN_RB
#define WB_CHAN_RB 2
#endif
__host__ __device__ constexpr int wb_strip_rows(int rh, int nw) { return WB_CHAN_RB * ((rh + nw * WB_CHAN_RB - 1) / (nw * WB_CHAN_RB)); }

template <typename T, typename G, bool REFLECT = false, typename RT = float>
__device__ __forceinline__ void resample_tile(const ChanArgs &a, const WbLevel &L, const T *src, const double mn,
                                              const double mx, const int ry0, const int rx0, const int rh, RT *R,
                                              unsigned char *uni, float4 *rowtab, const int tid) {
    static_assert(sizeof(RT) == 4 || sizeof(T) == 1, "byte R holds uint8 pixels");
    constexpr int RP = RPitch<RT, G::RW>::value;             // R's row pitch in elements
    auto rput = [&](int idx, float v) {
        if constexpr (sizeof(RT) == 1) R[idx] = (RT)(int)v; else R[idx] = v;
    };
    // rh <= RH: the tile rows that are needed (a tile on the bottom edge of its level uses fewer): wave-uniform, the
    // strips below are cut from it
    constexpr int RH = G::RH, RW = G::RW, PPITCH = G::PPITCH, NT = G::NT, NW = G::NW;
    const Tap *__restrict__ rtap = a.taps + L.tap_off;      // row taps [nh], then column taps [nw]
    const Tap *__restrict__ ctap = rtap + L.nh;
    constexpr int NCS = RW / 64, MAINW = NCS * 64, LEFT = RW - MAINW;
    // (readfirstlane: the wave index is the same in every lane -- said explicitly, the row loops below run on
    // scalar counters and branches instead of vector compares and exec masks)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Levels at their octave's own size (scale 1: every level i=0 with even dims) resample with
    // weights (1, 0): t = v*1*1 + 0 + 0 + 0 = v exactly -> plain copy.
    const bool ident = (L.src_h == L.nh) && (L.src_w == L.nw);
    if constexpr (sizeof(T) == 1) {
        // ... and for a tile that lies inside the level (no clamped coordinate) the copy is done four pixels at a time: one
        // (unaligned) dword load, four byte conversions, two 8-byte LDS stores -- every load of the tile in flight at
        // once, no taps.  One level in eight is such a level and it is the largest of its octave (21 % of all tiles).
        constexpr int RWD = (RW + 3) / 4;
        if (ident && ry0 >= 0 && ry0 + RH <= L.nh && rx0 >= 0 && rx0 + 4 * RWD <= L.nw) {
            static_assert(RW % 2 == 0, "pixel pairs");
            typedef uint32_t __attribute__((aligned(1))) u32u;
            constexpr int NE = RH * RWD, PER = (NE + NT - 1) / NT;
            uint32_t v[PER];
            int at[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                int e = tid + NT * i;
                e = e < NE ? e : NE - 1;                              // (duplicates rewrite the same values)
                const int k = e / RWD, d = e - k * RWD;
                v[i] = *reinterpret_cast<const u32u *>(src + (int64_t)(ry0 + k) * L.src_w + rx0 + 4 * d);
                at[i] = k * RP + 4 * d;
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                if constexpr (sizeof(RT) == 1) {
                    *reinterpret_cast<uint32_t *>(R + at[i]) = v[i];        // (the source bytes ARE the pixels; rows are whole dwords)
                } else {
                    float2 *dst = reinterpret_cast<float2 *>(R + at[i]);
                    dst[0] = make_float2((float)(v[i] & 0xffu), (float)((v[i] >> 8) & 0xffu));
                    if (at[i] % RW + 2 < RW) dst[1] = make_float2((float)((v[i] >> 16) & 0xffu), (float)(v[i] >> 24));
                }
            }
            WB_CSTAMP(1);
            WB_CSTAMP(2);
            WB_CSTAMP(3);
            return;
        }
    }
    // uint8 images: the tile's source patch (rows r_lo..r_hi, columns c_lo..c_hi of the octave) is
    // first copied to LDS with coalesced dword loads; the 4 taps of every pixel are then LDS byte
    // reads.  (Fetched straight from HBM they were 4 byte-gathers per pixel and the texture-address
    // unit, not the ALUs, set the pace.)  Falls back to direct loads if the patch would not fit
    // (strongly down-scaled tiny levels) and for float32 images.
    bool staged = false;
    int r_lo = 0, c_lo = 0;
    Tap tcs[NCS], trl, tleft;
    trl.i0 = trl.i1 = 0; trl.w0 = trl.w1 = 0.0;
    tleft = trl;
#pragma unroll
    for (int c = 0; c < NCS; ++c) tcs[c] = trl;
    if constexpr (sizeof(T) == 1) {
        // the patch extents: from the host's per-tile table when there is one (wb_channels_launch_x: a scalar load right
        // behind the tile record), else computed here -- four chains of fp64 arithmetic in front of every patch load
        int nrow, nbyte;
        if (a.patches) {
            const WbTilePatch tp = a.patches[blockIdx.x];
            r_lo = tp.r_lo;
            c_lo = tp.c_lo;
            nrow = tp.rows;
            nbyte = tp.bytes;
            staged = nrow != 0;
        } else {
            // (fp64 has no scalar unit: the values are computed by the vector ALU in every lane alike -- said explicitly,
            // so that everything derived from them, the staging loop's buffer descriptor included, stays in scalar registers)
            staged = tile_patch_extent<G, REFLECT>(L, ry0, rx0, rh, r_lo, c_lo, nrow, nbyte);
            r_lo = __builtin_amdgcn_readfirstlane(r_lo);
            c_lo = __builtin_amdgcn_readfirstlane(c_lo);
            nrow = __builtin_amdgcn_readfirstlane(nrow);
            nbyte = __builtin_amdgcn_readfirstlane(nbyte);
        }
        WB_CSTAMP(1);
        // the taps the resample below wants -- a lane's column taps, the row taps of the wave's strip (lane l: its row l), the taps of the
        // RW % 64 right-most columns -- are requested HERE, in front of the patch loads: behind the staging barrier each
        // of these loads was one more exposed memory round trip per workgroup
#pragma unroll
        for (int c = 0; c < NCS; ++c) {
            const int x = tile_coord<REFLECT>(rx0 + lane + 64 * c, L.nw);
            tcs[c] = ctap[x];
        }
        {
            static_assert(wb_strip_rows(RH, NW) <= 64, "one lane per row of the strip");
            const int RS = wb_strip_rows(rh, NW);               // rows of a wave's strip (see the row loop)
            const int kl = wave * RS + lane;
            const int ly = tile_coord<REFLECT>(ry0 + (kl < rh ? kl : rh - 1), L.nh);
            trl = rtap[ly];
            const int lx = tile_coord<REFLECT>(rx0 + MAINW + (lane < LEFT ? lane : 0), L.nw);
            tleft = ctap[lx];
        }
        if (staged) {
            // LDS row r = source row r_lo + r from column c_lo on: dword loads at byte granularity
            // (global memory takes unaligned dwords), aligned LDS stores
            // One patch row per wave at a time, one dword per lane (no index arithmetic per element);
            // UR rows are in flight together.  Lanes past the row end reload its last dword.
            constexpr int DWP = PPITCH / 4;                       // dwords per patch row
            const int ndw = (nbyte + 1 + 3) / 4;                  // + the (i0 + 1) neighbour of the last column
            uint32_t *pw = reinterpret_cast<uint32_t *>(uni);
            constexpr int UR = WB_CHAN_UR;
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
            for (int dw0 = 0; dw0 < ndw; dw0 += 64) {
                int dw = dw0 + ln;
                dw = dw < ndw ? dw : ndw - 1;
                // a row's address = the patch origin (a buffer descriptor built from wave-uniform values: scalar registers)
                // + the row's byte offset (a scalar: the instruction's soffset) + the lane's byte offset (a 32-bit vector
                // register): the buffer load's own addressing mode -- no 64-bit vector multiply-add per row (round 4; plain
                // pointer arithmetic is folded back into per-lane 64-bit pointers by the compiler)
                const int voff = 4 * dw;
                // (the origin is wave-uniform but the compiler cannot prove it and would wrap every load in a waterfall loop:
                // its two halves go through readfirstlane)
                const uint64_t origin = reinterpret_cast<uint64_t>(src + (int64_t)r_lo * L.src_w + c_lo);
                const uint64_t origin_u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(origin >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)origin);
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(origin_u), 0, 0x7fffffff, 0x00020000);
                for (int r0 = wv; r0 < nrow; r0 += NW * UR) {
                    uint32_t v[UR];
                    int rr[UR];
#pragma unroll
                    for (int k = 0; k < UR; ++k) {
                        rr[k] = r0 + NW * k < nrow ? r0 + NW * k : nrow - 1;
                        v[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, rr[k] * L.src_w, 0);
                    }
#pragma unroll
                    for (int k = 0; k < UR; ++k) pw[rr[k] * DWP + dw] = v[k];   // duplicates rewrite the same value
                }
            }
            // the row taps of the tile, one entry per tile row: {patch byte offset of the upper tap row, fp32 weights}.
            // Read back below with one wave-uniform (broadcast) LDS load per row: the values arrive in VECTOR registers
            // -- on gfx950 an fp32 add / multiply / fmac whose operands are all vector registers issues in 2 cycles,
            // with a scalar-register operand in 4 (tools/valu_class_probe.hip), and a v_readlane costs 4 as well
            {
                const int RS = wb_strip_rows(rh, NW);
                const int kl = wave * RS + lane;
                if (lane < RS && kl < rh) rowtab[kl] = make_float4(__int_as_float((trl.i0 - r_lo) * PPITCH), (float)trl.w0, (float)trl.w1, 0.0f);
            }
            if (LEFT > 0 && tid >= 64 && tid < 64 + LEFT)
                rowtab[RH + tid - 64] = make_float4(__int_as_float(tleft.i0 - c_lo), (float)tleft.w0, (float)tleft.w1, 0.0f);
            __syncthreads();
        }
    }
    WB_CSTAMP(2);
    if (staged) {
        if constexpr (sizeof(T) == 1) {
            const unsigned char *patch = uni;
            int ci0[NCS];
            float wc0f[NCS], wc1f[NCS];
            const Tap (&tc)[NCS] = tcs;
#pragma unroll
            for (int c = 0; c < NCS; ++c) {
                ci0[c] = tc[c].i0 - c_lo;
                wc0f[c] = (float)tc[c].w0;
                wc1f[c] = (float)tc[c].w1;
            }
            // Each wave owns a strip of consecutive tile rows and walks it RB rows per pass: every tap byte of the pass is
            // requested before the first is used, and the exact redo (fp64, the lane-held fp64 taps -- lane l holds the row
            // taps of row l of the strip: trl) is deferred behind all the fast-path arithmetic, one branch per pass.
            // Consecutive output rows of a down-scale by less than 2 usually share a source row (the lower taps of row k are
            // the upper taps of row k + 1): its horizontal interpolation is then taken over instead of read and computed
            // again; which rows share is wave-uniform.
            // Round 4: the passes are unrolled completely (a strip holds at most RSMAX rows), so nothing is carried
            // around a loop back-edge -- the rolled loop spent 30 of its 88 vector instructions per pass on register moves
            // (next pass's row entries, the previous row's interpolation and tap bytes) --, a pixel's four tap bytes hang
            // off ONE address register (volatile loads, see lds_byte_vol), the tap bytes are not kept for the redo (it reads
            // them again: a redo is rare per pixel), and nothing is left for the SLP vectoriser to pair.
            constexpr int RB = WB_CHAN_RB;
            constexpr int RSMAX = wb_strip_rows(RH, NW), NPASS = RSMAX / RB;
            const int RS = wb_strip_rows(rh, NW);
            const int k_lo = wave * RS, k_hi = k_lo + RS < rh ? k_lo + RS : rh;
            float hprev[NCS];                     // horizontal interpolation of the patch row at byte offset o_prev
            int o_prev = -1;
#pragma unroll
            for (int c = 0; c < NCS; ++c) hprev[c] = 0.0f;
            auto hlerp = [&](uint8_t x0, uint8_t x1, int c) {
                return scalar_only(__builtin_fmaf((float)x1, wc1f[c], scalar_only((float)x0 * wc0f[c])));
            };
            const int rrow = k_lo * RP + lane;    // this lane's first output of the strip
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int k0 = k_lo + RB * ps;
                if (k0 >= k_hi) break;                                          // wave-uniform
                float4 ent[RB];
                int o0[RB], av[RB][NCS];
                bool shared[RB];
                uint8_t b[RB][NCS][4];
                // every load of the pass first, unconditionally (a row past the strip's end repeats the last one; the upper
                // tap pair is fetched even where the previous row's interpolation will stand in for it -- a branch around
                // two byte loads made the compiler wait for them inside the branch, one LDS latency per row)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    int k = k0 + rb;
                    k = k < k_hi ? k : k_hi - 1;
                    ent[rb] = rowtab[k];                                        // wave-uniform address: a broadcast read
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    o0[rb] = __builtin_amdgcn_readfirstlane(__float_as_int(ent[rb].x));
                    // (a00, a01) / (a10, a11) sit at i0, i0 + 1 of two consecutive patch rows; the upper pair's interpolation
                    // is not computed again when it is the previous row's lower pair
                    const int o_above = rb == 0 ? o_prev : o0[rb - 1] + PPITCH;
                    shared[rb] = o0[rb] == o_above;
#pragma unroll
                    for (int c = 0; c < NCS; ++c) {
                        av[rb][c] = ci0[c] + o0[rb];
                        b[rb][c][0] = lds_byte_vol(patch, av[rb][c]);
                        b[rb][c][1] = lds_byte_vol(patch, av[rb][c] + 1);
                        b[rb][c][2] = lds_byte_vol(patch, av[rb][c] + PPITCH);
                        b[rb][c][3] = lds_byte_vol(patch, av[rb][c] + PPITCH + 1);
                    }
                }
                float out[RB][NCS];
                bool need[RB][NCS];
                bool redo = false;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    float top[NCS];
                    if (shared[rb]) {                                           // wave-uniform
#pragma unroll
                        for (int c = 0; c < NCS; ++c) top[c] = hprev[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < NCS; ++c) top[c] = hlerp(b[rb][c][0], b[rb][c][1], c);
                    }
#pragma unroll
                    for (int c = 0; c < NCS; ++c) {
                        const float bot = hlerp(b[rb][c][2], b[rb][c][3], c);
                        hprev[c] = bot;
                        need[rb][c] = !Src<T>::fast_rows(top[c], bot, ent[rb].y, ent[rb].z, out[rb][c]);
                        redo |= need[rb][c];
                    }
                    o_prev = o0[rb] + PPITCH;
                }
                if (__builtin_amdgcn_ballot_w64(redo) != 0) {              // rare: exact fp64 with the full taps
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        int k = k0 + rb;
                        k = k < k_hi ? k : k_hi - 1;
                        // the row's fp64 weights come from the lane that holds them (no memory access: a load
                        // from the tap table here stalled the whole pass behind an L2 round trip)
                        Tap tr;
                        tr.i0 = tr.i1 = 0;
                        tr.w0 = lane_f64(trl.w0, k - k_lo);
                        tr.w1 = lane_f64(trl.w1, k - k_lo);
#pragma unroll
                        for (int c = 0; c < NCS; ++c) {
                            if (need[rb][c])
                                out[rb][c] = Src<T>::finish(resample_f64((double)b[rb][c][0], (double)b[rb][c][1], (double)b[rb][c][2],
                                                                         (double)b[rb][c][3], tr, tc[c]), mn, mx, a.src_int);
                        }
                    }
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    if (k0 + rb < k_hi) {
#pragma unroll
                        for (int c = 0; c < NCS; ++c) rput(rrow + (RB * ps + rb) * RP + 64 * c, out[rb][c]);
                    }
                }
            }
            // the RW % 64 right-most columns of the wave's own strip, one pixel per lane: row and column entries from the
            // LDS tables, the four tap bytes off one address; coordinates and the fp64 taps only in the (rare) exact redo
            if constexpr (LEFT > 0) {
                const int nleft = (k_hi - k_lo) * LEFT;
                for (int p = lane; p < nleft; p += 64) {
                    const int kk = p / LEFT, q = p - kk * LEFT, k = k_lo + kk;
                    const float4 er = rowtab[k], ec = rowtab[RH + q];
                    const int o = __float_as_int(er.x) + __float_as_int(ec.x);
                    const uint8_t a00 = lds_byte_vol(patch, o), a01 = lds_byte_vol(patch, o + 1);
                    const uint8_t a10 = lds_byte_vol(patch, o + PPITCH), a11 = lds_byte_vol(patch, o + PPITCH + 1);
                    float out = 0.0f;
                    if (!Src<T>::fast((float)a00, (float)a01, (float)a10, (float)a11, er.y, er.z, ec.y, ec.z, out)) {
                        const int y = tile_coord<REFLECT>(ry0 + k, L.nh), x = tile_coord<REFLECT>(rx0 + MAINW + q, L.nw);
                        const Tap tr = rtap[y], tcl = ctap[x];
                        out = Src<T>::finish(resample_f64((double)a00, (double)a01, (double)a10, (double)a11, tr, tcl), mn, mx, a.src_int);
                    }
                    rput(k * RP + MAINW + q, out);
                }
            }
        }
    } else
    {
        Tap tc[NCS];
        float wc0f[NCS], wc1f[NCS];
#pragma unroll
        for (int c = 0; c < NCS; ++c) {
            const int x = tile_coord<REFLECT>(rx0 + lane + 64 * c, L.nw);
            tc[c] = ctap[x];
            wc0f[c] = (float)tc[c].w0;
            wc1f[c] = (float)tc[c].w1;
        }
        // RB rows per pass: all their source loads are in flight before the first one is used
        // (one row at a time, the loop was a chain of RH/4 memory latencies per wave)
        constexpr int RB = sizeof(T) == 8 ? 2 : 5;             // (a double pixel is two registers)
        for (int k0 = wave; k0 < rh; k0 += NW * RB) {
            Tap tr[RB];
            T v00[RB][NCS], v01[RB][NCS], v10[RB][NCS], v11[RB][NCS];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                int k = k0 + NW * rb;
                k = k < rh ? k : rh - 1;                                  // clamped, unconditional loads
                const int y = tile_coord<REFLECT>(ry0 + k, L.nh);
                tr[rb] = rtap[__builtin_amdgcn_readfirstlane(y)];
                const T *r0 = src + (int64_t)__builtin_amdgcn_readfirstlane(tr[rb].i0) * L.src_w;
                const T *r1 = src + (int64_t)__builtin_amdgcn_readfirstlane(tr[rb].i1) * L.src_w;
#pragma unroll
                for (int c = 0; c < NCS; ++c) {
                    v00[rb][c] = r0[tc[c].i0];
                    v01[rb][c] = r0[tc[c].i1];
                    v10[rb][c] = r1[tc[c].i0];
                    v11[rb][c] = r1[tc[c].i1];
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int k = k0 + NW * rb;
#pragma unroll
                for (int c = 0; c < NCS; ++c) {
                    float out = 0.0f;
                    if (ident && Src<T>::taps_finite(v01[rb][c], v10[rb][c], v11[rb][c])) {
                        // (float32 / integer results pass the clip unchanged: the pixel lies in its octave's range;
                        // a NaN bound turns every pixel of the level into NaN, also the copied ones)
                        out = (mn != mn || mx != mx) ? __builtin_nanf("") : (float)v00[rb][c];
                    } else {
                        bool ok = false;
                        if constexpr (Src<T>::kFastResample)
                            ok = Src<T>::fast((float)v00[rb][c], (float)v01[rb][c], (float)v10[rb][c], (float)v11[rb][c],
                                              (float)tr[rb].w0, (float)tr[rb].w1, wc0f[c], wc1f[c], out);
                        if (!ok)
                            out = Src<T>::finish(resample_f64((double)v00[rb][c], (double)v01[rb][c], (double)v10[rb][c],
                                                              (double)v11[rb][c], tr[rb], tc[c]), mn, mx, a.src_int);
                    }
                    if (k < rh) rput(k * RP + lane + 64 * c, out);
                }
            }
        }
    }
    WB_CSTAMP(3);
    if constexpr (LEFT > 0) {
        // (a staged tile has done these columns wave by wave above)
        for (int p = staged ? rh * LEFT : tid; p < rh * LEFT; p += NT) {
            const int k = p / LEFT, q = MAINW + p - k * LEFT;
            const int y = tile_coord<REFLECT>(ry0 + k, L.nh), x = tile_coord<REFLECT>(rx0 + q, L.nw);
            float out = 0.0f;
            bool ok = false;
            const Tap tr = rtap[y], tc = ctap[x];
            const T *r0 = src + (int64_t)tr.i0 * L.src_w;
            const T *r1 = src + (int64_t)tr.i1 * L.src_w;
            const T a00 = r0[tc.i0], a01 = r0[tc.i1], a10 = r1[tc.i0], a11 = r1[tc.i1];
            ok = ident && Src<T>::taps_finite(a01, a10, a11);
            if (ok) out = (mn != mn || mx != mx) ? __builtin_nanf("") : (float)a00;
            if constexpr (Src<T>::kFastResample)
                if (!ok) ok = Src<T>::fast((float)a00, (float)a01, (float)a10, (float)a11, (float)tr.w0, (float)tr.w1,
                                           (float)tc.w0, (float)tc.w1, out);
            if (!ok) out = Src<T>::finish(resample_f64((double)a00, (double)a01, (double)a10, (double)a11, tr, tc), mn, mx, a.src_int);
            rput(k * RP + q, out);
        }
    }
}

// (launch bound: 4 workgroups = 4 waves per SIMD is what the 39 KB of LDS admit; without it the register
// allocator may trade that occupancy for a few more registers -- measured: 138 VGPRs, 3 waves per SIMD, +17 % time)
// (NT threads per workgroup: 256 for the 16 x 64 tile, 512 for the 32 x 64 tile -- the same 4 waves per SIMD either way)
template <typename T, int S, int TU, int TV, bool SMOOTH, bool FAST, int NT>
__global__ __launch_bounds__(NT, sizeof(T) == 8 ? 1 : S == 4 ? (sizeof(T) == 1 ? WB_CHAN_S4_WAVES : 1) : 4) void channels_kernel(ChanArgs a) {
    using G = TileGeom<S, TU, TV, SMOOTH, NT>;
    constexpr int HS = G::HS, SV = G::SV, RH = G::RH, RW = G::RW, P = G::P;
    constexpr int PATCH_BYTES = sizeof(T) == 1 ? G::PATCH_BYTES : 0;
    constexpr int UNI_MIN = G::SH_BYTES > PATCH_BYTES ? G::SH_BYTES : PATCH_BYTES;
    constexpr int UNI_LUT = (S == 4 && sizeof(T) == 1 && WB_CHAN_S4_BYTES) ? ((G::SH_BYTES + 15) & ~15) + WB_BIN16_LUT_BYTES : 0;
    constexpr int UNI_BYTES = UNI_MIN > UNI_LUT ? UNI_MIN : UNI_LUT;
    // Shrink 4, uint8 images (round 4): R holds the resized pixels as BYTES (they are integers 0..255) -- 5.9 KB instead of
    // 23 KB, and the rank tables are parked behind the shrunk tile in `uni` (the dead source patch) instead of in R: 31 KB of
    // LDS per workgroup = five per CU instead of three.  The kernel at this shrink is latency-bound (16 resized pixels per
    // output: three workgroups kept the vector ALUs 42 % busy), so residency is what it wants; the price is one conversion
    // per R store and 36 per shrunk pixel's patch read.
    constexpr bool RBYTES = S == 4 && sizeof(T) == 1 && WB_CHAN_S4_BYTES;
    using RT = typename std::conditional<RBYTES, uint8_t, float>::type;
    constexpr int RP = RPitch<RT, RW>::value;
    constexpr bool LUT_IN_UNI = RBYTES;
    // (the tables of either width: WB_BIN16_LUT_BYTES is the larger)
    static_assert(WB_BIN16_LUT_BYTES >= WB_BIN_LUT_BYTES, "table sizes");
    constexpr int R_BYTES = LUT_IN_UNI ? RH * RP : (RH * RW * 4 > WB_BIN16_LUT_BYTES ? RH * RW * 4 : WB_BIN16_LUT_BYTES);   // (float R later holds the rank tables)
    __shared__ __attribute__((aligned(16))) unsigned char Rraw[R_BYTES];
    RT *R = reinterpret_cast<RT *>(Rraw);
    __shared__ __attribute__((aligned(16))) unsigned char uni[UNI_BYTES];
    unsigned char *lut_lds = LUT_IN_UNI ? uni + ((G::SH_BYTES + 15) & ~15) : Rraw;
    __shared__ uint32_t odd_values;      // set when a shrunk value lies outside the exact-sum range (see step 3)
    __shared__ float4 rowtab[sizeof(T) == 1 ? RH + RW % 64 : 1];   // row taps of the tile, taps of the RW % 64 last columns (uint8 images, staged patch)
    F4 *Sh = reinterpret_cast<F4 *>(uni);
    constexpr bool SEPARABLE = SMOOTH && FAST;
    if (SEPARABLE && threadIdx.x == 0) odd_values = 0;

    const WbTile tile = a.tiles[blockIdx.x];
    const WbLevel L = a.levels[tile.level];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int u0 = tile.ty * TU, v0 = tile.tx * TV;

    const T *src = (L.oct == 0) ? (const T *)a.img + (int64_t)b * a.img_stride
                                : (const T *)a.oct + (int64_t)b * a.oct_stride + L.src_off;
    double mn, mx;
    clip_range<T>(a, b, L.oct, mn, mx);

    const int ry0 = S * (u0 - HS) - 1, rx0 = S * (v0 - HS) - 1;
    // a tile on the bottom edge of its level holds fewer than TU output rows: the shrunk rows (su_need) and resized
    // rows (rh_need) behind them are all the steps below compute (7 % of the tiles' rows at 1080p lie past an edge)
    const int vrows = L.u - u0 < TU ? L.u - u0 : TU;
    const int su_need = vrows + 2 * HS, rh_need = S * su_need + 2;
    WB_CSTAMP(0);
    resample_tile<T, G, false, RT>(a, L, src, mn, mx, ry0, rx0, rh_need, R, uni, rowtab, tid);
    __syncthreads();
    WB_CSTAMP(4);
    if (a.dbg & 1) return;

    // ---- step 2: gradients -> 4 oriented channels -> shrink, one shrunk pixel per call
    //      Rp: the pixel's (S + 2) x (S + 2) patch of R, Shp: where its shrunk value goes
    auto shrunk_pixel = [&](const int ro, const int so) {     // (offsets, not pointers: R's alignment stays visible -- 8-byte reads)
        float pt[P][P];
        if constexpr (RBYTES) {
            // six bytes per patch row = two aligned dwords (the patch starts at column S * j = 4 j of a dword-padded row)
            static_assert(P == 6 && RP % 4 == 0, "shrink-4 patch rows");
#pragma unroll
            for (int y = 0; y < P; ++y) {
                const uint32_t *w = reinterpret_cast<const uint32_t *>(R + ro + y * RP);
                const uint32_t w0 = w[0], w1 = w[1];
                pt[y][0] = (float)(w0 & 0xffu);
                pt[y][1] = (float)((w0 >> 8) & 0xffu);
                pt[y][2] = (float)((w0 >> 16) & 0xffu);
                pt[y][3] = (float)(w0 >> 24);
                pt[y][4] = (float)(w1 & 0xffu);
                pt[y][5] = (float)((w1 >> 8) & 0xffu);
            }
        } else {
#pragma unroll
            for (int y = 0; y < P; ++y)
#pragma unroll
                for (int x = 0; x < P; ++x) pt[y][x] = R[ro + y * RP + x];
        }

        float hc[S][P];   // vertical [1,2,1] pass at patch rows 1..S
        float hr[P][S];   // horizontal [1,2,1] pass at patch cols 1..S
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
            for (int x = 0; x < P; ++x) hc[y][x] = scalar_only(Src<T>::hpass(pt[y][x], pt[y + 1][x], pt[y + 2][x]));
#pragma unroll
        for (int y = 0; y < P; ++y)
#pragma unroll
            for (int x = 0; x < S; ++x) hr[y][x] = scalar_only(Src<T>::hpass(pt[y][x], pt[y][x + 1], pt[y][x + 2]));

        float ch[S][S][4], gxs[S][S], gys[S][S];
        constexpr bool TWO_PASS = FAST && S > 1;          // ordinary values first, residues only where they can show
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
            for (int x = 0; x < S; ++x) {
                const float gx = scalar_only(Src<T>::dpass(hc[y][x], hc[y][x + 1], hc[y][x + 2]));
                const float gy = scalar_only(Src<T>::dpass(hr[y][x], hr[y + 1][x], hr[y + 2][x]));
                gxs[y][x] = gx;
                gys[y][x] = gy;
                if constexpr (TWO_PASS)
                    project_ordinary(gx, gy, a, ch[y][x]);
                else if constexpr (FAST)
                    project_int(gx, gy, a, ch[y][x]);
                else
                    project_f64(gx, gy, a, ch[y][x]);
            }

        float o[4];
        auto pool = [&]() {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (S == 1) {
                    o[k] = ch[0][0][k];
                } else if constexpr (S == 2) {
                    o[k] = scalar_only(scalar_only(scalar_only(scalar_only(ch[0][0][k] + ch[1][0][k]) + ch[0][1][k]) + ch[1][1][k]) * 0.25f);
                } else {  // S == 4 (extension): avg_pool_2 applied twice
                    float q[2][2];
#pragma unroll
                    for (int A = 0; A < 2; ++A)
#pragma unroll
                        for (int B = 0; B < 2; ++B)
                            q[A][B] = scalar_only(scalar_only(scalar_only(scalar_only(ch[2 * A][2 * B][k] + ch[2 * A + 1][2 * B][k]) +
                                                                  ch[2 * A][2 * B + 1][k]) + ch[2 * A + 1][2 * B + 1][k]) * 0.25f);
                    o[k] = scalar_only(scalar_only(scalar_only(scalar_only(q[0][0] + q[1][0]) + q[0][1]) + q[1][1]) * 0.25f);
                }
            }
        };
        // integer gradients: a shrunk value is 0, or in [0.17, 1443] (some pixel of the block had an
        // ordinary value), or a sum of the 1e-13-sized residues of the projection -- the smooth wants to
        // know about the last kind (step 3)
        auto flag_odd = [&]() {
            const uint32_t lo = __float_as_uint(0.125f) - 1u;
            const uint32_t m01 = min(__float_as_uint(o[0]) - 1u, __float_as_uint(o[1]) - 1u);
            const uint32_t m23 = min(__float_as_uint(o[2]) - 1u, __float_as_uint(o[3]) - 1u);
            if (min(m01, m23) < lo) odd_values = 1;
        };
        pool();
        if constexpr (TWO_PASS) {
            // a pooled 0 in a block that has a gradient (pooled |gx| != 0): every pixel of the block holds a
            // residue or 0 in that channel -- rare; redo the block with the exact values
            if (o[0] != 0.0f && fminf(fminf(o[1], o[2]), o[3]) == 0.0f) {
#pragma unroll
                for (int y = 0; y < S; ++y)
#pragma unroll
                    for (int x = 0; x < S; ++x) project_int(gxs[y][x], gys[y][x], a, ch[y][x]);
                pool();
                if constexpr (SEPARABLE) flag_odd();
            }
        } else if constexpr (SEPARABLE) {
            flag_odd();
        }
        Sh[so] = F4{o[0], o[1], o[2], o[3]};
    };
    // Tiles whose shrunk width is a wave or a little more (the 16 x 64 tiles: 66 columns): a wave owns whole rows, lane =
    // column, so a pixel's LDS addresses are the previous round's plus a constant (no division by the width, no 64-bit
    // multiply-add per pixel: round 4); the few columns beyond the 64th go to the last wave, which owns the fewest rows.
    constexpr bool ROWMAP = SV >= 64 && SV - 64 <= 8 && NT % 64 == 0;
    if constexpr (ROWMAP) {
        constexpr int NWV = NT / 64, XC = SV - 64;
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        int ro = S * (wave * RP + lane), so = wave * SV + lane;
#pragma nounroll
        for (int i = wave; i < su_need; i += NWV) {                               // wave-uniform trip count
            shrunk_pixel(ro, so);
            ro += NWV * S * RP;
            so += NWV * SV;
        }
        if constexpr (XC > 0) {
            if (wave == NWV - 1) {
                for (int p = lane; p < su_need * XC; p += 64) {
                    const int i = p / XC, j = 64 + p - i * XC;
                    shrunk_pixel(S * (i * RP + j), i * SV + j);
                }
            }
        }
    } else {
        for (int p = tid; p < su_need * SV; p += NT) {
            const int i = p / SV, j = p - i * SV;
            shrunk_pixel(S * (i * RP + j), p);
        }
    }
    // rank tables of the model (12 KiB, L2-resident): requested before the barrier, parked in R -- dead once every
    // thread has left step 2 -- right behind it
    constexpr int LUT_VECS = WB_BIN_LUT_BYTES / 16, LUT16_VECS = WB_BIN16_LUT_BYTES / 16;
    auto ranks_wide_tag = [](const ChanArgs &aa) { return aa.rank != nullptr && aa.rank_wide != 0; };
    static_assert(LUT_VECS == 768 && (NT == 256 || NT == 512), "three vectors per thread (256 threads), one or two (512)");
    const bool ranks = a.rank != nullptr;
    uint4 lut0 = make_uint4(0, 0, 0, 0), lut1 = lut0, lut2 = lut0, lut3 = lut0, lut4 = lut0;
    const bool wide_lut = ranks_wide_tag(a);
    if (ranks) {
        lut0 = a.rank_lut[tid];
        if (NT == 256 || tid < 256) lut1 = a.rank_lut[tid + NT];
        if (NT == 256) lut2 = a.rank_lut[tid + 512];
        if (wide_lut) {                                       // (the 16-bit tables: 1280 vectors)
            static_assert(LUT16_VECS == 1280, "five vectors per thread (256 threads)");
            if (NT == 256) {
                lut3 = a.rank_lut[tid + 768];
                lut4 = a.rank_lut[tid + 1024];
            } else {
                if (tid >= 256) lut1 = a.rank_lut[tid + 512];     // 768 .. 1023 (threads 256..511 held nothing there)
                if (tid < 256) lut2 = a.rank_lut[tid + 1024];     // 1024 .. 1279
            }
        }
    }
    __syncthreads();
    if (ranks) {
        uint4 *lut = reinterpret_cast<uint4 *>(lut_lds);
        lut[tid] = lut0;
        if (NT == 256 || tid < 256) lut[tid + NT] = lut1;
        if (NT == 256) lut[tid + 512] = lut2;
        if (wide_lut) {
            if (NT == 256) {
                lut[tid + 768] = lut3;
                lut[tid + 1024] = lut4;
            } else {
                if (tid >= 256) lut[tid + 512] = lut1;
                if (tid < 256) lut[tid + 1024] = lut2;
            }
        }
    }
    WB_CSTAMP(5);

    if (a.dbg & 2) return;
    // ---- step 3: 3x3 binomial smooth (fp64 sum in source order, /16, one rounding), border = 0.
    //      Each thread owns RPT vertically adjacent outputs of one column, so every shrunk value
    //      it needs is read and widened to fp64 once for up to three output rows.
    // (tiles that do not split into whole strips -- 8 x 30 on 256 threads: one output per thread, the last threads idle)
    constexpr bool WHOLE = TU * TV % NT == 0 && NT % TV == 0;
    constexpr int RPT = WHOLE ? TU * TV / NT : 1;
    static_assert(WHOLE || TU * TV <= NT, "one output per thread");
    float *out = reinterpret_cast<float *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    const int j = tid % TV, i0 = (tid / TV) * RPT;
    const int sv = v0 + j;
    float o[RPT][4];
    // (64-wide tiles: a wave owns whole output rows, so on a bottom-edge tile the waves whose rows lie past the level
    // skip the smooth and the ranks -- wave-uniform; they still meet the barrier below)
    const bool live = WHOLE ? (TV != 64 || u0 + __builtin_amdgcn_readfirstlane(i0) < L.u) : i0 < TU;
#pragma unroll
    for (int y = 0; y < RPT; ++y) o[y][0] = o[y][1] = o[y][2] = o[y][3] = 0.0f;
    if (!live) {
    } else if constexpr (SMOOTH) {
        if (SEPARABLE && odd_values == 0) {
            // Every value of the tile is 0 or a float32 in [2^-3, 2^11): all partial sums of the nine
            // weighted terms are multiples of 2^-26 below 2^15 -- exact in fp64 in ANY order.  So the
            // row sums are formed once, row by row, and shared by the three output rows that use them.
            // Channels 0 and 2 -- |gx| and |gy| of integer gradients, pooled -- are multiples of 2^-2 (2^-4 under the
            // shrink-4 extension) below 2^10: their nine-term sums have at most 18 significant bits and are exact in
            // fp32 too, so these two channels need neither the conversions nor the fp64 arithmetic (the fp64 sum is the
            // same real number, /16 is exact, and the result is representable: identical bits).
            double s[3][2];
            float s32[3][2];
#pragma unroll
            for (int y = 0; y < RPT + 2; ++y) {
                const F4 c0 = Sh[(i0 + y) * SV + j], c1 = Sh[(i0 + y) * SV + j + 1], c2 = Sh[(i0 + y) * SV + j + 2];
                const float a0[4] = {c0.x, c0.y, c0.z, c0.w}, a1[4] = {c1.x, c1.y, c1.z, c1.w}, a2[4] = {c2.x, c2.y, c2.z, c2.w};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    s32[y % 3][h] = __builtin_fmaf(2.0f, a1[2 * h], a0[2 * h]) + a2[2 * h];
                    s[y % 3][h] = __builtin_fma(2.0, (double)a1[2 * h + 1], (double)a0[2 * h + 1]) + (double)a2[2 * h + 1];
                }
                if (y >= 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        o[y - 2][2 * h] = (__builtin_fmaf(2.0f, s32[(y - 1) % 3][h], s32[(y - 2) % 3][h]) + s32[y % 3][h]) * 0.0625f;
                        o[y - 2][2 * h + 1] = (float)((__builtin_fma(2.0, s[(y - 1) % 3][h], s[(y - 2) % 3][h]) + s[y % 3][h]) * 0.0625);
                    }
                }
            }
        } else {
            double w[RPT + 2][3][4];
#pragma unroll
            for (int y = 0; y < RPT + 2; ++y)
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    F4 c = Sh[(i0 + y) * SV + (j + x)];
                    w[y][x][0] = (double)c.x; w[y][x][1] = (double)c.y; w[y][x][2] = (double)c.z; w[y][x][3] = (double)c.w;
                }
#pragma unroll
            for (int y = 0; y < RPT; ++y)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[y][k] = smooth9(w[y][0][k], w[y][1][k], w[y][2][k], w[y + 1][0][k], w[y + 1][1][k], w[y + 1][2][k],
                                      w[y + 2][0][k], w[y + 2][1][k], w[y + 2][2][k]);
        }
    } else {
#pragma unroll
        for (int y = 0; y < RPT; ++y) {
            F4 c = Sh[(i0 + y) * SV + j];
            o[y][0] = c.x; o[y][1] = c.y; o[y][2] = c.z; o[y][3] = c.w;
        }
    }
    WB_CSTAMP(6);
#pragma unroll
    for (int y = 0; y < RPT; ++y) {
        const int su = u0 + i0 + y;
        if ((!WHOLE && !live) || su >= L.u || sv >= L.v || (a.dbg & 4)) continue;
        if (SMOOTH && (su == 0 || sv == 0 || su == L.u - 1 || sv == L.v - 1)) o[y][0] = o[y][1] = o[y][2] = o[y][3] = 0.0f;
        // one float4 per pixel ([u][v][4]): 64 lanes store 1 KiB contiguous
        if (a.chn) {
            float4 *dst = reinterpret_cast<float4 *>(out + ((int64_t)su * L.v + sv) * 4);
            *dst = make_float4(o[y][0], o[y][1], o[y][2], o[y][3]);
        }
    }
    if (ranks) {
        // the same pixels as threshold ranks, one dword per pixel -- or, 16-bit ranks, eight bytes (wb_common.h: wb_bin_rank)
        __syncthreads();                                      // the tables are in LDS
        const float *Sthr = reinterpret_cast<const float *>(lut_lds);
        const int K = a.rank_iters;
        auto rank_pixels = [&](auto wide_tag) {
            constexpr bool WIDE = decltype(wide_tag)::value;
            constexpr int SLOTS = WIDE ? WB_BIN16_SLOTS : WB_BIN_SLOTS, CELLS = WIDE ? WB_BIN16_CELLS : WB_BIN_CELLS;
            const uint8_t *base8 = reinterpret_cast<const uint8_t *>(lut_lds) + 4 * SLOTS * 4;
            const uint16_t *base16 = reinterpret_cast<const uint16_t *>(base8);
            // all RPT x 4 values of the thread advance together: every step is RPT * 4 independent LDS lookups (one
            // value at a time, the 1 + K dependent lookups of each value were a chain of LDS latencies)
            uint32_t r[RPT][4];
            if (live) {
#pragma unroll
                for (int y = 0; y < RPT; ++y)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t cell = wb_bin_cell(o[y][k], a.rank_k[k], a.rank_b[k], (float)(CELLS - 1));
                        r[y][k] = WIDE ? (uint32_t)base16[k * CELLS + cell] : (uint32_t)base8[k * CELLS + cell];
                    }
                if (K <= 2) {
                    // the usual case -- at most two thresholds share a cell: both candidates S[r], S[r + 1] come with ONE
                    // LDS read (they are neighbours) and are compared independently: S is sorted, so the second test only
                    // passes where the first does; a threshold of a higher cell, or the +inf padding, never passes
                    // (r + 1 <= the table's last padding slot stays inside the channel's table)
#pragma unroll
                    for (int y = 0; y < RPT; ++y)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float *sp = Sthr + k * SLOTS + r[y][k];
                            const float s0 = sp[0], s1 = sp[1];
                            r[y][k] += (o[y][k] > s0 ? 1u : 0u) + (o[y][k] > s1 ? 1u : 0u);
                        }
                } else if (WIDE && K <= 4) {
                    // the coarser 16-bit grid: up to four thresholds per cell, all four candidates S[r .. r + 3] fetched at
                    // once (r + 3 <= 1023: the table's last four slots are +inf padding)
#pragma unroll
                    for (int y = 0; y < RPT; ++y)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float *sp = Sthr + k * SLOTS + r[y][k];
                            const float s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
                            r[y][k] += (o[y][k] > s0 ? 1u : 0u) + (o[y][k] > s1 ? 1u : 0u) + (o[y][k] > s2 ? 1u : 0u) + (o[y][k] > s3 ? 1u : 0u);
                        }
                } else {
                    for (int i = 0; i < K; ++i) {
#pragma unroll
                        for (int y = 0; y < RPT; ++y)
#pragma unroll
                            for (int k = 0; k < 4; ++k) r[y][k] += o[y][k] > Sthr[k * SLOTS + r[y][k]] ? 1u : 0u;
                    }
                }
            }
#pragma unroll
            for (int y = 0; y < RPT; ++y) {
                const int su = u0 + i0 + y;
                if (!live || su >= L.u || sv >= L.v || (a.dbg & 4)) continue;
                uint32_t rk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    rk[k] = r[y][k];
                    if constexpr (sizeof(T) != 1) rk[k] = o[y][k] != o[y][k] ? (WIDE ? 65535u : 255u) : rk[k];   // a NaN pixel fails every `v <= thr`
                }
                if constexpr (WIDE) {
                    uint2 *rout = reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(a.rank) + ((int64_t)b * a.rank_stride + L.chn_off));
                    rout[(int64_t)su * L.v + sv] = make_uint2(rk[0] | (rk[1] << 16), rk[2] | (rk[3] << 16));
                } else {
                    uint32_t *rout = reinterpret_cast<uint32_t *>(a.rank + (int64_t)b * a.rank_stride + L.chn_off);
                    rout[(int64_t)su * L.v + sv] = rk[0] | (rk[1] << 8) | (rk[2] << 16) | (rk[3] << 24);
                }
            }
        };
        if (a.rank_wide)
            rank_pixels(std::true_type{});
        else
            rank_pixels(std::false_type{});
    }
    WB_CSTAMP(7);
}

// -------------------------------------------------------------------------------------------
// Integer channel functions of the reference's FPGA flavour (fpga/channels.py:5-67) on uint8 images:
//   dx, dy   3x3 Sobel stencils in exact integer arithmetic; numba leaves the 1-pixel border of
//            the (resized) image at 0 -- no reflected halo here
//   NCH = 4  grad_hist_4_u1: y = (dx, trunc((dx-dy)/2), dy, trunc((dx+dy)/2)); min(|y| // 4, 255)
//   NCH = 1  grad_mag_u1:    min(max(|dx|, |dy|) // 4, 255)
// then channel_pyramid's generic tail on uint8 arrays: avg_pool_2 wraps its three uint8 adds
// mod 256 before the /4 (channels.py:61-64), the smooth stencil sums in int64 and the /16 is
// truncated by the store into the uint8 array (channels.py:78-90), border 0.
// Output [u][v][NCH] uint8: one dword (NCH = 4) or one byte per pixel.
template <int S, int TU, int TV, bool SMOOTH, int NCH>
__global__ __launch_bounds__(256) void channels_u1_kernel(ChanArgs a) {
    using T = uint8_t;
    using G = TileGeom<S, TU, TV, SMOOTH>;
    constexpr int HS = G::HS, SU = G::SU, SV = G::SV, RH = G::RH, RW = G::RW, P = G::P;
    constexpr int UNI_BYTES = G::SH_BYTES > G::PATCH_BYTES ? G::SH_BYTES : G::PATCH_BYTES;
    __shared__ float R[RH * RW];
    __shared__ __attribute__((aligned(16))) unsigned char uni[UNI_BYTES];
    __shared__ float4 rowtab[RH + RW % 64];
    uint32_t *Sh = reinterpret_cast<uint32_t *>(uni);     // packed channels of one shrunk pixel

    const WbTile tile = a.tiles[blockIdx.x];
    const WbLevel L = a.levels[tile.level];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int u0 = tile.ty * TU, v0 = tile.tx * TV;
    const T *src = (L.oct == 0) ? (const T *)a.img + (int64_t)b * a.img_stride
                                : (const T *)a.oct + (int64_t)b * a.oct_stride + L.src_off;
    double mn, mx;
    clip_range<T>(a, b, L.oct, mn, mx);
    const int ry0 = S * (u0 - HS) - 1, rx0 = S * (v0 - HS) - 1;
    resample_tile<T, G>(a, L, src, mn, mx, ry0, rx0, RH, R, uni, rowtab, tid);
    __syncthreads();
    if (a.dbg & 1) return;

    // ---- step 2: integer gradients -> channels -> shrink, one shrunk pixel per iteration
    for (int p = tid; p < SU * SV; p += 256) {
        const int i = p / SV, j = p - i * SV;
        // The stencils in fp32 (exact: integers below 2^11), shared [1,2,1] passes as in channels_kernel;
        // only the channel values are converted to integers.  trunc((dx -/+ dy) / 2) has the magnitude
        // floor(|dx -/+ dy| / 2), so  |y| // 4  is  |dx| >> 2, |dx - dy| >> 3, |dy| >> 2, |dx + dy| >> 3;
        // with 8 bit pixels |dx|, |dy| <= 1020, so none of them exceeds 255 and the clamp never acts.
        float pt[P][P];
#pragma unroll
        for (int y = 0; y < P; ++y)
#pragma unroll
            for (int x = 0; x < P; ++x) pt[y][x] = R[(S * i + y) * RW + (S * j + x)];
        float hc[S][P], hr[P][S];
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
            for (int x = 0; x < P; ++x) hc[y][x] = scalar_only(Src<T>::hpass(pt[y][x], pt[y + 1][x], pt[y + 2][x]));
#pragma unroll
        for (int y = 0; y < P; ++y)
#pragma unroll
            for (int x = 0; x < S; ++x) hr[y][x] = scalar_only(Src<T>::hpass(pt[y][x], pt[y][x + 1], pt[y][x + 2]));
        // numba leaves the 1-pixel border of the resized image at 0: only blocks on that border test their pixels
        const int by0 = ry0 + S * i + 1, bx0 = rx0 + S * j + 1;                    // first pixel of the block
        const bool on_border = by0 <= 0 || bx0 <= 0 || by0 + S - 1 >= L.nh - 1 || bx0 + S - 1 >= L.nw - 1;
        int ch[S][S][NCH];
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
            for (int x = 0; x < S; ++x) {
                float dx = scalar_only(hc[y][x + 2] - hc[y][x]);
                float dy = scalar_only(hr[y + 2][x] - hr[y][x]);
                if (on_border) {
                    const int gy = by0 + y, gx = bx0 + x;
                    if (gy <= 0 || gx <= 0 || gy >= L.nh - 1 || gx >= L.nw - 1) dx = dy = 0.0f;
                }
                if constexpr (NCH == 4) {
                    ch[y][x][0] = (int)(uint32_t)fabsf(dx) >> 2;
                    ch[y][x][1] = (int)(uint32_t)fabsf(dx - dy) >> 3;
                    ch[y][x][2] = (int)(uint32_t)fabsf(dy) >> 2;
                    ch[y][x][3] = (int)(uint32_t)fabsf(dx + dy) >> 3;
                } else {
                    ch[y][x][0] = (int)(uint32_t)fmaxf(fabsf(dx), fabsf(dy)) >> 2;
                }
            }
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            int v;
            if constexpr (S == 1) {
                v = ch[0][0][k];
            } else if constexpr (S == 2) {
                v = ((ch[0][0][k] + ch[1][0][k] + ch[0][1][k] + ch[1][1][k]) & 255) >> 2;
            } else {  // S == 4 (extension): avg_pool_2 applied twice
                int q[2][2];
#pragma unroll
                for (int A = 0; A < 2; ++A)
#pragma unroll
                    for (int B = 0; B < 2; ++B)
                        q[A][B] = ((ch[2 * A][2 * B][k] + ch[2 * A + 1][2 * B][k] + ch[2 * A][2 * B + 1][k] +
                                    ch[2 * A + 1][2 * B + 1][k]) & 255) >> 2;
                v = ((q[0][0] + q[1][0] + q[0][1] + q[1][1]) & 255) >> 2;
            }
            o |= (uint32_t)v << (8 * k);
        }
        Sh[p] = o;
    }
    __syncthreads();
    if (a.dbg & 2) return;

    // ---- step 3: 3x3 binomial smooth, integer sum >> 4, border = 0; strips as in channels_kernel
    constexpr int RPT = TU * TV / 256;
    static_assert(TU * TV % 256 == 0 && 256 % TV == 0, "tile must split into whole thread strips");
    uint8_t *out = reinterpret_cast<uint8_t *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    const int j = tid % TV, i0 = (tid / TV) * RPT;
    const int sv = v0 + j;
    uint32_t o[RPT];
    if constexpr (SMOOTH) {
        uint32_t w[RPT + 2][3];
#pragma unroll
        for (int y = 0; y < RPT + 2; ++y)
#pragma unroll
            for (int x = 0; x < 3; ++x) w[y][x] = Sh[(i0 + y) * SV + (j + x)];
#pragma unroll
        for (int y = 0; y < RPT; ++y) {
            o[y] = 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                auto at = [&](int yy, int xx) { return (int)((w[y + yy][xx] >> (8 * k)) & 255u); };
                const int sum = at(0, 0) + 2 * at(0, 1) + at(0, 2) + 2 * at(1, 0) + 4 * at(1, 1) + 2 * at(1, 2) +
                                at(2, 0) + 2 * at(2, 1) + at(2, 2);
                o[y] |= (uint32_t)(sum >> 4) << (8 * k);
            }
        }
    } else {
#pragma unroll
        for (int y = 0; y < RPT; ++y) o[y] = Sh[(i0 + y) * SV + j];
    }
#pragma unroll
    for (int y = 0; y < RPT; ++y) {
        const int su = u0 + i0 + y;
        if (su >= L.u || sv >= L.v || (a.dbg & 4)) continue;
        if (SMOOTH && (su == 0 || sv == 0 || su == L.u - 1 || sv == L.v - 1)) o[y] = 0;
        const int64_t at = (int64_t)su * L.v + sv;
        if constexpr (NCH == 4)
            reinterpret_cast<uint32_t *>(out)[at] = o[y];        // 64 lanes store 256 B contiguous
        else
            out[at] = (uint8_t)o[y];
    }
}

// Exhaustive device check of project_int against project_f64 over [-1020, 1020]^2.
__global__ void selftest_projection_kernel(ChanArgs a, uint32_t *mismatches) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 2041;
    if (i >= n * n) return;
    float gx = (float)(i / n - 1020), gy = (float)(i % n - 1020);
    float f[4], r[4];
    project_int(gx, gy, a, f);
    project_f64(gx, gy, a, r);
    for (int k = 0; k < 4; ++k)
        if (__float_as_uint(f[k]) != __float_as_uint(r[k])) atomicAdd(mismatches, 1u);
}

template <typename T, int S, int TU, int TV, bool FAST, int NT = 256>
void launch_variant(hipStream_t st, dim3 grid, const ChanArgs &a, bool smooth) {
    // diagnostic (WB_CHAN_XLDS=bytes): extra dynamic LDS per workgroup lowers the workgroups per CU, to tell a
    // latency-bound kernel (time ~ 1 / residency) from a throughput-bound one (time unchanged)
    static const size_t xlds = getenv("WB_CHAN_XLDS") ? (size_t)atoi(getenv("WB_CHAN_XLDS")) : 0;
    if (smooth)
        hipLaunchKernelGGL((channels_kernel<T, S, TU, TV, true, FAST, NT>), grid, dim3(NT), xlds, st, a);
    else
        hipLaunchKernelGGL((channels_kernel<T, S, TU, TV, false, FAST, NT>), grid, dim3(NT), xlds, st, a);
}

template <typename T, bool FAST>
int launch_dtype(hipStream_t st, dim3 grid, const ChanArgs &a, int shrink, bool smooth, bool tile32) {
    switch (shrink) {
        case 1: launch_variant<T, 1, 16, 64, FAST>(st, grid, a, smooth); break;
        case 2:
            if (tile32)
                launch_variant<T, 2, 32, 64, FAST, 512>(st, grid, a, smooth);
            else
                launch_variant<T, 2, 16, 64, FAST>(st, grid, a, smooth);
            break;
        case 4: launch_variant<T, 4, 8, WB_CHAN_S4_TV, FAST>(st, grid, a, smooth); break;
        default:
            wb_set_error("wb_channels_launch: shrink=%d unsupported (1, 2; 4 as an extension)", shrink);
            return WB_ERR_UNSUPPORTED;
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

// -------------------------------------------------------------------------------------------
// waldboost.channels.grad_mag (reference channels.py:11-37, defaults norm=5, eps=1e-3): one float32
// channel  mag / (triangle11(mag) + eps),  mag = sqrt(gx^2 + gy^2) in fp32.  The normaliser is
// scipy's convolve1d twice (rows, then columns) with the 11-tap triangle: symmetric-kernel branch
// of NI_Correlate1D, fp64 accumulation  t = x[l]*w[c];  t += (x[l+j] + x[l-j]) * w[c+j], j = -5..-1,
// one fp32 rounding per pass, 'reflect' borders.  The tile therefore carries a 5-pixel halo of mag
// (6 of the resized image); halo positions outside the level hold the REFLECTED coordinate's
// pixel, under which the gradient magnitude of the mirror position comes out exactly (the [1,2,1]
// pass is symmetric, the difference pass only changes sign).  Secondary channel function: plain
// per-pixel code, not tuned like channels_kernel.
struct GmGeom {
    static constexpr int NH = 5;      // half width of the 11-tap triangle
};

// (the geometry resample_tile wants: the resized tile and the staged source patch -- uint8 images, any down-scale
// below 2 -- which shares its memory with the magnitudes and the shrunk tile, both written after the resize)
template <int S, int TU, int TV, bool SMOOTH> struct GmTile {
    static constexpr int HS = SMOOTH ? 1 : 0, NH = GmGeom::NH;
    static constexpr int SU = TU + 2 * HS, SV = TV + 2 * HS;      // shrunk tile incl. smooth halo
    static constexpr int VH = S * SU, VW = S * SV;                // normalised magnitudes needed
    static constexpr int MH = VH + 2 * NH, MW = VW + 2 * NH;      // magnitudes incl. the triangle halo
    static constexpr int RH = MH + 2, RW = MW + 2;                // resized pixels incl. the gradient halo
    // (512 threads: at two workgroups per CU -- what the 60 KB of LDS admit -- 16 waves per CU, like the gradient kernels)
    static constexpr int NT = 512, NW = NT / 64;
    static constexpr int PROWS = 2 * RH + 4, PPITCH = (2 * RW + 12 + 3) & ~3;
};

template <typename T, int S, int TU, int TV, bool SMOOTH>
__global__ __launch_bounds__(512, 2) void channels_gm_kernel(ChanArgs a) {
    using G = GmTile<S, TU, TV, SMOOTH>;
    constexpr int HS = G::HS, NH = G::NH, SU = G::SU, SV = G::SV, VH = G::VH, VW = G::VW, MH = G::MH, MW = G::MW;
    constexpr int RH = G::RH, RW = G::RW, NT = G::NT;
    constexpr int MG_SH_BYTES = (MH * MW + SU * SV) * 4, PATCH_BYTES = sizeof(T) == 1 ? G::PROWS * G::PPITCH : 0;
    __shared__ __attribute__((aligned(16))) float R[RH * RW];   // resized tile; later the row-pass result [VH][MW]
    __shared__ __attribute__((aligned(16))) unsigned char uni[MG_SH_BYTES > PATCH_BYTES ? MG_SH_BYTES : PATCH_BYTES];
    __shared__ float4 rowtab[sizeof(T) == 1 ? RH + RW % 64 : 1];
    float *Mg = reinterpret_cast<float *>(uni);            // magnitudes; the centre is normalised in place
    float *Sh = Mg + MH * MW;
    static_assert(VH * MW <= RH * RW, "row-pass result reuses the resized tile");

    const WbTile tile = a.tiles[blockIdx.x];
    const WbLevel L = a.levels[tile.level];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int u0 = tile.ty * TU, v0 = tile.tx * TV;
    const T *src = (L.oct == 0) ? (const T *)a.img + (int64_t)b * a.img_stride
                                : (const T *)a.oct + (int64_t)b * a.oct_stride + L.src_off;
    double mn, mx;
    clip_range<T>(a, b, L.oct, mn, mx);
    const int ry0 = S * (u0 - HS) - NH - 1, rx0 = S * (v0 - HS) - NH - 1;

    // ---- resized pixels (reference channels.py:132), reflected outside the level: the channel kernels' own resample
    //      (uint8: source patch staged in LDS, shared interpolation between rows, exact redo where the fast path's
    //      band test asks for it) with mirrored instead of clamped coordinates
    resample_tile<T, G, true>(a, L, src, mn, mx, ry0, rx0, RH, R, uni, rowtab, tid);
    __syncthreads();
    if (a.dbg & 1) return;           // (WB_CHAN_DBG: phase timing -- 1 resize, 2 magnitudes, 8 / 16 the two triangle passes)

    // ---- gradient magnitude (channels.py:16-21, 31-32): fp32 squares, sum and square root
    for (int p = tid; p < MH * MW; p += NT) {
        const int k = p / MW, q = p - k * MW;
        const float *c = R + k * RW + q;                      // 3x3 patch, centre at (k+1, q+1)
        const float hc0 = Src<T>::hpass(c[0], c[RW], c[2 * RW]);              // vertical [1,2,1] at column q
        const float hc2 = Src<T>::hpass(c[2], c[RW + 2], c[2 * RW + 2]);      //                     column q+2
        const float hr0 = Src<T>::hpass(c[0], c[1], c[2]);                    // horizontal [1,2,1] at row k
        const float hr2 = Src<T>::hpass(c[2 * RW], c[2 * RW + 1], c[2 * RW + 2]);
        const float hc1 = Src<T>::hpass(c[1], c[RW + 1], c[2 * RW + 1]);      // the centre taps (weight 0: see dpass)
        const float hr1 = Src<T>::hpass(c[RW], c[RW + 1], c[RW + 2]);
        const float gx = Src<T>::dpass(hc0, hc1, hc2), gy = Src<T>::dpass(hr0, hr1, hr2);
        Mg[p] = sqrtf(gx * gx + gy * gy);
    }
    __syncthreads();
    if (a.dbg & 2) return;

    // ---- triangle filter along the rows' axis (convolve1d axis 0), result over the resized tile's memory.
    //      Each thread forms TG outputs that are neighbours ALONG the filter: the 10 + TG magnitudes they span are read
    //      and widened to fp64 once (one output at a time, every magnitude was read and converted eleven times);
    //      per output the sum is formed exactly as before, term by term in scipy's order.
    constexpr int TG = 4;
    float *Tv = R;
    {
        constexpr int GROUPS = (VH + TG - 1) / TG;
        for (int p = tid; p < GROUPS * MW; p += NT) {
            const int g = p / MW, q = p - g * MW, k0 = g * TG;
            double x[TG + 2 * NH];
#pragma unroll
            for (int i = 0; i < TG + 2 * NH; ++i) {
                const int row = k0 + i < MH ? k0 + i : MH - 1;     // (rows past the tile: read, never used)
                x[i] = (double)Mg[row * MW + q];
            }
#pragma unroll
            for (int o = 0; o < TG; ++o) {
                if (k0 + o >= VH) break;
                double t = x[o + NH] * a.tri[NH];
#pragma unroll
                for (int j = -NH; j < 0; ++j) t = t + (x[o + NH + j] + x[o + NH - j]) * a.tri[NH + j];
                Tv[(k0 + o) * MW + q] = (float)t;
            }
        }
    }
    __syncthreads();
    if (a.dbg & 8) return;
    // ---- ... along the columns' axis, then mag / (norm + eps), in place at the centre of Mg
    {
        constexpr int GROUPS = (VW + TG - 1) / TG;
        for (int p = tid; p < VH * GROUPS; p += NT) {
            // (neighbouring lanes take neighbouring ROWS: their reads are MW floats apart -- 2-way bank conflicts; TG
            // floats apart, along the row, they were 4-way)
            const int qg = p / VH, k = p - qg * VH, q0 = qg * TG;
            double x[TG + 2 * NH];
#pragma unroll
            for (int i = 0; i < TG + 2 * NH; ++i) {
                const int col = q0 + i < MW ? q0 + i : MW - 1;
                x[i] = (double)Tv[k * MW + col];
            }
#pragma unroll
            for (int o = 0; o < TG; ++o) {
                if (q0 + o >= VW) break;
                double t = x[o + NH] * a.tri[NH];
#pragma unroll
                for (int j = -NH; j < 0; ++j) t = t + (x[o + NH + j] + x[o + NH - j]) * a.tri[NH + j];
                float *m = Mg + (k + NH) * MW + q0 + o + NH;
                *m = *m / ((float)t + a.gm_eps);
            }
        }
    }
    __syncthreads();
    if (a.dbg & 16) return;

    // ---- shrink (channels.py:55-64, fp32 ((a+b)+c)+d then /4)
    for (int p = tid; p < SU * SV; p += NT) {
        const int i = p / SV, j = p - i * SV;
        auto at = [&](int y, int x) { return Mg[(S * i + y + NH) * MW + S * j + x + NH]; };
        float o;
        if constexpr (S == 1) {
            o = at(0, 0);
        } else if constexpr (S == 2) {
            o = (((at(0, 0) + at(1, 0)) + at(0, 1)) + at(1, 1)) * 0.25f;
        } else {
            float qd[2][2];
#pragma unroll
            for (int A = 0; A < 2; ++A)
#pragma unroll
                for (int B = 0; B < 2; ++B)
                    qd[A][B] = (((at(2 * A, 2 * B) + at(2 * A + 1, 2 * B)) + at(2 * A, 2 * B + 1)) + at(2 * A + 1, 2 * B + 1)) * 0.25f;
            o = (((qd[0][0] + qd[1][0]) + qd[0][1]) + qd[1][1]) * 0.25f;
        }
        Sh[p] = o;
    }
    __syncthreads();

    // ---- 3x3 smooth (fp64, source order), border 0, store [u][v][1]
    float *out = reinterpret_cast<float *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    for (int p = tid; p < TU * TV; p += NT) {
        const int i = p / TV, j = p - i * TV;
        const int su = u0 + i, sv = v0 + j;
        if (su >= L.u || sv >= L.v) continue;
        float o;
        if constexpr (SMOOTH) {
            const float *c = Sh + i * SV + j;
            o = smooth9(c[0], c[1], c[2], c[SV], c[SV + 1], c[SV + 2], c[2 * SV], c[2 * SV + 1], c[2 * SV + 2]);
            if (su == 0 || sv == 0 || su == L.u - 1 || sv == L.v - 1) o = 0.0f;
        } else {
            o = Sh[i * SV + j];
        }
        out[(int64_t)su * L.v + sv] = o;
    }
}

template <typename T>
int launch_gm(hipStream_t st, dim3 grid, const ChanArgs &a, int shrink, bool smooth) {
#define WB_GM(S, TU, TV)                                                                         \
    if (smooth)                                                                                  \
        hipLaunchKernelGGL((channels_gm_kernel<T, S, TU, TV, true>), grid, dim3(512), 0, st, a); \
    else                                                                                         \
        hipLaunchKernelGGL((channels_gm_kernel<T, S, TU, TV, false>), grid, dim3(512), 0, st, a);
    switch (shrink) {                      // same output tiles as the other channel kernels (wb_channels_tile)
        case 1: WB_GM(1, 16, 64) break;
        case 2: WB_GM(2, 16, 64) break;
        case 4: WB_GM(4, 8, 32) break;
        default:
            wb_set_error("wb_channels_launch: shrink=%d unsupported (1, 2; 4 as an extension)", shrink);
            return WB_ERR_UNSUPPORTED;
    }
#undef WB_GM
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

template <int NCH>
int launch_u1(hipStream_t st, dim3 grid, const ChanArgs &a, int shrink, bool smooth) {
#define WB_U1(S, TU, TV)                                                                        \
    if (smooth)                                                                                 \
        hipLaunchKernelGGL((channels_u1_kernel<S, TU, TV, true, NCH>), grid, dim3(256), 0, st, a);  \
    else                                                                                        \
        hipLaunchKernelGGL((channels_u1_kernel<S, TU, TV, false, NCH>), grid, dim3(256), 0, st, a);
    switch (shrink) {
        case 1: WB_U1(1, 16, 64) break;
        case 2: WB_U1(2, 16, 64) break;
        case 4: WB_U1(4, 8, 32) break;
        default:
            wb_set_error("wb_channels_launch: shrink=%d unsupported (1, 2; 4 as an extension)", shrink);
            return WB_ERR_UNSUPPORTED;
    }
#undef WB_U1
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

// experiment (WB_CHAN_TILE32=1): grad_hist at shrink 2 on a 32 x 64 tile of 512 threads -- 14 % of the resized pixels a
// workgroup computes are halo instead of 24 %, 5 % fewer vector instructions per image, and yet 39.6 against 33.0 us
// per image at batch 64 (47.3 -> 51.5 at batch 1): two workgroups of eight waves per CU overlap their load and
// compute phases worse than four of four.  Kept for A/B runs; the default is the 16 x 64 tile.
bool tile32() {
    static const bool t32 = getenv("WB_CHAN_TILE32") != nullptr;
    return t32;
}

// the canonical constants np.cos/np.sin(np.linspace(0, pi, 5)[:-1]) the integer fast path is proven for
const double kCanonCs[4] = {1.0, 0x1.6a09e667f3bcdp-1, 0x1.1a62633145c07p-54, -0x1.6a09e667f3bccp-1};
const double kCanonSn[4] = {0.0, 0x1.6a09e667f3bccp-1, 1.0, 0x1.6a09e667f3bcdp-1};

bool canonical_constants(const double *cs_sn) {
    for (int k = 0; k < 4; ++k)
        if (cs_sn[k] != kCanonCs[k] || cs_sn[4 + k] != kCanonSn[k]) return false;
    return true;
}

void set_constants(ChanArgs &a, const double *cs_sn) {
    for (int k = 0; k < 4; ++k) {
        a.cs[k] = cs_sn[k];
        a.sn[k] = cs_sn[4 + k];
    }
    a.chi = (float)cs_sn[5];
    a.clo = (float)(cs_sn[5] - (double)a.chi);
    a.c2hi = (float)cs_sn[2];
    a.c2lo = (float)(cs_sn[2] - (double)a.c2hi);
}

}  // namespace

// -------------------------------------------------------------------------------------------
// The pyramid around a channel function this build has no kernel for (reference channels.py:119,136 calls whatever
// callable channel_opts["channels"] holds): the steps on either side of the caller's function as plain kernels --
//   resize_level_kernel   one level's resized image (channels.py:132), cast back to the image dtype
//   pool2_kernel          avg_pool_2 of an [H][W][C] array (channels.py:55-64): uint8 adds wrap, float32 ((a+b)+c)+d
//   smooth_kernel         smooth_image_3d (channels.py:78-90): nine-term sum in source order (float32 channels: fp64;
//                         uint8 channels: integers), / 16, cast back; 1-pixel border 0
// One thread per output element, through global memory: correct rather than tuned (the callable between them runs on
// the host anyway).
namespace {

template <typename T, typename O>
__global__ __launch_bounds__(256) void resize_level_kernel(const T *src, int src_w, int nh, int nw, const WbTap *rtap, const WbTap *ctap,
                                                           double mn, double mx, int cast_mode, O *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)nh * nw) return;
    const int y = (int)(i / nw), x = (int)(i - (int64_t)y * nw);
    const WbTap tr = rtap[y], tc = ctap[x];
    const T *r0 = src + (int64_t)tr.i0 * src_w, *r1 = src + (int64_t)tr.i1 * src_w;
    const double t = resample_f64((double)r0[tc.i0], (double)r0[tc.i1], (double)r1[tc.i0], (double)r1[tc.i1], tr, tc);
    if constexpr (sizeof(T) == 8) {
        // float64-held dtypes: the value after the clip and the cast back, still as a double (Src<double>::finish
        // rounds to float32 for the channel kernels; here the caller gets the image dtype's own value)
        double v = t;
        if (mn != mn || mx != mx) v = __builtin_nan("");
        else v = v < mn ? mn : (v > mx ? mx : v);
        switch (cast_mode) {
            case WB_CAST_TRUNC: v = trunc(v); break;
            case WB_CAST_BOOL: v = v != 0.0 ? 1.0 : 0.0; break;
            case WB_CAST_F16: v = wb_round_f16(v); break;
        }
        out[i] = (O)v;
    } else {
        out[i] = (O)Src<T>::finish(t, mn, mx, cast_mode);
    }
}

template <typename E>
__global__ __launch_bounds__(256) void pool2_kernel(const E *in, int H, int W, int C, E *out) {
    const int oh = H >> 1, ow = W >> 1;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)oh * ow * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int y = (int)(p / ow), x = (int)(p - (int64_t)y * ow);
    auto at = [&](int dy, int dx) { return in[((int64_t)(2 * y + dy) * W + (2 * x + dx)) * C + c]; };
    if constexpr (sizeof(E) == 1)
        out[i] = (E)((((uint32_t)at(0, 0) + at(1, 0) + at(0, 1) + at(1, 1)) & 255u) >> 2);
    else
        out[i] = (((at(0, 0) + at(1, 0)) + at(0, 1)) + at(1, 1)) * 0.25f;
}

template <typename E>
__global__ __launch_bounds__(256) void smooth_kernel(const E *in, int H, int W, int C, E *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
    if (y == 0 || x == 0 || y == H - 1 || x == W - 1) {
        out[i] = (E)0;
        return;
    }
    auto at = [&](int dy, int dx) { return in[((int64_t)(y + dy) * W + (x + dx)) * C + c]; };
    if constexpr (sizeof(E) == 1) {
        const int s = at(-1, -1) + 2 * at(-1, 0) + at(-1, 1) + 2 * at(0, -1) + 4 * at(0, 0) + 2 * at(0, 1) + at(1, -1) + 2 * at(1, 0) + at(1, 1);
        out[i] = (E)(s >> 4);
    } else {
        out[i] = smooth9(at(-1, -1), at(-1, 0), at(-1, 1), at(0, -1), at(0, 0), at(0, 1), at(1, -1), at(1, 0), at(1, 1));
    }
}

}  // namespace

extern "C" int wb_resize_level_launch(void *stream, const void *img, const void *oct, int dtype, const WbLevel *level_host,
                                      const uint32_t *minmax_host, const WbTap *taps, void *out) {
    WB_REQUIRE(img && level_host && minmax_host && taps && out, "wb_resize_level_launch: null pointer");
    const WbLevel &L = *level_host;
    const dim3 grid((unsigned)(((int64_t)L.nh * L.nw + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    const WbTap *rtap = taps + L.tap_off, *ctap = rtap + L.nh;
    double mn, mx;
    if (dtype == WB_DTYPE_U8) {
        mn = (double)(~minmax_host[0]);
        mx = (double)minmax_host[1];
        const uint8_t *src = L.oct == 0 ? (const uint8_t *)img : (const uint8_t *)oct + L.src_off;
        hipLaunchKernelGGL((resize_level_kernel<uint8_t, uint8_t>), grid, dim3(256), 0, st, src, L.src_w, L.nh, L.nw, rtap, ctap, mn, mx, 0, (uint8_t *)out);
    } else if (dtype == WB_DTYPE_F32) {
        mn = (double)wb_key_f32(~minmax_host[0]);
        mx = (double)wb_key_f32(minmax_host[1]);
        const float *src = L.oct == 0 ? (const float *)img : (const float *)oct + L.src_off;
        hipLaunchKernelGGL((resize_level_kernel<float, float>), grid, dim3(256), 0, st, src, L.src_w, L.nh, L.nw, rtap, ctap, mn, mx, 0, (float *)out);
    } else if (wb_dtype_held_f64(dtype)) {
        const unsigned long long *mm = reinterpret_cast<const unsigned long long *>(minmax_host);
        auto dec = [](unsigned long long k) {
            const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
            double d;
            __builtin_memcpy(&d, &b, 8);
            return d;
        };
        mn = dec(~mm[0]);
        mx = dec(mm[1]);
        const double *src = L.oct == 0 ? (const double *)img : (const double *)oct + L.src_off;
        hipLaunchKernelGGL((resize_level_kernel<double, double>), grid, dim3(256), 0, st, src, L.src_w, L.nh, L.nw, rtap, ctap, mn, mx,
                           wb_cast_mode(dtype), (double *)out);
    } else {
        wb_set_error("wb_resize_level_launch: unsupported image dtype code %d", dtype);
        return WB_ERR_UNSUPPORTED;
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_pool_smooth_launch(void *stream, const void *in, int chn_dtype, int H, int W, int C, int shrink, int smooth,
                                     void *tmp, void *out) {
    WB_REQUIRE(in && out && H >= 1 && W >= 1 && C >= 1, "wb_pool_smooth_launch: bad argument");
    WB_REQUIRE(chn_dtype == WB_DTYPE_U8 || chn_dtype == WB_DTYPE_F32, "wb_pool_smooth_launch: channel dtype %d (uint8 or float32)", chn_dtype);
    WB_REQUIRE(shrink == 1 || shrink == 2, "wb_pool_smooth_launch: shrink %d (1 or 2)", shrink);
    WB_REQUIRE(!(shrink == 2 && smooth) || tmp, "wb_pool_smooth_launch: pooling and smoothing need the tmp buffer");
    hipStream_t st = (hipStream_t)stream;
    const void *cur = in;
    int h = H, w = W;
    if (shrink == 2) {
        void *dst = smooth ? tmp : out;
        const int64_t n = (int64_t)(H >> 1) * (W >> 1) * C;
        if (n > 0) {
            const dim3 grid((unsigned)((n + 255) / 256));
            if (chn_dtype == WB_DTYPE_U8)
                hipLaunchKernelGGL((pool2_kernel<uint8_t>), grid, dim3(256), 0, st, (const uint8_t *)cur, H, W, C, (uint8_t *)dst);
            else
                hipLaunchKernelGGL((pool2_kernel<float>), grid, dim3(256), 0, st, (const float *)cur, H, W, C, (float *)dst);
        }
        cur = dst;
        h = H >> 1;
        w = W >> 1;
    }
    if (smooth) {
        const int64_t n = (int64_t)h * w * C;
        if (n > 0) {
            const dim3 grid((unsigned)((n + 255) / 256));
            if (chn_dtype == WB_DTYPE_U8)
                hipLaunchKernelGGL((smooth_kernel<uint8_t>), grid, dim3(256), 0, st, (const uint8_t *)cur, h, w, C, (uint8_t *)out);
            else
                hipLaunchKernelGGL((smooth_kernel<float>), grid, dim3(256), 0, st, (const float *)cur, h, w, C, (float *)out);
        }
    } else if (shrink != 2) {
        WB_HIP_CHECK(hipMemcpyAsync(out, in, (size_t)H * W * C * (chn_dtype == WB_DTYPE_U8 ? 1 : 4), hipMemcpyDeviceToDevice, st));
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_channels_tile(int channel_func, int shrink, int *tile_u, int *tile_v) {
    WB_REQUIRE(tile_u && tile_v, "wb_channels_tile: null pointer");
    if (shrink == 1 || shrink == 2) {
        // (see tile32(): the 32 x 64 tile of 512 threads is an opt-in experiment)
        const bool big = shrink == 2 && channel_func == WB_CHN_GRAD_HIST && tile32();
        *tile_u = big ? 32 : 16;
        *tile_v = 64;
    } else if (shrink == 4) {
        *tile_u = 8;
        *tile_v = channel_func == WB_CHN_GRAD_HIST ? WB_CHAN_S4_TV : 32;
    } else {
        wb_set_error("wb_channels_tile: shrink=%d unsupported", shrink);
        return WB_ERR_UNSUPPORTED;
    }
    return WB_OK;
}

extern "C" int wb_channel_func_info(int channel_func, int *n_channels, int *chn_dtype) {
    WB_REQUIRE(n_channels && chn_dtype, "wb_channel_func_info: null pointer");
    switch (channel_func) {
        case WB_CHN_GRAD_HIST: *n_channels = 4; *chn_dtype = WB_DTYPE_F32; return WB_OK;
        case WB_CHN_GRAD_HIST_4_U1: *n_channels = 4; *chn_dtype = WB_DTYPE_U8; return WB_OK;
        case WB_CHN_GRAD_MAG_U1: *n_channels = 1; *chn_dtype = WB_DTYPE_U8; return WB_OK;
        case WB_CHN_GRAD_MAG: *n_channels = 1; *chn_dtype = WB_DTYPE_F32; return WB_OK;
    }
    wb_set_error("wb_channel_func_info: unknown channel function %d", channel_func);
    return WB_ERR_INVALID;
}

// Per tile, the source patch it stages (WbTilePatch): the kernels' own extent function, on the host, with the geometry of
// the kernel that channel_func / shrink / smooth select.
namespace {
template <typename G, bool FULL_ROWS>
void fill_patches(const WbLevel *levels, const WbTile *tiles, int n_tiles, WbTilePatch *out) {
    for (int i = 0; i < n_tiles; ++i) {
        const WbTile &t = tiles[i];
        const WbLevel &L = levels[t.level];
        const int u0 = t.ty * G::TU, v0 = t.tx * G::TV;
        const int ry0 = G::S * (u0 - G::HS) - 1, rx0 = G::S * (v0 - G::HS) - 1;
        // (channels_kernel computes only the rows a tile on the bottom edge of its level needs; the uint8 kernels all RH)
        const int vrows = L.u - u0 < G::TU ? L.u - u0 : G::TU;
        const int rh = FULL_ROWS ? G::RH : G::S * (vrows + 2 * G::HS) + 2;
        int r_lo, c_lo, nrow, nbyte;
        const bool staged = tile_patch_extent<G, false>(L, ry0, rx0, rh, r_lo, c_lo, nrow, nbyte);
        WbTilePatch &o = out[i];
        o.r_lo = staged ? r_lo : 0;
        o.c_lo = staged ? c_lo : 0;
        o.rows = staged ? (uint16_t)nrow : 0;
        o.bytes = staged ? (uint16_t)nbyte : 0;
        o.pad = 0;
    }
}
template <bool FULL_ROWS>
int fill_patches_for(int shrink, bool smooth, bool big, const WbLevel *levels, const WbTile *tiles, int n_tiles, WbTilePatch *out) {
    constexpr int TV4 = FULL_ROWS ? 32 : WB_CHAN_S4_TV;      // (the uint8 channel functions keep the 8 x 32 tile)
#define WB_FP(S, TU, TV)                                                                  \
    if (smooth) fill_patches<TileGeom<S, TU, TV, true>, FULL_ROWS>(levels, tiles, n_tiles, out); \
    else fill_patches<TileGeom<S, TU, TV, false>, FULL_ROWS>(levels, tiles, n_tiles, out);
    switch (shrink) {
        case 1: WB_FP(1, 16, 64) return WB_OK;
        case 2: if (big) { WB_FP(2, 32, 64) } else { WB_FP(2, 16, 64) } return WB_OK;
        case 4: WB_FP(4, 8, TV4) return WB_OK;
    }
#undef WB_FP
    wb_set_error("wb_channels_tile_patches: shrink=%d unsupported (1, 2; 4 as an extension)", shrink);
    return WB_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" int wb_channels_tile_patches(int channel_func, int shrink, int smooth, const WbLevel *levels_host, int n_levels,
                                        const WbTile *tiles_host, int n_tiles, WbTilePatch *out_host) {
    WB_REQUIRE(levels_host && tiles_host && out_host && n_levels >= 1 && n_tiles >= 0, "wb_channels_tile_patches: null pointer / empty plan");
    for (int i = 0; i < n_tiles; ++i)
        WB_REQUIRE(tiles_host[i].level >= 0 && tiles_host[i].level < n_levels, "wb_channels_tile_patches: tile %d names level %d of %d", i,
                   tiles_host[i].level, n_levels);
    if (channel_func == WB_CHN_GRAD_HIST)
        return fill_patches_for<false>(shrink, smooth != 0, tile32(), levels_host, tiles_host, n_tiles, out_host);
    if (channel_func == WB_CHN_GRAD_HIST_4_U1 || channel_func == WB_CHN_GRAD_MAG_U1)
        return fill_patches_for<true>(shrink, smooth != 0, false, levels_host, tiles_host, n_tiles, out_host);
    wb_set_error("wb_channels_tile_patches: channel function %d takes no patch table", channel_func);
    return WB_ERR_UNSUPPORTED;
}

extern "C" int wb_channels_launch(void *stream, const void *img, int64_t img_stride, const void *oct,
                                  int64_t oct_stride, int dtype, int batch, const WbLevel *levels,
                                  int n_levels, const WbTile *tiles, int n_tiles, const uint32_t *minmax,
                                  int n_oct, const WbTap *taps, int channel_func, int shrink, int smooth,
                                  const double *cs_sn, void *chn, int64_t chn_stride, const WbModel *rank_model,
                                  uint8_t *rank, int64_t rank_stride) {
    return wb_channels_launch_x(stream, img, img_stride, oct, oct_stride, dtype, batch, levels, n_levels, tiles, n_tiles, minmax,
                                n_oct, taps, channel_func, shrink, smooth, cs_sn, chn, chn_stride, rank_model, rank, rank_stride,
                                nullptr, WB_DTYPE_RANK8);
}

extern "C" int wb_channels_launch_x(void *stream, const void *img, int64_t img_stride, const void *oct,
                                    int64_t oct_stride, int dtype, int batch, const WbLevel *levels,
                                    int n_levels, const WbTile *tiles, int n_tiles, const uint32_t *minmax,
                                    int n_oct, const WbTap *taps, int channel_func, int shrink, int smooth,
                                    const double *cs_sn, void *chn, int64_t chn_stride, const WbModel *rank_model,
                                    uint8_t *rank, int64_t rank_stride, const WbTilePatch *patches, int rank_dtype) {
    WB_REQUIRE(img && levels && tiles && minmax && taps && (chn || rank), "wb_channels_launch: null pointer");
    WB_REQUIRE(!rank || rank_dtype == WB_DTYPE_RANK8 || rank_dtype == WB_DTYPE_RANK16, "wb_channels_launch_x: rank_dtype %d (WB_DTYPE_RANK8 or WB_DTYPE_RANK16)", rank_dtype);
    const bool wide = rank && rank_dtype == WB_DTYPE_RANK16;
    WB_REQUIRE(!wide || rank_model->bin16_ok, "wb_channels_launch_x: this model has no 16-bit rank tables (wb_model_info: rank16_ok)");
    WB_REQUIRE(!wide || reinterpret_cast<uintptr_t>(rank) % 8 == 0, "wb_channels_launch_x: 16-bit ranks must be 8-byte aligned");
    WB_REQUIRE(!patches || (dtype == WB_DTYPE_U8 && channel_func != WB_CHN_GRAD_MAG),
               "wb_channels_launch_x: the patch table goes with uint8 images and the gradient-histogram kernels");
    WB_REQUIRE(!rank == !rank_model, "wb_channels_launch: rank and rank_model go together");
    WB_REQUIRE(!rank || channel_func == WB_CHN_GRAD_HIST, "wb_channels_launch: ranks are written for grad_hist channels only");
    WB_REQUIRE(!rank || wide || rank_model->bin_ok, "wb_channels_launch: this model has no rank tables (wb_model_info: rank_ok)");
    WB_REQUIRE(!rank || reinterpret_cast<uintptr_t>(rank) % 4 == 0, "wb_channels_launch: rank must be 4-byte aligned");
    WB_REQUIRE(cs_sn || channel_func != WB_CHN_GRAD_HIST, "wb_channels_launch: grad_hist needs the orientation constants");
    WB_REQUIRE(batch >= 1 && n_levels >= 1 && n_tiles >= 1, "wb_channels_launch: empty launch");
    WB_REQUIRE(batch <= 65535, "wb_channels_launch: batch %d exceeds grid.y limit", batch);
    WB_REQUIRE(smooth == 0 || smooth == 1, "wb_channels_launch: smooth must be 0 or 1");
    WB_REQUIRE(n_oct >= 1 && n_oct <= WB_MAX_OCTAVES, "wb_channels_launch: n_oct out of range");
    ChanArgs a;
    a.img = img;
    a.oct = oct;
    a.img_stride = img_stride;
    a.oct_stride = oct_stride;
    a.levels = levels;
    a.tiles = tiles;
    a.minmax = minmax;
    a.taps = taps;
    a.n_oct = n_oct;
    a.chn = chn;
    a.chn_stride = chn_stride;
    a.src_int = 0;
    a.rank = rank;
    a.rank_stride = rank_stride;
    a.rank_lut = nullptr;
    a.rank_iters = 0;
    a.patches = patches;
    a.rank_wide = wide ? 1 : 0;
    if (rank) {
        a.rank_lut = reinterpret_cast<const uint4 *>(wide ? rank_model->bin16_lut_dev : rank_model->bin_lut_dev);
        a.rank_iters = wide ? rank_model->bin16_iters : rank_model->bin_iters;
        for (int k = 0; k < 4; ++k) {
            a.rank_k[k] = wide ? rank_model->bin16_k[k] : rank_model->bin_k[k];
            a.rank_b[k] = wide ? rank_model->bin16_b[k] : rank_model->bin_b[k];
        }
    }
    if (cs_sn) set_constants(a, cs_sn);
    static const int dbg = getenv("WB_CHAN_DBG") ? atoi(getenv("WB_CHAN_DBG")) : 0;
    a.dbg = dbg;
    dim3 grid((unsigned)n_tiles, (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    if (channel_func == WB_CHN_GRAD_HIST_4_U1 || channel_func == WB_CHN_GRAD_MAG_U1) {
        if (dtype != WB_DTYPE_U8) {
            wb_set_error("wb_channels_launch: the uint8 channel functions take uint8 images (8 bit input, fpga/channels.py:32)");
            return WB_ERR_UNSUPPORTED;
        }
        return channel_func == WB_CHN_GRAD_HIST_4_U1 ? launch_u1<4>(st, grid, a, shrink, smooth != 0)
                                                     : launch_u1<1>(st, grid, a, shrink, smooth != 0);
    }
    if (channel_func == WB_CHN_GRAD_MAG) {
        // H = (1,2,..,6,..,2,1) as float32, divided by its float32 sum (channels.py:11-13)
        for (int i = 0; i < 11; ++i) a.tri[i] = (double)((float)(i < 6 ? i + 1 : 11 - i) / 36.0f);
        a.gm_eps = 1e-3f;
        if (dtype == WB_DTYPE_U8) return launch_gm<uint8_t>(st, grid, a, shrink, smooth != 0);
        if (dtype == WB_DTYPE_F32) return launch_gm<float>(st, grid, a, shrink, smooth != 0);
        wb_set_error("wb_channels_launch: unsupported dtype %d (uint8 and float32 images only)", dtype);
        return WB_ERR_UNSUPPORTED;
    }
    if (channel_func != WB_CHN_GRAD_HIST) {
        wb_set_error("wb_channels_launch: channel function %d has no kernel", channel_func);
        return WB_ERR_UNSUPPORTED;
    }
    if (dtype == WB_DTYPE_U8) {
        // integer gradients + canonical constants: exact fp32 projection (see project_int)
        static const bool no_fast = getenv("WB_CHAN_NO_FAST") != nullptr;
        if (canonical_constants(cs_sn) && !no_fast) return launch_dtype<uint8_t, true>(st, grid, a, shrink, smooth != 0, tile32());
        return launch_dtype<uint8_t, false>(st, grid, a, shrink, smooth != 0, tile32());
    }
    if (dtype == WB_DTYPE_F32) return launch_dtype<float, false>(st, grid, a, shrink, smooth != 0, tile32());
    if (wb_dtype_held_f64(dtype)) {
        a.src_int = wb_cast_mode(dtype);
        return launch_dtype<double, false>(st, grid, a, shrink, smooth != 0, tile32());
    }
    wb_set_error("wb_channels_launch: unsupported image dtype code %d", dtype);
    return WB_ERR_UNSUPPORTED;
}

extern "C" int wb_selftest_projection(void *stream, uint32_t *mismatches) {
    WB_REQUIRE(mismatches, "wb_selftest_projection: null pointer");
    ChanArgs a = {};
    double cs_sn[8];
    for (int k = 0; k < 4; ++k) {
        cs_sn[k] = kCanonCs[k];
        cs_sn[4 + k] = kCanonSn[k];
    }
    set_constants(a, cs_sn);
    const int n = 2041 * 2041;
    hipLaunchKernelGGL(selftest_projection_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, mismatches);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}


#ifdef WB_CASC_STAMPS
#include <vector>
// Diagnostic build: mean microseconds between consecutive stamps over the first n_wg workgroups of
// the last channel launch (s_memrealtime ticks at 100 MHz).  Workgroups that skipped a phase
// (direct path) contribute 0 to it.
extern "C" int wb_debug_channel_stamps(int n_wg, double *mean_us7, double *lifetime_us) {
    if (n_wg > WB_CSTAMP_WGS) n_wg = WB_CSTAMP_WGS;
    std::vector<unsigned long long> h((size_t)n_wg * WB_CSTAMP_SLOTS);
    WB_HIP_CHECK(hipDeviceSynchronize());
    WB_HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_chan_stamps), h.size() * 8));
    double acc[7] = {0}, life = 0;
    for (int w = 0; w < n_wg; ++w) {
        const unsigned long long *s = &h[(size_t)w * WB_CSTAMP_SLOTS];
        for (int k = 0; k < 7; ++k) acc[k] += (double)(s[k + 1] - s[k]);
        life += (double)(s[7] - s[0]);
    }
    for (int k = 0; k < 7; ++k) mean_us7[k] = acc[k] / n_wg / 100.0;
    *lifetime_us = life / n_wg / 100.0;
    return WB_OK;
}
#endif
