// Octave pyramid of the raw image (reference channels.py:93-101 + avg_pool_2 :55-64)
// and the per-octave min/max that skimage.resize clips its output to (channels.py:132).
//
// HBM-bound integer/byte work: the image is read once, every octave is written once.
// One workgroup owns a 128x128 block of octave 0 and derives its share of octaves 1..7
// hierarchically in LDS (a 2^k x 2^k block of octave 0 fully determines one pixel of octave k,
// odd tails included, because floor(floor(h/2)/2) = floor(h/4)); octaves >= 8 of very large
// images come from a one-workgroup tail kernel.  min/max are reduced per workgroup and leave
// with two atomics per octave.
//
// minmax encoding: mm[0] = max over ~key(pixel)  (so min key = ~mm[0]),  mm[1] = max over key;
// both are plain atomicMax on a word the CALLER has zeroed (like the cascade's counters: one memset of one
// control block per step can then serve every kernel of the step).
#include "wb_common.h"

namespace {

// block side at octave 0 and the number of octaves derived inside the block: 128 -> octaves
// 1..7 for bytes (20 KiB of LDS), 64 -> 1..6 for float32 (20 KiB)
template <typename T> struct Blk;
template <> struct Blk<uint8_t> { static constexpr int OB = 128, LEVELS = 7; };
template <> struct Blk<float> { static constexpr int OB = 64, LEVELS = 6; };

template <typename T> struct PixKey;
template <> struct PixKey<uint8_t> {
    static __device__ uint32_t key(uint8_t v) { return v; }
};
template <> struct PixKey<float> {
    static __device__ uint32_t key(float v) { return wb_f32_key(v); }
};

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

// workgroup reduction of (lo-as-~key max, hi max) -> two atomics by thread 0 (one pair per wave was
// measured 2x slower at batch 1: hundreds of atomics queue up on each octave's two words)
__device__ inline void block_minmax_commit(uint32_t nlo, uint32_t hi, uint32_t *red, uint32_t *mm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t l2 = __shfl_xor(nlo, o), h2 = __shfl_xor(hi, o);
        nlo = l2 > nlo ? l2 : nlo;
        hi = h2 > hi ? h2 : hi;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[2 * wave] = nlo;
        red[2 * wave + 1] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            nlo = red[2 * w] > nlo ? red[2 * w] : nlo;
            hi = red[2 * w + 1] > hi ? red[2 * w + 1] : hi;
        }
        atomicMax(mm + 0, nlo);
        atomicMax(mm + 1, hi);
    }
    __syncthreads();
}

struct OctDims {
    int h[WB_MAX_OCTAVES], w[WB_MAX_OCTAVES];
    int64_t off[WB_MAX_OCTAVES];
};

// grid = (blocks_x * blocks_y, batch); block = 256 threads
template <typename T>
__global__ __launch_bounds__(256) void octaves_block_kernel(const T *img, int64_t img_stride, T *oct, int64_t oct_stride,
                                                            OctDims d, int n_oct, int blocks_x, uint32_t *minmax,
                                                            uint32_t *zero, int zero_words) {
    // words a LATER kernel of the step accumulates into (the cascade's counters and statistics): nobody touches them while
    // this kernel runs, so its first workgroup resets them -- a step then needs no memset launch of its own
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < zero_words; i += 256) zero[i] = 0u;
    constexpr int OB = Blk<T>::OB, OB_LEVELS = Blk<T>::LEVELS;
    __shared__ __attribute__((aligned(16))) T bufA[OB * OB];
    __shared__ T bufB[(OB / 2) * (OB / 2)];
    __shared__ uint32_t red[4][2 * (OB_LEVELS + 1)];      // per wave: (max ~key, max key) of octaves 0..OB_LEVELS

    const int b = blockIdx.y;
    const int by = blockIdx.x / blocks_x, bx = blockIdx.x - by * blocks_x;
    const int tid = threadIdx.x;
    const T *src = img + (int64_t)b * img_stride;
    T *obase = oct + (int64_t)b * oct_stride;
    uint32_t *mm = minmax + (int64_t)b * n_oct * 2;

    // ---- octave 0 block -> LDS, with its min/max (all pixels, odd tails included).  Loads are
    //      unconditional (clamped addresses) and issued 8 at a time; bytes travel as dwords when the
    //      rows are 4-byte aligned.
    const int H = d.h[0], W = d.w[0];
    const int y0 = by * OB, x0 = bx * OB;
    uint32_t nlo = 0u, hi = 0u;
    constexpr int U = 8;
    // bytes, rows 16-byte aligned (the usual image): a thread takes 16 pixels of TWO rows (two 16-byte loads, all of a
    // block's loads in flight at once), pools them into 8 pixels of octave 1 in registers -- two byte lanes per dword at a
    // time -- and stores those as two dwords: octave 0 never goes through LDS, octave 1 leaves with 8-byte instead of
    // 1-byte stores (at batch 1 the kernel is one round of workgroups whose serial depth is the image's octave chain)
    bool from_regs = false;
    uint32_t nlo1 = 0u, hi1 = 0u;
    if constexpr (sizeof(T) == 1) {
        T *dst1 = obase + (n_oct > 1 ? d.off[1] : 0);
        from_regs = n_oct > 1 && (W & 15) == 0 && (img_stride & 15) == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 &&
                    (d.w[1] & 3) == 0 && (reinterpret_cast<uintptr_t>(dst1) & 3) == 0;
        if (from_regs) {
            constexpr int NG = OB / 16, NITEM = (OB / 2) * NG, PER = NITEM / 256;
            static_assert(NITEM % 256 == 0, "whole items per thread");
            const int oh = d.h[1], ow = d.w[1];
            uint4 va[PER], vb[PER];
            int item_r[PER], item_g[PER];
            bool ok0[PER], ok1[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int it = tid + k * 256;
                const int rp = it / NG, g = it - rp * NG;
                item_r[k] = rp;
                item_g[k] = g;
                const int ya = y0 + 2 * rp, yb = ya + 1, x = x0 + 16 * g;
                ok0[k] = ya < H && x < W;
                ok1[k] = yb < H && x < W;
                const int yac = ya < H ? ya : H - 1, ybc = yb < H ? yb : H - 1, xc = x < W ? x : W - 16;
                va[k] = *reinterpret_cast<const uint4 *>(src + (int64_t)yac * W + xc);
                vb[k] = *reinterpret_cast<const uint4 *>(src + (int64_t)ybc * W + xc);
            }
            uint32_t *lds1 = reinterpret_cast<uint32_t *>(bufB);          // octave 1 of the block: [OB/2][OB/2] bytes
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t a[4] = {va[k].x, va[k].y, va[k].z, va[k].w}, b[4] = {vb[k].x, vb[k].y, vb[k].z, vb[k].w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const uint32_t pa = (a[j] >> (8 * bb)) & 255u, pb = (b[j] >> (8 * bb)) & 255u;
                        if (ok0[k]) { nlo = (~pa) > nlo ? (~pa) : nlo; hi = pa > hi ? pa : hi; }
                        if (ok1[k]) { nlo = (~pb) > nlo ? (~pb) : nlo; hi = pb > hi ? pb : hi; }
                    }
                    // two pooled pixels per dword: the byte pairs summed in 16-bit lanes ((a+b+c+d) & 255) >> 2
                    const uint32_t sum = (a[j] & 0x00ff00ffu) + ((a[j] >> 8) & 0x00ff00ffu) + (b[j] & 0x00ff00ffu) + ((b[j] >> 8) & 0x00ff00ffu);
                    o[j] = ((sum >> 2) & 0x3fu) | (((sum >> 18) & 0x3fu) << 8);
                }
                const uint32_t out0 = o[0] | (o[1] << 16), out1 = o[2] | (o[3] << 16);
                const int rp = item_r[k], g = item_g[k];
                lds1[(rp * (OB / 2) + 8 * g) / 4] = out0;
                lds1[(rp * (OB / 2) + 8 * g) / 4 + 1] = out1;
                const int oy = (y0 >> 1) + rp, ox = (x0 >> 1) + 8 * g;
                if (oy < oh && ox < ow) {                                  // (ow % 4 == 0 and ox % 8 == 0: whole dwords in or out)
                    uint32_t *gp = reinterpret_cast<uint32_t *>(dst1 + (int64_t)oy * ow + ox);
                    gp[0] = out0;
                    if (ox + 4 < ow) gp[1] = out1;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const uint32_t pv = ((q < 4 ? out0 : out1) >> (8 * (q & 3))) & 255u;
                        if (ox + q < ow) { nlo1 = (~pv) > nlo1 ? (~pv) : nlo1; hi1 = pv > hi1 ? pv : hi1; }
                    }
                }
            }
        }
    }
    if (from_regs) {
    } else if (sizeof(T) == 1 && (W & 3) == 0 && (img_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0) {
        constexpr int DW = OB / 4;                                   // dwords per block row
        const uint32_t *src32 = reinterpret_cast<const uint32_t *>(src);
        uint32_t *lds32 = reinterpret_cast<uint32_t *>(bufA);
        const int wdw = W >> 2;
        for (int i0 = tid; i0 < OB * DW; i0 += 256 * U) {
            uint32_t v[U];
            bool ok[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                int i = i0 + k * 256;
                i = i < OB * DW ? i : OB * DW - 1;
                int r = i / DW, c = i - r * DW;
                int y = y0 + r, xd = (x0 >> 2) + c;
                ok[k] = (y < H) && (xd < wdw);
                y = y < H ? y : H - 1;
                xd = xd < wdw ? xd : wdw - 1;
                v[k] = src32[(int64_t)y * wdw + xd];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                int i = i0 + k * 256;
                if (i < OB * DW) {
                    uint32_t w = ok[k] ? v[k] : 0u;
                    lds32[i] = w;
                    if (ok[k]) {
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) {
                            uint32_t kk = (w >> (8 * bb)) & 255u;
                            nlo = (~kk) > nlo ? (~kk) : nlo;
                            hi = kk > hi ? kk : hi;
                        }
                    }
                }
            }
        }
    } else {
        for (int i0 = tid; i0 < OB * OB; i0 += 256 * U) {
            T v[U];
            bool ok[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                int i = i0 + k * 256;
                i = i < OB * OB ? i : OB * OB - 1;
                int r = i / OB, c = i - r * OB;
                int y = y0 + r, x = x0 + c;
                ok[k] = (y < H) && (x < W);
                y = y < H ? y : H - 1;
                x = x < W ? x : W - 1;
                v[k] = src[(int64_t)y * W + x];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                int i = i0 + k * 256;
                if (i < OB * OB) {
                    bufA[i] = ok[k] ? v[k] : T(0);
                    if (ok[k]) {
                        uint32_t kk = PixKey<T>::key(v[k]);
                        nlo = (~kk) > nlo ? (~kk) : nlo;
                        hi = kk > hi ? kk : hi;
                    }
                }
            }
        }
    }
    __syncthreads();
    // per-thread partial (max ~key, max key) of every octave this block touches; ONE workgroup reduction
    // and one atomic per value at the very end (a reduction + two atomics per octave cost two barriers each)
    uint32_t pmm[2 * (OB_LEVELS + 1)];
#pragma unroll
    for (int j = 0; j < 2 * (OB_LEVELS + 1); ++j) pmm[j] = 0u;
    pmm[0] = nlo;
    pmm[1] = hi;
    pmm[2] = nlo1;                       // (octave 1 straight from registers: see above)
    pmm[3] = hi1;

    // ---- octaves 1..7: pool LDS -> LDS (+ global), ping-pong between bufA and bufB
    T *cur = from_regs ? bufB : bufA;
    T *nxt = from_regs ? bufA : bufB;
    int side = from_regs ? OB / 2 : OB;  // side of `cur`
    const int kmax = n_oct - 1 < OB_LEVELS ? n_oct - 1 : OB_LEVELS;
    int k = from_regs ? 2 : 1;
    for (; k <= kmax; ++k) {
        if ((side >> 1) * (side >> 1) <= 64) break;       // the small octaves: one wave, no barriers (below)
        const int ns = side >> 1;
        const int oh = d.h[k], ow = d.w[k];
        const int oy0 = y0 >> k, ox0 = x0 >> k;
        T *dst = obase + d.off[k];
        nlo = 0u;
        hi = 0u;
        for (int i = tid; i < ns * ns; i += 256) {
            int r = i / ns, c = i - r * ns;
            const T *p0 = cur + (2 * r) * side + 2 * c;
            T v = pool4<T>(p0[0], p0[side], p0[1], p0[side + 1]);
            nxt[r * ns + c] = v;
            int y = oy0 + r, x = ox0 + c;
            if (y < oh && x < ow) {
                dst[(int64_t)y * ow + x] = v;
                uint32_t key = PixKey<T>::key(v);
                nlo = (~key) > nlo ? (~key) : nlo;
                hi = key > hi ? key : hi;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 1; j <= OB_LEVELS; ++j)          // (static indexing keeps pmm in registers)
            if (j == k) {
                pmm[2 * j] = nlo;
                pmm[2 * j + 1] = hi;
            }
        T *t = cur;
        cur = nxt;
        nxt = t;
        side = ns;
    }
    // ---- the octaves of at most 64 pixels per block (the last four of a 128 x 128 block): wave 0 alone walks them, lane =
    //      pixel -- LDS serves a wave's accesses in order, so a level's reads see the previous level's writes without a
    //      workgroup barrier; the other waves go straight to the final reduction (at batch 1 the kernel is ONE round of
    //      workgroups whose length is this serial chain: four barriers and four sweeps of 256 threads over a handful of
    //      pixels less)
    if (tid < 64) {
        for (; k <= kmax; ++k) {
            const int ns = side >> 1;
            const int oh = d.h[k], ow = d.w[k];
            const int oy0 = y0 >> k, ox0 = x0 >> k;
            T *dst = obase + d.off[k];
            nlo = 0u;
            hi = 0u;
            if (tid < ns * ns) {
                const int r = tid / ns, c = tid - r * ns;
                const T *p0 = cur + (2 * r) * side + 2 * c;
                const T v = pool4<T>(p0[0], p0[side], p0[1], p0[side + 1]);
                nxt[r * ns + c] = v;
                const int y = oy0 + r, x = ox0 + c;
                if (y < oh && x < ow) {
                    dst[(int64_t)y * ow + x] = v;
                    const uint32_t key = PixKey<T>::key(v);
                    nlo = ~key;
                    hi = key;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 1; j <= OB_LEVELS; ++j)
                if (j == k) {
                    pmm[2 * j] = nlo;
                    pmm[2 * j + 1] = hi;
                }
            T *t = cur;
            cur = nxt;
            nxt = t;
            side = ns;
        }
    }
    // ---- min/max of all octaves: wave reduction, exchange through LDS, then lane j commits value j
    const int nval = 2 * (kmax + 1);
#pragma unroll
    for (int j = 0; j < 2 * (OB_LEVELS + 1); ++j) {
        uint32_t v = pmm[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            uint32_t w = __shfl_xor(v, o);
            v = w > v ? w : v;
        }
        if ((tid & 63) == 0) red[tid >> 6][j] = v;
    }
    __syncthreads();
    if (tid < nval) {
        uint32_t v = red[0][tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) v = red[w][tid] > v ? red[w][tid] : v;
        if (v) atomicMax(mm + tid, v);                 // mm is [octave][2]: index 2*k + {0: ~min key, 1: max key}
    }
}

// octaves beyond the block's reach (bytes: images of 2048+ px on the short side): one workgroup per image walks
// the remaining octaves in LDS, starting from octave 7 in HBM (written by the launch before).
template <typename T>
__global__ __launch_bounds__(256) void octaves_tail_kernel(T *oct, int64_t oct_stride, OctDims d, int n_oct,
                                                           uint32_t *minmax) {
    constexpr int OB_LEVELS = Blk<T>::LEVELS;
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    __shared__ uint32_t red[8];
    T *cur = reinterpret_cast<T *>(tail_smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    T *obase = oct + (int64_t)b * oct_stride;
    uint32_t *mm = minmax + (int64_t)b * n_oct * 2;
    int sh = d.h[OB_LEVELS], sw = d.w[OB_LEVELS];
    T *nxt = cur + sh * sw;
    for (int i = tid; i < sh * sw; i += 256) cur[i] = obase[d.off[OB_LEVELS] + i];
    __syncthreads();
    for (int k = OB_LEVELS + 1; k < n_oct; ++k) {
        const int oh = d.h[k], ow = d.w[k];
        T *dst = obase + d.off[k];
        uint32_t nlo = 0u, hi = 0u;
        for (int i = tid; i < oh * ow; i += 256) {
            int r = i / ow, c = i - r * ow;
            const T *p0 = cur + (2 * r) * sw + 2 * c;
            T v = pool4<T>(p0[0], p0[sw], p0[1], p0[sw + 1]);
            nxt[i] = v;
            dst[i] = v;
            uint32_t key = PixKey<T>::key(v);
            nlo = (~key) > nlo ? (~key) : nlo;
            hi = key > hi ? key : hi;
        }
        __syncthreads();
        block_minmax_commit(nlo, hi, red, mm + 2 * k);
        T *t = cur;
        cur = nxt;
        nxt = t;
        sh = oh;
        sw = ow;
    }
}

// -------------------------------------------------------------------------------------------
// Every other image dtype the reference accepts (channels.py:122: "dtype = image.dtype"): float64, and the
// integer types, held as float64 (exact: |v| < 2^53).  avg_pool_2 (channels.py:55-64) adds in the ARRAY's
// dtype -- integers wrap modulo 2^bits -- divides by 4 in float64 and casts back by truncation; float64 adds
// ((a+b)+c)+d and divides exactly.  min/max are 64-bit order-preserving keys (word 0: max of ~key, word 1:
// max of key, atomicMax on zeroed words, like the 32-bit ones).  One plain launch per octave: these are the
// rare dtypes, correct rather than tuned.
__device__ inline unsigned long long f64_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ inline void wave_minmax64_commit(unsigned long long nlo, unsigned long long hi, unsigned long long *mm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(nlo, o), h2 = __shfl_xor(hi, o);
        nlo = l2 > nlo ? l2 : nlo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        if (nlo) atomicMax(mm + 0, nlo);
        if (hi) atomicMax(mm + 1, hi);
    }
}

// wrap an exact integer sum (|s| < 2^36) into the integer type of `bits` bits (two's complement if sgn)
__device__ inline double wrap_int(double s, int bits, int sgn) {
    const double m = (double)(1ull << bits);
    double r = s - floor(s / m) * m;                    // s mod 2^bits, exact (powers of two, small integers)
    if (sgn && r >= m * 0.5) r -= m;
    return r;
}

__global__ __launch_bounds__(256) void minmax_f64_kernel(const double *img, int64_t img_stride, int64_t n, int n_oct,
                                                         unsigned long long *minmax, uint32_t *zero, int zero_words) {
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < zero_words; i += 256) zero[i] = 0u;          // (see octaves_block_kernel)
    const double *src = img + (int64_t)blockIdx.y * img_stride;
    unsigned long long nlo = 0, hi = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned long long k = f64_key(src[i]);
        nlo = ~k > nlo ? ~k : nlo;
        hi = k > hi ? k : hi;
    }
    wave_minmax64_commit(nlo, hi, minmax + (int64_t)blockIdx.y * n_oct * 2);
}

// octave k from octave k - 1 (src: sh x sw, dst: oh x ow = floor(sh/2) x floor(sw/2))
__global__ __launch_bounds__(256) void pool_f64_kernel(const double *src_base, int64_t src_stride, int sw, double *dst_base,
                                                       int64_t dst_stride, int oh, int ow, int bits, int sgn, int k, int n_oct,
                                                       unsigned long long *minmax) {
    const double *src = src_base + (int64_t)blockIdx.y * src_stride;
    double *dst = dst_base + (int64_t)blockIdx.y * dst_stride;
    unsigned long long nlo = 0, hi = 0;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)oh * ow) {
        const int r = (int)(i / ow), c = (int)(i - (int64_t)r * ow);
        const double *p0 = src + (int64_t)(2 * r) * sw + 2 * c;
        const double a = p0[0], b = p0[sw], cc = p0[1], d = p0[sw + 1];      // [2r,2c], [2r+1,2c], [2r,2c+1], [2r+1,2c+1]
        double v;
        if (bits > 0) {
            // each add wraps in the array's dtype; wrapping once at the end is the same residue
            v = trunc(wrap_int(((a + b) + cc) + d, bits, sgn) / 4.0);
        } else if (bits == -WB_CAST_TRUNC) {
            v = trunc((((a + b) + cc) + d) / 4.0);          // int64 / uint64 values exact in float64: no rounding, no wrap
        } else if (bits == -WB_CAST_BOOL) {
            v = (a != 0.0 || b != 0.0 || cc != 0.0 || d != 0.0) ? 1.0 : 0.0;    // bool + bool is logical or; x / 4 != 0
        } else if (bits == -WB_CAST_F16) {
            v = wb_round_f16(wb_round_f16(wb_round_f16(wb_round_f16(a + b) + cc) + d) / 4.0);   // every float16 add rounds
        } else {
            v = (((a + b) + cc) + d) / 4.0;
        }
        dst[i] = v;
        const unsigned long long key = f64_key(v);
        nlo = ~key;
        hi = key;
    }
    wave_minmax64_commit(nlo, hi, minmax + ((int64_t)blockIdx.y * n_oct + k) * 2);
}

int launch_octaves_f64(hipStream_t st, const double *img, int dtype, int batch, int H, int W, int64_t img_stride, double *oct,
                       int64_t oct_stride, const int64_t *oct_off, int n_oct, unsigned long long *minmax, uint32_t *zero, int zero_words) {
    int bits = 0, sgn = 0;
    switch (dtype) {
        case WB_DTYPE_F64: break;
        case WB_DTYPE_I8: bits = 8; sgn = 1; break;
        case WB_DTYPE_I16: bits = 16; sgn = 1; break;
        case WB_DTYPE_U16: bits = 16; break;
        case WB_DTYPE_I32: bits = 32; sgn = 1; break;
        case WB_DTYPE_U32: bits = 32; break;
        case WB_DTYPE_I64: case WB_DTYPE_U64: bits = -WB_CAST_TRUNC; break;     // (negative: no wrap, see pool_f64_kernel)
        case WB_DTYPE_BOOL: bits = -WB_CAST_BOOL; break;
        case WB_DTYPE_F16: bits = -WB_CAST_F16; break;
    }
    const int64_t n0 = (int64_t)H * W;
    int blocks = (int)((n0 + 255) / 256);
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(minmax_f64_kernel, dim3(blocks, batch), dim3(256), 0, st, img, img_stride, n0, n_oct, minmax, zero, zero_words);
    int sh = H, sw = W;
    for (int k = 1; k < n_oct; ++k) {
        const int oh = sh >> 1, ow = sw >> 1;
        const double *src = k == 1 ? img : oct + oct_off[k - 1];
        const int64_t sstride = k == 1 ? img_stride : oct_stride;
        hipLaunchKernelGGL(pool_f64_kernel, dim3((unsigned)(((int64_t)oh * ow + 255) / 256), batch), dim3(256), 0, st, src, sstride,
                           sw, oct + oct_off[k], oct_stride, oh, ow, bits, sgn, k, n_oct, minmax);
        sh = oh;
        sw = ow;
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

template <typename T>
int launch_octaves(hipStream_t st, const T *img, int batch, int H, int W, int64_t img_stride, T *oct,
                   int64_t oct_stride, const int64_t *oct_off, int n_oct, uint32_t *minmax, uint32_t *zero, int zero_words) {
    OctDims d;
    int h = H, w = W;
    for (int k = 0; k < n_oct; ++k) {
        d.h[k] = h;
        d.w[k] = w;
        d.off[k] = k ? oct_off[k] : 0;
        h >>= 1;
        w >>= 1;
    }
    constexpr int OB = Blk<T>::OB, OB_LEVELS = Blk<T>::LEVELS;
    const int bx = (W + OB - 1) / OB, by = (H + OB - 1) / OB;
    hipLaunchKernelGGL(octaves_block_kernel<T>, dim3(bx * by, batch), dim3(256), 0, st, img, img_stride, oct, oct_stride,
                       d, n_oct, bx, minmax, zero, zero_words);
    if (n_oct > OB_LEVELS + 1) {
        size_t px = (size_t)d.h[OB_LEVELS] * d.w[OB_LEVELS];
        size_t lds = (px + px / 4 + 16) * sizeof(T);
        if (lds > 150 * 1024) {
            wb_set_error("wb_octaves_launch: %dx%d image too large for the octave tail kernel", H, W);
            return WB_ERR_UNSUPPORTED;
        }
        if (lds > 48 * 1024)
            WB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&octaves_tail_kernel<T>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(octaves_tail_kernel<T>, dim3(batch), dim3(256), lds, st, oct, oct_stride, d, n_oct, minmax);
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

}  // namespace

extern "C" int wb_octaves_launch(void *stream, const void *img, int dtype, int batch, int H, int W,
                                 int64_t img_stride, void *oct, int64_t oct_stride, const int64_t *oct_off,
                                 int n_oct, uint32_t *minmax) {
    return wb_octaves_launch_z(stream, img, dtype, batch, H, W, img_stride, oct, oct_stride, oct_off, n_oct, minmax, nullptr, 0);
}

extern "C" int wb_octaves_launch_z(void *stream, const void *img, int dtype, int batch, int H, int W,
                                   int64_t img_stride, void *oct, int64_t oct_stride, const int64_t *oct_off,
                                   int n_oct, uint32_t *minmax, uint32_t *zero, int zero_words) {
    WB_REQUIRE(img && minmax, "wb_octaves_launch: null pointer");
    WB_REQUIRE(zero_words == 0 || (zero && zero_words > 0), "wb_octaves_launch_z: zero_words without a pointer");
    WB_REQUIRE(batch >= 1 && batch <= 65535 && H >= 1 && W >= 1, "wb_octaves_launch: bad shape batch=%d H=%d W=%d", batch, H, W);
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
        return launch_octaves<uint8_t>(st, (const uint8_t *)img, batch, H, W, img_stride, (uint8_t *)oct, oct_stride, oct_off, n_oct, minmax, zero, zero_words);
    if (dtype == WB_DTYPE_F32)
        return launch_octaves<float>(st, (const float *)img, batch, H, W, img_stride, (float *)oct, oct_stride, oct_off, n_oct, minmax, zero, zero_words);
    if (wb_dtype_held_f64(dtype))
        return launch_octaves_f64(st, (const double *)img, dtype, batch, H, W, img_stride, (double *)oct, oct_stride, oct_off, n_oct,
                                  reinterpret_cast<unsigned long long *>(minmax), zero, zero_words);
    wb_set_error("wb_octaves_launch: unsupported image dtype code %d", dtype);
    return WB_ERR_UNSUPPORTED;
}
