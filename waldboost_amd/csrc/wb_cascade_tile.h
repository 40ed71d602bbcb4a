// The tiled cascade kernel (see wb_cascade.hip for the design notes), as a device function template, in a header of
// its own so that the SAME source builds two ways:
//   * wb_cascade.hip includes it for the generic kernels (stage records read through the scalar cache);
//   * wb_jit.hip hands it to hiprtc, behind a generated prelude that defines WB_JIT_BAKED and the stage table of ONE
//     model as constants (wb_model_specialize) -- the model-specialised kernel.
// Under hiprtc (__HIPCC_RTC__) there are no system headers: the few ABI structs are repeated below and checked for size.
#pragma once
#ifdef __HIPCC_RTC__
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
typedef unsigned long size_t;
#define WB_DET_SHARDS 64
#define WB_CASC_TC 64
#define WB_STAGE_NI(D) ((1 << (D)) - 1)
#define WB_STAGE_NL(D) (1 << (D))
#define WB_STAGE_DWORDS(D) ((((2 * WB_STAGE_NI(D) + WB_STAGE_NL(D) + 1) + 3) / 4) * 4)
struct WbLevel {
    int32_t oct, src_h, src_w, nh, nw, u, v, tap_off;
    int64_t src_off, chn_off;
    double sy, sx;
};
struct WbTile {
    int32_t level;
    uint16_t ty, tx;
};
struct WbDet {
    int32_t image, level;
    uint16_t r, c;
    float score;
};
#define INFINITY __builtin_huge_valf()
#else
#include "wb_common.h"
#endif
static_assert(sizeof(WbLevel) == 64 && sizeof(WbTile) == 8 && sizeof(WbDet) == 16, "ABI structs (include/waldboost_hip.h)");

namespace {

// the stage table of a BAKED build (wb_jit.hip generates the two macros); a one-word stand-in otherwise
#ifdef WB_JIT_BAKED
static __device__ const int32_t kWbStages[] = {WB_JIT_STAGE_WORDS};
#ifndef WB_JIT_LDS_STAGES
#define WB_JIT_LDS_STAGES WB_JIT_T      // stage records mirrored in LDS: all of them, or 0 (a table beyond 16 KiB: long cascades)
#endif
#ifndef WB_JIT_WAVES
#define WB_JIT_WAVES 8
#endif
#else
static __device__ const int32_t kWbStages[1] = {0};
#define WB_JIT_SEGMENTS(X)
// (a BAKED build also knows the model's geometry at compile time: stage count, channels, LDS tile rows and pitch)
#define WB_JIT_T 0
#define WB_JIT_LDS_STAGES 0
#define WB_JIT_WAVES 8
#define WB_JIT_C 0
#define WB_JIT_ROWS 0
#define WB_JIT_PITCH 0
#endif
// Dynamic LDS layout of the tile kernel (ALL of its LDS: no static __shared__, so the region starts at LDS address 0
// and a BAKED build can use absolute LDS addresses as instruction offsets):
//   [ channel tile | per-wave queues TR*64*8 | hist T*4 | (16-aligned) stage-table mirror | control words 128 B ]
// eb: bytes per element of a byte tile (1: uint8 channels / 8-bit ranks, 2: 16-bit ranks), 0: the planar float32 tile
__host__ __device__ constexpr size_t wb_lds_tile_bytes(int eb, int C, int rows, int pitch) {
    return eb ? (((size_t)C * rows * pitch * eb + 15) & ~(size_t)15) : (size_t)C * rows * pitch * 4;
}
// Entries of the workgroup's survivor queue (8 bytes each).  Round 4: capped -- 64 * WAVES pooled chunks + 512 -- instead
// of one entry per window of the tile (2048 for the 32 x 64 tile: 16 KB of the workgroup's 36 KB): a cascade rejects most
// windows within its first eight stages (431 of 2048 survive them on the benchmark model), and the rare tile that keeps
// more goes on DENSELY (a lane mask per row, as in phase A) eight stages at a time until its survivors fit.  Under 32 KB
// of LDS a fifth workgroup fits a CU once waves of the resident four have left (they no longer wait at a final barrier).
#ifndef WB_CASC_QCAP_DEFINED
#define WB_CASC_QCAP_DEFINED
#ifndef WB_CASC_QFULL
#define WB_CASC_QFULL 0      // 1: the round-3 layout (A/B builds)
#endif
__host__ __device__ constexpr int wb_casc_qcap(int TR, int WAVES) {
    return (WB_CASC_QFULL || TR * 64 < 64 * WAVES + 512) ? TR * 64 : 64 * WAVES + 512;
}
#endif
__host__ __device__ constexpr size_t wb_lds_stab_off(int eb, int C, int rows, int pitch, int TR, int T, int WAVES) {
    return (wb_lds_tile_bytes(eb, C, rows, pitch) + (size_t)wb_casc_qcap(TR, WAVES) * 8 + (size_t)T * 4 + 15) & ~(size_t)15;
}
#define WB_LDS_CTL_BYTES 256
// experiment switches (A/B builds; the defaults are what measured best)
// every lambda of the kernel body is inlined by decree: left to its cost model the compiler made a real function of
// run_segments in a 1024-stage depth-3 build, with the captured state handed over through scratch memory
#define WB_INLINE_LAMBDA __attribute__((always_inline))
#ifndef WB_TAIL_W
#define WB_TAIL_W 2          // windows the stage-parallel tail walks side by side
#endif
#ifndef WB_TAIL_ALL
#define WB_TAIL_ALL 1        // tail: gather every node's feature up front (eval_all)
#endif
#ifndef WB_TAIL_PRIO
#define WB_TAIL_PRIO 3       // s_setprio of a wave inside the stage-parallel tail (0 = off)
#endif
#ifndef WB_CASC_END_BARRIER
#define WB_CASC_END_BARRIER 0    // 1: a workgroup barrier in front of the statistics flush instead of the arrival counter (A/B builds)
#endif
#ifndef WB_SEG_PREFETCH
#define WB_SEG_PREFETCH 1    // BAKED segments: next group's gathers before this group's rejection tests
#endif

template <bool BAKED> __device__ __forceinline__ const int32_t *wb_stage_table(const int32_t *stages) {
    if constexpr (BAKED)
        return kWbStages;
    else
        return stages;
}

struct CascArgs {
    const void *chn;            // [u][v][C] float32, or uint8 when chn_u8
    int chn_u8;
    int64_t chn_stride;
    const WbLevel *levels;
    const WbTile *tiles;
    int n_levels;
    const int32_t *stages;      // stage records with LDS float offsets
    int T, m, n, C;
    int lds_rows, lds_pitch;
    int lds_stages;             // stage records mirrored in LDS for the stage-parallel tail (0 = read them from HBM)
    WbDet *det;
    uint32_t *det_count;
    uint32_t det_cap;           // per shard
    uint32_t *alive;            // [batch][n_levels][T], accumulated into (nullptr: no statistics)
    int n_tiles;
    int spar_wg;                // the whole tile goes stage-parallel after phase A when it holds at most this many windows
    int spar[4];                // stage-parallel tail entry: (t >= spar[0] && n <= spar[1]) || (t >= spar[2] && n <= spar[3])
    uint32_t *zero;             // words the NEXT step's first kernel accumulates into (the octaves' min / max keys): reset here
    int zero_words;
    int dbg;                    // diagnostics (WB_CASC_DBG): 1 = skip the tile load, 2 = stop after the load
};

__device__ inline float as_f(int32_t x) { return __int_as_float(x); }

// theta == -inf for a wave-uniform theta, decided on the SCALAR unit: the bits go through an opaque scalar register so
// the comparison stays an integer one (written as a float test -- or as a plain bit test, which the compiler turns
// back into a float test -- it was a vector compare per stage)
template <bool BAKED = false> __device__ inline bool never_rejects(float theta) {
    int bits = __float_as_int(theta);
    if constexpr (!BAKED) asm volatile("" : "+s"(bits));       // (a baked theta is a constant: the test folds away)
    return bits == (int)0xff800000;
}

// Diagnostic build only (make STAMPS=1): wave 0 of every workgroup stores s_memrealtime at its phase
// boundaries into a private slot (plain stores, nothing reads them in the kernel); the host turns
// them into mean wall-clock per phase (wb_debug_cascade_stamps).  Never part of a measured build.
#ifdef WB_CASC_STAMPS
#define WB_STAMP_SLOTS 8
#define WB_STAMP_WGS (1 << 16)
__device__ unsigned long long g_stamps[WB_STAMP_WGS * WB_STAMP_SLOTS];
#define WB_STAMP(k)                                                                                       \
    do {                                                                                                  \
        unsigned long long _wg = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;                 \
        if (threadIdx.x == 0 && _wg < WB_STAMP_WGS) g_stamps[_wg * WB_STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define WB_STAMP(k) do {} while (0)
#endif


// a[path] for the root-to-node path bits[0..] (false = left, true = right), first decision first
template <int N, typename V> struct Sel {
    static __device__ inline V get(const V *a, const bool *bits) {
        V lo = Sel<N / 2, V>::get(a, bits + 1);
        V hi = Sel<N / 2, V>::get(a + N / 2, bits + 1);
        return bits[0] ? hi : lo;
    }
};
template <typename V> struct Sel<1, V> {
    static __device__ inline V get(const V *a, const bool *) { return a[0]; }
};

// The stage record (see wb_common.h): in SGPRs when every lane is at the same stage
// (wave-uniform address -> s_load), in VGPRs in the stage-parallel tail (one stage per lane).
template <int D> struct Stage {
    static constexpr int NI = WB_STAGE_NI(D), NL = WB_STAGE_NL(D);
    int off[NI];
    float thr[NI];
    float pred[NL];
    float theta;
    __device__ inline void load(const int32_t *sp) {
#pragma unroll
        for (int i = 0; i < NI; ++i) off[i] = sp[i];
#pragma unroll
        for (int i = 0; i < NI; ++i) thr[i] = as_f(sp[NI + i]);
#pragma unroll
        for (int i = 0; i < NL; ++i) pred[i] = as_f(sp[2 * NI + i]);
        theta = as_f(sp[2 * NI + NL]);
    }
    // walk the complete depth-D tree for the window whose origin is at BYTE offset `base` of the
    // LDS tile (offsets in the records are bytes too: one v_add per gather)
    // BYTES: the tile holds uint8 pixels ([row][col][C] bytes), the record's offsets address it and its
    // thresholds are integers (wb_api.hip: fill<true>): an 8-bit gather and an integer compare per node.
    template <int BYTES> static __device__ inline bool goes_right(const char *t8, int at, float th) {
        if constexpr (BYTES == 2) {                      // 16-bit ranks: offsets are bytes, always even
            const int v = *reinterpret_cast<const uint16_t *>(t8 + at);
            return !(v <= __float_as_int(th));
        } else if constexpr (BYTES == 1) {
            const int v = *reinterpret_cast<const uint8_t *>(t8 + at);
            return !(v <= __float_as_int(th));
        } else {
            const float v = *reinterpret_cast<const float *>(t8 + at);
            return !(v <= th);                         // NaN goes right, like the reference's `<=`
        }
    }
    // the same walk with EVERY node's feature gathered up front (depth <= 2): one LDS round trip instead of one per level,
    // no selects of offsets or thresholds -- for records that sit in vector registers (the stage-parallel tail: one stage
    // per lane), where the selects of the leaf values cost no moves
    template <int BYTES = 0> __device__ inline float eval_all(const float *tile, int base) const {
        if constexpr (D > 2) {
            return eval<BYTES>(tile, base);
        } else {
            const char *t8 = reinterpret_cast<const char *>(tile);
            const bool r0 = goes_right<BYTES>(t8, base + off[0], thr[0]);
            if constexpr (D == 1) {
                return r0 ? pred[1] : pred[0];
            } else {
                const bool rl = goes_right<BYTES>(t8, base + off[1], thr[1]), rr = goes_right<BYTES>(t8, base + off[2], thr[2]);
                const float pl = rl ? pred[1] : pred[0], pr = rr ? pred[3] : pred[2];
                return r0 ? pr : pl;
            }
        }
    }
    template <int BYTES = 0> __device__ inline float eval(const float *tile, int base) const {
        const char *t8 = reinterpret_cast<const char *>(tile);
        bool right[D];
#pragma unroll
        for (int d = 0; d < D; ++d) right[d] = false;
        right[0] = goes_right<BYTES>(t8, base + off[0], thr[0]);
        if constexpr (D > 1) {
            int o = Sel<2, int>::get(off + 1, right);
            float th = Sel<2, float>::get(thr + 1, right);
            right[1] = goes_right<BYTES>(t8, base + o, th);
        }
        if constexpr (D > 2) {
            int o = Sel<4, int>::get(off + 3, right);
            float th = Sel<4, float>::get(thr + 3, right);
            right[2] = goes_right<BYTES>(t8, base + o, th);
        }
        return Sel<NL, float>::get(pred, right);
    }
};

__device__ inline int lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Stage T of a BAKED build: every field of the record is a compile-time constant.
//   * byte tiles, depth <= 2 (FAST): no selects between constants at all.  A node test `x > thr` on bytes is bit 8 of
//     y = x + (255 - thr)  (x in 0..255, thr in -1..255).  Depth 2: BOTH children are gathered (LDS reads with the
//     feature offsets as instruction offsets: no address arithmetic), yL = xL + (255 - thrL), yR = xR + (767 - thrR)
//     -- bit 9 marks the right side -- and ONE select by the root's test gives y with (y >> 6) & 12 = 4 * leaf.  The leaf
//     value is read from the stage table's LDS mirror at that offset (one ds_read_b32, record offset as an immediate).
//     Per window and stage: 3 slow-class vector instructions (root compare, select, theta compare) + 5 fast ones, against
//     ~13 with the records in scalar registers.
//   * otherwise: the generic walk on the constant record (selects with literal operands).
typedef const __attribute__((address_space(3))) unsigned char *WbLdsU8;
typedef const __attribute__((address_space(3))) unsigned short *WbLdsU16;
typedef const __attribute__((address_space(3))) float *WbLdsF32;
template <int D, int EB, int TR, int T> struct StageAt {
    static constexpr int SD = WB_STAGE_DWORDS(D), NI = WB_STAGE_NI(D), NL = WB_STAGE_NL(D);
    static constexpr bool FAST = EB != 0 && D <= 2 && WB_JIT_LDS_STAGES != 0;     // (the leaf values come from the LDS mirror)
    // (16-bit elements: the same bit trick one byte wider -- `x > thr` is bit 16 of x + (65535 - thr), the right child's
    // sum carries bit 17, and 4 * leaf = (y >> 14) & 12)
    static constexpr uint32_t XMAX = EB == 2 ? 65535u : 255u, RIGHT = EB == 2 ? 131072u : 512u;
    static constexpr int LEAF_SHIFT = EB == 2 ? 14 : 6;
    static __device__ __forceinline__ int off(int i) { return kWbStages[T * SD + i]; }
    static __device__ __forceinline__ int thr(int i) { return kWbStages[T * SD + NI + i]; }
    static __device__ __forceinline__ float theta() { return as_f(kWbStages[T * SD + 2 * NI + NL]); }
    // tile: the LDS tile (at LDS address 0: FAST addresses it by number); base: byte offset of the window's origin
    static __device__ __forceinline__ float eval(const float *tile, int base) {
        if constexpr (FAST) {
            // absolute LDS addresses (the tile starts at 0, the stage mirror at a compile-time offset): every constant
            // part of an address is an instruction offset, nothing is added on the vector unit
            constexpr uint32_t PRED = (uint32_t)wb_lds_stab_off(EB, WB_JIT_C, WB_JIT_ROWS, WB_JIT_PITCH, TR, WB_JIT_T, WB_JIT_WAVES) + (T * SD + 2 * NI) * 4;
            auto px = [&](int i) {
                if constexpr (EB == 2)
                    return (uint32_t)*(WbLdsU16)(uint32_t)(base + off(i));
                else
                    return (uint32_t)*(WbLdsU8)(uint32_t)(base + off(i));
            };
            uint32_t y;
            if constexpr (D == 1) {
                y = px(0) + (uint32_t)((int)XMAX - thr(0));
                y = (y >> LEAF_SHIFT) & 4u;
            } else {
                const uint32_t x0 = px(0), xl = px(1), xr = px(2);
                const uint32_t yl = xl + (uint32_t)((int)XMAX - thr(1)), yr = xr + (uint32_t)((int)(XMAX + RIGHT) - thr(2));
                y = ((int)x0 > thr(0)) ? yr : yl;
                y = (y >> LEAF_SHIFT) & 12u;
            }
            return *(WbLdsF32)(PRED + y);
        } else {
            Stage<D> st;
            st.load(kWbStages + T * SD);
            return st.template eval<EB>(tile, base);
        }
    }
};

// Stages are evaluated in groups of G: the G tree walks of a window are independent (only the
// fp32 accumulation and the rejection tests are sequential), so their 2*G LDS gathers and the G
// scalar record loads are all in flight together and the wave's latency chain per stage drops
// G-fold.  A window that dies inside a group has had a few stages evaluated in vain; nothing it
// produced is ever used.  The stage table is padded with G no-op records so a group may start
// at any stage < T.
template <int D> struct GroupSize { static constexpr int G = (D >= 3) ? 2 : 4; };

// U8: the channels are bytes -- uint8 channels as the reference's integer channel functions produce them, or the
// threshold RANKS of float32 channels (wb_channels_launch with a rank model; the stage records then carry the
// thresholds' indices): the tile is the pixels as they are, a quarter of the float tile.
// BAKED (the JIT build, wb_jit.hip): the stage records of ONE model are compile-time constants (kWbStages) -- feature offsets,
// thresholds, leaf values and theta become instruction immediates, no scalar loads, no moves in front of the selects --
// for phase A and for the wave-synchronous segments; the stage-parallel tail reads per-lane records as before.
template <bool V> struct WbTag { static constexpr bool value = V; };
template <bool B, int TB, int TE> struct WbSeg { static constexpr bool baked = B; static constexpr int tb = TB, te = TE; };
template <int V> struct WbInt { static constexpr int value = V; };
// f(WbInt<B>{}), f(WbInt<B + S>{}), ... while f returns true and the index stays below E: a loop whose index is a constant
template <int B, int E, int S, class F> __device__ __forceinline__ void wb_static_for(F &&f) {
    if constexpr (B < E) {
        if (f(WbInt<B>{})) wb_static_for<B + S, E, S>(f);
    }
}

template <int D, int RPW, int WAVES, int EB, bool BAKED>
__device__ __forceinline__ void cascade_tile_body(const CascArgs &a, const int32_t *__restrict__ stages) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert((2 * WAVES + 2) * 4 <= WB_LDS_CTL_BYTES, "control words");
    static_assert(!BAKED || WAVES == WB_JIT_WAVES, "the prelude's wave count");
    constexpr int NT = WAVES * 64;
    constexpr int TR = RPW * WAVES;
    constexpr int SD = WB_STAGE_DWORDS(D);
    constexpr int G = GroupSize<D>::G;
    constexpr int S0 = 8;                                  // stages in phase A (multiple of G; 4 and 12 measured slower)

    // (values that are the same in every lane of a wave but derived from threadIdx or read from LDS are passed
    // through readfirstlane: the compiler then keeps them -- and every count, bound and branch computed from them --
    // in scalar registers; left as "per-lane" values they turned the queue loops below into vector code with
    // exec-mask branches: 471 of a wave's 1251 vector instructions were in the segments, as many as in phase A)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WbTile tile_d = a.tiles[blockIdx.x];
    const WbLevel L = a.levels[tile_d.level];
    const int b = blockIdx.y;
    const int pitch = BAKED ? WB_JIT_PITCH : a.lds_pitch, rows = BAKED ? WB_JIT_ROWS : a.lds_rows;
    const int T = BAKED ? WB_JIT_T : a.T;
    const int nC = BAKED ? WB_JIT_C : a.C;

    float *tile = reinterpret_cast<float *>(smem);
    // float32 channels: planar float tile [C][rows][pitch]; uint8 channels: the pixels as they are, [rows][pitch][C] bytes
    constexpr bool U8 = EB != 0;                            // a byte tile (8- or 16-bit elements)
    const size_t tile_bytes = wb_lds_tile_bytes(EB, nC, rows, pitch);
    const int px_stride = EB ? nC * EB : 4;                 // bytes between horizontally adjacent windows' origins
    constexpr int QCAP = wb_casc_qcap(TR, WAVES);           // entries of the survivor queue
    constexpr bool QPREFIX = QCAP < TR * 64;                // capped: a wave's queue starts at the count of the waves before it
    uint2 *queue = reinterpret_cast<uint2 *>(smem + tile_bytes) + (QPREFIX ? 0 : wave * (RPW * 64));
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + tile_bytes + (size_t)QCAP * 8);

    const int nr = L.u - a.m > 0 ? L.u - a.m : 0;          // window grid
    const int nc = L.v - a.n > 0 ? L.v - a.n : 0;
    const int r0 = tile_d.ty * TR, c0 = tile_d.tx * WB_CASC_TC;

    // LDS mirror of the stage table (when it is small enough): the tail reads one record per lane
    const size_t stab_off = wb_lds_stab_off(EB, nC, rows, pitch, TR, T, WAVES);
    int4 *stab = reinterpret_cast<int4 *>(smem + stab_off);
    // control words behind the mirror: the per-wave counts exchanged at stage 8 and (their own words) at stage 16
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + stab_off + (size_t)(BAKED ? WB_JIT_LDS_STAGES : a.lds_stages) * SD * 4);
    uint32_t *wcnt2 = wcnt + WAVES;
    uint32_t *ticket = wcnt2 + WAVES;           // next unclaimed entry of the workgroup's survivor list (the stage-parallel tail)
    uint32_t *arrived = ticket + 1;             // waves that have finished (the last one flushes the tile's statistics)
    if (tid == 0) {
        *ticket = 0u;                           // (visible behind the tile load's barrier)
        *arrived = 0u;
    }
    // (nothing of this step reads the octaves' min / max keys any more -- the channel kernel has finished: the first
    // workgroup resets them for the next step's octave kernel, which then needs no memset launch in front of it)
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = tid; i < a.zero_words; i += NT) a.zero[i] = 0u;
    if constexpr (BAKED) {
        // the specialised stages address LDS by number: the dynamic region must start at LDS address 0
        if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    }
    WB_STAMP(0);
    for (int t = tid; t < T; t += NT) hist[t] = 0;
    // the stage mirror: its first NT vectors are REQUESTED here and stored behind the tile (as a loop of its own in front
    // of the tile loads it put one more memory round trip on every workgroup's way to the first barrier)
    const int n_stab = (BAKED ? WB_JIT_LDS_STAGES : a.lds_stages) * (SD / 4);
    int4 stab_mine = make_int4(0, 0, 0, 0);
    if (tid < n_stab) stab_mine = reinterpret_cast<const int4 *>(stages)[tid];

    // ---- stage the channel block into LDS (planar [C][rows][pitch])
    const float *chn = reinterpret_cast<const float *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    const uint8_t *chn8 = reinterpret_cast<const uint8_t *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    if (a.dbg & 1) {
    } else if (EB == 2 && a.C == 4 && (pitch & 1) == 0) {
        // 16-bit ranks, 8 bytes per pixel: 16 bytes = TWO pixels per lane, one 16-byte LDS write (as below, a group may read
        // one pixel past the end of a level row: the buffers carry 16 spare elements)
        constexpr int U = 2;
        const int ngrp = (WB_CASC_TC + a.n - 1 + 1) >> 1;            // 2-pixel groups per tile row (<= pitch / 2)
        const int total = rows * ngrp;
        const uint32_t m_ngrp = 0xFFFFFFFFu / (uint32_t)ngrp + 1u;
        struct __attribute__((aligned(8))) Px2 { uint32_t x, y, z, w; };      // two pixels, 8-byte aligned only
        const uint16_t *chn16 = reinterpret_cast<const uint16_t *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
        uint32_t *tile32 = reinterpret_cast<uint32_t *>(smem);
        for (int e0 = tid; e0 < total; e0 += NT * U) {
            Px2 v[U];
            int dst[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                uint32_t e = (uint32_t)(e0 + k * NT);
                e = e < (uint32_t)total ? e : (uint32_t)total - 1u;  // duplicates rewrite the same values
                const uint32_t row = __umulhi(e, m_ngrp), grp = e - row * (uint32_t)ngrp;
                int gr = r0 + (int)row, gc = c0 + 2 * (int)grp;
                gr = gr < L.u ? gr : L.u - 1;
                gc = gc < L.v ? gc : L.v - 1;
                v[k] = *reinterpret_cast<const Px2 *>(chn16 + ((int64_t)gr * L.v + gc) * 4);
                dst[k] = (int)(2u * (row * (uint32_t)pitch + 2u * grp));       // dwords: two per pixel
            }
#pragma unroll
            for (int k = 0; k < U; ++k) *reinterpret_cast<uint4 *>(tile32 + dst[k]) = make_uint4(v[k].x, v[k].y, v[k].z, v[k].w);
        }
    } else if (EB == 2) {
        // 16-bit elements of any channel count: the tile [rows][pitch][C], element by element
        uint16_t *tile16 = reinterpret_cast<uint16_t *>(smem);
        const uint16_t *chn16 = reinterpret_cast<const uint16_t *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
        const int total = a.C * rows * pitch;
        for (int idx = tid; idx < total; idx += NT) {
            const int ch = idx % a.C;
            const int rc = idx / a.C;
            const int col = rc % pitch, row = rc / pitch;
            const int gr = r0 + row, gc = c0 + col;
            uint16_t v = 0;
            if (gr < L.u && gc < L.v) v = chn16[((int64_t)gr * L.v + gc) * a.C + ch];
            tile16[idx] = v;
        }
    } else if (U8 && a.C == 4 && (pitch & 3) == 0) {
        // uint8 channels, one dword per pixel, kept as they are: the tile is [rows][pitch] dwords, a quarter of
        // the float tile (twice the workgroups per CU), loaded 16 bytes = FOUR pixels per lane, stored with one
        // 16-byte LDS write.  A group may read up to 12 bytes past the end of a level row (the buffers carry
        // 16 spare bytes); those pixels land in columns no window of the level reads.
        constexpr int U = 2;
        const int ngrp = (WB_CASC_TC + a.n - 1 + 3) >> 2;            // 4-pixel groups per tile row (<= pitch / 4)
        const int total = rows * ngrp;
        const uint32_t m_ngrp = 0xFFFFFFFFu / (uint32_t)ngrp + 1u;
        struct __attribute__((aligned(4))) Px4 { uint32_t x, y, z, w; };      // four pixels, dword-aligned only
        uint32_t *tile32 = reinterpret_cast<uint32_t *>(smem);
        for (int e0 = tid; e0 < total; e0 += NT * U) {
            Px4 v[U];
            int dst[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                uint32_t e = (uint32_t)(e0 + k * NT);
                e = e < (uint32_t)total ? e : (uint32_t)total - 1u;  // duplicates rewrite the same values
                const uint32_t row = __umulhi(e, m_ngrp), grp = e - row * (uint32_t)ngrp;
                int gr = r0 + (int)row, gc = c0 + 4 * (int)grp;
                gr = gr < L.u ? gr : L.u - 1;
                gc = gc < L.v ? gc : L.v - 1;
                v[k] = *reinterpret_cast<const Px4 *>(chn8 + ((int64_t)gr * L.v + gc) * 4);
                dst[k] = (int)(row * (uint32_t)pitch + 4u * grp);
            }
#pragma unroll
            for (int k = 0; k < U; ++k) *reinterpret_cast<uint4 *>(tile32 + dst[k]) = make_uint4(v[k].x, v[k].y, v[k].z, v[k].w);
        }
    } else if (U8) {
        // uint8 channels of any count: the byte tile [rows][pitch][C], element by element
        uint8_t *tile8 = reinterpret_cast<uint8_t *>(smem);
        const int total = a.C * rows * pitch;
        for (int idx = tid; idx < total; idx += NT) {
            const int ch = idx % a.C;
            const int rc = idx / a.C;
            const int col = rc % pitch, row = rc / pitch;
            const int gr = r0 + row, gc = c0 + col;
            uint8_t v = 0;
            if (gr < L.u && gc < L.v) v = chn8[((int64_t)gr * L.v + gc) * a.C + ch];
            tile8[idx] = v;
        }
    } else if (a.C == 4) {
        // Channels live in HBM as one float4 per pixel ([u][v][4]): a tile row is ONE contiguous
        // run of (64+n-1)*16 bytes.  Each thread loads U pixels back to back (straight-line code:
        // a branch around a load or a store makes the compiler sink each load next to its use and
        // wait for it alone), then scatters each pixel's 4 values to the 4 LDS planes (consecutive
        // lanes -> consecutive LDS addresses in every plane).  Out-of-level pixels receive some
        // other valid pixel of the level -- no existing window reads them -- and elements past
        // the end of the tile land in a spare slot behind each plane's last row.
        constexpr int U = 8;
        const int ncol = WB_CASC_TC + a.n - 1;               // pixels per tile row
        const int total = rows * ncol;
        const uint32_t m_ncol = 0xFFFFFFFFu / (uint32_t)ncol + 1u;   // exact e / ncol for e < 2^16
        const float4 *src = reinterpret_cast<const float4 *>(chn);
        const int plane = rows * pitch;
        for (int e0 = tid; e0 < total; e0 += NT * U) {
            float4 v[U];
            int dst[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                uint32_t e = (uint32_t)(e0 + k * NT);
                const bool in = e < (uint32_t)total;
                e = in ? e : (uint32_t)total - 1u;
                uint32_t row = __umulhi(e, m_ncol), col = e - row * (uint32_t)ncol;
                int gr = r0 + (int)row, gc = c0 + (int)col;
                gr = gr < L.u ? gr : L.u - 1;
                gc = gc < L.v ? gc : L.v - 1;
                v[k] = src[(int64_t)gr * L.v + gc];
                dst[k] = in ? (int)(row * (uint32_t)pitch + col) : plane - 1;   // last pad column of the last row: never read
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                tile[dst[k]] = v[k].x;
                tile[dst[k] + plane] = v[k].y;
                tile[dst[k] + 2 * plane] = v[k].z;
                tile[dst[k] + 3 * plane] = v[k].w;
            }
        }
    } else {  // any other channel count (caller-supplied arrays): generic element loop
        const int total = a.C * rows * pitch;
        for (int idx = tid; idx < total; idx += NT) {
            int ch = idx % a.C;
            int rc = idx / a.C;
            int col = rc % pitch, row = rc / pitch;
            int gr = r0 + row, gc = c0 + col;
            float v = 0.f;
            if (gr < L.u && gc < L.v) {
                const int64_t at = ((int64_t)gr * L.v + gc) * a.C + ch;
                v = chn[at];
            }
            tile[(ch * rows + row) * pitch + col] = v;
        }
    }
    if (tid < n_stab) stab[tid] = stab_mine;
    for (int i = tid + NT; i < n_stab; i += NT) stab[i] = reinterpret_cast<const int4 *>(stages)[i];
    __syncthreads();
    WB_STAMP(1);
    if (a.dbg & 2) return;

    // ---- phase A: RPW windows per lane through stages [0, S0)
    float hs[RPW];
    unsigned long long lm[RPW];          // liveness of the 64 windows of row j as a lane mask: the bookkeeping
    int base[RPW];                       // (counts, rejection) is then scalar work, not per-lane VALU
    const int wr = wave * RPW;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        hs[j] = 0.f;
        lm[j] = __ballot((c0 + lane < nc) && (r0 + wr + j < nr));
        base[j] = ((wr + j) * pitch + lane) * px_stride;
    }
    const int tA = T < S0 ? T : S0;
    static_assert(S0 <= 64, "one lane per phase-A stage");
    uint32_t entered = 0;                 // windows of this wave entering stage `lane` (phase A)
    // FULL: the cascade has at least S0 stages (the usual case) -- phase A is then straight-line code, no per-stage
    // bound checks: the scheduler is free to request a stage's record while the previous stage is being evaluated
    auto phase_a = [&](auto full_tag) WB_INLINE_LAMBDA {
        constexpr bool FULL = decltype(full_tag)::value;
        if constexpr (BAKED) {
            wb_static_for<0, S0, G>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                if (!FULL && t >= tA) return false;
                float p[G][RPW];
                wb_static_for<0, G, 1>([&](auto gg) {
                    constexpr int g = decltype(gg)::value;
#pragma unroll
                    for (int j = 0; j < RPW; ++j) p[g][j] = StageAt<D, EB, TR, t + g>::eval(tile, base[j]);
                    return true;
                });
                bool more = true;
                wb_static_for<0, G, 1>([&](auto gg) {
                    constexpr int g = decltype(gg)::value;
                    if (!FULL && t + g >= tA) return more = false;
                    int cnt = 0;
#pragma unroll
                    for (int j = 0; j < RPW; ++j) cnt += __popcll(lm[j]);
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(entered) : "s"(cnt), "n"(t + g));
                    const float theta = StageAt<D, EB, TR, t + g>::theta();
                    const unsigned long long never = never_rejects<true>(theta) ? ~0ull : 0ull;
#pragma unroll
                    for (int j = 0; j < RPW; ++j) {
                        hs[j] = hs[j] + p[g][j];
                        lm[j] &= __ballot(hs[j] >= theta) | never;
                    }
                    return true;
                });
                return more;
            });
            return;
        }
#pragma unroll
        for (int t = 0; t < S0; t += G) {
            if (!FULL && t >= tA) break;
            Stage<D> st[G];
            const int32_t *sp = stages + (size_t)t * SD;
#pragma unroll
            for (int g = 0; g < G; ++g) st[g].load(sp + g * SD);
            float p[G][RPW];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < RPW; ++j) p[g][j] = st[g].template eval<EB>(tile, base[j]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!FULL && t + g >= tA) break;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < RPW; ++j) cnt += __popcll(lm[j]);
                // lane t keeps stage t's count (one LDS atomic per wave after the phase): both operands are scalars, so
                // this is ONE v_writelane instead of a move, a compare and a select
                // (the loops are fully unrolled: the lane index is an immediate)
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(entered) : "s"(cnt), "n"(t + g));
                // theta == -inf never rejects (a NaN sum would fail `>=`): folded into the mask, not a branch,
                // so the RPW rows stay in one basic block and share the stage's constants in registers
                // (the bits of the scalar theta compared as an integer: a scalar compare; as a float compare it was
                // a vector instruction per stage)
                const unsigned long long never = never_rejects<BAKED>(st[g].theta) ? ~0ull : 0ull;
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    hs[j] = hs[j] + p[g][j];                      // (a dead window's sum is never read again)
                    lm[j] &= __ballot(hs[j] >= st[g].theta) | never;
                }
            }
        }
    };
    if (tA == S0)
        phase_a(WbTag<true>{});
    else
        phase_a(WbTag<false>{});
    if (entered) atomicAdd(&hist[lane], entered);
    if (a.dbg & 4) return;

    // ---- survivors of phase A.  If more stages follow, the survivors of the whole workgroup are
    //      pooled: by stage 8 a wave keeps only a fraction of its windows (half-empty chunks in
    //      every wave); pooled, they fill whole chunks of 64 for a few waves and the others are
    //      done.  (A second pooling at stage 16 was measured slower.)
    const uint32_t shard = blockIdx.x % WB_DET_SHARDS;
    int my_cnt = 0;
#pragma unroll
    for (int j = 0; j < RPW; ++j) my_cnt += __popcll(lm[j]);
    uint32_t total = (uint32_t)my_cnt, before = 0;
    bool pooled = false;
    uint2 *wgq = reinterpret_cast<uint2 *>(smem + tile_bytes);
    int t_done = tA;                                              // stages evaluated so far (densely)
    // the workgroup's survivors: total, and how many sit in the waves before this one
    auto recount = [&]() WB_INLINE_LAMBDA {
        if (lane == 0) wcnt[wave] = (uint32_t)my_cnt;
        __syncthreads();
        total = 0;
        before = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            uint32_t c = wcnt[w];
            if (w < wave) before += c;
            total += c;
        }
        total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
        before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    };
    // stages [t0, t1) on the dense state (hs, lm), records through the scalar cache: the capped queue's fallback.  ONE stage
    // at a time (a group of G records in scalar registers beside phase A's state spilled scalar registers into the hot
    // path: +4 % on every tile for the sake of the rare one that comes here)
    auto dense_more = [&](int t0, int t1) WB_INLINE_LAMBDA {
#pragma nounroll
        for (int t = t0; t < t1; ++t) {
            Stage<D> st;
            st.load(stages + (size_t)__builtin_amdgcn_readfirstlane(t) * SD);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < RPW; ++j) cnt += __popcll(lm[j]);
            if (lane == 0 && cnt) atomicAdd(&hist[t], (uint32_t)cnt);
            const unsigned long long never = never_rejects<BAKED>(st.theta) ? ~0ull : 0ull;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                hs[j] = hs[j] + st.template eval<EB>(tile, base[j]);
                lm[j] &= __ballot(hs[j] >= st.theta) | never;
            }
        }
    };
    if (T > t_done) {
        recount();
        if constexpr (QPREFIX) {
            // more survivors than the queue holds (rare: a tile that a cascade hardly thins out): on densely, eight stages
            // at a time, until they fit -- or the cascade ends
            while (total > (uint32_t)QCAP) {
                const int t1 = t_done + S0 < T ? t_done + S0 : T;
                dense_more(t_done, t1);
                t_done = t1;
                my_cnt = 0;
#pragma unroll
                for (int j = 0; j < RPW; ++j) my_cnt += __popcll(lm[j]);
                __syncthreads();                                  // every wave has read wcnt
                if (t_done >= T) break;
                recount();
            }
        }
        pooled = t_done < T && total <= 64u * WAVES;              // same decision in every wave
    }
    int n_q = my_cnt;
    if (QPREFIX && t_done >= T) {
        // every stage has been evaluated on the dense state (a cascade of at most S0 stages, or a tile whose survivors never
        // fitted the queue): the windows still alive are detections -- one returning atomic per wave, records straight out
        if (my_cnt > 0) {
            uint32_t s0 = 0;
            if (lane == 0) s0 = atomicAdd(a.det_count + shard, (uint32_t)my_cnt);
            s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0);
            WbDet *dst = a.det + (size_t)shard * a.det_cap;
            uint32_t n_loc = 0;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const unsigned long long mask = lm[j];
                if ((mask >> lane) & 1ull) {
                    const uint32_t at = s0 + n_loc + (uint32_t)lane_rank(mask);
                    if (at < a.det_cap) {
                        WbDet d;
                        d.image = b;
                        d.level = tile_d.level;
                        d.r = (uint16_t)(r0 + wr + j);
                        d.c = (uint16_t)(c0 + lane);
                        d.score = hs[j];
                        dst[at] = d;
                    }
                }
                n_loc += (uint32_t)__popcll(mask);
            }
        }
        n_q = 0;
    } else {
        // (capped queue: every wave's entries start at the count of the waves before it, pooled or not)
        uint2 *dstq = (pooled || QPREFIX) ? wgq + before : queue;
        if (QPREFIX && !pooled) queue = wgq + before;
        int n_loc = 0;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const unsigned long long mask = lm[j];
            int cnt = __popcll(mask);
            if (cnt == 0) continue;
            if ((mask >> lane) & 1ull)
                dstq[n_loc + lane_rank(mask)] = make_uint2((uint32_t)((wr + j) * 64 + lane), __float_as_uint(hs[j]));
            n_loc += cnt;
        }
    }
    bool scatter = false;
    const uint2 *dyn_list = nullptr;                              // scatter: the workgroup's shared survivor list ...
    int dyn_total = 0;                                            // ... and its length
    if (T > t_done) {
        __syncthreads();                                          // pooled entries visible; wcnt free again
        if (pooled) {
            // Few survivors in the whole tile (the usual case for a rejecting cascade: a few dozen of 2048):
            // deal them out one by one to ALL waves, which take them straight to the stage-parallel evaluator
            // below -- every wave works, instead of one or two waves walking ~100 stages in groups of G while
            // the others wait at the final barrier.  Many survivors: whole chunks of 64 per wave, as dense
            // wave-synchronous segments.
            scatter = total <= (uint32_t)a.spar_wg;
            if (scatter) {
                dyn_list = wgq;                                       // (claimed entry by entry in the tail below)
                dyn_total = (int)total;
                n_q = 0;
            } else {
                queue = wgq + 64 * wave;
                int left = (int)total - 64 * wave;
                n_q = left < 0 ? 0 : (left > 64 ? 64 : left);
            }
        }
    }
    n_q = __builtin_amdgcn_readfirstlane(n_q);
    WB_STAMP(2);
    WB_STAMP(3);

    // ---- phase B: dense re-packed survivors, stage segments [S0,2S0), [2S0,4S0), ...
    int t_begin = t_done;
    // wave-synchronous segments from t_begin up to (at most) t_stop, compacting after each
    // (a specialised build of depth-3 trees, or of a long cascade, keeps every register it may have full of constants; the
    // two lane addresses the segments use -- queue entry and counter word -- were the values its register allocator then
    // sent to scratch memory, at any occupancy -- and a specialised kernel is meant to live in registers and LDS (wb_jit.hip:
    // build_checked).  An opaque copy of the lane index makes them cheap to recompute instead.  The 128-stage depth-2
    // kernel never spilled them and keeps its code as it was)
    auto relane = [&]() {
        int l = lane;
#ifdef WB_JIT_BAKED
        if constexpr (BAKED && (D >= 3 || WB_JIT_T > 256)) asm volatile("" : "+v"(l));
#endif
        return l;
    };
    auto run_segments = [&](int t_stop) WB_INLINE_LAMBDA {
        while (t_begin < t_stop && n_q > 0) {
            // few windows left: the stage-parallel tail is cheaper than walking groups of G
            if ((t_begin >= a.spar[0] && n_q <= a.spar[1]) || (t_begin >= a.spar[2] && n_q <= a.spar[3])) break;
            int t_end = 2 * t_begin < t_stop ? 2 * t_begin : t_stop;
            t_end = t_end < t_begin + 64 ? t_end : t_begin + 64;          // one counter lane per stage of the segment
            int n_out = 0;
            uint32_t entered_b = 0;       // windows entering stage t_begin + lane, over all chunks: one LDS atomic per segment
            // the chunks of the queue through the segment's stages.  seg: WbSeg<false> -- the stage range is a run-time
            // value and the records come in through the scalar cache; WbSeg<true, TB, TE> (BAKED builds) -- this very
            // segment [TB, TE) unrolled, its records compile-time constants
            auto chunk_loop = [&](auto seg) WB_INLINE_LAMBDA {
                using Seg = decltype(seg);
                for (int qb = 0; qb < n_q; qb += 64) {
                    int i = qb + relane();
                    const bool mine = i < n_q;
                    unsigned long long am = __ballot(mine);                    // alive lanes of this chunk, as a mask
                    uint2 e = mine ? queue[i] : make_uint2(0u, 0u);
                    int pos = (int)e.x;
                    float h = __uint_as_float(e.y);
                    int wbase = ((pos >> 6) * pitch + (pos & 63)) * px_stride;
                    uint32_t ent_c = 0;           // this chunk's windows entering stage t_begin + lane
                    if constexpr (Seg::baked) {
                        constexpr int TB = Seg::tb, TE = Seg::te;
                        // (compile-time recursion instead of `#pragma unroll`: the optimiser declines to unroll a long
                        // loop with an early exit, and every index below must be a constant)
                        // (software-pipelined: the gathers of group t + G are issued before group t's sums and rejection
                        // tests -- the early exit between groups otherwise puts a full LDS round trip in front of every group)
                        float pc[G];
                        wb_static_for<0, G, 1>([&](auto gg) {
                            constexpr int g = decltype(gg)::value;
                            pc[g] = StageAt<D, EB, TR, TB + g>::eval(tile, wbase);
                            return true;
                        });
                        wb_static_for<TB, TE, G>([&](auto tt) {
                            constexpr int t = decltype(tt)::value;
                            float pn[G];
                            wb_static_for<0, G, 1>([&](auto gg) {
                                constexpr int g = decltype(gg)::value;
                                if constexpr (WB_SEG_PREFETCH && t + G < TE)
                                    pn[g] = StageAt<D, EB, TR, t + G + g>::eval(tile, wbase);
                                else
                                    pn[g] = 0.0f;
                                return true;
                            });
                            wb_static_for<0, G, 1>([&](auto gg) {
                                constexpr int g = decltype(gg)::value;
                                if constexpr (t + g < TE) {
                                    const int cnt = __popcll(am);
                                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(ent_c) : "s"(cnt), "n"(t + g - TB));
                                    const float theta = StageAt<D, EB, TR, t + g>::theta();
                                    h = h + pc[g];
                                    am &= __ballot(h >= theta) | (never_rejects<true>(theta) ? ~0ull : 0ull);
                                }
                                return true;
                            });
                            if (am == 0ull) return false;
                            wb_static_for<0, G, 1>([&](auto gg) {
                                constexpr int g = decltype(gg)::value;
                                if constexpr (WB_SEG_PREFETCH)
                                    pc[g] = pn[g];
                                else if constexpr (t + G < TE)
                                    pc[g] = StageAt<D, EB, TR, t + G + g>::eval(tile, wbase);
                                return true;
                            });
                            return true;
                        });
                    } else {
                        for (int t = t_begin; t < t_end; t += G) {
                            if (am == 0ull) break;
                            Stage<D> st[G];
                            const int32_t *sp = stages + (size_t)__builtin_amdgcn_readfirstlane(t) * SD;
#pragma unroll
                            for (int g = 0; g < G; ++g) st[g].load(sp + g * SD);
                            float p[G];
#pragma unroll
                            for (int g = 0; g < G; ++g) p[g] = st[g].template eval<EB>(tile, wbase);
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                if (t + g >= t_end) break;
                                // the chunk's count for stage t + g goes into lane (t + g - t_begin) of ent_c with ONE v_writelane
                                // (value and lane index are both scalars; two scalar operands exceed the constant-bus limit, so
                                // the index travels in m0) instead of a move, a compare and a select
                                const int cnt = __popcll(am);
                                if constexpr (BAKED)        // (a specialised build only lands here for a segment off its list: plain code)
                                    ent_c = lane == t + g - t_begin ? (uint32_t)cnt : ent_c;
                                else
                                    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(ent_c) : "s"(cnt), "s"(t + g - t_begin) : "m0");
                                h = h + p[g];                        // (a dead window's sum is never read again)
                                am &= __ballot(h >= st[g].theta) | (never_rejects<BAKED>(st[g].theta) ? ~0ull : 0ull);
                            }
                        }
                    }
                    entered_b += ent_c;
                    int cnt = __popcll(am);
                    if (cnt) {
                        // in place: n_out + rank <= qb + lane, and this wave already holds chunk qb in registers
                        if ((am >> lane) & 1ull) queue[n_out + lane_rank(am)] = make_uint2((uint32_t)pos, __float_as_uint(h));
                        n_out += cnt;
                    }
                }
            };
            bool handled = false;
            if constexpr (BAKED) {
#define WB_X(TB, TE)                                             \
    if (!handled && t_begin == (TB) && t_end == (TE)) {          \
        chunk_loop(WbSeg<true, (TB), (TE)>{});                   \
        handled = true;                                          \
    }
                WB_JIT_SEGMENTS(WB_X)
#undef WB_X
            }
            if (!handled) chunk_loop(WbSeg<false, 0, 0>{});
            if (entered_b) atomicAdd(&hist[t_begin + relane()], entered_b);
            n_q = n_out;
            t_begin = t_end;
        }
    };

    static_assert(S0 == 8, "the pooling above assumes phase A ends at stage 8");
    if (a.dbg & 8) return;
    if (!scatter) {
        // After the segment [8, 16) the tile usually holds a few dozen windows in a few sparse chunks, with ~100
        // stages to go: count them workgroup-wide once more and, if they are few, deal them out to all waves for
        // the stage-parallel evaluator (as above after phase A).
        constexpr int S1 = 2 * S0;
        if (RPW >= 2 && pooled && T > S1 && t_begin == S0) {
            run_segments(S1);
            if (lane == 0) wcnt2[wave] = (uint32_t)n_q;
            __syncthreads();
            uint32_t total2 = 0, before2 = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = wcnt2[w];
                if (w < wave) before2 += c;
                total2 += c;
            }
            total2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)total2);
            before2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)before2);
            const uint32_t room = (uint32_t)QCAP - 64u * WAVES;         // queue entries behind the pooled chunks
            if (total2 <= ((uint32_t)a.spar_wg < room ? (uint32_t)a.spar_wg : room)) {   // same decision in every wave
                uint2 *list2 = wgq + 64 * WAVES;                      // (the pooled chunks occupy the first 64 * WAVES entries)
                for (int i = lane; i < n_q; i += 64) list2[before2 + i] = queue[i];
                __syncthreads();
                scatter = true;
                dyn_list = list2;
                dyn_total = (int)total2;
                n_q = 0;
                t_begin = S1;
            }
        }
        if (!scatter) run_segments(T);
    }
    WB_STAMP(4);
    if (a.dbg & 16) return;

    // ---- stage-parallel tail: lane i evaluates stage rs + i of a window (two windows side by side)
    // this lane's record for stage rs + lane (the LDS mirror, or HBM when the table is too large for it)
    auto lane_stage = [&](int rs) WB_INLINE_LAMBDA {
        Stage<D> st;
        const int t = rs + lane;
        const int tt = t < T ? t : T - 1;
        int32_t rec[SD];
        if (BAKED ? WB_JIT_LDS_STAGES != 0 : a.lds_stages != 0) {
#pragma unroll
            for (int q = 0; q < SD / 4; ++q) {
                int4 v = stab[tt * (SD / 4) + q];
                rec[4 * q] = v.x; rec[4 * q + 1] = v.y; rec[4 * q + 2] = v.z; rec[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SD / 4; ++q) {
                int4 v = reinterpret_cast<const int4 *>(stages)[(size_t)tt * (SD / 4) + q];
                rec[4 * q] = v.x; rec[4 * q + 1] = v.y; rec[4 * q + 2] = v.z; rec[4 * q + 3] = v.w;
            }
        }
        st.load(rec);
        return st;
    };
    // W windows (origins pos[], scores h[] on entering stage rs) through the stages rs .. rs + nvalid - 1, one stage per
    // lane: rej[w] = a stage rejected it; else h[w] = its score behind the last of them.  entered += the number of
    // these windows that entered stage rs + lane.  A window's chain -- gathers, leaf select, the ripple -- is all
    // latency: two independent chains interleave in the same issue slots.
    auto pass = [&](auto w_tag, const Stage<D> &st, int nvalid, const int *pos, float *h, bool *rej, uint32_t &entered) WB_INLINE_LAMBDA {
        constexpr int W = decltype(w_tag)::value;
        float pk[W], hk[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int wbase = ((pos[w] >> 6) * pitch + (pos[w] & 63)) * px_stride;
            const float p = WB_TAIL_ALL ? st.template eval_all<EB>(tile, wbase) : st.template eval<EB>(tile, wbase);
            // Replay in stage order: lane k accumulates p_0 .. p_k one after the other -- the same
            // additions in the same order as the reference's running `hs +=` -- so it ends up
            // with the score the rejection test of stage rs+k sees.
            // (ripple through the wave with DPP wave_shr:1 -- lane k takes lane k-1's running sum
            // and adds its own p; after step j lanes 0..j are final and later steps recompute the
            // same value, so 63 steps settle every lane)
            // Lane 0 folds the incoming score into its addend (0 + x == x exactly), so the shifted-in
            // value of the out-of-range lane can be the DPP zero (bound_ctrl) and each step is ONE
            // v_add_f32 with a wave_shr:1 source.
            pk[w] = lane == 0 ? h[w] + p : p;
            hk[w] = pk[w];
        }
        // Most windows are rejected within a few stages, so the ripple runs in blocks of 8 steps and stops
        // at the first block whose settled lanes hold a rejection: the lowest such lane is the first
        // rejecting stage (every earlier stage is settled and passed).  (A window that is done keeps rippling
        // beside its partner: its settled lanes recompute the same sums, its verdict is frozen.)
        unsigned long long rmask[W];
        bool done[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            rmask[w] = 0ull;
            done[w] = false;
        }
        for (int settled = 1;;) {                            // lanes [0, settled) hold their final sums
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    int prev = __builtin_amdgcn_update_dpp(0, __float_as_int(hk[w]), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
                    hk[w] = __int_as_float(prev) + pk[w];
                }
            }
            settled += 8;
            const int upto = settled < nvalid ? settled : nvalid;
            bool all = true;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (!done[w]) {
                    rmask[w] = __ballot((lane < upto) && (st.theta != -INFINITY) && !(hk[w] >= st.theta));
                    done[w] = rmask[w] || settled >= nvalid;
                }
                all = all && done[w];
            }
            if (all) break;
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int last = rmask[w] ? (int)__builtin_ctzll(rmask[w]) : nvalid - 1;   // last stage entered
            entered += lane <= last ? 1u : 0u;                                         // (lanes >= nvalid never count: last < nvalid)
            rej[w] = rmask[w] != 0ull;
            h[w] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hk[w]), nvalid - 1));
        }
    };
    // The ripple is a chain of dependent vector instructions: on a SIMD that seven other waves keep busy it advances one
    // step per round of the issue arbiter -- a 64-stage pass then takes thousands of cycles while the rest of the
    // workgroup waits at the final barrier.  Raised priority puts the chain's next instruction in front (it issues few).
    __builtin_amdgcn_s_setprio(WB_TAIL_PRIO);
    if (scatter) {
        // The workgroup's few survivors sit in ONE list: every wave claims the next two entries (an LDS ticket) and
        // takes them through ALL remaining stages, then claims again -- the waves finish within one window of each other
        // whatever the survivors cost (dealt out in fixed shares, the wave with the longest-lived windows kept the other
        // seven waiting at the final barrier for a fifth of the workgroup's lifetime).  A window that passes the last
        // stage is appended to the output right here (one atomic each: they are rare).
        for (;;) {
            uint32_t i0 = 0;
            if (lane == 0) i0 = atomicAdd(ticket, 2u);
            const int i = __builtin_amdgcn_readfirstlane((int)i0);
            if (i >= dyn_total) break;
            const bool two = i + 1 < dyn_total;
            int pos[2];
            float h[2];
            bool alive[2] = {true, two};
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const uint2 e = dyn_list[two || w == 0 ? i + w : i];     // same entry in every lane
                pos[w] = __builtin_amdgcn_readfirstlane((int)e.x);
                h[w] = __uint_as_float(e.y);
            }
            for (int rs = t_begin; rs < T && (alive[0] || alive[1]); rs += 64) {
                const int nvalid = T - rs < 64 ? T - rs : 64;
                const Stage<D> st = lane_stage(rs);
                uint32_t entered_t = 0;
                bool rej[2] = {false, false};
                if (alive[0] && alive[1]) {
                    pass(WbInt<2>{}, st, nvalid, pos, h, rej, entered_t);
                } else {
                    const int k = alive[0] ? 0 : 1;
                    pass(WbInt<1>{}, st, nvalid, pos + k, h + k, rej + k, entered_t);
                }
                if (entered_t) atomicAdd(&hist[rs + lane], entered_t);
                alive[0] = alive[0] && !rej[0];
                alive[1] = alive[1] && !rej[1];
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                if (!alive[w]) continue;                              // wave-uniform
                if (lane == 0) {
                    const uint32_t slot = atomicAdd(a.det_count + shard, 1u);
                    if (slot < a.det_cap) {
                        WbDet d;
                        d.image = b;
                        d.level = tile_d.level;
                        d.r = (uint16_t)(r0 + (pos[w] >> 6));
                        d.c = (uint16_t)(c0 + (pos[w] & 63));
                        d.score = h[w];
                        a.det[(size_t)shard * a.det_cap + slot] = d;
                    }
                }
            }
        }
    } else {
        // a wave's own queue (it left the segments with a few windows): pass by pass, the survivors re-packed in place
        for (int rs = t_begin; rs < T && n_q > 0; rs += 64) {
            const int nvalid = T - rs < 64 ? T - rs : 64;
            const Stage<D> st = lane_stage(rs);
            int n_out = 0;
            uint32_t entered_t = 0;                              // windows that entered stage rs + lane in this pass
            auto windows = [&](auto w_tag, int i) WB_INLINE_LAMBDA {
                constexpr int W = decltype(w_tag)::value;
                int pos[W];
                float h[W];
                bool rej[W];
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const uint2 e = queue[i + w];                    // same entry in every lane
                    pos[w] = __builtin_amdgcn_readfirstlane((int)e.x);
                    h[w] = __uint_as_float(e.y);
                }
                pass(w_tag, st, nvalid, pos, h, rej, entered_t);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (rej[w]) continue;
                    if (lane == 0) queue[n_out] = make_uint2((uint32_t)pos[w], __float_as_uint(h[w]));   // n_out <= i + w: entries read above
                    ++n_out;
                }
            };
            int i = 0;
            if constexpr (WB_TAIL_W == 2)
                for (; i + 1 < n_q; i += 2) windows(WbInt<2>{}, i);
            for (; WB_TAIL_W != 2 && i + 1 < n_q; ++i) windows(WbInt<1>{}, i);
            if (i < n_q) windows(WbInt<1>{}, i);
            if (entered_t) atomicAdd(&hist[rs + lane], entered_t);
            n_q = n_out;
        }
    }

    __builtin_amdgcn_s_setprio(0);
    WB_STAMP(5);
    if (a.dbg & 32) return;
    // ---- epilogue: the wave queues now hold the windows alive after stage T-1.  Every wave that still holds some --
    //      few do -- reserves their slots in one of the sharded output buffers itself (one returning atomic per such
    //      wave, in flight across the barrier below) and copies its records out: no count exchange, no second barrier,
    //      nobody waits for another wave's atomic.  (The order of the records inside a shard was never defined.)
    uint32_t slot0 = 0;
    if (n_q > 0 && lane == 0) slot0 = atomicAdd(a.det_count + shard, (uint32_t)n_q);
    if (n_q > 0) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
        WbDet *dst = a.det + (size_t)shard * a.det_cap;
        for (int i = lane; i < n_q; i += 64) {
            uint2 e = queue[i];
            if (o + i < a.det_cap) {
                WbDet d;
                d.image = b;
                d.level = tile_d.level;
                d.r = (uint16_t)(r0 + ((int)e.x >> 6));
                d.c = (uint16_t)(c0 + ((int)e.x & 63));
                d.score = __uint_as_float(e.y);
                dst[o + i] = d;
            }
        }
    }
    WB_STAMP(6);
    // per-stage alive counts of this tile -> alive[image][level][stage]: one fire-and-forget atomic per stage the
    // tile reached, once per workgroup (the workgroup's waves have summed in LDS; an atomic per wave and stage made
    // every wave of every tile queue up behind the others').  Round 4: no barrier in front of it -- a wave that is done
    // says so (one LDS atomic; its additions to hist are in before it counts) and
    // LEAVES; the last one to arrive flushes the sums.  The waves that carried nothing through the late stages -- most of
    // them -- used to wait here for a fifth of the workgroup's lifetime holding their wave slots.
#if WB_CASC_END_BARRIER
    __syncthreads();                                          // (A/B build: the round-3 ending -- every wave waits, all flush)
    if (a.alive) {
        uint32_t *al = a.alive + ((int64_t)b * a.n_levels + tile_d.level) * a.T;
        for (int t = tid; t < T; t += NT) {
            const uint32_t c = hist[t];
            if (c) atomicAdd(al + t, c);
        }
    }
    return;
#endif
    uint32_t earlier = 0;
    if (a.dbg & 64) __syncthreads();          // diagnostic (WB_CASC_DBG=64): every wave stays until all are done, in the SAME binary
    // (LDS serves a wave's operations in order: this wave's additions to hist are performed before its arrival is; the
    // compiler must not move them across it either -- hence the two barriers around a plain atomic)
    asm volatile("" ::: "memory");
    if (lane == 0) earlier = atomicAdd(arrived, 1u);
    asm volatile("" ::: "memory");
    earlier = (uint32_t)__builtin_amdgcn_readfirstlane((int)earlier);
    if (earlier == (uint32_t)WAVES - 1u && a.alive) {
        uint32_t *al = a.alive + ((int64_t)b * a.n_levels + tile_d.level) * a.T;
        for (int t = lane; t < T; t += 64) {
            const uint32_t c = hist[t];
            if (c) atomicAdd(al + t, c);
        }
    }
    WB_STAMP(7);
}

}  // namespace
