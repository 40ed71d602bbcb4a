// Dense sliding-window evaluation of the WaldBoost decision-tree cascade on gfx950.
//
// Replaces reference model.py:216-259 (Model.predict_on_image: window grid, stage loop,
// rejection, compaction, n_loc/n_weak statistics) and training.py:84-96
// (DTree.predict_on_image: the tree walk on all alive windows).
//
// One workgroup owns a tile of TR x 64 windows of one level of one image:
//   * the (TR+m-1) x (64+n-1) x C channel block is staged once into LDS, planar, so that a
//     wavefront's 64 lanes (64 adjacent window columns) gather from 64 adjacent banks;
//   * wave-synchronous stages: every lane of a wave is at the same stage, so the stage records
//     come in through the scalar cache (s_load) and cost no vector memory or LDS traffic; stages
//     are evaluated in groups of G with all 2*G gathers in flight (only the fp32 accumulation
//     and the rejection tests are sequential);
//   * phase A runs the first stages with RPW windows per lane; survivors are compacted with
//     wave ballot + mbcnt into the wave's own LDS queue; phase B re-packs them densely for
//     geometrically growing stage segments, compacting in place after each segment, and pools
//     the survivors of the whole workgroup once, at stage 8, so that a few waves hold full
//     chunks of 64 and the rest retire;
//   * stage-parallel tail: once a wave is down to a handful of windows (about 1e-3 of the
//     windows of the benchmark cascade reach stage 32, with ~100 stages to go) the roles flip:
//     one window at a time, 64 stages AT ONCE, one stage per lane (per-lane stage records,
//     gathers from the same LDS tile); the fp32 accumulation and the rejection tests are then
//     replayed in stage order (a DPP wave_shr:1 ripple: lane i ends up with ((h+p_0)+p_1)+...+p_i,
//     exactly the reference's running sum) -- so a nearly empty wave no longer walks 100 stages serially
//     while the workgroup's LDS tile sits idle;
//   * the windows alive after the last stage stay in the wave's queue; one thread reserves room
//     for the whole workgroup with ONE atomic on one of WB_DET_SHARDS counters and the waves
//     copy their records out.
//
// Scores are accumulated in fp32 strictly in stage order and compared with `>=`, so they are
// bit-identical to the reference's `hs += ...; mask = hs >= theta`.
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "wb_common.h"

namespace {

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
    int dbg;                    // diagnostics (WB_CASC_DBG): 1 = skip the tile load, 2 = stop after the load
};

__device__ inline float as_f(int32_t x) { return __int_as_float(x); }

// theta == -inf for a wave-uniform theta, decided on the SCALAR unit: the bits go through an opaque scalar register so
// the comparison stays an integer one (written as a float test -- or as a plain bit test, which the compiler turns
// back into a float test -- it was a vector compare per stage)
__device__ inline bool never_rejects(float theta) {
    int bits = __float_as_int(theta);
    asm volatile("" : "+s"(bits));
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
    template <bool BYTES> static __device__ inline bool goes_right(const char *t8, int at, float th) {
        if constexpr (BYTES) {
            const int v = *reinterpret_cast<const uint8_t *>(t8 + at);
            return !(v <= __float_as_int(th));
        } else {
            const float v = *reinterpret_cast<const float *>(t8 + at);
            return !(v <= th);                         // NaN goes right, like the reference's `<=`
        }
    }
    template <bool BYTES = false> __device__ inline float eval(const float *tile, int base) const {
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
template <int D, int RPW, int WAVES, bool U8>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_sgpr(80))) void cascade_tile_kernel(CascArgs a, const int32_t *__restrict__ stages) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t wcnt[WAVES];
    __shared__ uint32_t wcnt2[WAVES];    // the count exchange at stage 16 (its own words: a wave may already be writing wcnt for the epilogue)
    __shared__ uint32_t wg_base;
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
    const int pitch = a.lds_pitch, rows = a.lds_rows;
    const int T = a.T;

    float *tile = reinterpret_cast<float *>(smem);
    // float32 channels: planar float tile [C][rows][pitch]; uint8 channels: the pixels as they are, [rows][pitch][C] bytes
    const size_t tile_bytes = U8 ? (((size_t)a.C * rows * pitch + 15) & ~(size_t)15) : (size_t)a.C * rows * pitch * 4;
    const int px_stride = U8 ? a.C : 4;                      // bytes between horizontally adjacent windows' origins
    uint2 *queue = reinterpret_cast<uint2 *>(smem + tile_bytes) + wave * (RPW * 64);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + tile_bytes + (size_t)TR * 64 * 8);

    const int nr = L.u - a.m > 0 ? L.u - a.m : 0;          // window grid
    const int nc = L.v - a.n > 0 ? L.v - a.n : 0;
    const int r0 = tile_d.ty * TR, c0 = tile_d.tx * WB_CASC_TC;

    // LDS mirror of the stage table (when it is small enough): the tail reads one record per lane
    int4 *stab = reinterpret_cast<int4 *>(smem + ((tile_bytes + (size_t)TR * 64 * 8 + (size_t)T * 4 + 15) & ~(size_t)15));
    WB_STAMP(0);
    for (int t = tid; t < T; t += NT) hist[t] = 0;
    for (int i = tid; i < a.lds_stages * (SD / 4); i += NT) stab[i] = reinterpret_cast<const int4 *>(stages)[i];

    // ---- stage the channel block into LDS (planar [C][rows][pitch])
    const float *chn = reinterpret_cast<const float *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    const uint8_t *chn8 = reinterpret_cast<const uint8_t *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    if (a.dbg & 1) {
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
    auto phase_a = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
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
                for (int j = 0; j < RPW; ++j) p[g][j] = st[g].template eval<U8>(tile, base[j]);
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
                const unsigned long long never = never_rejects(st[g].theta) ? ~0ull : 0ull;
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    hs[j] = hs[j] + p[g][j];                      // (a dead window's sum is never read again)
                    lm[j] &= __ballot(hs[j] >= st[g].theta) | never;
                }
            }
        }
    };
    if (tA == S0)
        phase_a(std::true_type{});
    else
        phase_a(std::false_type{});
    if (entered) atomicAdd(&hist[lane], entered);
    if (a.dbg & 4) return;

    // ---- survivors of phase A.  If more stages follow, the survivors of the whole workgroup are
    //      pooled: by stage 8 a wave keeps only a fraction of its windows (half-empty chunks in
    //      every wave); pooled, they fill whole chunks of 64 for a few waves and the others are
    //      done.  (A second pooling at stage 16 was measured slower.)
    int my_cnt = 0;
#pragma unroll
    for (int j = 0; j < RPW; ++j) my_cnt += __popcll(lm[j]);
    uint32_t total = (uint32_t)my_cnt, before = 0;
    bool pooled = false;
    uint2 *wgq = reinterpret_cast<uint2 *>(smem + tile_bytes);
    if (T > S0) {
        if (lane == 0) wcnt[wave] = (uint32_t)my_cnt;
        __syncthreads();
        total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            uint32_t c = wcnt[w];
            if (w < wave) before += c;
            total += c;
        }
        total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
        before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
        pooled = total <= 64u * WAVES;                            // same decision in every wave
    }
    {
        uint2 *dstq = pooled ? wgq + before : queue;
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
    int n_q = my_cnt;
    int qs = 1;                                                   // stride of this wave's queue entries
    bool scatter = false;
    if (T > S0) {
        __syncthreads();                                          // pooled entries visible; wcnt free again
        if (pooled) {
            // Few survivors in the whole tile (the usual case for a rejecting cascade: a few dozen of 2048):
            // deal them out one by one to ALL waves, which take them straight to the stage-parallel evaluator
            // below -- every wave works, instead of one or two waves walking ~100 stages in groups of G while
            // the others wait at the final barrier.  Many survivors: whole chunks of 64 per wave, as dense
            // wave-synchronous segments.
            scatter = total <= (uint32_t)a.spar_wg;
            if (scatter) {
                queue = wgq + wave;
                qs = WAVES;
                n_q = (int)total > wave ? ((int)total - wave + WAVES - 1) / WAVES : 0;
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
    int t_begin = tA;
    // wave-synchronous segments from t_begin up to (at most) t_stop, compacting after each
    auto run_segments = [&](int t_stop) {
        while (t_begin < t_stop && n_q > 0) {
            // few windows left: the stage-parallel tail is cheaper than walking groups of G
            if ((t_begin >= a.spar[0] && n_q <= a.spar[1]) || (t_begin >= a.spar[2] && n_q <= a.spar[3])) break;
            int t_end = 2 * t_begin < t_stop ? 2 * t_begin : t_stop;
            t_end = t_end < t_begin + 64 ? t_end : t_begin + 64;          // one counter lane per stage of the segment
            int n_out = 0;
            uint32_t entered_b = 0;       // windows entering stage t_begin + lane, over all chunks: one LDS atomic per segment
            for (int qb = 0; qb < n_q; qb += 64) {
                int i = qb + lane;
                const bool mine = i < n_q;
                unsigned long long am = __ballot(mine);                    // alive lanes of this chunk, as a mask
                uint2 e = mine ? queue[i] : make_uint2(0u, 0u);
                int pos = (int)e.x;
                float h = __uint_as_float(e.y);
                int wbase = ((pos >> 6) * pitch + (pos & 63)) * px_stride;
                uint32_t ent_c = 0;           // this chunk's windows entering stage t_begin + lane
                for (int t = t_begin; t < t_end; t += G) {
                    if (am == 0ull) break;
                    Stage<D> st[G];
                    const int32_t *sp = stages + (size_t)__builtin_amdgcn_readfirstlane(t) * SD;
#pragma unroll
                    for (int g = 0; g < G; ++g) st[g].load(sp + g * SD);
                    float p[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) p[g] = st[g].template eval<U8>(tile, wbase);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (t + g >= t_end) break;
                        // the chunk's count for stage t + g goes into lane (t + g - t_begin) of ent_c with ONE v_writelane
                        // (value and lane index are both scalars; two scalar operands exceed the constant-bus limit, so
                        // the index travels in m0) instead of a move, a compare and a select
                        const int cnt = __popcll(am);
                        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(ent_c) : "s"(cnt), "s"(t + g - t_begin) : "m0");
                        h = h + p[g];                        // (a dead window's sum is never read again)
                        am &= __ballot(h >= st[g].theta) | (never_rejects(st[g].theta) ? ~0ull : 0ull);
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
            if (entered_b) atomicAdd(&hist[t_begin + lane], entered_b);
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
        if (RPW >= 2 && pooled && T > S1) {
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
            const uint32_t room = 64u * WAVES * (RPW - 1);             // queue entries behind the pooled chunks
            if (total2 <= ((uint32_t)a.spar_wg < room ? (uint32_t)a.spar_wg : room)) {   // same decision in every wave
                uint2 *list2 = wgq + 64 * WAVES;                      // (the pooled chunks occupy the first 64 * WAVES entries)
                for (int i = lane; i < n_q; i += 64) list2[before2 + i] = queue[i];
                __syncthreads();
                scatter = true;
                queue = list2 + wave;
                qs = WAVES;
                n_q = (int)total2 > wave ? ((int)total2 - wave + WAVES - 1) / WAVES : 0;
                t_begin = S1;
            }
        }
        if (!scatter) run_segments(T);
    }
    WB_STAMP(4);
    if (a.dbg & 16) return;

    // ---- stage-parallel tail: one window at a time, lane i evaluates stage rs+i
    for (int rs = t_begin; rs < T && n_q > 0; rs += 64) {
        const int t = rs + lane;
        const int nvalid = T - rs < 64 ? T - rs : 64;
        Stage<D> st;                                             // this lane's own stage
        {
            const int tt = t < T ? t : T - 1;
            int32_t rec[SD];
            if (a.lds_stages) {
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
        }
        int n_out = 0;
        uint32_t entered_t = 0;                                  // windows that entered stage rs + lane in this pass
        for (int i = 0; i < n_q; ++i) {
            const uint2 e = queue[i * qs];                       // same entry in every lane
            const int pos = __builtin_amdgcn_readfirstlane((int)e.x);
            const int wbase = ((pos >> 6) * pitch + (pos & 63)) * px_stride;
            const float p = st.template eval<U8>(tile, wbase);
            // Replay in stage order: lane k accumulates p_0 .. p_k one after the other -- the same
            // additions in the same order as the reference's running `hs +=` -- so it ends up
            // with the score the rejection test of stage rs+k sees.
            // (ripple through the wave with DPP wave_shr:1 -- lane k takes lane k-1's running sum
            // and adds its own p; after step j lanes 0..j are final and later steps recompute the
            // same value, so 63 steps settle every lane)
            // Lane 0 folds the incoming score into its addend (0 + x == x exactly), so the shifted-in
            // value of the out-of-range lane can be the DPP zero (bound_ctrl) and each step is ONE
            // v_add_f32 with a wave_shr:1 source.
            // Most windows are rejected within a few stages, so the ripple runs in blocks of 8 steps and stops
            // at the first block whose settled lanes hold a rejection: the lowest such lane is the first
            // rejecting stage (every earlier stage is settled and passed).
            const float h_in = __uint_as_float(e.y);
            const float pk = lane == 0 ? h_in + p : p;
            float hk = pk;
            unsigned long long rmask;
            for (int settled = 1;;) {                            // lanes [0, settled) hold their final sums
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int prev = __builtin_amdgcn_update_dpp(0, __float_as_int(hk), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
                    hk = __int_as_float(prev) + pk;
                }
                settled += 8;
                const int upto = settled < nvalid ? settled : nvalid;
                rmask = __ballot((lane < upto) && (st.theta != -INFINITY) && !(hk >= st.theta));
                if (rmask || settled >= nvalid) break;
            }
            const int last = rmask ? (int)__builtin_ctzll(rmask) : nvalid - 1;   // last stage entered
            entered_t += lane <= last ? 1u : 0u;
            if (!rmask) {
                float hl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hk), nvalid - 1));
                if (lane == 0) queue[n_out * qs] = make_uint2((uint32_t)pos, __float_as_uint(hl));   // n_out <= i
                ++n_out;
            }
        }
        if (entered_t) atomicAdd(&hist[t], entered_t);           // (lanes >= nvalid never count: last < nvalid)
        n_q = n_out;
    }

    WB_STAMP(5);
    if (a.dbg & 32) return;
    // ---- epilogue: the wave queues now hold the windows alive after stage T-1.  One atomic per
    //      workgroup reserves their slots in one of the sharded output buffers.
    if (lane == 0) wcnt[wave] = (uint32_t)n_q;
    __syncthreads();
    const uint32_t shard = blockIdx.x % WB_DET_SHARDS;
    if (tid == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) total += wcnt[w];
        wg_base = total ? atomicAdd(a.det_count + shard, total) : 0u;
    }
    // per-stage alive counts of this tile -> alive[image][level][stage]: one fire-and-forget atomic per stage the
    // tile reached, once per workgroup (the workgroup's waves have summed in LDS; an atomic per wave and stage made
    // every wave of every tile queue up behind the others')
    if (a.alive) {
        uint32_t *al = a.alive + ((int64_t)b * a.n_levels + tile_d.level) * a.T;
        for (int t = tid; t < T; t += NT) {
            const uint32_t c = hist[t];
            if (c) atomicAdd(al + t, c);
        }
    }
    __syncthreads();
    WB_STAMP(6);
    if (n_q > 0) {
        uint32_t o = wg_base;
        for (int w = 0; w < wave; ++w) o += wcnt[w];
        o = (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
        WbDet *dst = a.det + (size_t)shard * a.det_cap;
        for (int i = lane; i < n_q; i += 64) {
            uint2 e = queue[i * qs];
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
    WB_STAMP(7);
}

// -------------------------------------------------------------------------------------------
// Generic fallback for trees deeper than WB_CASC_MAX_DEPTH (or any shape): one thread per window,
// 4 x 64 windows per workgroup, the reference's flat node arrays walked as training.py:84-96 does,
// features gathered straight from HBM/L2.  Wave-synchronous in the stage index (dead lanes idle),
// so the per-stage alive counts are ballots; survivors leave through an LDS list and one sharded
// atomic per workgroup, like the tiled kernel.  Correctness fallback, not a tuned path.
struct GenArgs {
    const void *chn;
    int chn_u8;
    int64_t chn_stride;
    const WbLevel *levels;
    const WbTile *tiles;
    int n_levels, n_tiles;
    int T, m, n, C;
    const int32_t *node_off, *feat, *left, *right;
    const float *thr, *pred, *theta;
    WbDet *det;
    uint32_t *det_count;
    uint32_t det_cap;
    uint32_t *alive;
};

__global__ __launch_bounds__(256) void cascade_generic_kernel(GenArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(gsm);                       // T counters
    uint2 *list = reinterpret_cast<uint2 *>(gsm + (((size_t)a.T * 4 + 15) & ~(size_t)15));   // 256 entries
    __shared__ uint32_t n_list, base_slot;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WbTile tile_d = a.tiles[blockIdx.x];
    const WbLevel L = a.levels[tile_d.level];
    const int b = blockIdx.y;
    const int nr = L.u - a.m > 0 ? L.u - a.m : 0, nc = L.v - a.n > 0 ? L.v - a.n : 0;
    const int r = tile_d.ty * 4 + wave, c = tile_d.tx * 64 + lane;
    for (int t = tid; t < a.T; t += 256) hist[t] = 0;
    if (tid == 0) n_list = 0;
    __syncthreads();
    const float *chn = reinterpret_cast<const float *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    const uint8_t *chn8 = reinterpret_cast<const uint8_t *>(a.chn) + (int64_t)b * a.chn_stride + L.chn_off;
    bool alive = r < nr && c < nc;
    float h = 0.f;
    for (int t = 0; t < a.T; ++t) {
        int cnt = __popcll(__ballot(alive));
        if (cnt == 0) break;
        if (lane == 0) atomicAdd(&hist[t], (uint32_t)cnt);
        if (alive) {
            const int o = a.node_off[t], k = a.node_off[t + 1] - o;
            int node = 0;
            for (int step = 0; step < k; ++step) {                            // a walk visits a node at most once
                int l = a.left[o + node];
                if (l < 0) break;
                int f = a.feat[o + node];
                const int64_t at = ((int64_t)(r + (f & 255)) * L.v + (c + ((f >> 8) & 255))) * a.C + ((f >> 16) & 255);
                float v = a.chn_u8 ? (float)chn8[at] : chn[at];
                node = (v <= a.thr[o + node]) ? l : a.right[o + node];
            }
            h = h + a.pred[o + node];
            const float th = a.theta[t];
            alive = (th == -INFINITY) || (h >= th);
        }
    }
    if (alive) {
        uint32_t s = atomicAdd(&n_list, 1u);
        list[s] = make_uint2((uint32_t)(wave * 64 + lane), __float_as_uint(h));
    }
    __syncthreads();
    const uint32_t shard = blockIdx.x % WB_DET_SHARDS;
    if (tid == 0) base_slot = n_list ? atomicAdd(a.det_count + shard, n_list) : 0u;
    if (a.alive) {
        uint32_t *al = a.alive + ((int64_t)b * a.n_levels + tile_d.level) * a.T;
        for (int t = tid; t < a.T; t += 256)
            if (hist[t]) atomicAdd(al + t, hist[t]);
    }
    __syncthreads();
    if ((uint32_t)tid < n_list) {
        uint2 e = list[tid];
        uint32_t slot = base_slot + tid;
        if (slot < a.det_cap) {
            WbDet d;
            d.image = b;
            d.level = tile_d.level;
            d.r = (uint16_t)(tile_d.ty * 4 + (int)(e.x >> 6));
            d.c = (uint16_t)(tile_d.tx * 64 + (int)(e.x & 63));
            d.score = __uint_as_float(e.y);
            a.det[(size_t)shard * a.det_cap + slot] = d;
        }
    }
}

// -------------------------------------------------------------------------------------------
// DTree.predict_on_image on explicit window lists (reference training.py:84-96)
__global__ void tree_eval_kernel(const void *Xv, int x_u8, int u, int v, int C, const int32_t *rs, const int32_t *cs,
                                 int64_t n_pos, const uint8_t *feature, const float *threshold,
                                 const int8_t *left, const int8_t *right, const float *prediction,
                                 int n_nodes, float *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos) return;
    int r = rs[i], c = cs[i];
    int node = 0;
    for (int step = 0; step < n_nodes; ++step) {          // bounded: a walk visits each node at most once
        int l = left[node];
        if (l < 0) break;
        int fr = feature[node * 3 + 0], fc = feature[node * 3 + 1], ch = feature[node * 3 + 2];
        const int64_t at = ((int64_t)(r + fr) * v + (c + fc)) * C + ch;
        float val = x_u8 ? (float)reinterpret_cast<const uint8_t *>(Xv)[at] : reinterpret_cast<const float *>(Xv)[at];
        node = (val <= threshold[node]) ? l : (int)right[node];
    }
    out[i] = prediction[node];
}

// -------------------------------------------------------------------------------------------
// Training-time callers of the hot path (reference samples.py:14-43, model.py:181-214, training.py:73-83)

// gather_samples: one wave per sample copies its m x n x C crop, row by row (a crop row is n*C
// contiguous elements in X); VEC = elements moved per lane and step
template <typename E>
__global__ __launch_bounds__(64) void gather_samples_kernel(const E *X, int v, int rowlen, const int32_t *rs,
                                                            const int32_t *cs, int m, int xstride, E *out) {
    const int64_t i = blockIdx.x;
    const E *src = X + ((int64_t)rs[i] * v + cs[i]) * xstride;
    E *dst = out + i * (int64_t)m * rowlen;
    const int total = m * rowlen;
    for (int e = threadIdx.x; e < total; e += 64) {
        const int y = e / rowlen, x = e - y * rowlen;
        dst[e] = src[(int64_t)y * v * xstride + x];
    }
}

// Model.predict on samples: one thread per sample, the reference's flat node arrays
__global__ __launch_bounds__(256) void samples_predict_kernel(GenArgs a, int64_t n_samples, float *H, uint8_t *mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_samples) return;
    const int64_t base = i * (int64_t)a.m * a.n * a.C;
    const float *xf = reinterpret_cast<const float *>(a.chn) + base;
    const uint8_t *x8 = reinterpret_cast<const uint8_t *>(a.chn) + base;
    float h = 0.f;
    bool alive = true;
    for (int t = 0; t < a.T && alive; ++t) {
        const int o = a.node_off[t], k = a.node_off[t + 1] - o;
        int node = 0;
        for (int step = 0; step < k; ++step) {
            const int l = a.left[o + node];
            if (l < 0) break;
            const int f = a.feat[o + node];
            const int at = ((f & 255) * a.n + ((f >> 8) & 255)) * a.C + ((f >> 16) & 255);
            const float val = a.chn_u8 ? (float)x8[at] : xf[at];
            node = (val <= a.thr[o + node]) ? l : a.right[o + node];
        }
        h = h + a.pred[o + node];
        const float th = a.theta[t];
        alive = (th == -INFINITY) || (h >= th);
    }
    H[i] = alive ? h : -INFINITY;
    mask[i] = alive ? 1 : 0;
}

// DTree.apply on samples
__global__ __launch_bounds__(256) void tree_apply_kernel(const void *Xv, int x_u8, int64_t n_samples, int m, int n, int C,
                                                         const uint8_t *feature, const float *threshold,
                                                         const int8_t *left, const int8_t *right, int n_nodes,
                                                         int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_samples) return;
    const int64_t base = i * (int64_t)m * n * C;
    int node = 0;
    for (int step = 0; step < n_nodes; ++step) {
        const int l = left[node];
        if (l < 0) break;
        const int at = (feature[node * 3] * n + feature[node * 3 + 1]) * C + feature[node * 3 + 2];
        const float val = x_u8 ? (float)reinterpret_cast<const uint8_t *>(Xv)[base + at]
                               : reinterpret_cast<const float *>(Xv)[base + at];
        node = (val <= threshold[node]) ? l : (int)right[node];
    }
    out[i] = node;
}

// Model.get_boxes (reference model.py:136-147): [c, r, c+n, r+m] as fp32, times fp32(1/scale)
__global__ void boxes_kernel(const WbDet *det, int64_t n_det, const float *inv_scale, int m, int n,
                             float *boxes, float *scores) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_det) return;
    WbDet d = det[i];
    float s = inv_scale[d.level];
    float x1 = (float)d.c, y1 = (float)d.r;
    float x2 = (float)((int)d.c + n), y2 = (float)((int)d.r + m);
    reinterpret_cast<float4 *>(boxes)[i] = make_float4(x1 * s, y1 * s, x2 * s, y2 * s);
    scores[i] = d.score;
}

// The valid records of all detection shards, back to back behind a 4-word header -- what a host read-back or a
// collective wants: ONE contiguous prefix whose length the header gives.  One workgroup per shard; every workgroup
// reads all WB_DET_SHARDS counters (256 B) and derives its own output offset, so there is no second pass.
__global__ __launch_bounds__(256) void det_pack_kernel(const WbDet *det, const uint32_t *det_count, uint32_t cap,
                                                        int32_t *out, uint32_t out_cap) {
    static_assert(WB_DET_SHARDS == 64, "one counter per lane of a wave");
    const int shard = blockIdx.x, lane = threadIdx.x & 63;
    const uint32_t raw = det_count[lane];
    const uint32_t mine = raw < cap ? raw : cap;
    uint32_t before = 0, total = 0, worst = 0;
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mine, s);
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)raw, s);
        before += s < shard ? c : 0u;
        total += c;
        worst = r > worst ? r : worst;
    }
    const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)mine, shard);
    if (shard == 0 && threadIdx.x == 0) {
        out[0] = (int32_t)total;                              // valid records in all shards
        out[1] = (int32_t)worst;                              // fullest shard (> cap: records were dropped, scan again)
        out[2] = (int32_t)(total < out_cap ? total : out_cap);  // records present behind this header
        out[3] = (int32_t)cap;
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(det + (size_t)shard * cap);
    uint4 *dst = reinterpret_cast<uint4 *>(out) + 1;
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        if (before + i < out_cap) dst[before + i] = src[i];
}

// Model.detect's last step on the device (reference model.py:136-147 get_boxes, :173-179 the concatenated result):
// for every valid record of every shard, at its packed position i < out_cap,
//   keys[i]   = level << 54 | r << 40 | c << 26 | i     sorting these 64-bit words IS the reference order (level, r, c),
//                                                       and the low 26 bits say where the sorted record's box and score lie
//   boxes[i]  = (c, r, c + n, r + m) * fp32(1 / scale[level])   (boxes_kernel's arithmetic)
//   scores[i] = score
// behind det_pack_kernel's 4-word header, in ONE buffer: header | keys[out_cap] | boxes[out_cap] | scores[out_cap].
// The host reads it back with one copy, sorts the keys and gathers -- no per-field arithmetic on the host.
__global__ __launch_bounds__(256) void det_finish_kernel(const WbDet *det, const uint32_t *det_count, uint32_t cap,
                                                          const float *inv_scale, int m, int n, int32_t *out, uint32_t out_cap) {
    static_assert(WB_DET_SHARDS == 64, "one counter per lane of a wave");
    const int shard = blockIdx.x, lane = threadIdx.x & 63;
    const uint32_t raw = det_count[lane];
    const uint32_t mine = raw < cap ? raw : cap;
    uint32_t before = 0, total = 0, worst = 0;
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mine, s);
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)raw, s);
        before += s < shard ? c : 0u;
        total += c;
        worst = r > worst ? r : worst;
    }
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)mine, shard);
    if (shard == 0 && threadIdx.x == 0) {
        out[0] = (int32_t)total;
        out[1] = (int32_t)worst;
        out[2] = (int32_t)(total < out_cap ? total : out_cap);
        out[3] = (int32_t)cap;
    }
    const WbDet *src = det + (size_t)shard * cap;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(out + 4);
    float4 *boxes = reinterpret_cast<float4 *>(keys + out_cap);
    float *scores = reinterpret_cast<float *>(boxes + out_cap);
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
        const uint32_t at = before + i;
        if (at >= out_cap) break;
        const WbDet d = src[i];
        const float sc = inv_scale[d.level];
        keys[at] = ((unsigned long long)(uint32_t)d.level << 54) | ((unsigned long long)d.r << 40) |
                   ((unsigned long long)d.c << 26) | (unsigned long long)at;
        boxes[at] = make_float4((float)d.c * sc, (float)d.r * sc, (float)((int)d.c + n) * sc, (float)((int)d.r + m) * sc);
        scores[at] = d.score;
    }
}

#define WB_CASC_CONFIGS(X) X(8, 4) X(4, 4) X(2, 4) X(1, 4) X(8, 8) X(4, 8) X(2, 8) X(1, 8) X(2, 16) X(1, 16)

template <int D>
int launch_depth(hipStream_t st, dim3 grid, const CascArgs &a, int rpw, int waves, size_t lds) {
#define WB_X(R, W)                                                                                          \
    if (rpw == R && waves == W) {                                                                           \
        if (a.chn_u8)                                                                                       \
            hipLaunchKernelGGL((cascade_tile_kernel<D, R, W, true>), grid, dim3(W * 64), lds, st, a, a.stages);  \
        else                                                                                                \
            hipLaunchKernelGGL((cascade_tile_kernel<D, R, W, false>), grid, dim3(W * 64), lds, st, a, a.stages); \
        WB_HIP_CHECK(hipGetLastError());                                                                    \
        return WB_OK;                                                                                       \
    }
    WB_CASC_CONFIGS(WB_X)
#undef WB_X
    wb_set_error("cascade: no kernel for rows-per-wave %d x %d waves", rpw, waves);
    return WB_ERR_INVALID;
}

template <int D>
int prepare_depth(int rpw, int waves) {
#define WB_X(R, W)                                                                                          \
    if (rpw == R && waves == W) {                                                                           \
        WB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cascade_tile_kernel<D, R, W, false>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));          \
        WB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cascade_tile_kernel<D, R, W, true>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));          \
        return WB_OK;                                                                                       \
    }
    WB_CASC_CONFIGS(WB_X)
#undef WB_X
    wb_set_error("cascade: no kernel for rows-per-wave %d x %d waves", rpw, waves);
    return WB_ERR_UNSUPPORTED;
}

}  // namespace

int wb_cascade_group(int depth) { return depth >= 3 ? 2 : 4; }

int wb_cascade_prepare(int depth, int rpw, int waves) {
    switch (depth) {
        case 1: return prepare_depth<1>(rpw, waves);
        case 2: return prepare_depth<2>(rpw, waves);
        case 3: return prepare_depth<3>(rpw, waves);
    }
    wb_set_error("cascade: no kernel for depth %d", depth);
    return WB_ERR_UNSUPPORTED;
}

extern "C" int wb_cascade_launch(void *stream, const WbModel *model, const void *chn, int chn_dtype,
                                 int64_t chn_stride, int batch, const WbLevel *levels, int n_levels,
                                 const WbTile *tiles, int n_tiles, WbDet *det, uint32_t *det_count,
                                 uint32_t shard_capacity, uint32_t *alive) {
    WB_REQUIRE(model && chn && levels && tiles && det_count, "wb_cascade_launch: null pointer");
    WB_REQUIRE(det || shard_capacity == 0, "wb_cascade_launch: det is null but capacity > 0");
    WB_REQUIRE(batch >= 1 && batch <= 65535, "wb_cascade_launch: batch %d out of range", batch);
    WB_REQUIRE(n_levels >= 1 && n_tiles >= 1, "wb_cascade_launch: empty launch");
    WB_REQUIRE(chn_dtype == WB_DTYPE_F32 || chn_dtype == WB_DTYPE_U8 || chn_dtype == WB_DTYPE_RANK8,
               "wb_cascade_launch: channel dtype %d (float32, uint8 or ranks)", chn_dtype);
    CascArgs a;
    a.chn = chn;
    a.chn_u8 = chn_dtype != WB_DTYPE_F32;
    a.chn_stride = chn_stride;
    a.levels = levels;
    a.tiles = tiles;
    a.n_levels = n_levels;
    // WB_DTYPE_RANK8: the bytes are threshold ranks of this model (wb_channels_launch wrote them): the uint8 tile
    // kernel with the rank records
    const bool ranks = chn_dtype == WB_DTYPE_RANK8;
    WB_REQUIRE(!ranks || (model->bin_ok && !model->generic), "wb_cascade_launch: this model has no rank tables (wb_model_info: rank_ok)");
    a.stages = ranks ? model->stages_bin_dev : (a.chn_u8 ? model->stages_u8_dev : model->stages_dev);
    a.T = model->n_stages;
    a.m = model->m;
    a.n = model->n;
    a.C = model->C;
    a.lds_rows = model->lds_rows;
    a.lds_pitch = model->lds_pitch;
    a.lds_stages = model->lds_stages;
    a.det = det;
    a.det_count = det_count;
    a.det_cap = shard_capacity;
    a.alive = alive;
    a.n_tiles = n_tiles;
    static const int dbg = getenv("WB_CASC_DBG") ? atoi(getenv("WB_CASC_DBG")) : 0;
    a.dbg = dbg;
    // a wave flips to the stage-parallel tail when few windows are left (measured flat around these)
    struct Spar { int v[4]; };
    static const Spar spar = [] {
        Spar s = {{32, 8, 16, 2}};
        if (const char *e = getenv("WB_CASC_SPAR")) sscanf(e, "%d,%d,%d,%d", &s.v[0], &s.v[1], &s.v[2], &s.v[3]);
        return s;
    }();
    for (int i = 0; i < 4; ++i) a.spar[i] = spar.v[i];
    // the workgroup-wide re-count behind the segment [8, 16) takes every wave's queue as evaluated up to stage 16: a
    // wave must not leave run_segments for the tail before that stage (an override below 16 would skip stages 8..15)
    a.spar[0] = a.spar[0] < 16 ? 16 : a.spar[0];
    a.spar[2] = a.spar[2] < 16 ? 16 : a.spar[2];
    static const int spar_wg = getenv("WB_CASC_SPAR_WG") ? atoi(getenv("WB_CASC_SPAR_WG")) : 64;
    a.spar_wg = spar_wg;
    dim3 grid((unsigned)n_tiles, (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    if (model->generic) {
        GenArgs g;
        g.chn = chn; g.chn_u8 = a.chn_u8; g.chn_stride = chn_stride; g.levels = levels; g.tiles = tiles;
        g.n_levels = n_levels; g.n_tiles = n_tiles;
        g.T = model->n_stages; g.m = model->m; g.n = model->n; g.C = model->C;
        g.node_off = model->g_node_off; g.feat = model->g_feat; g.left = model->g_left; g.right = model->g_right;
        g.thr = model->g_thr; g.pred = model->g_pred; g.theta = model->g_theta;
        g.det = det; g.det_count = det_count; g.det_cap = shard_capacity; g.alive = alive;
        size_t lds = (((size_t)g.T * 4 + 15) & ~(size_t)15) + 256 * 8;
        WB_REQUIRE(lds <= 64 * 1024, "wb_cascade_launch: %d stages exceed the generic kernel's LDS", g.T);
        hipLaunchKernelGGL(cascade_generic_kernel, grid, dim3(256), lds, st, g);
        WB_HIP_CHECK(hipGetLastError());
        return WB_OK;
    }
    const size_t lds = (size_t)(a.chn_u8 ? model->lds_bytes_u8 : model->lds_bytes);
    switch (model->depth) {
        case 1: return launch_depth<1>(st, grid, a, model->rpw, model->waves, lds);
        case 2: return launch_depth<2>(st, grid, a, model->rpw, model->waves, lds);
        case 3: return launch_depth<3>(st, grid, a, model->rpw, model->waves, lds);
    }
    wb_set_error("wb_cascade_launch: model depth %d has no kernel", model->depth);
    return WB_ERR_UNSUPPORTED;
}

extern "C" int wb_tree_eval_launch(void *stream, const void *X, int x_dtype, int u, int v, int C, const int32_t *rs,
                                   const int32_t *cs, int64_t n_pos, const uint8_t *feature,
                                   const float *threshold, const int8_t *left, const int8_t *right,
                                   const float *prediction, int n_nodes, float *out) {
    WB_REQUIRE(n_pos >= 0, "wb_tree_eval_launch: negative count");
    if (n_pos == 0) return WB_OK;
    WB_REQUIRE(X && rs && cs && feature && threshold && left && right && prediction && out,
               "wb_tree_eval_launch: null pointer");
    WB_REQUIRE(u > 0 && v > 0 && C > 0 && n_nodes > 0 && n_nodes <= 127, "wb_tree_eval_launch: bad shape");
    WB_REQUIRE(x_dtype == WB_DTYPE_F32 || x_dtype == WB_DTYPE_U8, "wb_tree_eval_launch: channel dtype %d (float32 or uint8)", x_dtype);
    int64_t blocks = (n_pos + 255) / 256;
    hipLaunchKernelGGL(tree_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X,
                       (int)(x_dtype == WB_DTYPE_U8), u, v, C, rs, cs, n_pos, feature, threshold, left, right, prediction, n_nodes, out);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_gather_samples_launch(void *stream, const void *X, int x_dtype, int u, int v, int C,
                                        const int32_t *rs, const int32_t *cs, int64_t n_pos, int m, int n, void *out) {
    WB_REQUIRE(n_pos >= 0, "wb_gather_samples_launch: negative count");
    if (n_pos == 0) return WB_OK;
    WB_REQUIRE(X && rs && cs && out, "wb_gather_samples_launch: null pointer");
    WB_REQUIRE(u > 0 && v > 0 && C > 0 && m > 0 && n > 0 && m <= u && n <= v, "wb_gather_samples_launch: bad shape");
    WB_REQUIRE(x_dtype == WB_DTYPE_F32 || x_dtype == WB_DTYPE_U8, "wb_gather_samples_launch: channel dtype %d (float32 or uint8)", x_dtype);
    WB_REQUIRE(n_pos <= 0x7fffffff, "wb_gather_samples_launch: too many samples for one launch");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_pos);
    const int esz = x_dtype == WB_DTYPE_U8 ? 1 : 4;
    const size_t px = (size_t)C * esz;                       // bytes per pixel
    const bool al16 = px % 16 == 0 && (reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    const bool al4 = px % 4 == 0 && (reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out)) % 4 == 0;
    if (al16)       // whole pixels as 16-byte vectors (float32 x 4 channels: one per pixel)
        hipLaunchKernelGGL((gather_samples_kernel<uint4>), grid, dim3(64), 0, st, (const uint4 *)X, v, n * (int)(px / 16), rs, cs, m,
                           (int)(px / 16), (uint4 *)out);
    else if (al4)
        hipLaunchKernelGGL((gather_samples_kernel<uint32_t>), grid, dim3(64), 0, st, (const uint32_t *)X, v, n * (int)(px / 4), rs, cs,
                           m, (int)(px / 4), (uint32_t *)out);
    else
        hipLaunchKernelGGL((gather_samples_kernel<uint8_t>), grid, dim3(64), 0, st, (const uint8_t *)X, v, n * (int)px, rs, cs, m,
                           (int)px, (uint8_t *)out);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_samples_predict_launch(void *stream, const WbModel *model, const void *X, int x_dtype,
                                         int64_t n_samples, float *H, uint8_t *mask) {
    WB_REQUIRE(n_samples >= 0, "wb_samples_predict_launch: negative count");
    if (n_samples == 0) return WB_OK;
    WB_REQUIRE(model && X && H && mask, "wb_samples_predict_launch: null pointer");
    WB_REQUIRE(x_dtype == WB_DTYPE_F32 || x_dtype == WB_DTYPE_U8, "wb_samples_predict_launch: sample dtype %d (float32 or uint8)", x_dtype);
    GenArgs g = {};
    g.chn = X; g.chn_u8 = x_dtype == WB_DTYPE_U8;
    g.T = model->n_stages; g.m = model->m; g.n = model->n; g.C = model->C;
    g.node_off = model->g_node_off; g.feat = model->g_feat; g.left = model->g_left; g.right = model->g_right;
    g.thr = model->g_thr; g.pred = model->g_pred; g.theta = model->g_theta;
    const int64_t blocks = (n_samples + 255) / 256;
    WB_REQUIRE(blocks <= 0x7fffffff, "wb_samples_predict_launch: too many samples for one launch");
    hipLaunchKernelGGL(samples_predict_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n_samples, H, mask);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_tree_apply_launch(void *stream, const void *X, int x_dtype, int64_t n_samples, int m, int n, int C,
                                    const uint8_t *feature, const float *threshold, const int8_t *left,
                                    const int8_t *right, int n_nodes, int32_t *node) {
    WB_REQUIRE(n_samples >= 0, "wb_tree_apply_launch: negative count");
    if (n_samples == 0) return WB_OK;
    WB_REQUIRE(X && feature && threshold && left && right && node, "wb_tree_apply_launch: null pointer");
    WB_REQUIRE(m > 0 && n > 0 && C > 0 && n_nodes > 0 && n_nodes <= 127, "wb_tree_apply_launch: bad shape");
    WB_REQUIRE(x_dtype == WB_DTYPE_F32 || x_dtype == WB_DTYPE_U8, "wb_tree_apply_launch: sample dtype %d (float32 or uint8)", x_dtype);
    const int64_t blocks = (n_samples + 255) / 256;
    WB_REQUIRE(blocks <= 0x7fffffff, "wb_tree_apply_launch: too many samples for one launch");
    hipLaunchKernelGGL(tree_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X,
                       (int)(x_dtype == WB_DTYPE_U8), n_samples, m, n, C, feature, threshold, left, right, n_nodes, node);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_det_pack_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                                  int32_t *packed, uint32_t packed_capacity) {
    WB_REQUIRE(det_count && packed, "wb_det_pack_launch: null pointer");
    WB_REQUIRE(det || shard_capacity == 0, "wb_det_pack_launch: det is null but capacity > 0");
    WB_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0, "wb_det_pack_launch: packed must be 16-byte aligned");
    hipLaunchKernelGGL(det_pack_kernel, dim3(WB_DET_SHARDS), dim3(256), 0, (hipStream_t)stream, det, det_count,
                       shard_capacity, packed, packed_capacity);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_det_finish_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                                    const float *inv_scale, int n_levels, int max_rows, int max_cols, int m, int n,
                                    void *out, uint32_t out_capacity) {
    WB_REQUIRE(det_count && out && inv_scale, "wb_det_finish_launch: null pointer");
    WB_REQUIRE(det || shard_capacity == 0, "wb_det_finish_launch: det is null but capacity > 0");
    WB_REQUIRE(reinterpret_cast<uintptr_t>(out) % 16 == 0, "wb_det_finish_launch: out must be 16-byte aligned");
    WB_REQUIRE(out_capacity % 2 == 0, "wb_det_finish_launch: out_capacity must be even (16-byte aligned sections)");
    // the sort key holds level in 10 bits, r and c in 14 bits each, the packed position in 26
    if (n_levels > (1 << 10) || max_rows > (1 << 14) || max_cols > (1 << 14) || out_capacity > (1u << 26)) {
        wb_set_error("wb_det_finish_launch: %d levels of up to %d x %d windows, %u records do not fit the 10/14/14/26-bit key",
                     n_levels, max_rows, max_cols, out_capacity);
        return WB_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(det_finish_kernel, dim3(WB_DET_SHARDS), dim3(256), 0, (hipStream_t)stream, det, det_count,
                       shard_capacity, inv_scale, m, n, reinterpret_cast<int32_t *>(out), out_capacity);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_boxes_launch(void *stream, const WbDet *det, int64_t n_det, const float *inv_scale, int m,
                               int n, float *boxes, float *scores) {
    WB_REQUIRE(n_det >= 0, "wb_boxes_launch: negative count");
    if (n_det == 0) return WB_OK;
    WB_REQUIRE(det && inv_scale && boxes && scores, "wb_boxes_launch: null pointer");
    int64_t blocks = (n_det + 255) / 256;
    hipLaunchKernelGGL(boxes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, det, n_det,
                       inv_scale, m, n, boxes, scores);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}


#ifdef WB_CASC_STAMPS
// Diagnostic build: mean microseconds between consecutive stamps over the first n_wg workgroups
// of the last cascade launch (s_memrealtime ticks at 100 MHz).
extern "C" int wb_debug_cascade_stamps(int n_wg, double *mean_us7, double *lifetime_us) {
    if (n_wg > WB_STAMP_WGS) n_wg = WB_STAMP_WGS;
    std::vector<unsigned long long> h((size_t)n_wg * WB_STAMP_SLOTS);
    WB_HIP_CHECK(hipDeviceSynchronize());
    WB_HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
    double acc[7] = {0}, life = 0;
    for (int w = 0; w < n_wg; ++w) {
        const unsigned long long *s = &h[(size_t)w * WB_STAMP_SLOTS];
        for (int k = 0; k < 7; ++k) acc[k] += (double)(s[k + 1] - s[k]);
        life += (double)(s[7] - s[0]);
    }
    for (int k = 0; k < 7; ++k) mean_us7[k] = acc[k] / n_wg / 100.0;
    *lifetime_us = life / n_wg / 100.0;
    return WB_OK;
}
#endif
