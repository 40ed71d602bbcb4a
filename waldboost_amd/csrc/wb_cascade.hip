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

#include <vector>

#include "wb_common.h"

#include "wb_cascade_tile.h"

namespace {

template <int D, int RPW, int WAVES, int EB>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_sgpr(80))) void cascade_tile_kernel(CascArgs a, const int32_t *__restrict__ stages) {
    cascade_tile_body<D, RPW, WAVES, EB, false>(a, stages);
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
    uint32_t *zero;
    int zero_words;
};

__global__ __launch_bounds__(256) void cascade_generic_kernel(GenArgs a) {
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < a.zero_words; i += 256) a.zero[i] = 0u;      // (see cascade_tile_body)
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

// det_finish_kernel with the ordering done here as well (wb_det_finish_sorted_launch).  The keys are unique, so a record's
// place in the reference's order is the NUMBER OF SMALLER KEYS: every workgroup gathers all n <= WB_FINISH_SORT_MAX keys
// into LDS (50 KB of L2 reads, every load in flight at once: a thread finds the shard of its flat index by bisection of
// the shards' prefix sums), ranks its own 16 records against them -- sixteen threads per record, each over a sixteenth of
// the keys, the keys as LDS broadcast reads -- and writes key, box and score straight to the record's rank: up to 256
// workgroups of 16 records, one per CU.  (One workgroup sorting in LDS -- a bitonic network, built first --
// took 37 us for the same: 78 stages x 64 KB through ONE CU's LDS.)  The host takes slices instead of sorting and
// gathering (0.03 ms of a 0.23 ms Model.detect call, and the step that bounded Model.detect_stream at batch 1).
// header[3] = 1 says so.  More valid records than WB_FINISH_SORT_MAX (or than out_cap): the sections are written
// unordered, exactly as det_finish_kernel leaves them, header[3] = 0.
#define WB_FINISH_SORT_MAX 4096
// TPR threads per record, 256 / TPR records per workgroup, WB_FINISH_SORT_MAX * TPR / 256 workgroups (>= WB_DET_SHARDS: the
// unordered form wants a workgroup per shard).  One image: TPR = 16, 256 workgroups -- the latency of Model.detect's last
// step; a batch: TPR = 4, 64 workgroups per image (every workgroup gathers all of its image's keys: fewer, longer ones).
// blockIdx.y: the image of a batch (wb_det_order_batch_launch) -- its own 64 counters, record region and output block
// (img_det / img_out: their distances in records / int32 words); a single image launches one row.
template <int TPR>
__global__ __launch_bounds__(256) void det_finish_sorted_kernel(const WbDet *det, const uint32_t *det_count, uint32_t cap,
                                                                 const float *inv_scale, int m, int n, int32_t *out, uint32_t out_cap,
                                                                 size_t img_det, size_t img_out, const int32_t *tail, uint32_t tail_words) {
    static_assert(WB_DET_SHARDS == 64, "one counter per lane of a wave, one workgroup per shard");
    __shared__ unsigned long long skey[WB_FINISH_SORT_MAX];
    __shared__ float sscore[WB_FINISH_SORT_MAX];
    __shared__ uint32_t sbefore[65];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wg = blockIdx.x;
    det_count += (size_t)blockIdx.y * WB_DET_SHARDS;
    det += (size_t)blockIdx.y * img_det;
    out += (size_t)blockIdx.y * img_out;
    // the caller's tail words (the scan's alive[] statistics) behind the scores: ONE read-back carries everything
    if (tail != nullptr)
        for (uint32_t i = (uint32_t)wg * 256u + (uint32_t)tid; i < tail_words; i += gridDim.x * 256u) out[4 + 7 * (size_t)out_cap + i] = tail[i];
    const uint32_t raw = det_count[lane];
    const uint32_t mine = raw < cap ? raw : cap;
    uint32_t before = 0, total = 0, worst = 0;                // before: valid records in the shards in front of shard `lane`
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mine, s);
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)raw, s);
        before += s < lane ? c : 0u;
        total += c;
        worst = r > worst ? r : worst;
    }
    const bool ordered = total <= out_cap && total <= WB_FINISH_SORT_MAX;
    if (wg == 0 && tid == 0) {
        out[0] = (int32_t)total;
        out[1] = (int32_t)worst;
        out[2] = (int32_t)(total < out_cap ? total : out_cap);
        out[3] = ordered ? 1 : 0;
    }
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(out + 4);
    float4 *boxes = reinterpret_cast<float4 *>(keys + out_cap);
    float *scores = reinterpret_cast<float *>(boxes + out_cap);
    auto key_of = [](const WbDet &d, uint32_t at) {
        return ((unsigned long long)(uint32_t)d.level << 54) | ((unsigned long long)d.r << 40) | ((unsigned long long)d.c << 26) |
               (unsigned long long)at;
    };
    auto box_of = [&](uint32_t level, uint32_t r, uint32_t c) {
        const float sc = inv_scale[level];
        return make_float4((float)c * sc, (float)r * sc, (float)((int)c + n) * sc, (float)((int)r + m) * sc);
    };
    if (!ordered) {                                           // (grid-uniform) det_finish_kernel's body: this workgroup's shard
        if (wg >= WB_DET_SHARDS) return;
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)mine, wg), b0 = (uint32_t)__builtin_amdgcn_readlane((int)before, wg);
        const WbDet *src = det + (size_t)wg * cap;
        for (uint32_t i = tid; i < cnt; i += 256) {
            const uint32_t at = b0 + i;
            if (at >= out_cap) break;
            const WbDet d = src[i];
            keys[at] = key_of(d, at);
            boxes[at] = box_of((uint32_t)d.level, d.r, d.c);
            scores[at] = d.score;
        }
        return;
    }
    constexpr int RPW = 256 / TPR;
    if (total <= (uint32_t)(RPW * wg)) return;                // (this workgroup's records start behind the last one)
    if (tid < 64) sbefore[tid] = before;
    if (tid == 0) sbefore[64] = total;
    __syncthreads();
    // where flat position q lies: the last shard s with sbefore[s] <= q (empty shards share a prefix with their successor
    // and are stepped over: the LAST such shard is the one that holds records)
    auto locate = [&](uint32_t q) {
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t step = 32; step > 0; step >>= 1)
            if (sbefore[lo + step] <= q) lo += step;
        return det + (size_t)lo * cap + (q - sbefore[lo]);
    };
    // all keys into LDS: WB_FINISH_SORT_MAX / 256 records per thread, every load requested before the first is used
    constexpr int PER = WB_FINISH_SORT_MAX / 256;
    {
        uint4 lr[PER];                                        // (image, level, r | c << 16, score)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            // (unconditional: positions past the end load the last record again -- with a branch around it every
            // bisection, six dependent LDS reads, ran alone: sixteen of them in a row were a quarter of the kernel)
            const uint32_t q = (uint32_t)tid + 256u * k;
            lr[k] = *reinterpret_cast<const uint4 *>(locate(q < total ? q : total - 1u));
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t q = (uint32_t)tid + 256u * k;
            if (q < total) {
                skey[q] = ((unsigned long long)lr[k].y << 54) | ((unsigned long long)(lr[k].z & 0xffffu) << 40) |
                          ((unsigned long long)(lr[k].z >> 16) << 26) | (unsigned long long)q;
                sscore[q] = __uint_as_float(lr[k].w);        // (the record's second visit below needs no memory)
            }
        }
    }
    __syncthreads();
    // TPR threads per record, each over the keys j = part, part + TPR, ... (a wave's records read the same TPR keys at a
    // time: LDS broadcasts), eight keys per thread and pass in flight
    constexpr uint32_t UN = 8;
    const uint32_t q = (uint32_t)(RPW * wg) + (uint32_t)tid / TPR, part = (uint32_t)tid % TPR;
    const bool live = q < total;
    const unsigned long long me = skey[live ? q : 0u];
    uint32_t smaller = 0;
    const uint32_t nfull = total - total % (TPR * UN);        // whole passes; the rest key by key
    for (uint32_t j = part; j < nfull; j += TPR * UN) {
        unsigned long long kk[UN];
#pragma unroll
        for (uint32_t u = 0; u < UN; ++u) kk[u] = skey[j + TPR * u];
#pragma unroll
        for (uint32_t u = 0; u < UN; ++u) smaller += kk[u] < me ? 1u : 0u;
    }
    for (uint32_t j = nfull + part; j < total; j += TPR) smaller += skey[j] < me ? 1u : 0u;
#pragma unroll
    for (uint32_t d = 1; d < TPR; d <<= 1) smaller += (uint32_t)__shfl_xor((int)smaller, (int)d);
    if (live && part == 0) {
        keys[smaller] = me;
        boxes[smaller] = box_of((uint32_t)(me >> 54), (uint32_t)(me >> 40) & 0x3fffu, (uint32_t)(me >> 26) & 0x3fffu);
        scores[smaller] = sscore[q];
    }
}

// A batch's detections by image (the step in front of det_finish_sorted_kernel for a batch): workgroup b walks ALL valid
// records of the shards -- flat positions, the shard of a position by bisection of the prefix sums, several loads in
// flight per thread -- and appends those of image b to bucket b (wave-aggregated: one LDS atomic per wave and pass).
// The order inside a bucket is whatever the atomics gave; ranking by key does not depend on it.  bucket_count[b][0] =
// the image's record count (above bucket_cap: the finishing kernel reports the overflow), [b][1..63] = 0: a bucket reads
// as a shard set whose first shard holds everything.  info = (valid records, fullest shard, images, bucket_cap).
__global__ __launch_bounds__(1024) void det_bucket_kernel(const WbDet *det, const uint32_t *det_count, uint32_t cap, WbDet *bucket,
                                                           uint32_t bucket_cap, uint32_t *bucket_count, int32_t *info) {
    static_assert(WB_DET_SHARDS == 64, "one counter per lane of a wave");
    __shared__ uint32_t sbefore[65];
    __shared__ uint32_t n_img;
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const uint32_t raw = det_count[lane];
    const uint32_t mine = raw < cap ? raw : cap;
    uint32_t before = 0, total = 0, worst = 0;
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mine, s);
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)raw, s);
        before += s < lane ? c : 0u;
        total += c;
        worst = r > worst ? r : worst;
    }
    if (b == 0 && tid == 0) {
        info[0] = (int32_t)total;
        info[1] = (int32_t)worst;
        info[2] = (int32_t)gridDim.x;
        info[3] = (int32_t)bucket_cap;
    }
    if (tid < 64) sbefore[tid] = before;
    if (tid == 0) {
        sbefore[64] = total;
        n_img = 0;
    }
    __syncthreads();
    auto locate = [&](uint32_t q) {
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t step = 32; step > 0; step >>= 1)
            if (sbefore[lo + step] <= q) lo += step;
        return det + (size_t)lo * cap + (q - sbefore[lo]);
    };
    WbDet *dst = bucket + (size_t)b * bucket_cap;
    constexpr int U = 4;
    for (uint32_t q0 = 0; q0 < total; q0 += 1024 * U) {      // (workgroup-uniform bounds)
        uint4 rec[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + (uint32_t)tid + 1024u * u;
            ok[u] = q < total;
            rec[u] = make_uint4(0xffffffffu, 0u, 0u, 0u);
            if (ok[u]) rec[u] = *reinterpret_cast<const uint4 *>(locate(q));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool take = ok[u] && (int)rec[u].x == b;
            const unsigned long long mask = __ballot(take);
            if (mask == 0ull) continue;                       // (wave-uniform)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&n_img, (uint32_t)__popcll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (take) {
                const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                if (slot < bucket_cap) *reinterpret_cast<uint4 *>(dst + slot) = rec[u];
            }
        }
    }
    __syncthreads();
    if (tid < 64) bucket_count[(size_t)b * WB_DET_SHARDS + tid] = tid == 0 ? n_img : 0u;
}

#define WB_CASC_CONFIGS(X) X(8, 4) X(4, 4) X(2, 4) X(1, 4) X(8, 8) X(4, 8) X(2, 8) X(1, 8) X(2, 16) X(1, 16)

template <int D>
int launch_depth(hipStream_t st, dim3 grid, const CascArgs &a, int rpw, int waves, size_t lds) {
#define WB_X(R, W)                                                                                          \
    if (rpw == R && waves == W) {                                                                           \
        if (a.chn_u8 == 2)                                                                                  \
            hipLaunchKernelGGL((cascade_tile_kernel<D, R, W, 2>), grid, dim3(W * 64), lds, st, a, a.stages);     \
        else if (a.chn_u8)                                                                                  \
            hipLaunchKernelGGL((cascade_tile_kernel<D, R, W, 1>), grid, dim3(W * 64), lds, st, a, a.stages);     \
        else                                                                                                \
            hipLaunchKernelGGL((cascade_tile_kernel<D, R, W, 0>), grid, dim3(W * 64), lds, st, a, a.stages);     \
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
        WB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cascade_tile_kernel<D, R, W, 0>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));          \
        WB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cascade_tile_kernel<D, R, W, 1>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));          \
        WB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cascade_tile_kernel<D, R, W, 2>),  \
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
    return wb_cascade_launch_z(stream, model, chn, chn_dtype, chn_stride, batch, levels, n_levels, tiles, n_tiles, det, det_count,
                               shard_capacity, alive, nullptr, 0);
}

extern "C" int wb_cascade_launch_z(void *stream, const WbModel *model, const void *chn, int chn_dtype,
                                   int64_t chn_stride, int batch, const WbLevel *levels, int n_levels,
                                   const WbTile *tiles, int n_tiles, WbDet *det, uint32_t *det_count,
                                   uint32_t shard_capacity, uint32_t *alive, uint32_t *zero, int zero_words) {
    WB_REQUIRE(zero_words == 0 || (zero && zero_words > 0), "wb_cascade_launch_z: zero_words without a pointer");
    WB_REQUIRE(model && chn && levels && tiles && det_count, "wb_cascade_launch: null pointer");
    WB_REQUIRE(det || shard_capacity == 0, "wb_cascade_launch: det is null but capacity > 0");
    WB_REQUIRE(batch >= 1 && batch <= 65535, "wb_cascade_launch: batch %d out of range", batch);
    WB_REQUIRE(n_levels >= 1 && n_tiles >= 1, "wb_cascade_launch: empty launch");
    WB_REQUIRE(chn_dtype == WB_DTYPE_F32 || chn_dtype == WB_DTYPE_U8 || chn_dtype == WB_DTYPE_RANK8 || chn_dtype == WB_DTYPE_RANK16,
               "wb_cascade_launch: channel dtype %d (float32, uint8 or ranks)", chn_dtype);
    CascArgs a;
    a.chn = chn;
    a.chn_u8 = chn_dtype == WB_DTYPE_F32 ? 0 : (chn_dtype == WB_DTYPE_RANK16 ? 2 : 1);      // element bytes of a byte tile
    a.chn_stride = chn_stride;
    a.levels = levels;
    a.tiles = tiles;
    a.n_levels = n_levels;
    // WB_DTYPE_RANK8: the bytes are threshold ranks of this model (wb_channels_launch wrote them): the uint8 tile
    // kernel with the rank records
    const bool ranks = chn_dtype == WB_DTYPE_RANK8, ranks16 = chn_dtype == WB_DTYPE_RANK16;
    WB_REQUIRE(!ranks || (model->bin_ok && !model->generic), "wb_cascade_launch: this model has no rank tables (wb_model_info: rank_ok)");
    WB_REQUIRE(!ranks16 || (model->bin16_ok && !model->generic), "wb_cascade_launch: this model has no 16-bit rank tables (wb_model_info: rank16_ok)");
    a.stages = ranks16 ? model->stages_bin16_dev : ranks ? model->stages_bin_dev : (a.chn_u8 ? model->stages_u8_dev : model->stages_dev);
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
    a.zero = zero;
    a.zero_words = zero_words;
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
    static const int spar_wg = getenv("WB_CASC_SPAR_WG") ? atoi(getenv("WB_CASC_SPAR_WG")) : 32;
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
        g.zero = zero; g.zero_words = zero_words;
        size_t lds = (((size_t)g.T * 4 + 15) & ~(size_t)15) + 256 * 8;
        WB_REQUIRE(lds <= 64 * 1024, "wb_cascade_launch: %d stages exceed the generic kernel's LDS", g.T);
        hipLaunchKernelGGL(cascade_generic_kernel, grid, dim3(256), lds, st, g);
        WB_HIP_CHECK(hipGetLastError());
        return WB_OK;
    }
    static const size_t xlds = getenv("WB_CASC_XLDS") ? (size_t)atoi(getenv("WB_CASC_XLDS")) : 0;   // diagnostic: fewer workgroups per CU
    const size_t lds = (size_t)(ranks16 ? model->lds_bytes_u16 : a.chn_u8 ? model->lds_bytes_u8 : model->lds_bytes) + xlds;
    // the model-specialised kernel, when wb_model_specialize has built one for this kind of byte tile (WB_CASC_JIT=0:
    // diagnostic, stay on the generic kernel)
    static const bool jit_off = getenv("WB_CASC_JIT") && atoi(getenv("WB_CASC_JIT")) == 0;
    if (void *jf = a.chn_u8 && !jit_off && !model->jit_off ? (ranks16 ? model->jit_bin16 : ranks ? model->jit_bin : model->jit_u8) : nullptr) {
        const int32_t *stages = a.stages;
        void *params[] = {&a, &stages};
        WB_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)jf, grid.x, grid.y, 1, (unsigned)model->waves * 64, 1, 1, (unsigned)lds, st,
                                           params, nullptr));
        return WB_OK;
    }
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

extern "C" int wb_det_finish_sorted_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                                           const float *inv_scale, int n_levels, int max_rows, int max_cols, int m, int n,
                                           void *out, uint32_t out_capacity, const int32_t *tail, uint32_t tail_words) {
    WB_REQUIRE(tail || tail_words == 0, "wb_det_finish_sorted_launch: tail is null but tail_words > 0");
    WB_REQUIRE(det_count && out && inv_scale, "wb_det_finish_sorted_launch: null pointer");
    WB_REQUIRE(det || shard_capacity == 0, "wb_det_finish_sorted_launch: det is null but capacity > 0");
    WB_REQUIRE(reinterpret_cast<uintptr_t>(out) % 16 == 0, "wb_det_finish_sorted_launch: out must be 16-byte aligned");
    WB_REQUIRE(out_capacity % 2 == 0, "wb_det_finish_sorted_launch: out_capacity must be even (16-byte aligned sections)");
    if (n_levels > (1 << 10) || max_rows > (1 << 14) || max_cols > (1 << 14) || out_capacity > (1u << 26)) {
        wb_set_error("wb_det_finish_sorted_launch: %d levels of up to %d x %d windows, %u records do not fit the 10/14/14/26-bit key",
                     n_levels, max_rows, max_cols, out_capacity);
        return WB_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(det_finish_sorted_kernel<16>, dim3(WB_FINISH_SORT_MAX * 16 / 256), dim3(256), 0, (hipStream_t)stream, det, det_count,
                       shard_capacity, inv_scale, m, n, reinterpret_cast<int32_t *>(out), out_capacity, (size_t)0, (size_t)0, tail, tail_words);
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

extern "C" int wb_det_order_batch_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                                         int n_images, const float *inv_scale, int n_levels, int max_rows, int max_cols, int m,
                                         int n, void *scratch, size_t scratch_bytes, void *out, uint32_t out_capacity) {
    WB_REQUIRE(det_count && out && inv_scale && scratch, "wb_det_order_batch_launch: null pointer");
    WB_REQUIRE(det || shard_capacity == 0, "wb_det_order_batch_launch: det is null but capacity > 0");
    WB_REQUIRE(n_images >= 1 && n_images <= 65535, "wb_det_order_batch_launch: 1 .. 65535 images");
    WB_REQUIRE(reinterpret_cast<uintptr_t>(out) % 16 == 0 && reinterpret_cast<uintptr_t>(scratch) % 16 == 0,
               "wb_det_order_batch_launch: out and scratch must be 16-byte aligned");
    WB_REQUIRE(out_capacity % 4 == 0 && out_capacity >= 4, "wb_det_order_batch_launch: out_capacity must be a multiple of 4 (16-byte aligned blocks)");
    if (n_levels > (1 << 10) || max_rows > (1 << 14) || max_cols > (1 << 14) || out_capacity > (1u << 26)) {
        wb_set_error("wb_det_order_batch_launch: %d levels of up to %d x %d windows, %u records do not fit the 10/14/14/26-bit key",
                     n_levels, max_rows, max_cols, out_capacity);
        return WB_ERR_UNSUPPORTED;
    }
    const size_t counts_bytes = (size_t)n_images * WB_DET_SHARDS * 4, need = counts_bytes + (size_t)n_images * out_capacity * sizeof(WbDet);
    if (scratch_bytes < need) {
        wb_set_error("wb_det_order_batch_launch: scratch holds %zu bytes, %d images of %u records want %zu", scratch_bytes, n_images,
                     out_capacity, need);
        return WB_ERR_INVALID;
    }
    uint32_t *bucket_count = reinterpret_cast<uint32_t *>(scratch);
    WbDet *bucket = reinterpret_cast<WbDet *>(reinterpret_cast<unsigned char *>(scratch) + counts_bytes);
    int32_t *info = reinterpret_cast<int32_t *>(out);
    hipLaunchKernelGGL(det_bucket_kernel, dim3(n_images), dim3(1024), 0, (hipStream_t)stream, det, det_count, shard_capacity, bucket,
                       out_capacity, bucket_count, info);
    static_assert(WB_FINISH_SORT_MAX * 4 / 256 >= WB_DET_SHARDS, "a workgroup per shard for the unordered form");
    hipLaunchKernelGGL(det_finish_sorted_kernel<4>, dim3(WB_FINISH_SORT_MAX * 4 / 256, n_images), dim3(256), 0, (hipStream_t)stream, bucket, bucket_count,
                       out_capacity, inv_scale, m, n, info + 4, out_capacity, (size_t)out_capacity, (size_t)(4 + 7 * (size_t)out_capacity),
                       (const int32_t *)nullptr, 0u);
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
