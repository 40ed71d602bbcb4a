// Dense sliding-window evaluation of the WaldBoost decision-tree cascade on gfx950.
//
// Replaces reference model.py:216-259 (Model.predict_on_image: window grid, stage loop,
// rejection, compaction, n_loc/n_weak statistics) and training.py:84-96
// (DTree.predict_on_image: the tree walk on all alive windows).
//
// One workgroup (4 wavefronts) owns a tile of TR x 64 windows of one level of one image:
//   * the (TR+m-1) x (64+n-1) x C channel block is staged once into LDS, planar, so that a
//     wavefront's 64 lanes (64 adjacent window columns) gather from 64 adjacent banks;
//   * the stage loop is wave-synchronous: every lane of a wave is at the same stage, so the
//     stage record (feature offsets, thresholds, leaf values, theta) comes in through the
//     scalar cache (s_load) and costs no vector memory or LDS traffic;
//   * phase A runs the first stages with RPW windows per lane (ILP hides the LDS latency);
//     survivors are compacted with wave ballot + mbcnt into the wave's own LDS queue;
//   * phase B re-packs the survivors densely (64 per wave-iteration) for geometrically
//     growing stage segments, compacting in place after each segment;
//   * windows alive after the last stage are appended to the detection buffer with one
//     wave-aggregated global atomic; per-stage alive counts go through an LDS histogram.
//
// Scores are accumulated in fp32 strictly in stage order and compared with `>=`, so they are
// bit-identical to the reference's `hs += ...; mask = hs >= theta`.
#include "wb_common.h"

namespace {

struct CascArgs {
    const float *chn;
    int64_t chn_stride;
    int layout;
    const WbLevel *levels;
    const WbTile *tiles;
    int n_levels;
    const int32_t *stages;
    int T, m, n, C;
    int lds_rows, lds_pitch;
    WbDet *det;
    uint32_t *det_count;
    uint32_t capacity;
    uint32_t *alive;
};

__device__ inline float as_f(int32_t x) { return __int_as_float(x); }

template <int N, typename V> struct Sel {
    // a[path] for path in [0, N) with the first decision in the most significant bit
    static __device__ inline V get(const V *a, int path) {
        V lo = Sel<N / 2, V>::get(a, path);
        V hi = Sel<N / 2, V>::get(a + N / 2, path);
        return (path & (N / 2)) ? hi : lo;
    }
};
template <typename V> struct Sel<1, V> {
    static __device__ inline V get(const V *a, int) { return a[0]; }
};

// The stage record, pulled into SGPRs by the caller (wave-uniform address).
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
    template <int d> __device__ inline int step(const float *tile, int base, int path) const {
        constexpr int first = (1 << d) - 1;
        int o = Sel<(1 << d), int>::get(off + first, path);
        float th = Sel<(1 << d), float>::get(thr + first, path);
        float v = tile[base + o];
        return 2 * path + ((v <= th) ? 0 : 1);   // NaN goes right, like the reference's `<=`
    }
    // walk the complete depth-D tree for the window whose origin is tile[base]
    __device__ inline float eval(const float *tile, int base) const {
        int path = step<0>(tile, base, 0);
        if constexpr (D > 1) path = step<1>(tile, base, path);
        if constexpr (D > 2) path = step<2>(tile, base, path);
        return Sel<NL, float>::get(pred, path);
    }
};

__device__ inline int lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

template <int D, int RPW>
__global__ __launch_bounds__(256) void cascade_kernel(CascArgs a, const int32_t *__restrict__ stages) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TR = RPW * WB_CASC_WAVES;
    constexpr int SD = WB_STAGE_DWORDS(D);
    constexpr int S0 = 4;                                  // stages in phase A

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WbTile tile_d = a.tiles[blockIdx.x];
    const WbLevel L = a.levels[tile_d.level];
    const int b = blockIdx.y;
    const int pitch = a.lds_pitch, rows = a.lds_rows;

    float *tile = reinterpret_cast<float *>(smem);
    const int tile_floats = a.C * rows * pitch;
    uint2 *queue = reinterpret_cast<uint2 *>(smem + (size_t)tile_floats * 4) + wave * (RPW * 64);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + (size_t)tile_floats * 4 + (size_t)TR * 64 * 8);

    const int nr = L.u - a.m > 0 ? L.u - a.m : 0;          // window grid
    const int nc = L.v - a.n > 0 ? L.v - a.n : 0;
    const int r0 = tile_d.ty * TR, c0 = tile_d.tx * WB_CASC_TC;

    for (int t = tid; t < a.T; t += 256) hist[t] = 0;

    // ---- stage the channel block into LDS (planar [C][rows][pitch])
    const float *chn = a.chn + (int64_t)b * a.chn_stride + L.chn_off;
    if (a.layout == WB_LAYOUT_PLANAR) {
        const int p4 = pitch >> 2;
        const int per_ch = rows * p4;
        const int64_t plane = (int64_t)L.u * L.vp;
        for (int ch = 0; ch < a.C; ++ch) {
            const float *src = chn + ch * plane;
            float *dst = tile + ch * rows * pitch;
            int row = tid / p4, q = tid - row * p4;
            const int drow = 256 / p4, dq = 256 - drow * p4;
            for (int idx = tid; idx < per_ch; idx += 256) {
                int gr = r0 + row, gc = c0 + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gr < L.u && gc < L.vp) v = *reinterpret_cast<const float4 *>(src + (int64_t)gr * L.vp + gc);
                *reinterpret_cast<float4 *>(dst + row * pitch + 4 * q) = v;
                row += drow;
                q += dq;
                if (q >= p4) { q -= p4; ++row; }
            }
        }
    } else {  // HWC arrays handed in by a caller (Model.predict_on_image on host data)
        const int total = a.C * rows * pitch;
        for (int idx = tid; idx < total; idx += 256) {
            int ch = idx % a.C;
            int rc = idx / a.C;
            int col = rc % pitch, row = rc / pitch;
            int gr = r0 + row, gc = c0 + col;
            float v = 0.f;
            if (gr < L.u && gc < L.v) v = chn[((int64_t)gr * L.v + gc) * a.C + ch];
            tile[(ch * rows + row) * pitch + col] = v;
        }
    }
    __syncthreads();

    // ---- phase A: RPW windows per lane through stages [0, S0)
    float hs[RPW];
    bool live[RPW];
    int base[RPW];
    const int wr = wave * RPW;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        hs[j] = 0.f;
        live[j] = (c0 + lane < nc) && (r0 + wr + j < nr);
        base[j] = (wr + j) * pitch + lane;
    }
    const int tA = a.T < S0 ? a.T : S0;
    for (int t = 0; t < tA; ++t) {
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < RPW; ++j) cnt += __popcll(__ballot(live[j]));
        if (cnt == 0) break;
        if (lane == 0) atomicAdd(&hist[t], (uint32_t)cnt);
        Stage<D> st;
        st.load(stages + (size_t)__builtin_amdgcn_readfirstlane(t) * SD);
        const bool rejects = st.theta != -INFINITY;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            float p = st.eval(tile, base[j]);
            float h = hs[j] + p;
            hs[j] = live[j] ? h : hs[j];
            live[j] = live[j] && (!rejects || h >= st.theta);
        }
    }

    // survivors of phase A -> this wave's queue (or straight out when the model is short)
    int n_q = 0;
    const bool last_seg_A = (tA >= a.T);
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        unsigned long long mask = __ballot(live[j]);
        int cnt = __popcll(mask);
        if (cnt == 0) continue;
        int rank = lane_rank(mask);
        if (last_seg_A) {
            uint32_t gbase = 0;
            if (lane == 0) gbase = atomicAdd(a.det_count, (uint32_t)cnt);
            gbase = __builtin_amdgcn_readfirstlane(gbase);
            if (live[j] && gbase + rank < a.capacity) {
                WbDet d;
                d.image = b;
                d.level = tile_d.level;
                d.r = (uint16_t)(r0 + wr + j);
                d.c = (uint16_t)(c0 + lane);
                d.score = hs[j];
                a.det[gbase + rank] = d;
            }
        } else if (live[j]) {
            queue[n_q + rank] = make_uint2((uint32_t)((wr + j) * 64 + lane), __float_as_uint(hs[j]));
        }
        n_q += cnt;
    }

    // ---- phase B: dense re-packed survivors, stage segments [S0,2S0), [2S0,4S0), ...
    int t_begin = tA;
    while (t_begin < a.T && n_q > 0) {
        int t_end = 2 * t_begin < a.T ? 2 * t_begin : a.T;
        const bool last = (t_end == a.T);
        int n_out = 0;
        for (int qb = 0; qb < n_q; qb += 64) {
            int i = qb + lane;
            bool alive = i < n_q;
            uint2 e = alive ? queue[i] : make_uint2(0u, 0u);
            int pos = (int)e.x;
            float h = __uint_as_float(e.y);
            int wbase = (pos >> 6) * pitch + (pos & 63);
            for (int t = t_begin; t < t_end; ++t) {
                int cnt = __popcll(__ballot(alive));
                if (cnt == 0) break;
                if (lane == 0) atomicAdd(&hist[t], (uint32_t)cnt);
                Stage<D> st;
                st.load(stages + (size_t)__builtin_amdgcn_readfirstlane(t) * SD);
                float p = st.eval(tile, wbase);
                float h2 = h + p;
                h = alive ? h2 : h;
                alive = alive && (st.theta == -INFINITY || h2 >= st.theta);
            }
            unsigned long long mask = __ballot(alive);
            int cnt = __popcll(mask);
            if (cnt) {
                int rank = lane_rank(mask);
                if (last) {
                    uint32_t gbase = 0;
                    if (lane == 0) gbase = atomicAdd(a.det_count, (uint32_t)cnt);
                    gbase = __builtin_amdgcn_readfirstlane(gbase);
                    if (alive && gbase + rank < a.capacity) {
                        WbDet d;
                        d.image = b;
                        d.level = tile_d.level;
                        d.r = (uint16_t)(r0 + (pos >> 6));
                        d.c = (uint16_t)(c0 + (pos & 63));
                        d.score = h;
                        a.det[gbase + rank] = d;
                    }
                } else if (alive) {
                    // in place: n_out + rank <= qb + lane, and this wave already holds chunk qb in registers
                    queue[n_out + rank] = make_uint2((uint32_t)pos, __float_as_uint(h));
                }
                n_out += cnt;
            }
        }
        n_q = n_out;
        t_begin = t_end;
    }

    // ---- per-stage alive counts of this tile -> global statistics
    __syncthreads();
    uint32_t *al = a.alive + ((int64_t)b * a.n_levels + tile_d.level) * a.T;
    for (int t = tid; t < a.T; t += 256) {
        uint32_t v = hist[t];
        if (v) atomicAdd(al + t, v);
    }
}

// -------------------------------------------------------------------------------------------
// DTree.predict_on_image on explicit window lists (reference training.py:84-96)
__global__ void tree_eval_kernel(const float *X, int u, int v, int C, const int32_t *rs, const int32_t *cs,
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
        float val = X[((int64_t)(r + fr) * v + (c + fc)) * C + ch];
        node = (val <= threshold[node]) ? l : (int)right[node];
    }
    out[i] = prediction[node];
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

template <int D>
int launch_depth(hipStream_t st, dim3 grid, const CascArgs &a, int rpw, size_t lds) {
    switch (rpw) {
        case 8: hipLaunchKernelGGL((cascade_kernel<D, 8>), grid, dim3(256), lds, st, a, a.stages); break;
        case 4: hipLaunchKernelGGL((cascade_kernel<D, 4>), grid, dim3(256), lds, st, a, a.stages); break;
        case 2: hipLaunchKernelGGL((cascade_kernel<D, 2>), grid, dim3(256), lds, st, a, a.stages); break;
        case 1: hipLaunchKernelGGL((cascade_kernel<D, 1>), grid, dim3(256), lds, st, a, a.stages); break;
        default: wb_set_error("cascade: bad rows-per-wave %d", rpw); return WB_ERR_INVALID;
    }
    WB_HIP_CHECK(hipGetLastError());
    return WB_OK;
}

template <int D, int RPW> int set_lds_attr() {
    WB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cascade_kernel<D, RPW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return WB_OK;
}

}  // namespace

int wb_cascade_prepare(int depth, int rpw) {
#define WB_CASE(D, R) if (depth == D && rpw == R) return set_lds_attr<D, R>();
    WB_CASE(1, 8) WB_CASE(1, 4) WB_CASE(1, 2) WB_CASE(1, 1)
    WB_CASE(2, 8) WB_CASE(2, 4) WB_CASE(2, 2) WB_CASE(2, 1)
    WB_CASE(3, 8) WB_CASE(3, 4) WB_CASE(3, 2) WB_CASE(3, 1)
#undef WB_CASE
    wb_set_error("cascade: no kernel for depth %d / rows-per-wave %d", depth, rpw);
    return WB_ERR_UNSUPPORTED;
}

extern "C" int wb_cascade_launch(void *stream, const WbModel *model, const float *chn, int64_t chn_stride,
                                 int layout, int batch, const WbLevel *levels, int n_levels,
                                 const WbTile *tiles, int n_tiles, WbDet *det, uint32_t *det_count,
                                 uint32_t capacity, uint32_t *alive) {
    WB_REQUIRE(model && chn && levels && tiles && det_count && alive, "wb_cascade_launch: null pointer");
    WB_REQUIRE(det || capacity == 0, "wb_cascade_launch: det is null but capacity > 0");
    WB_REQUIRE(batch >= 1 && batch <= 65535, "wb_cascade_launch: batch %d out of range", batch);
    WB_REQUIRE(n_levels >= 1 && n_tiles >= 1, "wb_cascade_launch: empty launch");
    WB_REQUIRE(layout == WB_LAYOUT_PLANAR || layout == WB_LAYOUT_HWC, "wb_cascade_launch: bad layout %d", layout);
    CascArgs a;
    a.chn = chn;
    a.chn_stride = chn_stride;
    a.layout = layout;
    a.levels = levels;
    a.tiles = tiles;
    a.n_levels = n_levels;
    a.stages = model->stages_dev;
    a.T = model->n_stages;
    a.m = model->m;
    a.n = model->n;
    a.C = model->C;
    a.lds_rows = model->lds_rows;
    a.lds_pitch = model->lds_pitch;
    a.det = det;
    a.det_count = det_count;
    a.capacity = capacity;
    a.alive = alive;
    dim3 grid((unsigned)n_tiles, (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    switch (model->depth) {
        case 1: return launch_depth<1>(st, grid, a, model->rpw, (size_t)model->lds_bytes);
        case 2: return launch_depth<2>(st, grid, a, model->rpw, (size_t)model->lds_bytes);
        case 3: return launch_depth<3>(st, grid, a, model->rpw, (size_t)model->lds_bytes);
    }
    wb_set_error("wb_cascade_launch: model depth %d has no kernel", model->depth);
    return WB_ERR_UNSUPPORTED;
}

extern "C" int wb_tree_eval_launch(void *stream, const float *X, int u, int v, int C, const int32_t *rs,
                                   const int32_t *cs, int64_t n_pos, const uint8_t *feature,
                                   const float *threshold, const int8_t *left, const int8_t *right,
                                   const float *prediction, int n_nodes, float *out) {
    WB_REQUIRE(n_pos >= 0, "wb_tree_eval_launch: negative count");
    if (n_pos == 0) return WB_OK;
    WB_REQUIRE(X && rs && cs && feature && threshold && left && right && prediction && out,
               "wb_tree_eval_launch: null pointer");
    WB_REQUIRE(u > 0 && v > 0 && C > 0 && n_nodes > 0 && n_nodes <= 127, "wb_tree_eval_launch: bad shape");
    int64_t blocks = (n_pos + 255) / 256;
    hipLaunchKernelGGL(tree_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, u, v, C, rs,
                       cs, n_pos, feature, threshold, left, right, prediction, n_nodes, out);
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
