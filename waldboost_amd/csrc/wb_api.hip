// C-ABI glue: error reporting and the cascade model handle (host-side canonicalisation of the
// reference's flat-array decision trees into the complete-tree stage records the kernels read).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "wb_common.h"

int wb_cascade_prepare(int depth, int rpw, int waves);  // wb_cascade.hip
int wb_cascade_group(int depth);                        // stages evaluated per group
int wb_jit_get(const int32_t *words, size_t n_words, int T, int D, int rpw, int waves, int C, int rows, int pitch,
               void **func_out);   // wb_jit.hip

static thread_local char g_err[512] = "";

void wb_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *wb_last_error(void) { return g_err; }
extern "C" int wb_abi_version(void) { return WB_ABI_VERSION; }

namespace {

struct TreeView {
    int k;  // nodes
    const uint8_t *feature;
    const float *threshold;
    const int8_t *left, *right;
    const float *prediction;
    const int32_t *rank;   // binned tiles: index of the node's threshold among its channel's sorted distinct thresholds (-1: NaN)
};

// cell of the linear lookup grid a value falls into -- the host mirror of bin_cell() in wb_cascade.hip
// (fmaf is correctly rounded on both sides, so both map every float to the same cell)
inline uint32_t bin_cell(float v, float k, float b, int N) {
    const float q = fmaf(v, k, b);
    if (!(q > 0.0f)) return 0u;
    if (q >= (float)(N - 1)) return (uint32_t)(N - 1);
    return (uint32_t)q;
}

int tree_depth(const TreeView &t, int node) {
    if (t.left[node] < 0) return 0;
    int dl = tree_depth(t, t.left[node]), dr = tree_depth(t, t.right[node]);
    return 1 + (dl > dr ? dl : dr);
}

// Fill the complete depth-D tree rooted at canonical node `ci` (BFS numbering: children of i are
// 2i+1, 2i+2) from reference node `node`.  A reference leaf above depth D becomes a dummy split
// (feature offset 0, both subtrees = that leaf), which cannot change the leaf value reached.
// BYTES: records for uint8 channels -- offsets address the interleaved byte tile [row][col][C], thresholds are
// integers (stored in the float slots): for an integer pixel v, `v <= thr` is `v <= floor(thr)`; a NaN or negative
// threshold is never met (-1), anything from 255 up always (255).
// BYTES == 2: the byte tile holds threshold ranks of float32 pixels (WbModel::bin_*): the integer is the node's rank.
template <int BYTES>
void fill(const TreeView &t, int node, int ci, int d, int D, int rows, int pitch, int C, int32_t *off, float *thr,
          float *pred) {
    const int NI = (1 << D) - 1;
    if (d == D) {
        pred[ci - NI] = t.prediction[node];
        return;
    }
    if (t.left[node] < 0) {
        off[ci] = 0;
        thr[ci] = 0.0f;
        fill<BYTES>(t, node, 2 * ci + 1, d + 1, D, rows, pitch, C, off, thr, pred);
        fill<BYTES>(t, node, 2 * ci + 2, d + 1, D, rows, pitch, C, off, thr, pred);
        return;
    }
    int fr = t.feature[node * 3 + 0], fc = t.feature[node * 3 + 1], ch = t.feature[node * 3 + 2];
    if (BYTES) {
        off[ci] = (fr * pitch + fc) * C + ch;
        const float th = t.threshold[node];
        int32_t ti = !(th >= 0.0f) ? -1 : (th >= 255.0f ? 255 : (int32_t)floorf(th));
        if (BYTES == 2) ti = t.rank[node];
        memcpy(&thr[ci], &ti, 4);
    } else {
        off[ci] = ((ch * rows + fr) * pitch + fc) * 4;   // byte offset inside the LDS tile
        thr[ci] = t.threshold[node];
    }
    fill<BYTES>(t, t.left[node], 2 * ci + 1, d + 1, D, rows, pitch, C, off, thr, pred);
    fill<BYTES>(t, t.right[node], 2 * ci + 2, d + 1, D, rows, pitch, C, off, thr, pred);
}

}  // namespace

extern "C" int wb_model_create(int n_stages, const int32_t *node_off, const uint8_t *feature,
                               const float *threshold, const int8_t *left, const int8_t *right,
                               const float *prediction, const float *theta, int m, int n, int C,
                               WbModel **out) {
    WB_REQUIRE(out, "wb_model_create: out is null");
    *out = nullptr;
    WB_REQUIRE(n_stages >= 0 && n_stages <= 16384, "wb_model_create: n_stages=%d out of range", n_stages);
    WB_REQUIRE(m >= 1 && n >= 1 && C >= 1 && m <= 256 && n <= 256 && C <= 256,
               "wb_model_create: window shape (%d,%d,%d) out of range (features are uint8)", m, n, C);
    WB_REQUIRE(n_stages == 0 || (node_off && feature && threshold && left && right && prediction && theta),
               "wb_model_create: null array");

    // ---- validate the trees the way the reference walks them (training.py:84-96)
    int D = 1;
    std::vector<TreeView> trees((size_t)n_stages);
    for (int s = 0; s < n_stages; ++s) {
        int o = node_off[s], k = node_off[s + 1] - node_off[s];
        WB_REQUIRE(o >= 0 && k >= 1 && k <= 127, "wb_model_create: stage %d has %d nodes (1..127 allowed: int8 links)", s, k);
        TreeView t{k, feature + (size_t)o * 3, threshold + o, left + o, right + o, prediction + o, nullptr};
        for (int i = 0; i < k; ++i) {
            if (t.left[i] < 0) continue;
            WB_REQUIRE(t.left[i] > i && t.left[i] < k && t.right[i] > i && t.right[i] < k,
                       "wb_model_create: stage %d node %d: children (%d,%d) must satisfy parent < child < %d",
                       s, i, (int)t.left[i], (int)t.right[i], k);
            WB_REQUIRE(t.feature[i * 3] < m && t.feature[i * 3 + 1] < n && t.feature[i * 3 + 2] < C,
                       "wb_model_create: stage %d node %d: feature (%d,%d,%d) outside window (%d,%d,%d)", s, i,
                       (int)t.feature[i * 3], (int)t.feature[i * 3 + 1], (int)t.feature[i * 3 + 2], m, n, C);
        }
        int d = tree_depth(t, 0);
        if (d > D) D = d;
        trees[s] = t;
    }
    // deep trees, or windows whose LDS tile would not fit a CU: generic node-walk kernel
    const int min_lds = C * (4 + m - 1) * (WB_CASC_TC + n) * 4 + 4 * WB_CASC_TC * 8 + n_stages * 4;
    const bool generic = D > WB_CASC_MAX_DEPTH || min_lds > 150 * 1024 || getenv("WB_CASC_GENERIC") != nullptr;
    const int Dreal = D;
    if (generic) D = 1;                            // (geometry fields below are unused in generic mode)

    // ---- cascade tile geometry: the largest tile whose LDS footprint leaves two workgroups per CU
    WbModel *M = new WbModel();
    memset(M, 0, sizeof(*M));
    M->n_stages = n_stages;
    M->depth = generic ? Dreal : D;
    M->generic = generic ? 1 : 0;
    M->m = m;
    M->n = n;
    M->C = C;
    M->lds_pitch = WB_CASC_TC + n;      // tile row (64 + n - 1 pixels) + one pad column (spare slot of the tile load)
    // tile = (rpw * waves) x 64 windows; WB_CASC_RPW / WB_CASC_WAVES override the default for tuning
    const int budget = 80 * 1024;
    int rpw = 4, waves = 8;
    if (const char *e = getenv("WB_CASC_RPW")) rpw = atoi(e);
    if (const char *e = getenv("WB_CASC_WAVES")) waves = atoi(e);
    for (;; rpw >>= 1) {
        M->rpw = rpw;
        M->waves = waves;
        M->tile_rows = rpw * waves;
        M->lds_rows = M->tile_rows + m - 1;
        M->lds_stages = (n_stages * WB_STAGE_DWORDS(D) * 4 <= 16 * 1024) ? n_stages : 0;
        // (layout: wb_cascade_tile.h, wb_lds_stab_off; 128 bytes of control words behind the stage mirror)
        M->lds_bytes = ((C * M->lds_rows * M->lds_pitch * 4 + M->tile_rows * WB_CASC_TC * 8 + n_stages * 4 + 15) & ~15) +
                       M->lds_stages * WB_STAGE_DWORDS(D) * 4 + 128;
        M->lds_bytes_u8 = ((((C * M->lds_rows * M->lds_pitch + 15) & ~15) + M->tile_rows * WB_CASC_TC * 8 + n_stages * 4 + 15) & ~15) +
                          M->lds_stages * WB_STAGE_DWORDS(D) * 4 + 128;
        if (M->lds_bytes <= budget || rpw <= 1) break;
    }
    if (!generic && M->lds_bytes > 160 * 1024) {
        wb_set_error("wb_model_create: window (%d,%d,%d) with %d stages needs %d B of LDS (> 160 KiB)", m, n, C,
                     n_stages, M->lds_bytes);
        delete M;
        return WB_ERR_UNSUPPORTED;
    }
    M->stage_dwords = WB_STAGE_DWORDS(D);
    // the reference's own flat node arrays: walked by the generic cascade kernel and by the
    // per-sample cascade (wb_samples_predict_launch), so every model carries them (a few KiB)
    {
        const int n_nodes = n_stages ? node_off[n_stages] : 0;
        std::vector<int32_t> feat((size_t)n_nodes), lft((size_t)n_nodes), rgt((size_t)n_nodes);
        for (int i = 0; i < n_nodes; ++i) {
            feat[i] = feature[i * 3] | (feature[i * 3 + 1] << 8) | (feature[i * 3 + 2] << 16);
            lft[i] = left[i];
            rgt[i] = right[i];
        }
        const int32_t zero = 0;
        hipError_t e = hipSuccess;
        auto up = [&](void **dst, const void *src, size_t bytes) {
            if (e == hipSuccess) e = hipMalloc(dst, bytes ? bytes : 4);
            if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        };
        up((void **)&M->g_node_off, n_stages ? node_off : &zero, (size_t)(n_stages + 1) * 4);
        up((void **)&M->g_feat, feat.data(), (size_t)n_nodes * 4);
        up((void **)&M->g_thr, threshold, (size_t)n_nodes * 4);
        up((void **)&M->g_left, lft.data(), (size_t)n_nodes * 4);
        up((void **)&M->g_right, rgt.data(), (size_t)n_nodes * 4);
        up((void **)&M->g_pred, prediction, (size_t)n_nodes * 4);
        up((void **)&M->g_theta, theta, (size_t)n_stages * 4);
        if (e != hipSuccess) {
            wb_set_error("wb_model_create: uploading the node arrays failed: %s", hipGetErrorString(e));
            wb_model_destroy(M);
            return WB_ERR_HIP;
        }
    }
    if (generic) {
        // one thread per window, 4 x 64 windows per workgroup, features gathered from HBM/L2
        M->rpw = 1;
        M->waves = 4;
        M->tile_rows = 4;
        M->lds_rows = M->lds_pitch = 0;
        M->lds_bytes = 0;
        *out = M;
        return WB_OK;
    }

    // ---- rank tables for float32 channels (wb_common.h: WbModel::bin_*): sorted distinct thresholds per channel
    std::vector<int32_t> rank((size_t)(n_stages ? node_off[n_stages] : 0), -1);
    std::vector<uint8_t> lut;
    {
        const int N = WB_BIN_CELLS;
        bool ok = C == 4 && n_stages > 0 && getenv("WB_NO_RANKS") == nullptr;
        std::vector<float> S[4];
        if (ok) {
            for (int s = 0; s < n_stages; ++s)
                for (int i = 0; i < trees[s].k; ++i)
                    if (trees[s].left[i] >= 0 && trees[s].threshold[i] == trees[s].threshold[i])
                        S[trees[s].feature[i * 3 + 2]].push_back(trees[s].threshold[i]);
            for (int c = 0; c < 4 && ok; ++c) {
                std::sort(S[c].begin(), S[c].end());
                S[c].erase(std::unique(S[c].begin(), S[c].end()), S[c].end());     // (== merges -0.0 and 0.0)
                ok = (int)S[c].size() <= WB_BIN_MAX;
            }
        }
        int K = 1;
        if (ok) {
            lut.assign((size_t)4 * WB_BIN_SLOTS * 4 + (size_t)4 * N, 0);
            float *Stab = reinterpret_cast<float *>(lut.data());
            uint8_t *base = lut.data() + 4 * WB_BIN_SLOTS * 4;
            for (int c = 0; c < 4 && ok; ++c) {
                float lo = INFINITY, hi = -INFINITY;
                for (float v : S[c])
                    if (isfinite(v)) { lo = fminf(lo, v); hi = fmaxf(hi, v); }
                double k = 1.0, b = 1.0;
                if (hi > lo) k = (double)(N - 2) / ((double)hi - (double)lo);
                if (lo <= hi) b = 1.0 - (double)lo * k;
                M->bin_k[c] = (float)k;
                M->bin_b[c] = (float)b;
                ok = isfinite(M->bin_k[c]) && isfinite(M->bin_b[c]) && M->bin_k[c] > 0.0f;
                if (!ok) break;
                std::vector<int> cnt((size_t)N, 0);
                for (float v : S[c]) cnt[bin_cell(v, M->bin_k[c], M->bin_b[c], N)]++;    // non-decreasing in v
                int run = 0;
                for (int j = 0; j < N; ++j) {
                    base[(size_t)c * N + j] = (uint8_t)run;
                    run += cnt[j];
                    if (cnt[j] > K) K = cnt[j];
                }
                for (int j = 0; j < WB_BIN_SLOTS; ++j) Stab[c * WB_BIN_SLOTS + j] = j < (int)S[c].size() ? S[c][j] : INFINITY;
            }
            ok = ok && K <= 16;
        }
        if (ok) {
            for (int s = 0; s < n_stages; ++s) {
                for (int i = 0; i < trees[s].k; ++i) {
                    const float th = trees[s].threshold[i];
                    if (trees[s].left[i] < 0 || th != th) continue;
                    const std::vector<float> &v = S[trees[s].feature[i * 3 + 2]];
                    rank[(size_t)node_off[s] + i] = (int32_t)(std::lower_bound(v.begin(), v.end(), th) - v.begin());
                }
                trees[s].rank = rank.data() + node_off[s];
            }
            M->bin_ok = 1;
            M->bin_cells = N;
            M->bin_iters = K;
            M->bin_lut_vec = (int)(lut.size() / 16);
        }
    }

    // ---- pack and upload the stage records
    const int NI = (1 << D) - 1, NL = 1 << D, SD = M->stage_dwords;
    // G trailing no-op records (offset 0, prediction 0, theta -inf) so a group load never leaves the table
    const int G = wb_cascade_group(D);
    std::vector<int32_t> packs[3];
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<int32_t> &packed = packs[mode];
        packed.assign((size_t)(n_stages + G) * SD, 0);
        for (int s = n_stages; s < n_stages + G; ++s)
            reinterpret_cast<float *>(packed.data() + (size_t)s * SD)[2 * NI + NL] = -INFINITY;
        if (mode == 2 && !M->bin_ok) continue;
        for (int s = 0; s < n_stages; ++s) {
            int32_t *rec = packed.data() + (size_t)s * SD;
            int32_t *off = rec;
            float *thr = reinterpret_cast<float *>(rec + NI);
            float *pred = reinterpret_cast<float *>(rec + 2 * NI);
            if (mode == 2)
                fill<2>(trees[s], 0, 0, 0, D, M->lds_rows, M->lds_pitch, C, off, thr, pred);
            else if (mode == 1)
                fill<1>(trees[s], 0, 0, 0, D, M->lds_rows, M->lds_pitch, C, off, thr, pred);
            else
                fill<0>(trees[s], 0, 0, 0, D, M->lds_rows, M->lds_pitch, C, off, thr, pred);
            reinterpret_cast<float *>(rec)[2 * NI + NL] = theta[s];
        }
    }
    if (const char *path = getenv("WB_DUMP_STAGES")) {       // diagnostic: the three record tables as raw int32
        if (FILE *f = fopen(path, "wb")) {
            const int32_t hdr[4] = {n_stages + G, SD, D, M->bin_ok};
            fwrite(hdr, 4, 4, f);
            for (int mode = 0; mode < 3; ++mode) fwrite(packs[mode].data(), 4, packs[mode].size(), f);
            fclose(f);
        }
    }
    M->stage_words = packs[1].size();
    M->stages_u8_host = static_cast<int32_t *>(malloc(packs[1].size() * 4 + 4));
    memcpy(M->stages_u8_host, packs[1].data(), packs[1].size() * 4);
    if (M->bin_ok) {
        M->stages_bin_host = static_cast<int32_t *>(malloc(packs[2].size() * 4 + 4));
        memcpy(M->stages_bin_host, packs[2].data(), packs[2].size() * 4);
    }
    {
        const std::vector<int32_t> &packed = packs[0];
        hipError_t e = hipMalloc((void **)&M->stages_dev, packed.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(M->stages_dev, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&M->stages_u8_dev, packs[1].size() * 4);
        if (e == hipSuccess) e = hipMemcpy(M->stages_u8_dev, packs[1].data(), packs[1].size() * 4, hipMemcpyHostToDevice);
        if (M->bin_ok) {
            if (e == hipSuccess) e = hipMalloc((void **)&M->stages_bin_dev, packs[2].size() * 4);
            if (e == hipSuccess) e = hipMemcpy(M->stages_bin_dev, packs[2].data(), packs[2].size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **)&M->bin_lut_dev, lut.size());
            if (e == hipSuccess) e = hipMemcpy(M->bin_lut_dev, lut.data(), lut.size(), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            wb_set_error("wb_model_create: uploading %zu stage bytes failed: %s", packed.size() * 4, hipGetErrorString(e));
            wb_model_destroy(M);
            return WB_ERR_HIP;
        }
    }
    int rc = wb_cascade_prepare(D, M->rpw, M->waves);
    if (rc != WB_OK) {
        wb_model_destroy(M);
        return rc;
    }
    *out = M;
    return WB_OK;
}

extern "C" int wb_model_destroy(WbModel *model) {
    if (!model) return WB_OK;
    if (model->stages_dev) (void)hipFree(model->stages_dev);
    if (model->stages_u8_dev) (void)hipFree(model->stages_u8_dev);
    if (model->stages_bin_dev) (void)hipFree(model->stages_bin_dev);
    if (model->bin_lut_dev) (void)hipFree(model->bin_lut_dev);
    free(model->stages_u8_host);
    free(model->stages_bin_host);
    void *g[] = {model->g_node_off, model->g_feat, model->g_thr, model->g_left, model->g_right, model->g_pred, model->g_theta};
    for (void *p : g)
        if (p) (void)hipFree(p);
    delete model;
    return WB_OK;
}

extern "C" int wb_model_info(const WbModel *model, WbModelInfo *info) {
    WB_REQUIRE(model && info, "wb_model_info: null pointer");
    info->n_stages = model->n_stages;
    info->depth = model->depth;
    info->m = model->m;
    info->n = model->n;
    info->C = model->C;
    info->tile_rows = model->tile_rows;
    info->tile_cols = WB_CASC_TC;
    info->lds_bytes = model->lds_bytes;
    info->rank_ok = model->bin_ok;
    info->specialized = (model->jit_u8 ? 1 : 0) | (model->jit_bin ? 2 : 0);
    return WB_OK;
}

// The model-specialised kernel for one kind of byte tile (wb_jit.hip): compiled with hiprtc on first use (a couple of
// seconds), then taken from the process / disk cache.  wb_cascade_launch uses it from then on for that channel dtype.
extern "C" int wb_model_specialize(WbModel *model, int chn_dtype) {
    WB_REQUIRE(model, "wb_model_specialize: null model");
    if (chn_dtype != WB_DTYPE_U8 && chn_dtype != WB_DTYPE_RANK8) {
        wb_set_error("wb_model_specialize: channel dtype %d has no specialised kernel (uint8 channels and threshold ranks do)", chn_dtype);
        return WB_ERR_UNSUPPORTED;
    }
    if (model->generic || model->n_stages == 0) {
        wb_set_error("wb_model_specialize: this model runs on the generic node-walk kernel (depth %d, %d stages)", model->depth, model->n_stages);
        return WB_ERR_UNSUPPORTED;
    }
    const bool ranks = chn_dtype == WB_DTYPE_RANK8;
    if (ranks && !model->bin_ok) {
        wb_set_error("wb_model_specialize: this model has no rank tables (wb_model_info: rank_ok)");
        return WB_ERR_UNSUPPORTED;
    }
    void **slot = ranks ? &model->jit_bin : &model->jit_u8;
    if (*slot) return WB_OK;
    if (!model->lds_stages) {
        wb_set_error("wb_model_specialize: %d stages exceed the LDS mirror of the stage table the specialised kernel reads leaf values from", model->n_stages);
        return WB_ERR_UNSUPPORTED;
    }
    return wb_jit_get(ranks ? model->stages_bin_host : model->stages_u8_host, model->stage_words, model->n_stages,
                      model->depth, model->rpw, model->waves, model->C, model->lds_rows, model->lds_pitch, slot);
}
