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
int wb_jit_get(const int32_t *words, size_t n_words, int T, int D, int rpw, int waves, int C, int rows, int pitch, int eb,
               int lds_stages, int compiler, int allow_scratch, void **func_out);   // wb_jit.hip
void wb_jit_release(void *func);                        // wb_jit.hip

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
// BYTES == 3: the same with 16-bit ranks (WbModel::bin16_*): byte offsets into a tile of two-byte elements.
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
        off[ci] = ((fr * pitch + fc) * C + ch) * (BYTES == 3 ? 2 : 1);       // (BYTES == 3: the tile's elements are 16-bit ranks)
        const float th = t.threshold[node];
        int32_t ti = !(th >= 0.0f) ? -1 : (th >= 255.0f ? 255 : (int32_t)floorf(th));
        if (BYTES >= 2) ti = t.rank[node];
        memcpy(&thr[ci], &ti, 4);
    } else {
        off[ci] = ((ch * rows + fr) * pitch + fc) * 4;   // byte offset inside the LDS tile
        thr[ci] = t.threshold[node];
    }
    fill<BYTES>(t, t.left[node], 2 * ci + 1, d + 1, D, rows, pitch, C, off, thr, pred);
    fill<BYTES>(t, t.right[node], 2 * ci + 2, d + 1, D, rows, pitch, C, off, thr, pred);
}


// Rank tables of a SET of cascades (one model, or the members of a WbRankGroup): per channel the sorted distinct
// thresholds of every internal node, the linear cell grid over them and the per-cell base counts (wb_common.h).
struct RankTables {
    std::vector<float> S[4];
    float k[4], b[4];
    int K = 1;
    std::vector<uint8_t> lut;      // float S[4][slots], then base[4][cells]: uint8 (narrow) or uint16 (wide)
};

// wide: the 16-bit form (WB_BIN16_*: up to 1022 thresholds per channel, 512 cells, uint16 base counts)
bool build_rank_tables(const std::vector<const std::vector<TreeView> *> &sets, RankTables &rt, bool wide = false) {
    const int N = wide ? WB_BIN16_CELLS : WB_BIN_CELLS, SLOTS = wide ? WB_BIN16_SLOTS : WB_BIN_SLOTS;
    const int MAXT = wide ? WB_BIN16_MAX : WB_BIN_MAX;
    for (const std::vector<TreeView> *trees : sets)
        for (const TreeView &t : *trees)
            for (int i = 0; i < t.k; ++i)
                if (t.left[i] >= 0 && t.threshold[i] == t.threshold[i]) {
                    if (t.feature[i * 3 + 2] >= 4) return false;
                    rt.S[t.feature[i * 3 + 2]].push_back(t.threshold[i]);
                }
    for (int c = 0; c < 4; ++c) {
        std::sort(rt.S[c].begin(), rt.S[c].end());
        rt.S[c].erase(std::unique(rt.S[c].begin(), rt.S[c].end()), rt.S[c].end());     // (== merges -0.0 and 0.0)
        if ((int)rt.S[c].size() > MAXT) return false;
    }
    rt.lut.assign((size_t)4 * SLOTS * 4 + (size_t)4 * N * (wide ? 2 : 1), 0);
    float *Stab = reinterpret_cast<float *>(rt.lut.data());
    uint8_t *base8 = rt.lut.data() + (size_t)4 * SLOTS * 4;
    uint16_t *base16 = reinterpret_cast<uint16_t *>(base8);
    rt.K = 1;
    const int KMAX = wide ? 64 : 16;
    for (int c = 0; c < 4; ++c) {
        // the grid spans [lo, hi] of the channel's finite thresholds -- or, when a few far-out thresholds (1e30 next to
        // values around 10) would squeeze all the others into one cell, a trimmed range: whatever lies outside lands in
        // the two end cells (cell() clamps; it stays non-decreasing in v for any k > 0, which is all the ranks need)
        std::vector<float> fin;
        for (float v : rt.S[c])
            if (isfinite(v)) fin.push_back(v);
        std::vector<int> cnt((size_t)N, 0);
        const double trims[] = {0.0, 0.01, 0.03, 0.1, 0.25};
        bool placed = false;
        for (double q : trims) {
            float lo = INFINITY, hi = -INFINITY;
            if (!fin.empty()) {
                const size_t n = fin.size(), cut = (size_t)(q * (double)n);
                lo = fin[cut < n ? cut : n - 1];
                hi = fin[n - 1 - (cut < n ? cut : n - 1)];
                if (hi < lo) { const float t = lo; lo = hi; hi = t; }
            }
            double k = 1.0, b = 1.0;
            if (hi > lo) k = (double)(N - 2) / ((double)hi - (double)lo);
            if (lo <= hi) b = 1.0 - (double)lo * k;
            rt.k[c] = (float)k;
            rt.b[c] = (float)b;
            if (!(isfinite(rt.k[c]) && isfinite(rt.b[c]) && rt.k[c] > 0.0f)) continue;
            std::fill(cnt.begin(), cnt.end(), 0);
            int worst = 0;
            for (float v : rt.S[c]) {
                const int at = ++cnt[bin_cell(v, rt.k[c], rt.b[c], N)];    // non-decreasing in v
                worst = at > worst ? at : worst;
            }
            if (worst <= KMAX) { placed = true; break; }
        }
        if (!placed) return false;
        int run = 0;
        for (int j = 0; j < N; ++j) {
            if (wide)
                base16[(size_t)c * N + j] = (uint16_t)run;
            else
                base8[(size_t)c * N + j] = (uint8_t)run;
            run += cnt[j];
            if (cnt[j] > rt.K) rt.K = cnt[j];
        }
        for (int j = 0; j < SLOTS; ++j) Stab[c * SLOTS + j] = j < (int)rt.S[c].size() ? rt.S[c][j] : INFINITY;
    }
    return rt.K <= KMAX;
}

// rank[node] = index of the node's threshold in its channel's table (-1: leaf or NaN threshold); sets TreeView::rank
void assign_ranks(std::vector<TreeView> &trees, const int32_t *node_off, const RankTables &rt, std::vector<int32_t> &rank) {
    for (size_t s = 0; s < trees.size(); ++s) {
        for (int i = 0; i < trees[s].k; ++i) {
            const float th = trees[s].threshold[i];
            if (trees[s].left[i] < 0 || th != th) continue;
            const std::vector<float> &v = rt.S[trees[s].feature[i * 3 + 2]];
            rank[(size_t)node_off[s] + i] = (int32_t)(std::lower_bound(v.begin(), v.end(), th) - v.begin());
        }
        trees[s].rank = rank.data() + node_off[s];
    }
}

// the stage records of a cascade for byte tiles of threshold ranks (fill<2>): (n_stages + G) records of SD dwords
void pack_rank_stages(const std::vector<TreeView> &trees, const float *theta, int D, int rows, int pitch, int C, int SD, int G,
                      std::vector<int32_t> &packed, bool wide = false) {
    const int NI = (1 << D) - 1, NL = 1 << D, n_stages = (int)trees.size();
    packed.assign((size_t)(n_stages + G) * SD, 0);
    for (int s = n_stages; s < n_stages + G; ++s) reinterpret_cast<float *>(packed.data() + (size_t)s * SD)[2 * NI + NL] = -INFINITY;
    for (int s = 0; s < n_stages; ++s) {
        int32_t *rec = packed.data() + (size_t)s * SD;
        if (wide)
            fill<3>(trees[s], 0, 0, 0, D, rows, pitch, C, rec, reinterpret_cast<float *>(rec + NI), reinterpret_cast<float *>(rec + 2 * NI));
        else
            fill<2>(trees[s], 0, 0, 0, D, rows, pitch, C, rec, reinterpret_cast<float *>(rec + NI), reinterpret_cast<float *>(rec + 2 * NI));
        reinterpret_cast<float *>(rec)[2 * NI + NL] = theta[s];
    }
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
        // (layout: wb_cascade_tile.h, wb_lds_stab_off; 256 bytes of control words behind the stage mirror)
        M->lds_bytes = ((C * M->lds_rows * M->lds_pitch * 4 + wb_casc_qcap(M->tile_rows, waves) * 8 + n_stages * 4 + 15) & ~15) +
                       M->lds_stages * WB_STAGE_DWORDS(D) * 4 + 256;
        M->lds_bytes_u8 = ((((C * M->lds_rows * M->lds_pitch + 15) & ~15) + wb_casc_qcap(M->tile_rows, waves) * 8 + n_stages * 4 + 15) & ~15) +
                          M->lds_stages * WB_STAGE_DWORDS(D) * 4 + 256;
        M->lds_bytes_u16 = ((((C * M->lds_rows * M->lds_pitch * 2 + 15) & ~15) + wb_casc_qcap(M->tile_rows, waves) * 8 + n_stages * 4 + 15) & ~15) +
                           M->lds_stages * WB_STAGE_DWORDS(D) * 4 + 256;
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
    RankTables rt;
    {
        std::vector<const std::vector<TreeView> *> all = {&trees};
        if (C == 4 && n_stages > 0 && getenv("WB_NO_RANKS") == nullptr && build_rank_tables(all, rt)) {
            assign_ranks(trees, node_off, rt, rank);
            M->bin_ok = 1;
            M->bin_cells = WB_BIN_CELLS;
            M->bin_iters = rt.K;
            M->bin_lut_vec = (int)(rt.lut.size() / 16);
            for (int c = 0; c < 4; ++c) {
                M->bin_k[c] = rt.k[c];
                M->bin_b[c] = rt.b[c];
            }
        }
    }
    const std::vector<uint8_t> &lut = rt.lut;
    // ... and the 16-bit form (WB_DTYPE_RANK16): for cascades whose thresholds do not fit a byte's ranks -- built for every
    // model that qualifies (a few KiB), used by the engine when the 8-bit tables are not there
    std::vector<int32_t> rank16((size_t)(n_stages ? node_off[n_stages] : 0), -1);
    RankTables rt16;
    std::vector<int32_t> pack16;
    {
        std::vector<const std::vector<TreeView> *> all = {&trees};
        if (C == 4 && n_stages > 0 && getenv("WB_NO_RANKS") == nullptr && build_rank_tables(all, rt16, true)) {
            std::vector<TreeView> trees16 = trees;
            assign_ranks(trees16, node_off, rt16, rank16);
            pack_rank_stages(trees16, theta, D, M->lds_rows, M->lds_pitch, C, WB_STAGE_DWORDS(D), wb_cascade_group(D), pack16, true);
            M->bin16_ok = 1;
            M->bin16_iters = rt16.K;
            for (int c = 0; c < 4; ++c) {
                M->bin16_k[c] = rt16.k[c];
                M->bin16_b[c] = rt16.b[c];
            }
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
    {   // host copy of the trees as the caller gave them (wb_rankgroup_create)
        const int n_nodes = n_stages ? node_off[n_stages] : 0;
        auto dup = [](const void *src, size_t bytes) {
            void *p = malloc(bytes ? bytes : 4);
            if (bytes) memcpy(p, src, bytes);
            return p;
        };
        const int32_t zero = 0;
        M->n_nodes = n_nodes;
        M->h_node_off = static_cast<int32_t *>(dup(n_stages ? node_off : &zero, (size_t)(n_stages + 1) * 4));
        M->h_feature = static_cast<uint8_t *>(dup(feature, (size_t)n_nodes * 3));
        M->h_threshold = static_cast<float *>(dup(threshold, (size_t)n_nodes * 4));
        M->h_prediction = static_cast<float *>(dup(prediction, (size_t)n_nodes * 4));
        M->h_left = static_cast<int8_t *>(dup(left, (size_t)n_nodes));
        M->h_right = static_cast<int8_t *>(dup(right, (size_t)n_nodes));
        M->h_theta = static_cast<float *>(dup(theta, (size_t)n_stages * 4));
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
        if (M->bin16_ok) {
            M->stages_bin16_host = static_cast<int32_t *>(malloc(pack16.size() * 4 + 4));
            memcpy(M->stages_bin16_host, pack16.data(), pack16.size() * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&M->stages_bin16_dev, pack16.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(M->stages_bin16_dev, pack16.data(), pack16.size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **)&M->bin16_lut_dev, rt16.lut.size());
            if (e == hipSuccess) e = hipMemcpy(M->bin16_lut_dev, rt16.lut.data(), rt16.lut.size(), hipMemcpyHostToDevice);
        }
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
    if (model->proxy) {
        wb_set_error("wb_model_destroy: this handle is a member view of a rank group (wb_rankgroup_destroy frees it)");
        return WB_ERR_INVALID;
    }
    void *h[] = {model->h_node_off, model->h_feature, model->h_threshold, model->h_prediction, model->h_left, model->h_right, model->h_theta};
    for (void *p : h) free(p);
    if (model->stages_dev) (void)hipFree(model->stages_dev);
    if (model->stages_u8_dev) (void)hipFree(model->stages_u8_dev);
    if (model->stages_bin_dev) (void)hipFree(model->stages_bin_dev);
    if (model->bin_lut_dev) (void)hipFree(model->bin_lut_dev);
    if (model->stages_bin16_dev) (void)hipFree(model->stages_bin16_dev);
    if (model->bin16_lut_dev) (void)hipFree(model->bin16_lut_dev);
    free(model->stages_bin16_host);
    free(model->stages_u8_host);
    free(model->stages_bin_host);
    wb_jit_release(model->jit_u8);
    wb_jit_release(model->jit_bin);
    wb_jit_release(model->jit_bin16);
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
    info->specialized = (model->jit_u8 ? 1 : 0) | (model->jit_bin ? 2 : 0) | (model->jit_bin16 ? 4 : 0);
    info->rank16_ok = model->bin16_ok;
    return WB_OK;
}

// The model-specialised kernel for one kind of byte tile (wb_jit.hip): compiled with hiprtc on first use (a couple of
// seconds), then taken from the process / disk cache.  wb_cascade_launch uses it from then on for that channel dtype.
// The loaded specialised kernels of a model off (0) or on (1) for wb_cascade_launch: off, the generic kernel scans -- what a
// caller needs to cross-check a specialised kernel on its own data (engine.py: _live_check), or to retire one it distrusts.
extern "C" int wb_model_use_specialized(WbModel *model, int enable) {
    WB_REQUIRE(model, "wb_model_use_specialized: null model");
    model->jit_off = enable ? 0 : 1;
    return WB_OK;
}

// A specialised kernel is run-time compiled code for ONE model: before it is trusted it scans a synthetic two-level
// pyramid of byte tiles several times and must give, every time, exactly what the generic kernel of the library gives on
// the same bytes: per-stage alive counts and the detection records (window, score bits).  The levels hold tiles of every
// kind the kernel distinguishes -- regions of different byte statistics, so that under most cascades some tiles keep more
// windows than the capped queue holds, some a few hundred, some a handful --, a ragged right and bottom edge, and a level
// of a single partial tile.  Round 4 (profiles/r04/jit_selftest.txt): the code one hiprtc produced for some cascades of
// depth-3 trees gave wrong records on nine scans of ten; nothing in the source explains it (the toolkit's own compiler, or
// any of four unrelated build switches, gave bit-exact kernels).  A kernel that fails is not used: WB_ERR_UNSUPPORTED,
// the model stays on the generic kernel.  WB_JIT_SELFTEST=<passes> (default 6; 0 = skip the test).
static int jit_selftest(WbModel *model, int chn_dtype, void **slot) {
    static const int passes = getenv("WB_JIT_SELFTEST") ? atoi(getenv("WB_JIT_SELFTEST")) : 6;
    if (passes <= 0) return WB_OK;
    const int eb = chn_dtype == WB_DTYPE_RANK16 ? 2 : 1, C = model->C, m = model->m, n = model->n, T = model->n_stages, TR = model->tile_rows;
    struct Lv { int gh, gw; } lv[2] = {{2 * TR + TR / 2 + 1, 64 + 37}, {TR / 2 - 3 > 0 ? TR / 2 - 3 : 1, 24}};
    std::vector<WbLevel> levels(2);
    std::vector<WbTile> tiles;
    int64_t elems = 0, windows = 0;
    for (int l = 0; l < 2; ++l) {
        memset(&levels[l], 0, sizeof(WbLevel));
        levels[l].u = lv[l].gh + m;
        levels[l].v = lv[l].gw + n;
        levels[l].chn_off = elems;
        elems += (int64_t)levels[l].u * levels[l].v * C;
        windows += (int64_t)lv[l].gh * lv[l].gw;
        for (int ty = 0; ty * TR < lv[l].gh; ++ty)
            for (int tx = 0; tx * 64 < lv[l].gw; ++tx) tiles.push_back(WbTile{l, (uint16_t)ty, (uint16_t)tx});
    }
    // bytes: four kinds of 16 x 16 regions (the whole range, dark, middle, bright), LCG noise inside
    std::vector<uint8_t> host((size_t)elems * eb + 16, 0);
    uint32_t x = 2463534242u;
    for (int l = 0; l < 2; ++l)
        for (int r = 0; r < levels[l].u; ++r)
            for (int c = 0; c < levels[l].v; ++c)
                for (int ch = 0; ch < C; ++ch) {
                    x = x * 1664525u + 1013904223u;
                    const int kind = ((r >> 4) + 2 * (c >> 4) + l) & 3;
                    const uint32_t lo = kind == 0 ? 0u : kind == 1 ? 0u : kind == 2 ? 100u : 200u, span = kind == 0 ? 256u : kind == 1 ? 32u : kind == 2 ? 60u : 56u;
                    const uint32_t v = (lo + (x >> 16) % span) * (eb == 2 ? 3u : 1u);
                    const size_t at = (size_t)levels[l].chn_off + ((size_t)r * levels[l].v + c) * C + ch;
                    if (eb == 1) host[at] = (uint8_t)v;
                    else memcpy(&host[at * 2], &v, 2);
                }
    const uint32_t cap = (uint32_t)windows;                  // per shard: every window of the pyramid may survive in one
    uint8_t *chn = nullptr;
    WbLevel *d_levels = nullptr;
    WbTile *d_tiles = nullptr;
    WbDet *det = nullptr;
    uint32_t *ctr = nullptr;                                 // [WB_DET_SHARDS] counts, then alive [2][T]
    const size_t ctr_words = WB_DET_SHARDS + 2 * (size_t)T;
    hipStream_t st = nullptr;
    auto cleanup = [&]() {
        if (st) (void)hipStreamDestroy(st);
        (void)hipFree(chn); (void)hipFree(d_levels); (void)hipFree(d_tiles); (void)hipFree(det); (void)hipFree(ctr);
    };
#define WB_ST_CHECK(expr)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            wb_set_error("wb_model_specialize (self-test): %s: %s", #expr, hipGetErrorString(e_));   \
            cleanup();                                                                               \
            return WB_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)
    WB_ST_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    WB_ST_CHECK(hipMalloc(&chn, host.size()));
    WB_ST_CHECK(hipMalloc(&d_levels, levels.size() * sizeof(WbLevel)));
    WB_ST_CHECK(hipMalloc(&d_tiles, tiles.size() * sizeof(WbTile)));
    WB_ST_CHECK(hipMalloc(&det, (size_t)WB_DET_SHARDS * cap * sizeof(WbDet)));
    WB_ST_CHECK(hipMalloc(&ctr, ctr_words * 4));
    WB_ST_CHECK(hipMemcpyAsync(chn, host.data(), host.size(), hipMemcpyHostToDevice, st));
    WB_ST_CHECK(hipMemcpyAsync(d_levels, levels.data(), levels.size() * sizeof(WbLevel), hipMemcpyHostToDevice, st));
    WB_ST_CHECK(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(WbTile), hipMemcpyHostToDevice, st));
    struct Result {
        std::vector<uint32_t> ctr;
        std::vector<WbDet> det;
    };
    auto scan = [&](Result &out) -> int {
        hipError_t e = hipMemsetAsync(ctr, 0, ctr_words * 4, st);
        if (e != hipSuccess) return WB_ERR_HIP;
        const int rc = wb_cascade_launch(st, model, chn, chn_dtype, 0, 1, d_levels, 2, d_tiles, (int)tiles.size(), det, ctr, cap, ctr + WB_DET_SHARDS);
        if (rc != WB_OK) return rc;
        out.ctr.resize(ctr_words);
        e = hipMemcpyAsync(out.ctr.data(), ctr, ctr_words * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return WB_ERR_HIP;
        out.det.clear();
        for (int s = 0; s < WB_DET_SHARDS; ++s) {
            const uint32_t k = out.ctr[s] < cap ? out.ctr[s] : cap;
            const size_t at = out.det.size();
            out.det.resize(at + k);
            if (k && hipMemcpy(out.det.data() + at, det + (size_t)s * cap, (size_t)k * sizeof(WbDet), hipMemcpyDeviceToHost) != hipSuccess) return WB_ERR_HIP;
        }
        std::sort(out.det.begin(), out.det.end(), [](const WbDet &p, const WbDet &q) {
            if (p.level != q.level) return p.level < q.level;
            if (p.r != q.r) return p.r < q.r;
            return p.c < q.c;
        });
        return WB_OK;
    };
    Result want, got;
    void *const jf = *slot;
    *slot = nullptr;                                         // the generic kernel ...
    int rc = scan(want);
    *slot = jf;                                              // ... and the specialised one
    int bad = 0;
    uint64_t n_det = 0;
    for (int p = 0; rc == WB_OK && p < passes; ++p) {
        rc = scan(got);
        if (rc != WB_OK) break;
        bool same = got.det.size() == want.det.size() && memcmp(got.ctr.data() + WB_DET_SHARDS, want.ctr.data() + WB_DET_SHARDS, 2 * (size_t)T * 4) == 0;
        for (size_t i = 0; same && i < got.det.size(); ++i)
            same = got.det[i].level == want.det[i].level && got.det[i].r == want.det[i].r && got.det[i].c == want.det[i].c &&
                   memcmp(&got.det[i].score, &want.det[i].score, 4) == 0 && got.det[i].image == want.det[i].image;
        bad += !same;
        n_det = want.det.size();
        if (!same && getenv("WB_JIT_VERBOSE")) {              // the first records that differ, both sides
            int shown = 0;
            for (size_t i = 0; i < got.det.size() && i < want.det.size() && shown < 6; ++i) {
                const WbDet &g = got.det[i], &w = want.det[i];
                if (g.level == w.level && g.r == w.r && g.c == w.c && memcmp(&g.score, &w.score, 4) == 0) continue;
                fprintf(stderr, "[wb_jit] self-test pass %d, record %zu: specialised (level %d, r %d, c %d, score %.6g) generic (level %d, r %d, c %d, score %.6g)\n",
                        p, i, g.level, (int)g.r, (int)g.c, g.score, w.level, (int)w.r, (int)w.c, w.score);
                ++shown;
            }
            if (got.det.size() != want.det.size()) fprintf(stderr, "[wb_jit] self-test pass %d: %zu records against %zu\n", p, got.det.size(), want.det.size());
        }
    }
    cleanup();
#undef WB_ST_CHECK
    if (rc != WB_OK) {
        if (rc == WB_ERR_HIP) wb_set_error("wb_model_specialize (self-test): a HIP call failed");
        return rc;
    }
    if (getenv("WB_JIT_VERBOSE"))
        fprintf(stderr, "[wb_jit] self-test: %d of %d scans differ from the generic kernel (%llu windows, %llu detections)\n", bad, passes,
                (unsigned long long)windows, (unsigned long long)n_det);
    if (bad) {
        wb_set_error("wb_model_specialize: the specialised kernel disagreed with the generic kernel on %d of %d self-test scans; "
                     "the model stays on the generic kernel", bad, passes);
        return WB_ERR_UNSUPPORTED;
    }
    return WB_OK;
}

extern "C" int wb_model_specialize(WbModel *model, int chn_dtype) {
    WB_REQUIRE(model, "wb_model_specialize: null model");
    if (chn_dtype != WB_DTYPE_U8 && chn_dtype != WB_DTYPE_RANK8 && chn_dtype != WB_DTYPE_RANK16) {
        wb_set_error("wb_model_specialize: channel dtype %d has no specialised kernel (uint8 channels and threshold ranks do)", chn_dtype);
        return WB_ERR_UNSUPPORTED;
    }
    if (model->generic || model->n_stages == 0) {
        wb_set_error("wb_model_specialize: this model runs on the generic node-walk kernel (depth %d, %d stages)", model->depth, model->n_stages);
        return WB_ERR_UNSUPPORTED;
    }
    const bool ranks = chn_dtype == WB_DTYPE_RANK8, ranks16 = chn_dtype == WB_DTYPE_RANK16;
    if ((ranks && !model->bin_ok) || (ranks16 && !model->bin16_ok)) {
        wb_set_error("wb_model_specialize: this model has no rank tables of that width (wb_model_info: rank_ok / rank16_ok)");
        return WB_ERR_UNSUPPORTED;
    }
    void **slot = ranks16 ? &model->jit_bin16 : ranks ? &model->jit_bin : &model->jit_u8;
    if (*slot) return WB_OK;
    const int bit = ranks16 ? 4 : ranks ? 2 : 1;
    if (model->jit_refused & bit) {
        wb_set_error("wb_model_specialize: this model's specialised kernel failed its self-test earlier; it stays on the generic kernel");
        return WB_ERR_UNSUPPORTED;
    }
    // Candidates: the build of the compiler in the process (its code is what the benchmark runs on) that keeps to registers
    // and LDS, then its build with scratch memory; a candidate is used once it has passed the self-test.  With
    // WB_JIT_COMPILERS=both the toolkit's compiler (wb_jit.hip: a second hiprtc in a link-map namespace of its own) is a
    // second source of candidates, =toolkit the only one.  It is NOT on by default: its code passed where the first
    // compiler's failed (60 scans of 60), but the compiler itself, running on that namespace's private copy of libc,
    // crashed with a segmentation fault in about every second run of the full GPU test suite (never in a short process;
    // the backtrace ends in libhiprtc.so.7 -> libc of the namespace).  A model whose first-compiler build fails the
    // self-test stays on the generic kernel.
    const char *which = getenv("WB_JIT_COMPILERS");
    const char *only = which && strcmp(which, "both") == 0 ? nullptr : which ? which : "process";
    int rc = WB_ERR_UNSUPPORTED;
    char first_err[sizeof(g_err)] = "";
    for (int allow_scratch = 0; allow_scratch < 2; ++allow_scratch) {
        for (int compiler = 0; compiler < 2; ++compiler) {
            if (only && strcmp(only, compiler == 0 ? "toolkit" : "process") == 0) continue;
            if (model->jit_refused & (bit << (8 + 4 * compiler))) continue;      // (this compiler's build failed the self-test in the first round)
            rc = wb_jit_get(ranks16 ? model->stages_bin16_host : ranks ? model->stages_bin_host : model->stages_u8_host, model->stage_words,
                            model->n_stages, model->depth, model->rpw, model->waves, model->C, model->lds_rows, model->lds_pitch,
                            ranks16 ? 2 : 1, model->lds_stages, compiler, allow_scratch, slot);
            if (rc == WB_OK) {
                rc = jit_selftest(model, chn_dtype, slot);
                if (rc == WB_OK) return WB_OK;
                wb_jit_release(*slot);                     // (a build that is not used is idle: wb_jit.hip unloads idle modules when it holds too many)
                *slot = nullptr;
                if (rc == WB_ERR_UNSUPPORTED) model->jit_refused |= bit << (8 + 4 * compiler);
            }
            if (getenv("WB_JIT_VERBOSE")) fprintf(stderr, "[wb_jit] compiler %d%s: %s\n", compiler, allow_scratch ? " (scratch allowed)" : "", g_err);
            if (rc != WB_ERR_UNSUPPORTED) return rc;        // (a compiler or HIP error: report it, do not mask it with the next attempt)
            if (!first_err[0]) snprintf(first_err, sizeof(first_err), "%s", g_err);
        }
    }
    model->jit_refused |= bit;
    if (first_err[0]) wb_set_error("%s", first_err);
    return rc;
}


// -------------------------------------------------------------------------------------------
// Several cascades scanning ONE pyramid of threshold ranks (reference waldboost/__init__.py:120-124: detect(image,
// *models) computes the channels once): the rank table of every channel is built from the UNION of the members'
// thresholds -- `v <= S_k  <=>  rank(v) <= k` holds for any sorted superset of a model's thresholds -- and every member
// gets stage records whose thresholds index that union.  A member is handed out as a VIEW of its model: a WbModel that
// shares every pointer with it except the rank tables, usable wherever a model is (wb_channels_launch's rank_model -- any
// member: they hold the same table --, wb_cascade_launch with WB_DTYPE_RANK8, wb_model_specialize, wb_model_info).
struct WbRankGroup {
    int n;
    std::vector<WbModel *> views;
    uint8_t *lut_dev;
};

extern "C" int wb_rankgroup_destroy(WbRankGroup *g) {
    if (!g) return WB_OK;
    for (WbModel *v : g->views) {
        if (!v) continue;
        if (v->stages_bin_dev) (void)hipFree(v->stages_bin_dev);
        free(v->stages_bin_host);
        wb_jit_release(v->jit_bin);                         // (a view's own specialised kernels: its thresholds index the group's union)
        wb_jit_release(v->jit_u8);
        delete v;
    }
    if (g->lut_dev) (void)hipFree(g->lut_dev);
    delete g;
    return WB_OK;
}

extern "C" int wb_rankgroup_create(const WbModel *const *models, int n, WbRankGroup **out) {
    WB_REQUIRE(out, "wb_rankgroup_create: out is null");
    *out = nullptr;
    WB_REQUIRE(models && n >= 1 && n <= 64, "wb_rankgroup_create: 1..64 models");
    std::vector<std::vector<TreeView>> trees((size_t)n);
    std::vector<const std::vector<TreeView> *> sets;
    for (int i = 0; i < n; ++i) {
        const WbModel *m = models[i];
        WB_REQUIRE(m && !m->proxy, "wb_rankgroup_create: model %d is null or itself a member view", i);
        if (m->generic || m->C != 4 || m->n_stages == 0) {
            wb_set_error("wb_rankgroup_create: model %d has no rank form (node-walk kernel, %d channels, %d stages)", i, m->C, m->n_stages);
            return WB_ERR_UNSUPPORTED;
        }
        for (int s = 0; s < m->n_stages; ++s) {
            const int o = m->h_node_off[s];
            trees[i].push_back(TreeView{m->h_node_off[s + 1] - o, m->h_feature + (size_t)o * 3, m->h_threshold + o, m->h_left + o,
                                        m->h_right + o, m->h_prediction + o, nullptr});
        }
        sets.push_back(&trees[i]);
    }
    RankTables rt;
    if (getenv("WB_NO_RANKS") != nullptr || !build_rank_tables(sets, rt)) {
        wb_set_error("wb_rankgroup_create: the models' thresholds do not fit one rank table (more than %d distinct per channel)", WB_BIN_MAX);
        return WB_ERR_UNSUPPORTED;
    }
    WbRankGroup *g = new WbRankGroup();
    g->n = n;
    g->lut_dev = nullptr;
    g->views.assign((size_t)n, nullptr);
    hipError_t e = hipMalloc((void **)&g->lut_dev, rt.lut.size());
    if (e == hipSuccess) e = hipMemcpy(g->lut_dev, rt.lut.data(), rt.lut.size(), hipMemcpyHostToDevice);
    for (int i = 0; i < n && e == hipSuccess; ++i) {
        const WbModel *m = models[i];
        std::vector<int32_t> rank((size_t)m->n_nodes, -1), packed;
        assign_ranks(trees[i], m->h_node_off, rt, rank);
        pack_rank_stages(trees[i], m->h_theta, m->depth, m->lds_rows, m->lds_pitch, m->C, m->stage_dwords, wb_cascade_group(m->depth), packed);
        WbModel *v = new WbModel(*m);                       // shares every device / host pointer of the model ...
        v->proxy = 1;
        v->bin_ok = 1;                                      // ... but the rank tables: the group's
        v->bin_cells = WB_BIN_CELLS;
        v->bin_iters = rt.K;
        v->bin_lut_vec = (int)(rt.lut.size() / 16);
        for (int c = 0; c < 4; ++c) {
            v->bin_k[c] = rt.k[c];
            v->bin_b[c] = rt.b[c];
        }
        v->bin_lut_dev = g->lut_dev;
        v->jit_bin = nullptr;                               // (a specialised kernel bakes the thresholds' indices: per view)
        v->jit_u8 = nullptr;                                // (whatever a view has specialised is the view's to release)
        v->jit_off = 0;
        v->jit_refused = 0;
        v->bin16_ok = 0;                                    // (the group ranks in one byte; the member's own 16-bit tables are not the union's)
        v->jit_bin16 = nullptr;
        v->stages_bin_dev = nullptr;
        v->stages_bin_host = static_cast<int32_t *>(malloc(packed.size() * 4 + 4));
        memcpy(v->stages_bin_host, packed.data(), packed.size() * 4);
        g->views[(size_t)i] = v;
        e = hipMalloc((void **)&v->stages_bin_dev, packed.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(v->stages_bin_dev, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        wb_set_error("wb_rankgroup_create: uploading the tables failed: %s", hipGetErrorString(e));
        wb_rankgroup_destroy(g);
        return WB_ERR_HIP;
    }
    *out = g;
    return WB_OK;
}

extern "C" int wb_rankgroup_model(WbRankGroup *group, int i, WbModel **view) {
    WB_REQUIRE(group && view && i >= 0 && i < group->n, "wb_rankgroup_model: bad argument");
    *view = group->views[(size_t)i];
    return WB_OK;
}
