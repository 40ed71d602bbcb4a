// Internal helpers shared by the gfx950 kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/waldboost_hip.h"

#define WB_WAVE 64

// ---- error plumbing (thread-local message, never throws across the ABI) ----
void wb_set_error(const char *fmt, ...);

#define WB_HIP_CHECK(expr)                                                             \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            wb_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return WB_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

#define WB_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            wb_set_error(__VA_ARGS__);   \
            return WB_ERR_INVALID;       \
        }                                \
    } while (0)

// ---- order-preserving uint32 keys for min/max atomics ----
// uint8 pixels use their value; float32 uses the usual sign-flip encoding so that
// unsigned comparison of keys == float comparison of values.
__host__ __device__ inline uint32_t wb_f32_key(float f) {
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float wb_key_f32(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// ---- how a float64-held image dtype is cast back after the resize / pooled in the octaves (WB_DTYPE_F64 .. WB_DTYPE_F16) ----
#define WB_CAST_NONE 0     // float64
#define WB_CAST_TRUNC 1    // integers: truncation toward zero
#define WB_CAST_BOOL 2     // bool: != 0
#define WB_CAST_F16 3      // float16: round to nearest even
inline int wb_cast_mode(int dtype) {
    switch (dtype) {
        case WB_DTYPE_F64: return WB_CAST_NONE;
        case WB_DTYPE_BOOL: return WB_CAST_BOOL;
        case WB_DTYPE_F16: return WB_CAST_F16;
        default: return WB_CAST_TRUNC;
    }
}
inline bool wb_dtype_held_f64(int dtype) { return dtype == WB_DTYPE_F64 || (dtype >= WB_DTYPE_I8 && dtype <= WB_DTYPE_F16); }
// a double rounded to binary16 (round to nearest even, one rounding): to float32 toward zero with the sticky bit kept
// in the last place (round to odd), then the hardware's float32 -> float16 conversion
__device__ inline double wb_round_f16(double x) {
    float f = __double2float_rz(x);
    if ((double)f != x) f = __uint_as_float(__float_as_uint(f) | 1u);     // (inexact; a NaN stays a NaN)
    return (double)(float)(_Float16)f;
}

// ---- cascade geometry ----
#define WB_CASC_TC 64        // windows per tile row = one per lane
#define WB_CASC_MAX_DEPTH 3
#define WB_BIN_MAX 254       // distinct thresholds per channel a binned tile can rank in one byte: ranks 0..254, 255 = NaN
                             // pixel; S[254] and S[255] of a channel's table are then always the +inf padding, so the
                             // channel kernel may read S[r] and S[r + 1] together for every rank r <= 254
#define WB_BIN_SLOTS 256     // entries of a channel's sorted threshold table
#define WB_BIN_CELLS 2048    // cells of a channel's lookup grid
#define WB_BIN_LUT_BYTES (4 * WB_BIN_SLOTS * 4 + 4 * WB_BIN_CELLS)   // float S[4][256], then uint8 base[4][N]
// ... and in two bytes (WB_DTYPE_RANK16): cascades with more distinct thresholds per channel than a byte ranks -- long soft
// cascades (reference __init__.py:230-269 appends stages without bound), deep trees.  The tables still live in LDS while
// a channel tile is ranked, so the count is bounded by that: 1020 thresholds per channel (S[1020..1023] = +inf padding: the
// channel kernel reads S[r .. r + 3] together), 512 grid cells with 16-bit base counts = 20 KiB.
#define WB_BIN16_MAX 1020
#define WB_BIN16_SLOTS 1024
#define WB_BIN16_CELLS 512
#define WB_BIN16_LUT_BYTES (4 * WB_BIN16_SLOTS * 4 + 4 * WB_BIN16_CELLS * 2)   // float S[4][1024], then uint16 base[4][512]

// entries of the cascade workgroup's survivor queue (wb_cascade_tile.h: the same definition, for the hiprtc build)
#ifndef WB_CASC_QCAP_DEFINED
#define WB_CASC_QCAP_DEFINED
#ifndef WB_CASC_QFULL
#define WB_CASC_QFULL 0
#endif
__host__ __device__ constexpr int wb_casc_qcap(int TR, int WAVES) {
    return (WB_CASC_QFULL || TR * 64 < 64 * WAVES + 512) ? TR * 64 : 64 * WAVES + 512;
}
#endif
#ifndef WB_CASC_END_BARRIER
#define WB_CASC_END_BARRIER 0
#endif

// The canonical stage record the cascade kernels read with scalar loads:
//   int   off[NI]   LDS byte offset of each internal node's feature (BFS order)
//   float thr[NI]
//   float pred[NL]  leaf predictions, left to right
//   float theta
// NI = 2^D - 1, NL = 2^D; padded to WB_STAGE_DWORDS(D) dwords.
#define WB_STAGE_NI(D) ((1 << (D)) - 1)
#define WB_STAGE_NL(D) (1 << (D))
#define WB_STAGE_DWORDS(D) ((((2 * WB_STAGE_NI(D) + WB_STAGE_NL(D) + 1) + 3) / 4) * 4)

struct WbModel {
    int n_stages, depth, m, n, C;
    int rpw;          // window rows per wave in the cascade tile
    int waves;        // wavefronts per workgroup
    int tile_rows;    // = rpw * waves
    int lds_rows;     // tile_rows + m - 1
    int lds_pitch;    // WB_CASC_TC + n - 1 pixels + 1 pad column
    int lds_bytes;
    int lds_stages;   // stage records mirrored in LDS (n_stages if the table is <= 16 KiB, else 0)
    int stage_dwords;
    int32_t *stages_dev;        // (n_stages + G) stage records with LDS byte offsets (planar float32 tile)
    int32_t *stages_u8_dev;     // the same for uint8 channels: offsets into the interleaved byte tile, integer thresholds
    int lds_bytes_u8;           // dynamic LDS of the kernel on uint8 channels
    // float32 channels as threshold RANKS (written by the channel kernel, scanned by the uint8 cascade tile): per
    // channel the model's distinct thresholds are sorted, a pixel is replaced by the number of them below it (its
    // rank, one byte), and a node test `v <= thr` becomes `rank(v) <= index(thr)` -- the same decision for every
    // float, in a quarter of the bytes
    int bin_ok;                 // 1 = every channel has <= WB_BIN_MAX distinct thresholds and the tables fit
    int bin_cells;              // N: cells of the linear lookup grid per channel
    int bin_iters;              // K: most thresholds that share one cell (refinement steps per pixel)
    float bin_k[4], bin_b[4];   // cell(v) = trunc(clamp(fma(v, k[c], b[c]), 0, N - 1))
    int bin_lut_vec;            // size of the table block in 16-byte units
    uint8_t *bin_lut_dev;       // float S[4][256] (sorted thresholds, +inf padded), then uint8 base[4][N]
    int32_t *stages_bin_dev;    // stage records for the binned tile: byte-tile offsets, thresholds = ranks
    // the same with 16-bit ranks (WB_DTYPE_RANK16): up to WB_BIN16_MAX distinct thresholds per channel
    int bin16_ok, bin16_iters;
    float bin16_k[4], bin16_b[4];
    uint8_t *bin16_lut_dev;     // float S[4][WB_BIN16_SLOTS], then uint16 base[4][WB_BIN16_CELLS]
    int32_t *stages_bin16_dev;  // stage records for the 16-bit tile [rows][pitch][C] x 2 bytes: byte offsets, thresholds = ranks
    int32_t *stages_bin16_host;
    int lds_bytes_u16;          // dynamic LDS of the kernel on the 16-bit tile
    void *jit_bin16;
    // host copy of the caller's tree arrays (wb_rankgroup_create derives stage records for a shared rank table from them)
    int n_nodes;
    int32_t *h_node_off;        // [n_stages + 1]
    uint8_t *h_feature;         // [n_nodes][3]
    float *h_threshold, *h_prediction, *h_theta;
    int8_t *h_left, *h_right;
    int proxy;                  // 1 = a member view of a WbRankGroup: shares every pointer but the rank tables with its model
    // model-specialised kernels (wb_jit.hip, wb_model_specialize): hipFunction_t per byte-tile stage table, or null
    int32_t *stages_u8_host, *stages_bin_host;   // host copies of the two byte-tile tables the generator bakes in
    size_t stage_words;                          // (n_stages + G) * stage_dwords
    void *jit_u8, *jit_bin;
    int jit_off;                // 1 = wb_cascade_launch ignores the loaded specialised kernels (wb_model_use_specialized)
    int jit_refused;            // bit per byte-tile kind (1 uint8, 2 ranks, 4 16-bit ranks): its specialised kernel was built and failed the self-test
    // trees deeper than WB_CASC_MAX_DEPTH: generic node-walk kernel on the reference's own flat arrays
    int generic;                // 1 = use cascade_generic_kernel
    int32_t *g_node_off;        // [n_stages + 1]
    int32_t *g_feat;            // [n_nodes] row | col << 8 | channel << 16
    float *g_thr;               // [n_nodes]
    int32_t *g_left, *g_right;  // [n_nodes] child index inside the stage's tree, -1 on leaves
    float *g_pred;              // [n_nodes]
    float *g_theta;             // [n_stages]
};

// ---- threshold ranks of float32 channel values (WbModel::bin_*, WB_DTYPE_RANK8) ----
// cell of a channel's lookup grid (host mirror: bin_cell() in wb_api.hip): non-decreasing in v
__device__ inline uint32_t wb_bin_cell(float v, float k, float b, float top = (float)(WB_BIN_CELLS - 1)) {
    const float q = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, k, b), 0.0f, top);
    return (uint32_t)q;
}
// rank of v among the channel's sorted distinct thresholds S (+inf padded): the number of them below v.
// base[cell] counts the thresholds of lower cells (all below v); the thresholds of v's own cell follow in S, in
// order, at most K of them; thresholds of higher cells are above v.  (NaN: the caller substitutes 255.)
__device__ inline uint32_t wb_bin_rank(float v, float k, float b, const uint8_t *base, const float *S, int K) {
    uint32_t r = base[wb_bin_cell(v, k, b)];
    for (int i = 0; i < K; ++i) r += v > S[r] ? 1u : 0u;
    return r;
}
