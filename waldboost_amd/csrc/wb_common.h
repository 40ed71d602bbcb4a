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

// ---- cascade geometry ----
#define WB_CASC_TC 64        // windows per tile row = one per lane
#define WB_CASC_MAX_DEPTH 3

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
    // trees deeper than WB_CASC_MAX_DEPTH: generic node-walk kernel on the reference's own flat arrays
    int generic;                // 1 = use cascade_generic_kernel
    int32_t *g_node_off;        // [n_stages + 1]
    int32_t *g_feat;            // [n_nodes] row | col << 8 | channel << 16
    float *g_thr;               // [n_nodes]
    int32_t *g_left, *g_right;  // [n_nodes] child index inside the stage's tree, -1 on leaves
    float *g_pred;              // [n_nodes]
    float *g_theta;             // [n_stages]
};
