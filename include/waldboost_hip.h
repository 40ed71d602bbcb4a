/*
 * waldboost_hip.h -- C ABI of the MI355X (gfx950) waldboost detection hot path.
 *
 * The reference (RomanJuranek/waldboost) is pure Python and has no FFI/plugin
 * seam: the hot path sits behind plain Python calls.  This ABI is the seam the
 * build puts *beneath* that Python surface; each entry point names the
 * reference code it replaces.  All pointers marked "dev" are device pointers
 * supplied by the caller (e.g. torch ``data_ptr()``); no entry point allocates
 * result memory, synchronises the device, or throws.  Every function returns
 * WB_OK (0) or a negative error code; wb_last_error() gives the thread-local
 * message.  ``stream`` is a hipStream_t passed as void* (NULL = default stream).
 *
 *   reference waldboost/channels.py:93-101 (_image_octaves) + :55-64 (avg_pool_2)
 *        -> wb_octaves_launch
 *   reference waldboost/channels.py:111-146 (channel_pyramid: resize :132,
 *        grad_hist :40-52, gradients :16-21, avg_pool_2 :55-64, smooth :78-90)
 *        -> wb_channels_launch
 *   reference waldboost/fpga/channels.py:5-67 (grad_hist_4_u1, grad_mag_u1) and
 *        waldboost/channels.py:30-37 (grad_mag) as channel_opts["channels"]
 *        -> wb_channels_launch with channel_func = WB_CHN_*
 *   reference waldboost/channels.py:40-52, :30-37 called directly with non-default arguments
 *        -> wb_grad_hist_launch, wb_grad_mag_launch
 *   reference waldboost/model.py:62-67,272-283 (Model ctor/append) +
 *        waldboost/training.py:24-31 (DTree.__init__)
 *        -> wb_model_create / wb_model_destroy / wb_model_info
 *   reference waldboost/model.py:216-259 (Model.predict_on_image) +
 *        waldboost/training.py:84-96 (DTree.predict_on_image)
 *        -> wb_cascade_launch
 *   reference waldboost/training.py:84-96 called on explicit (rs, cs) lists
 *        -> wb_tree_eval_launch
 *   reference waldboost/model.py:136-147 (Model.get_boxes)
 *        -> wb_boxes_launch
 *   reference waldboost/model.py:173-179 (Model.detect: per-level results concatenated for the caller)
 *        -> wb_det_pack_launch
 *   reference waldboost/model.py:136-147 + :173-179 (get_boxes on the concatenated, ordered detections)
 *        -> wb_det_finish_launch; wb_det_finish_sorted_launch (the ordering on the device too);
 *           wb_det_order_batch_launch (the same per image of a batch)
 *   reference waldboost/samples.py:14-43 (gather_samples), waldboost/model.py:181-214 (Model.predict),
 *        waldboost/training.py:73-83 (DTree.apply/predict): the training-time callers of the hot path
 *        -> wb_gather_samples_launch, wb_samples_predict_launch, wb_tree_apply_launch
 */
#ifndef WALDBOOST_HIP_H
#define WALDBOOST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WB_ABI_VERSION 8

#define WB_OK 0
#define WB_ERR_INVALID (-1)     /* bad argument / malformed model */
#define WB_ERR_HIP (-2)         /* a HIP runtime call failed */
#define WB_ERR_UNSUPPORTED (-3) /* valid input this build has no kernel for */

#define WB_DTYPE_U8 0
#define WB_DTYPE_F32 1
/* Channel buffers only: one byte per element holding the RANK of the float32 channel value among the distinct
 * thresholds a given model tests on that channel (the number of them below the value; 255 for NaN).  For that
 * model `v <= threshold` (training.py:92) is `rank(v) <= index(threshold)` for every float v, so the cascade
 * decides exactly as on the float32 channels while reading a quarter of the bytes.  Written by
 * wb_channels_launch(rank_model, rank), read by wb_cascade_launch(chn_dtype = WB_DTYPE_RANK8) with the SAME
 * model.  Available when the model has 4 channels and at most 254 distinct thresholds per channel
 * (WbModelInfo.rank_ok). */
#define WB_DTYPE_RANK8 2
/* The same in TWO bytes per element (ABI 7): for models with up to 1020 distinct thresholds per channel -- long soft
 * cascades (reference __init__.py:230-269 appends stages without bound), deep trees -- whose channels would otherwise
 * stay float32.  65535 marks a NaN pixel.  Written by wb_channels_launch_x(rank_dtype = WB_DTYPE_RANK16), read by
 * wb_cascade_launch(chn_dtype = WB_DTYPE_RANK16); available when WbModelInfo.rank16_ok. */
#define WB_DTYPE_RANK16 13
/* Image buffers only: the other dtypes the reference's channel_pyramid accepts (channels.py:122 keeps
 * image.dtype).  All of them are held as float64 elements (exact for these integer types); the code tells the
 * octave kernel how avg_pool_2's adds wrap (integers wrap modulo 2^bits in the array's dtype, channels.py:61-64)
 * and the channel kernel how the resize result is cast back (.astype(dtype), channels.py:132: truncation toward
 * zero for integers; float64 stays float64 and is rounded to float32 by grad_hist's astype("f")).
 * Their per-octave (min, max) keys are 64-bit: minmax is then uint64 [batch][n_oct][2]. */
#define WB_DTYPE_F64 3
#define WB_DTYPE_I8 4
#define WB_DTYPE_I16 5
#define WB_DTYPE_U16 6
#define WB_DTYPE_I32 7
#define WB_DTYPE_U32 8
/* int64 / uint64 images whose values are exact in float64 (|v| < 2^51: the caller checks; their 2x2 sums then neither
 * round nor wrap), bool images (NumPy adds bools with logical OR, so a pooled pixel is "any of the four"; the resize
 * result is cast back with != 0) and float16 images (every add of avg_pool_2 and the cast back round to binary16). */
#define WB_DTYPE_I64 9
#define WB_DTYPE_U64 10
#define WB_DTYPE_BOOL 11
#define WB_DTYPE_F16 12

/* Channel functions (channel_opts["channels"] of the reference) the channel kernel implements:
 *   WB_CHN_GRAD_HIST       waldboost.channels.grad_hist (n_bins=4)    4 x float32  channels.py:40-52
 *   WB_CHN_GRAD_HIST_4_U1  waldboost.fpga.grad_hist_4_u1              4 x uint8    fpga/channels.py:29-53
 *   WB_CHN_GRAD_MAG_U1     waldboost.fpga.grad_mag_u1                 1 x uint8    fpga/channels.py:56-67
 *   WB_CHN_GRAD_MAG        waldboost.channels.grad_mag (norm=5,eps=1e-3) 1 x float32 channels.py:30-37 */
#define WB_CHN_GRAD_HIST 0
#define WB_CHN_GRAD_HIST_4_U1 1
#define WB_CHN_GRAD_MAG_U1 2
#define WB_CHN_GRAD_MAG 3

/* Channel images are [u][v][C] everywhere (the layout channel_pyramid hands to callers), in the
 * dtype the channel function produces (float32 or uint8): with C = 4 a pixel is one aligned
 * float4 (or one dword), so the channel kernel stores one vector per lane and a cascade tile row
 * is one contiguous run in HBM.  Offsets and strides of channel buffers (WbLevel.chn_off,
 * chn_stride) count ELEMENTS of that dtype; every level starts on a multiple of 4 elements. */

#define WB_MAX_OCTAVES 24

/* Detection / work-queue buffers are split into WB_DET_SHARDS independent append regions, each
 * with its own counter, so that concurrent workgroups do not serialise on one atomic word:
 *   records of shard s : det[s*shard_capacity .. s*shard_capacity + min(count[s], shard_capacity))
 * A workgroup appends to shard (blockIdx.x % WB_DET_SHARDS). */
#define WB_DET_SHARDS 64

/* One pyramid level (reference channels.py:127-146).  Built on the host: the level
 * plan is Python-float arithmetic in the reference and stays there. */
typedef struct WbLevel {
    int32_t oct;      /* source octave index                                        */
    int32_t src_h;    /* octave base size                                           */
    int32_t src_w;
    int32_t nh;       /* resized size (multiple of shrink)                          */
    int32_t nw;
    int32_t u;        /* channel image size = nh/shrink, nw/shrink                  */
    int32_t v;
    int32_t tap_off;  /* first entry of this level's resampling taps in the WbTap table:
                         nh row taps, then nw column taps                           */
    int64_t src_off;  /* element offset of the octave in the per-image octave buffer;
                         octave 0 lives in the image buffer itself (src_off unused) */
    int64_t chn_off;  /* float offset of this level in the per-image channel buffer */
    double sy;        /* zoom step src_h/nh (fp64 division, as scipy zoom does)     */
    double sx;        /* src_w/nw                                                   */
} WbLevel;            /* 64 bytes */

/* One axis of the order-1 resample of an output coordinate k (scipy.ndimage.zoom, grid_mode=True,
 * mode='mirror' -- what skimage.transform.resize runs for channels.py:132), built on the host in
 * fp64 exactly as NI_ZoomShift does: cc = ((k + 0.5) * step) - 0.5; i0 = floor(cc);
 * w0 = 1 - (cc - i0); w1 = 1 - w0; i1 = i0 + 1; indices mirror-mapped into the source. */
typedef struct WbTap {
    int32_t i0, i1;
    double w0, w1;
} WbTap; /* 24 bytes */

/* One workgroup's tile: level index + tile coordinates (in tiles). */
typedef struct WbTile {
    int32_t level;
    uint16_t ty;
    uint16_t tx;
} WbTile; /* 8 bytes */

/* One detection: window (r, c) of pyramid level `level` of image `image`
 * survived every stage with accumulated response `score` (model.py:255-259). */
typedef struct WbDet {
    int32_t image;
    int32_t level;
    uint16_t r;
    uint16_t c;
    float score;
} WbDet; /* 16 bytes */

typedef struct WbModel WbModel; /* opaque */

typedef struct WbModelInfo {
    int32_t n_stages;
    int32_t depth;      /* depth of the canonical complete trees the kernels walk  */
    int32_t m, n, C;    /* window shape (rows, cols, channels)                     */
    int32_t tile_rows;  /* cascade tile: tile_rows x 64 windows per workgroup      */
    int32_t tile_cols;
    int32_t lds_bytes;  /* dynamic LDS per workgroup of the cascade kernel         */
    int32_t rank_ok;    /* 1 = the model has rank tables (WB_DTYPE_RANK8 channels)  */
    int32_t specialized; /* bit 0: a model-specialised kernel is loaded for uint8 channels, bit 1: for ranks, bit 2: for 16-bit ranks */
    int32_t rank16_ok;   /* 1 = the model has 16-bit rank tables (WB_DTYPE_RANK16 channels) */
} WbModelInfo;

int wb_abi_version(void);
const char *wb_last_error(void);

/* Tile sizes of the channel kernel (outputs per workgroup) for a channel function (WB_CHN_*) and shrink: what the
 * tile table handed to wb_channels_launch must be cut to. */
int wb_channels_tile(int channel_func, int shrink, int *tile_u, int *tile_v);

/* Channel count and element dtype (WB_DTYPE_*) a channel function produces. */
int wb_channel_func_info(int channel_func, int *n_channels, int *chn_dtype);

/* Octave pyramid of the raw image + per-octave min/max (the clip range of the resize).
 *   img      dev  [batch][H][W] of dtype, image b at img + b*img_stride elements
 *   oct      dev  per-image octave buffer for octaves 1..n_oct-1 (octave k at
 *                 oct + b*oct_stride + oct_off[k]); oct_off is a HOST array of n_oct
 *                 element offsets (oct_off[0] ignored)
 *   minmax   dev  uint32 [batch][n_oct][2] order-preserving keys of (min, max); uint64 [batch][n_oct][2]
 *                 for WB_DTYPE_F64 and the integer codes.  ACCUMULATED into (atomic max): the caller zeroes it
 * uint8 pooling wraps mod 256 before the divide, as the reference does under NumPy
 * (SURVEY S2); float32 pooling is ((a+b)+c)+d then /4. */
int wb_octaves_launch(void *stream, const void *img, int dtype, int batch, int H, int W,
                      int64_t img_stride, void *oct, int64_t oct_stride, const int64_t *oct_off,
                      int n_oct, uint32_t *minmax);
/* The same, and the launch's first workgroup also zeroes zero[0 .. zero_words): accumulators a LATER kernel of the step
 * adds into (the cascade's shard counters and alive[] statistics) -- with wb_cascade_launch_z a step needs no memset
 * launch at all.  Nothing else may touch those words while this launch runs. */
int wb_octaves_launch_z(void *stream, const void *img, int dtype, int batch, int H, int W,
                        int64_t img_stride, void *oct, int64_t oct_stride, const int64_t *oct_off,
                        int n_oct, uint32_t *minmax, uint32_t *zero, int zero_words);

/* All levels of all images: bilinear resize (fp64) -> channel function (grad_hist: Sobel
 * gradients -> 4 oriented channels with fp64 projection) -> shrink -> 3x3 smooth, fused per tile.
 *   channel_func  WB_CHN_*; the uint8 functions take uint8 images only (WB_ERR_UNSUPPORTED otherwise)
 *   levels   dev  WbLevel[n_levels];  tiles dev WbTile[n_tiles] (tile = wb_channels_tile)
 *   taps     dev  WbTap table addressed by WbLevel.tap_off
 *   img/oct  as for wb_octaves_launch; both buffers must extend 16 bytes past their last element
 *            (source rows are fetched with 4-byte-aligned dword loads)
 *   cs_sn    HOST double[8]: cos(theta_k), k=0..3 then sin(theta_k) (channels.py:43-46); used by
 *            WB_CHN_GRAD_HIST only
 *   chn      dev  [u][v][C] per level in the channel function's dtype, image b at
 *                 chn + b*chn_stride, level l at + levels[l].chn_off (elements); NULL (with `rank`) = do not
 *                 write the float32 channels at all
 *   rank_model, rank   optional (WB_CHN_GRAD_HIST only): also write the channels as WB_DTYPE_RANK8 bytes for that
 *                 model, same [u][v][4] layout and element offsets, image b at rank + b*rank_stride; the buffer
 *                 must extend 16 bytes past its last element (the cascade's 16-byte group loads).  Model.detect
 *                 (model.py:149-179) needs nothing else: the float32 pyramid then never touches HBM */
int wb_channels_launch(void *stream, const void *img, int64_t img_stride, const void *oct,
                       int64_t oct_stride, int dtype, int batch, const WbLevel *levels, int n_levels,
                       const WbTile *tiles, int n_tiles, const uint32_t *minmax, int n_oct,
                       const WbTap *taps, int channel_func, int shrink, int smooth, const double *cs_sn,
                       void *chn, int64_t chn_stride, const WbModel *rank_model, uint8_t *rank,
                       int64_t rank_stride);

/* The same launch with a per-tile side table (ABI 7): patches = dev WbTilePatch[n_tiles], entry i for tiles[i] -- the
 * source patch that tile stages in LDS for its resample, as wb_channels_tile_patches computes it on the HOST from the
 * library's own tile geometry (the level plan and its tiles are host data anyway, channels.py:124-131).  With the table a
 * workgroup starts its patch loads right behind its tile record instead of behind four chains of fp64 arithmetic that
 * every workgroup of a level would repeat.  patches = NULL: exactly wb_channels_launch.  Results are identical either way. */
typedef struct WbTilePatch {
    int32_t r_lo;    /* first source row / column of the patch in the level's octave                               */
    int32_t c_lo;
    uint16_t rows;   /* source rows r_lo .. r_lo + rows - 1; 0 = this tile stages no patch (an identity level, an   */
    uint16_t bytes;  /* up-scale, a patch beyond the LDS budget): it takes the kernel's other paths                 */
    uint32_t pad;
} WbTilePatch;       /* 16 bytes */
int wb_channels_tile_patches(int channel_func, int shrink, int smooth, const WbLevel *levels_host, int n_levels,
                             const WbTile *tiles_host, int n_tiles, WbTilePatch *out_host);
int wb_channels_launch_x(void *stream, const void *img, int64_t img_stride, const void *oct,
                         int64_t oct_stride, int dtype, int batch, const WbLevel *levels, int n_levels,
                         const WbTile *tiles, int n_tiles, const uint32_t *minmax, int n_oct,
                         const WbTap *taps, int channel_func, int shrink, int smooth, const double *cs_sn,
                         void *chn, int64_t chn_stride, const WbModel *rank_model, uint8_t *rank,
                         int64_t rank_stride, const WbTilePatch *patches, int rank_dtype);
/* rank_dtype: WB_DTYPE_RANK8 (what wb_channels_launch writes) or WB_DTYPE_RANK16 -- `rank` is then uint16 [u][v][4] per
 * level (rank_stride and the level offsets count ELEMENTS, as for every channel buffer), 8-byte aligned. */

/* The pyramid around a channel function this library has no kernel for (channels.py:119,136 calls whatever callable
 * channel_opts["channels"] holds -- the caller runs it, between these two):
 *   wb_resize_level_launch  one level's resized image, cast back to the image dtype (channels.py:132): level_host = HOST
 *                           WbLevel of that level, minmax_host = HOST copy of the level's octave's (min, max) keys as
 *                           wb_octaves_launch left them (two 32-bit words; two 64-bit words for the float64-held dtypes),
 *                           img / oct / taps dev as for wb_channels_launch (batch 1); out dev [nh][nw] of uint8 / float32 /
 *                           float64 (the float64-held dtypes: the cast value, as a double)
 *   wb_pool_smooth_launch   avg_pool_2 (shrink = 2, channels.py:55-64) and smooth_image_3d (smooth = 1, :78-90) of the
 *                           callable's result in dev [H][W][C] uint8 or float32 -> out dev [H/shrink][W/shrink][C]; tmp =
 *                           dev scratch of the pooled size when both steps run */
int wb_resize_level_launch(void *stream, const void *img, const void *oct, int dtype, const WbLevel *level_host,
                           const uint32_t *minmax_host, const WbTap *taps, void *out);
int wb_pool_smooth_launch(void *stream, const void *in, int chn_dtype, int H, int W, int C, int shrink, int smooth, void *tmp,
                          void *out);

/* The channel functions called directly on one float32 image (the caller's image.astype("f")) WITH ARGUMENTS
 * (reference channels.py:40-52 grad_hist(image, n_bins, full, bias) and :30-37 grad_mag(image, norm, eps); with
 * their default arguments, and inside channel_pyramid, they run in wb_channels_launch).
 *   wb_grad_hist_launch  cs_sn HOST double[2*n_bins]: cos(theta_k) then sin(theta_k) of np.linspace(0, pi or 2*pi,
 *                        n_bins+1)[:-1]; wide = 0: bias is a float32 value (a Python scalar is one under NumPy-2
 *                        promotion), out dev float32 [H][W][n_bins]; wide = 1: bias is a float64 / int64 NumPy scalar --
 *                        |chns| - bias and everything after it are float64, out dev float64 [H][W][n_bins]
 *   wb_grad_mag_launch   n_taps = 0: the plain magnitude (norm None or <= 1); else taps HOST float32[n_taps] =
 *                        triangle_kernel(norm), scratch dev float32[2*H*W]; out dev float32 [H][W]; wide = 0: eps is a
 *                        float32 value, mag / (norm + eps) in float32; wide = 1: eps is a float64 NumPy scalar -- the
 *                        in-place `mag /= norm + eps` divides in float64 and rounds once to float32 */
int wb_grad_hist_launch(void *stream, const float *img, int H, int W, int n_bins, int full, double bias, int wide,
                        const double *cs_sn, void *out);
int wb_grad_mag_launch(void *stream, const float *img, int H, int W, int n_taps, const float *taps, double eps, int wide,
                       float *scratch, float *out);

/* Build the device-side cascade from the reference's tree arrays (all HOST pointers).
 *   node_off  int32[n_stages+1]  first node of each stage's tree in the flat arrays
 *   feature   uint8[n_nodes][3]  (row, col, channel) inside the window
 *   threshold float[n_nodes], left/right int8[n_nodes] (-1 on leaves), prediction float[n_nodes]
 *   theta     float[n_stages]    rejection thresholds, -inf = stage never rejects
 * Trees must have parent index < child index (the order the reference walks them in,
 * training.py:88).  Trees up to depth 3 run in the LDS-tiled kernel; deeper ones in a generic
 * node-walk kernel (correct, not tuned). */
int wb_model_create(int n_stages, const int32_t *node_off, const uint8_t *feature,
                    const float *threshold, const int8_t *left, const int8_t *right,
                    const float *prediction, const float *theta, int m, int n, int C,
                    WbModel **out);
int wb_model_destroy(WbModel *model);
int wb_model_info(const WbModel *model, WbModelInfo *info);

/* Compile (hiprtc, a couple of seconds the first time; afterwards from the cache directory $WB_JIT_CACHE, default
 * ~/.cache/waldboost_amd) and load the model-specialised tile kernel for one kind of byte tile: chn_dtype
 * WB_DTYPE_RANK8, WB_DTYPE_RANK16 or WB_DTYPE_U8.  The model's stage records -- feature offsets, thresholds (model.py:62-67,
 * training.py:24-31), leaf values and theta -- are compile-time constants in it.  wb_cascade_launch uses it from then
 * on for that dtype; results are bit-identical to the generic kernel's -- which the call verifies before it returns: the
 * new kernel and the generic one scan a synthetic pyramid of byte tiles (8,493 windows, six passes; $WB_JIT_SELFTEST) and
 * must agree on every per-stage alive count and detection record.  WB_ERR_UNSUPPORTED for float32 channels, for models on
 * the node-walk kernel, and for a model none of whose builds passes that test; a failed compilation or a refused build
 * leaves the model on the generic kernel. */
int wb_model_specialize(WbModel *model, int chn_dtype);
/* ABI 8.  Make wb_cascade_launch ignore (enable = 0) or use again (1) the specialised kernels this model has loaded: with
 * them off the generic kernel scans.  For callers that cross-check a specialised kernel on their own data before they
 * rely on it (the Python engine does, on the image at hand, right after wb_model_specialize), or retire one. */
int wb_model_use_specialized(WbModel *model, int enable);

/* Several cascades over ONE pyramid of threshold ranks (waldboost.detect(image, *models), reference __init__.py:120-124:
 * the channels are computed once for all models): one rank table per channel from the UNION of the members'
 * thresholds.  wb_rankgroup_model hands out member i as a VIEW of its model -- a WbModel handle owned by the group that
 * shares everything with the model but the rank tables; pass any view as wb_channels_launch's rank_model (they hold
 * the same table) and view i to wb_cascade_launch(WB_DTYPE_RANK8) / wb_model_specialize / wb_model_info.  The models
 * must outlive the group.  WB_ERR_UNSUPPORTED when a channel's union exceeds 254 thresholds or a member has no rank
 * form (node-walk models, C != 4): scan float32 channels then. */
typedef struct WbRankGroup WbRankGroup;
int wb_rankgroup_create(const WbModel *const *models, int n, WbRankGroup **out);
int wb_rankgroup_model(WbRankGroup *group, int i, WbModel **view);
int wb_rankgroup_destroy(WbRankGroup *group);

/* Build check of the specialised kernel's source without a GPU: a synthetic cascade of n_stages depth-`depth` trees
 * through the generator and hiprtc for `arch` (e.g. "gfx950"); *code_bytes = size of the code object. */
int wb_jit_compile_check(int depth, int n_stages, const char *arch, int64_t *code_bytes);
/* The same for a tile of elem_bytes-wide elements (1: uint8 channels / WB_DTYPE_RANK8, 2: WB_DTYPE_RANK16). */
int wb_jit_compile_check2(int depth, int n_stages, int elem_bytes, const char *arch, int64_t *code_bytes);

/* Dense sliding-window cascade over all levels of all images.
 *   chn           [u][v][C] per level as written by wb_channels_launch (or caller-provided), of
 *                 chn_dtype WB_DTYPE_F32 or WB_DTYPE_U8 (uint8 values compare against the float32
 *                 thresholds as their exact float32 values, like NumPy's uint8 <= float32), or
 *                 WB_DTYPE_RANK8 (ranks of float32 channels for THIS model); a byte
 *                 buffer must extend 16 bytes past its last element (16-byte group loads)
 *   tiles         dev WbTile[n_tiles]: tiles of tile_rows x tile_cols WINDOWS over the
 *                 (u-m) x (v-n) window grid of each level
 *   det           dev WbDet[WB_DET_SHARDS][shard_capacity]; det_count dev uint32[WB_DET_SHARDS]:
 *                 survivors per shard (a count may exceed shard_capacity: the records beyond it
 *                 are dropped, the count stays exact -- grow the buffer and launch again)
 *   alive         dev uint32 [batch][n_levels][n_stages]: windows entering each stage (the reference's n_weak
 *                 is its sum, n_loc the sum of column 0: model.py:248,252); NULL = no statistics
 * det_count and alive are ACCUMULATED into: the caller zeroes them.  Record order is unspecified; sort by
 * (image, level, r, c) to obtain the reference order. */
int wb_cascade_launch(void *stream, const WbModel *model, const void *chn, int chn_dtype,
                      int64_t chn_stride, int batch, const WbLevel *levels, int n_levels,
                      const WbTile *tiles, int n_tiles, WbDet *det, uint32_t *det_count,
                      uint32_t shard_capacity, uint32_t *alive);
/* The same, and the launch's first workgroup also zeroes zero[0 .. zero_words): the octaves' (min, max) keys, which
 * nothing of this step reads any more once the channel kernel has finished -- ready for the next step's
 * wb_octaves_launch(_z). */
int wb_cascade_launch_z(void *stream, const WbModel *model, const void *chn, int chn_dtype,
                        int64_t chn_stride, int batch, const WbLevel *levels, int n_levels,
                        const WbTile *tiles, int n_tiles, WbDet *det, uint32_t *det_count,
                        uint32_t shard_capacity, uint32_t *alive, uint32_t *zero, int zero_words);

/* The valid records of all shards of a detection buffer (as wb_cascade_launch fills it), packed back to back:
 *   packed  dev int32, 16-byte aligned: a 4-word header {valid records in all shards, fullest shard's count
 *           (> shard_capacity: records were dropped -- grow and scan again), records present behind the header
 *           = min(valid, packed_capacity), shard_capacity}, then that many WbDet records (shard order)
 * One contiguous prefix for a single host read-back (Model.detect, model.py:173-179) or one collective
 * (multi-GPU gather of detections); no host synchronisation. */
int wb_det_pack_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                       int32_t *packed, uint32_t packed_capacity);

/* Model.detect's last step in one launch (model.py:136-147, :173-179): for the valid records of all shards, at their
 * packed positions i < out_capacity,
 *   out  dev, 16-byte aligned:  int32 header[4] as wb_det_pack_launch writes it
 *                             | uint64 keys[out_capacity]   level << 54 | r << 40 | c << 26 | i
 *                             | float  boxes[out_capacity][4]   (c, r, c + n, r + m) * inv_scale[level]
 *                             | float  scores[out_capacity]
 * Sorting the keys gives the reference order (level, r, c); a sorted key's low 26 bits index boxes / scores.
 *   inv_scale  dev float32[n_levels] = float32(1 / scale) per level; m, n = window rows, cols
 *   n_levels, max_rows, max_cols  the scan's extent, checked against the key's 10 / 14 / 14 bits
 *   out_capacity  even, <= 2^26.   WB_ERR_UNSUPPORTED when the extent does not fit the key (use wb_det_pack_launch +
 *   wb_boxes_launch then).  No host synchronisation. */
int wb_det_finish_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                         const float *inv_scale, int n_levels, int max_rows, int max_cols, int m, int n,
                         void *out, uint32_t out_capacity);

/* wb_det_finish_launch with the ordering done on the device too (model.py:173-179: the concatenated result is in level
 * order, row-major inside a level): same arguments, same buffer layout, but when the valid records number at most
 * min(out_capacity, 4096) the three sections are written IN KEY ORDER -- keys[i] ascending, boxes[i] / scores[i] the
 * i-th detection of the reference's order (a key's low 26 bits then name the packed position the record came from and
 * carry no meaning for the caller) -- and header[3] = 1.  Otherwise header[3] = 0 and the sections are exactly what
 * wb_det_finish_launch writes (sort the keys, gather).  header[0..2] as wb_det_pack_launch writes them.
 *   tail, tail_words  optional (NULL, 0): dev int32 words copied behind the scores section, out + 16 + 28 * out_capacity
 *                     bytes -- the scan's alive[] statistics (model.py:248,252), so that ONE read-back carries all a
 *                     Model.detect call returns
 * No host synchronisation. */
int wb_det_finish_sorted_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                                const float *inv_scale, int n_levels, int max_rows, int max_cols, int m, int n,
                                void *out, uint32_t out_capacity, const int32_t *tail, uint32_t tail_words);

/* The same for a BATCH (the reference's detection loop over files, scripts/waldboost-detect.py:64-67, taken a batch at a
 * time): the shards' records are first split by image into buckets (scratch), then every image's bucket is ordered and
 * finished like a single image's shards.  Two launches, no host synchronisation.
 *   scratch  dev, 16-byte aligned, >= n_images * (256 + 16 * out_capacity) bytes
 *   out      dev, 16-byte aligned:  int32 info[4] = (valid records of all shards, fullest shard -- above
 *            shard_capacity: records were dropped, scan again --, n_images, out_capacity)
 *            | per image b, 16 + 28 * out_capacity bytes: header[4] | keys | boxes | scores exactly as
 *              wb_det_finish_sorted_launch writes them for ONE image with capacity out_capacity: header[0] = the image's
 *              detections, header[1] the same (above out_capacity: they did not fit), header[3] = 1 when in key order
 *   out_capacity  per image; a multiple of 4, <= 2^26 (ordered up to 4096 detections per image) */
int wb_det_order_batch_launch(void *stream, const WbDet *det, const uint32_t *det_count, uint32_t shard_capacity,
                              int n_images, const float *inv_scale, int n_levels, int max_rows, int max_cols, int m, int n,
                              void *scratch, size_t scratch_bytes, void *out, uint32_t out_capacity);

/* One tree evaluated at explicit window origins (rs[i], cs[i]) of an HWC channel image
 * X[u][v][C] of x_dtype (WB_DTYPE_F32 / WB_DTYPE_U8); out[i] = prediction of the leaf reached
 * (training.py:84-96). Tree arrays and rs/cs/out are dev pointers. */
int wb_tree_eval_launch(void *stream, const void *X, int x_dtype, int u, int v, int C, const int32_t *rs,
                        const int32_t *cs, int64_t n_pos, const uint8_t *feature,
                        const float *threshold, const int8_t *left, const int8_t *right,
                        const float *prediction, int n_nodes, float *out);

/* Crops of m x n windows of an HWC channel image X[u][v][C] (x_dtype: WB_DTYPE_F32 / WB_DTYPE_U8) at the
 * origins (rs[i], cs[i]) -> out[n_pos][m][n][C] of the same dtype (reference samples.py:14-43
 * gather_samples, the crop step of hard-negative mining).  Origins must satisfy rs+m <= u, cs+n <= v
 * (the host wrapper checks).  All pointers dev. */
int wb_gather_samples_launch(void *stream, const void *X, int x_dtype, int u, int v, int C, const int32_t *rs,
                             const int32_t *cs, int64_t n_pos, int m, int n, void *out);

/* The cascade on per-sample arrays X[n_samples][m][n][C] (reference model.py:181-214 Model.predict, the
 * re-scoring step of the training sample pool): H[i] = accumulated response in stage order while the
 * sample is alive, -inf once a stage rejected it; mask[i] (uint8) = 1 if it passed every stage. */
int wb_samples_predict_launch(void *stream, const WbModel *model, const void *X, int x_dtype,
                              int64_t n_samples, float *H, uint8_t *mask);

/* One tree on per-sample arrays X[n_samples][m][n][C]: node[i] = index of the leaf reached (reference
 * training.py:73-83 DTree.apply; DTree.predict is prediction[node]).  Tree arrays dev, as for
 * wb_tree_eval_launch. */
int wb_tree_apply_launch(void *stream, const void *X, int x_dtype, int64_t n_samples, int m, int n, int C,
                         const uint8_t *feature, const float *threshold, const int8_t *left,
                         const int8_t *right, int n_nodes, int32_t *node);

/* XYXY float32 boxes of detections: [c, r, c+n, r+m] * (1/scale[level]) (model.py:136-147).
 *   inv_scale  dev float[n_levels] = float32(1.0/scale) computed on the host in fp64 */
int wb_boxes_launch(void *stream, const WbDet *det, int64_t n_det, const float *inv_scale,
                    int m, int n, float *boxes /* [n_det][4] */, float *scores /* [n_det] */);

/* Device self-test: the uint8 fast path of the orientation projection (fp32 arithmetic that is
 * proven equal to the reference's fp64 formula for integer gradients) is compared with the fp64
 * formula for every (gx, gy) in [-1020, 1020]^2; *mismatches (dev uint32, zeroed by the caller)
 * receives the number of differing channel values -- must stay 0. */
int wb_selftest_projection(void *stream, uint32_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* WALDBOOST_HIP_H */
