/*
 * v2ce_hip.h -- C ABI of libv2ce_hip.so, the MI355X (gfx950) implementation of the V2CE hot path.
 *
 * The reference (ucsd-hdsi-dvs/V2CE-Toolbox) is pure Python/PyTorch and has no FFI of its own; the
 * hot path sits behind three Python call boundaries.  This header is the native
 * boundary underneath them: every entry point names the reference code it replaces.  All pointers
 * are DEVICE pointers unless the name ends in _host; the caller owns every buffer; `stream` is a
 * hipStream_t passed as void* (NULL = the null stream); functions only enqueue work and never
 * synchronise; they return V2CE_OK or a negative error code and never throw.
 *
 * Activation layout everywhere: [B][T][C][H][W] f32 ("frame-major planar"): the reference's own
 * input layout [B,L,2,H,W] (scripts/v2ce_3d.py:26) and output layout [B,L,20,H,W]
 * (scripts/v2ce_3d.py:29), which is also LDATI's [frames,2,10,H,W] (v2ce.py:351-352).
 */
#ifndef V2CE_HIP_H
#define V2CE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2CE_OK 0
#define V2CE_ERR_BAD_ARG (-1)      /* NULL pointer, non-positive size, bad enum */
#define V2CE_ERR_UNSUPPORTED (-2)  /* shape / option outside what the kernels cover */
#define V2CE_ERR_HIP (-3)          /* a HIP runtime call failed; see v2ce_last_error() */
#define V2CE_ERR_WORKSPACE (-4)    /* workspace too small */

#define V2CE_RNG_REPLAY 0 /* uniforms supplied: dense [B,2,9,H,W,replay_max_n] f32 (LDATI.py:171) */
#define V2CE_RNG_PHILOX 1 /* Philox4x32-10, counter (pixel, j>>2, p*9+c, frame_base+b), key seed */

#define V2CE_STRATEGY_SLOPE 0 /* additional_events_strategy='slope' (the CLI's, v2ce.py:356) */
#define V2CE_STRATEGY_NONE 1  /* 'none': voxels with more than one event emit nothing (LDATI.py:241) */
#define V2CE_STRATEGY_RANDOM 2 /* 'random': the raw uniform is the time offset, in SECONDS (LDATI.py:173-174) */

#define V2CE_POOL_NONE 0
#define V2CE_POOL_AVG 1      /* nn.AvgPool2d(k, stride 1, padding k//2) of the counts before the slope (LDATI.py:181) */
#define V2CE_POOL_WEIGHTED 2 /* 3x3 [[1,2,1],[2,4,2],[1,2,1]]/16 conv, zero padding (LDATI.py:178-180) */

#define V2CE_ACT_NONE 0
#define V2CE_ACT_RELU 1   /* torch.relu / nn.ReLU           (submodules.py:105,232) */
#define V2CE_ACT_LEAKY 2  /* nn.LeakyReLU(0.01)             (submodules.py:101-103) */

typedef void *v2ce_stream_t;

const char *v2ce_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char *v2ce_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Stage 2 -- LDATI.  Replaces scripts/LDATI.py:126-310 (sample_voxel_statistical, y_relocate,
 * calculate_statistical_linear_params_for_stage2, pick_elements, pick_and_sort), every option value of
 * sample_voxel_statistical (v2ce_ldati_options below; the CLI's are 'slope', pooling 'none', bidirectional=False).
 * Two-phase, because the output length is data dependent:
 *   count -> (caller reads seg_offsets + stats, allocates) -> emit
 * ---------------------------------------------------------------------------------------------- */

/* Per-(frame, time-bin) event counts, their exclusive prefix, and the statistics the caller needs to
 * allocate.  Replaces y_relocate (LDATI.py:80-106) + torch.max (LDATI.py:169) + the sizes implied by
 * pick_elements' selections (LDATI.py:228,239).
 * vox [B,2,10,H,W] f32.  seg_offsets [B*9+1] i64: exclusive prefix of the segment counts (last =
 * total events).  stats [4] i64 = {max count of any voxel (LDATI.py:169 max_n), most events of one
 * (2048-pixel tile, bin), largest segment, total events}.  tile_ws (>= v2ce_ldati_tile_ws_bytes):
 * per-tile counts and offsets (and the offsets once more as one contiguous row per segment, for the bucket sort's
 * setup), consumed by v2ce_ldati_emit's two-level path. */
/* The keyword options of sample_voxel_statistical (LDATI.py:126).  NULL = the CLI's call (v2ce.py:356):
 * additional_events_strategy='slope', pooling_type='none', bidirectional=False. */
typedef struct {
    int32_t strategy;            /* V2CE_STRATEGY_* (additional_events_strategy) */
    int32_t bidirectional;       /* y_relocate(bidirectional=True), LDATI.py:107-122 */
    int32_t pooling_type;        /* V2CE_POOL_*; only shapes the 'slope' strategy */
    int32_t pooling_kernel_size; /* 'avg' only; odd, 1..15 */
} v2ce_ldati_options;

size_t v2ce_ldati_tile_ws_bytes(int B, int H, int W);
int v2ce_ldati_count(const float *vox, int B, int H, int W, const v2ce_ldati_options *options, void *tile_ws,
                     size_t tile_ws_bytes, int64_t *seg_offsets, int64_t *stats, v2ce_stream_t stream);

/* Bytes of LDS the sweep (fallback) kernel needs per workgroup for this fps/t0; 0 = the key range
 * of one time bin does not fit its LDS histogram (fps < ~12): only the two-level path runs then. */
size_t v2ce_ldati_lds_bytes(double fps, double t0);

/* Emit all events, each (frame, bin) segment stably sorted by timestamp, at seg_offsets.
 * Replaces LDATI.py:156-165 (single-event times), :171-212 (slope-distributed times) and
 * :217-310 (pick, concat negative-then-positive, sort by timestamp, pack to 13-byte records).  Tie
 * order is the STABLE order [neg singles row-major, neg multis row-major then draw, pos singles,
 * pos multis] -- what the reference's CPU argsort yields for segments >= 32768 events.
 * uniforms/replay_max_n: REPLAY mode only.  frame_ts_add [B] i64 or NULL: added to every timestamp
 * of frame b (v2ce.py:365 per-frame offset, fused).
 * Output: either the four SoA arrays (ts, x, y, p; length seg_offsets[B*9]) or `packed` (13-byte
 * records {i8 timestamp, i2 x, i2 y, i1 polarity} = the numpy recarray layout of LDATI.py:308-309,
 * 4-byte aligned, total*13 bytes); the other must be NULL.
 * total_events / max_segment_events / max_tile_events: host copies of stats[3], [2], [1].
 * workspace (>= v2ce_ldati_workspace_bytes, which returns 0 when shape or density is outside the
 * path: > 512 tiles per frame, > 15360 events in one tile-bin) + the tile_ws of v2ce_ldati_count
 * select the two-level path (tile pass -> coarse buckets -> LDS counting sort -> coalesced output);
 * workspace NULL runs the one-workgroup-per-segment sweep for every segment (no scratch, slower,
 * needs v2ce_ldati_lds_bytes != 0).  Both produce bit-identical output.  Coarse buckets beyond the LDS
 * capacity of a sort workgroup (degenerate ties) are ordered by a dedicated, slower kernel inside the
 * same call; the device status word (v2ce_ldati_status) stays 0 unless an internal limit is hit.
 * Options: bidirectional relocation and the pooled slope (a pre-pass writes {k, b} per voxel) run on the
 * two-level path (workspace required); 'random' spreads the timestamps of a bin over a whole second, far
 * beyond the bucket machinery's key range: it takes a generic path (tile pass writing one 64-bit key per
 * event, a library radix sort -- rocPRIM --, decode), workspace required. */
size_t v2ce_ldati_workspace_bytes(int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                                  int64_t total_events, int64_t max_segment_events, int64_t max_tile_events,
                                  int packed_output);
int v2ce_ldati_emit(const float *vox, int B, int H, int W, double fps, double t0,
                    const v2ce_ldati_options *options, int rng_mode,
                    const float *uniforms, int replay_max_n, uint64_t seed, int64_t frame_base,
                    const int64_t *seg_offsets, const int64_t *frame_ts_add, int64_t *ts,
                    int16_t *x, int16_t *y, int8_t *p, uint8_t *packed, int64_t total_events,
                    int64_t max_segment_events, int64_t max_tile_events, const void *tile_ws,
                    void *workspace, size_t workspace_bytes, v2ce_stream_t stream);
/* Device address of the status word (int32) inside a workspace used with the same arguments. */
int v2ce_ldati_status(const void *workspace, int B, int H, int W, double fps, double t0,
                      const v2ce_ldati_options *options, int64_t total_events, int64_t max_segment_events,
                      int64_t max_tile_events, const int32_t **status_dev);

/* The two-level plan v2ce_ldati_emit would use (introspection for tools/tests): info [10] = {ok, fine-key
 * bits (shift), coarse buckets per segment, tiles per frame, tile-pass capacity, sort capacity, entries
 * of the per-tile tables, entries of the per-bucket tables, LDS bytes of the tile pass, of the sort}.
 * Workspace layout (u32 units): bofs [n_bkt = B*9*(NB+1)] | groups [B*9*NB] | big_list [B*9*NB] | ngroups [B*9] |
 * seg_flag [B*9] | status [4] = {status, number of big buckets, -, -} | records [total] |
 * roff [n_tab = B*9*T*(NB+1)] as u16. */
/* Fused count + sparse tile pass (round 4).  v2ce_ldati_count reads the voxel grid once to
 * count and v2ce_ldati_emit reads it again to compute the timestamps.  On real UNet output every 2048-pixel tile is sparse
 * (<= 8192 events over the nine bins): v2ce_ldati_count_fused does both in ONE pass -- it returns exactly what
 * v2ce_ldati_count returns (seg_offsets, stats[0..3]; stats is int64[8] here, stats[4] = the largest tile's events over all
 * nine bins) and leaves every sparse tile's bucket-grouped records in its own slot of `fused_ws` (needs the random-draw
 * arguments of v2ce_ldati_emit for that).  v2ce_ldati_emit_fused is v2ce_ldati_emit with that workspace: when no tile
 * exceeded its slot (largest_tile_events = stats[4] <= 8192) and the call's bucket geometry is the one the fused pass assumed
 * (the geometry follows from the densest segment's events: expected_max_segment_events is the caller's guess, normally the
 * previous batch's stats[2]; the same value goes to all three functions), only the bucket scan and the bucket sort remain; otherwise it ignores fused_ws and runs
 * the two-pass path -- same bytes either way (tests/test_gpu_ldati.py::test_fused_count_equals_two_pass).
 * v2ce_ldati_fused_ws_bytes = 0: no fused path for these options ('random', pooled slope) -- use v2ce_ldati_count.
 * Dense regime (expected_max_tile_bin_events > 0: the caller's guess of the densest (tile, bin) run, normally the previous
 * batch's stats[1] plus a margin; 0 selects the sparse form above): the dense tile kernel is the count pass and the tile pass
 * at once, every (tile, bin) run in its own slot of that many records (rounded up to 256); v2ce_ldati_emit_fused uses the
 * slots when max_tile_events = stats[1] fits them and the geometry matches, and repeats the tile pass on the two-pass path
 * otherwise -- same bytes either way.  The same two expectations go to all three functions of a call.  No dense form
 * (v2ce_ldati_fused_ws_bytes = 0) for bidirectional relocation, 'random', pooled slope, or when the slots would exceed 4 GiB. */
size_t v2ce_ldati_fused_ws_bytes(int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                                 int64_t expected_max_segment_events, int64_t expected_max_tile_bin_events);
int v2ce_ldati_count_fused(const float *vox, int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                           int rng_mode, const float *uniforms, int replay_max_n, uint64_t seed, int64_t frame_base,
                           int64_t expected_max_segment_events, int64_t expected_max_tile_bin_events, void *tile_ws, size_t tile_ws_bytes,
                           void *fused_ws, size_t fused_ws_bytes,
                           int64_t *seg_offsets /* [B*9+1] */, int64_t *stats /* [8] */, v2ce_stream_t stream);
int v2ce_ldati_emit_fused(const float *vox, int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                          int rng_mode, const float *uniforms, int replay_max_n, uint64_t seed, int64_t frame_base,
                          const int64_t *seg_offsets, const int64_t *frame_ts_add, int64_t *ts, int16_t *x, int16_t *y,
                          int8_t *p, uint8_t *packed, int64_t total_events, int64_t max_segment_events,
                          int64_t max_tile_events, const void *tile_ws, void *workspace, size_t workspace_bytes,
                          const void *fused_ws, size_t fused_ws_bytes, int64_t largest_tile_events,
                          int64_t expected_max_segment_events, int64_t expected_max_tile_bin_events, v2ce_stream_t stream);

int v2ce_ldati_plan_info(int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                         int64_t total_events, int64_t max_segment_events, int64_t max_tile_events, int64_t *info);

/* Diagnostic (tests/test_gpu_ldati.py::test_exact_math_helpers), never called by the product: the dense tile kernel's
 * Philox path evaluates LDATI.py:195's  (-b + sqrt(b^2 + 2 k u)) / k  with hand-scheduled correctly rounded sequences
 * (csrc/ldati.hip: sqrt_rn_nr, div_rn_nr) and its Philox rounds on three-input xors.  This entry compares them with the
 * compiler's IEEE square root / division for EVERY slope-table entry (|count difference| <= 31, count <= 31) and EVERY
 * uniform m * 2^-24, and with the plain Philox rounds: mismatches[0] = differing times, mismatches[1] = differing
 * Philox outputs; mismatches[2] = tendencies (every f32 in [0, 1) and (-2^-18, 0)) for which the single-event time's two
 * constant f64 divisions differ between their fused-multiply-add form (single_time_fast) and the IEEE divisions -- the
 * check the library itself runs once per device and fps before it uses that form.  All three must be 0.  Synchronous; ~0.1 s. */
int v2ce_ldati_selfcheck(double fps, int64_t *mismatches /* host [3] */);

/* How the stable tie order inside the LDS counting sorts is obtained on the current device: 1 = straight from the
 * lane order in which one ds_add_rtn_u32 wave-instruction serves equal addresses (checked once per device by a probe
 * kernel enqueued in front of the first v2ce_ldati_count; gfx950 passes), 0 = ballot match-any ranks (probe failed
 * or not yet run, or V2CE_LDATI_NO_ATOMIC_ORDER=1 in the environment: the fallback, forced for tests).  Both give
 * identical output.  Synchronises the device. */
int v2ce_ldati_rank_mode(int32_t *mode);

/* SoA <-> packed 13-byte records {i8 timestamp, i2 x, i2 y, i1 polarity}: the numpy recarray layout
 * of LDATI.py:308-309 (numpy.core.records.fromarrays, itemsize 13).  packed: n*13 bytes. */
int v2ce_events_pack(const int64_t *ts, const int16_t *x, const int16_t *y, const int8_t *p,
                     int64_t n, uint8_t *packed, v2ce_stream_t stream);
int v2ce_events_unpack(const uint8_t *packed, int64_t n, int64_t *ts, int16_t *x, int16_t *y, int8_t *p,
                       v2ce_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Frame ingest.  Replaces v2ce.py:45-64 (image_pre_processing) for frames already at the target
 * height: frames [N][H][W] u8 -> units [N-1][2][H][W] f32 = ((u8/255) - mean) / std of (frame i,
 * frame i+1); separate correctly rounded f32 operations, bit-identical to the host path.
 * ---------------------------------------------------------------------------------------------- */
int v2ce_preprocess_pairs(const uint8_t *frames, int N, int H, int W, float mean, float stdv,
                          float *units, v2ce_stream_t stream);
/* The same for frames of another size: v2ce.py:57-58 cv2.resize(img, (out_w, out_h)), INTER_LINEAR, of x = u8/255 first;
 * units [N-1][2][out_h][out_w].  Restates OpenCV's published scalar algorithm (resize.cpp: float coefficient
 * (float)((d + 0.5) * scale - 0.5), floor, subtraction in float; horizontal pass, then vertical; the 2 x 2 decimation
 * is its INTER_AREA fast path) operation by operation and bit-identically to the host path glue._resize_bilinear;
 * pinned by hand-derived vectors -- OpenCV itself is not installed in the build image. */
int v2ce_preprocess_pairs_resize(const uint8_t *frames, int N, int H, int W, int out_h, int out_w, float mean,
                                 float stdv, float *units, v2ce_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Stage 1 -- V2ce3d building blocks.  Replaces the ATen ops behind scripts/unet_2layer.py:335-379,
 * scripts/submodules.py:115-124,249-264 and scripts/spectral_norm.py:19-31.
 * ---------------------------------------------------------------------------------------------- */

/* One fused 3-D convolution: y = act( conv(x) * scale[co] + shift[co] (+ residual) ).
 * conv is nn.Conv3d with a cubic kernel (1 or 3), padding ksize/2, stride (1, s, s)
 * (submodules.py:96,226-247); scale/shift fold the conv bias and the eval-mode BatchNorm3d
 * (submodules.py:229-231,246); residual/act implement submodules.py:262-263.
 * The input is a VIRTUAL concat of two sources (unet_2layer.py:358-364): channels [0,C0) come from
 * x0 [B,T,C0,H0,W0], nearest-resampled to (Hin,Win) through hmap/wmap (source row/col per logical
 * row/col; NULL = identity, requires H0==Hin, W0==Win); channels [C0,C0+C1) from x1 [B,T,C1,Hin,Win].
 * w_packed: [Cin][k^3][Cout] f32 (see v2ce_pack_weights).  y: [B,T,Cout,Hout,Wout]. */
typedef struct {
    int32_t B, T;
    int32_t C0, H0, W0;
    int32_t C1;
    int32_t Hin, Win;
    int32_t Cout, Hout, Wout;
    int32_t ksize;      /* 1 or 3 */
    int32_t stride_hw;  /* 1 or 2 */
    int32_t act;        /* V2CE_ACT_* */
    int32_t tile_t, tile_h, tile_w; /* output tile per workgroup; 0 = choose automatically */
    int32_t precision;  /* V2CE_PRECISION_F32 (exact f32 MFMA) or V2CE_PRECISION_F16X2 (see below) */
    /* Row pitches in floats (0 = the width itself): x0 rows are W0_pitch apart, x1 rows Win_pitch, the rows of y,
     * residual and sc_y Wout_pitch (pred_y is always dense).  Tensors are then [B][T][C][H][pitch] with only the
     * first W columns of a row meaningful (the rest is never read or written).  With pitches that are multiples
     * of 32 floats the 32-position pieces a wave stores / gathers are whole 128-byte cache lines: the store
     * path is 1.75x faster than on rows of 346 floats (tools/micro/store_rate.hip). */
    int32_t W0_pitch, Win_pitch, Wout_pitch;
    /* Activation layout of x0, x1, y, residual and sc_y (pred_y is always planar):
     *   V2CE_LAYOUT_PLANAR  [B][T][C][H][pitch]           -- the reference's own layout; exact-f32 kernels
     *   V2CE_LAYOUT_C16     [B][T][C/16][H][pitch][16]    -- channel groups of 16 innermost; split-half kernels
     * The split-half kernels REQUIRE C16 (all channel counts are multiples of 16 there): a 16-channel chunk of a
     * halo element is then one 64-byte cache line (four 16-byte loads instead of sixteen dword loads from sixteen
     * planes), and a lane's four consecutive output channels are one 16-byte store: -5.8 % on the forward pass.
     * The exact-f32 kernels require PLANAR, with one bridge: the 2-channel head convolution (C0 == 2, Cout == 32,
     * 3x3x3) reads the planar network input and writes y in the layout named here. */
    int32_t layout;
    /* Range-tracking slots per BATCH ELEMENT: 0 = y_absmax / x0_absmax / x1_absmax are one slot per tensor (below);
     * s >= 2 = element b of the batch uses the slot at + b * s floats (y_absmax then spans B * s floats: max |y| of
     * element b at [b * s], its range-guard value at [b * s + 1]).  Every sequence then gets its own power-of-two
     * pre-scale, so its result is bit-identical whatever else shares the launch (batch-invariant numerics: what lets
     * one reference batch be sharded over GPUs sequence by sequence). */
    int32_t absmax_batch_stride;
} v2ce_conv3d_desc;
#define V2CE_LAYOUT_PLANAR 0
#define V2CE_LAYOUT_C16 1

/* V2CE_PRECISION_F16X2 (3x3x3 kernels, channel counts multiples of 16): every operand is split into
 * two fp16 numbers (22 bits) and each k-step is three v_mfma_f32_32x32x16_f16 with f32 accumulation;
 * w_packed must then be the buffer written by v2ce_pack_weights_f16x2
 * (v2ce_pack_weights_f16x2_bytes: two fp16 planes of Cout*Cin*27 + {max |w|, pre-scale} as 2 floats;
 * the weight pre-scale is the power of two that puts max |w/sigma| in [2^14, 2^15)).  Error against an
 * f64 evaluation equals the exact-f32 path's (profiles/r01_d_precision_report.json). */
#define V2CE_PRECISION_F32 0
#define V2CE_PRECISION_F16X2 1
size_t v2ce_pack_weights_f16x2_bytes(int Cout, int Cin, int k3);
int v2ce_pack_weights_f16x2(const float *w, int Cout, int Cin, int k3, const float *sigma, void *w_f16x2,
                            v2ce_stream_t stream);

/* Range tracking (all three may be NULL): y_absmax [2] device floats, zeroed by the caller: [0] receives
 * max |y| of the launch by atomic max; [1] receives the split-half launch's RANGE GUARD value E (below;
 * exact-f32 launches leave it 0; v2ce_conv3d_fwd_pred with y == NULL writes only [1]).
 * x0_absmax / x1_absmax [1] are the slots the launches that produced x0 / x1 wrote (any upper bound of
 * max |x| works); the split-half kernel derives its power-of-two activation pre-scale from them on the
 * device, so no magnitude can overflow fp16.  With x0_absmax NULL the split-half kernel uses the fixed
 * pre-scale 16 and requires |x| < 4094.  Ignored (inputs) by the exact-f32 kernels; both precisions
 * record y_absmax[0].
 *
 * Range guard.  One power-of-two scale per tensor puts max |x| (max |w|) in [2^14, 2^15); an element
 * more than ~18 binades below the maximum has an fp16-subnormal lo half and is represented to 2^-25 of
 * the SCALED range instead of to 2^-22 relative.  The kernel reports the worst case this can cost one
 * of its outputs:  E = max_co |scale[co]| * K * 2^-25 * (max|w| / x_scale + max|x| / w_scale),
 * K = Cin * ksize^3 (fused shortcut: the larger of the two convs' values).  E <= V2CE_RANGE_GUARD_LIMIT
 * guarantees the split-half result within that absolute distance of the exact-f32 kernels' for this
 * layer, whatever the distribution of the values (synthetic weights, 346x260: max E 3e-7); the host
 * model accumulates the maximum over a clip and repeats the clip on the exact-f32 kernels when it is
 * exceeded (v2ce_3d.V2ce3d.range_guard_value, glue.run_guarded).  The bound is a worst case: a single
 * 1e4 outlier among O(1) activations trips it, although the measured error there is still < 1e-6. */
#define V2CE_RANGE_GUARD_LIMIT 2.5e-6f
int v2ce_conv3d_fwd(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                    const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                    const float *scale, const float *shift, const float *residual, float *y,
                    const float *x0_absmax, const float *x1_absmax, float *y_absmax,
                    v2ce_stream_t stream);

/* v2ce_conv3d_fwd with a fused 1x1x1 head (the UNet's last two layers in one launch: the 32-channel
 * 3x3x3 conv of the last decoder block, scripts/submodules.py:249-264, and `pred`,
 * scripts/unet_2layer.py:374 / submodules.py:85-124 with relu): pred_y[b][t][o][h][w] =
 * relu(sum_c pred_w[o][c] * y[b][t][c][h][w] + pred_b[o]), o < pred_cout <= 32, evaluated on the conv's
 * accumulators before they leave the registers.  Requires precision F16X2, ksize 3, stride 1, Cout 32,
 * act RELU.  pred_w = table of v2ce_pack_pred_weights_f16x2 (from the [pred_cout][32] f32 weights),
 * pred_b = 32 floats (zero padded).  y may be NULL when only the head's output is wanted. */
size_t v2ce_pack_pred_weights_f16x2_bytes(void);
int v2ce_pack_pred_weights_f16x2(const float *w, int cout, int cin, void *table, v2ce_stream_t stream);
int v2ce_conv3d_fwd_pred(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                         const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                         const float *scale, const float *shift, const float *residual, float *y,
                         const float *x0_absmax, const float *x1_absmax, float *y_absmax,
                         const void *pred_w, const float *pred_b, int pred_cout, float *pred_y,
                         v2ce_stream_t stream);

/* v2ce_conv3d_fwd with the block's 1x1x1 shortcut fused in (scripts/submodules.py:249-264: conv1 and
 * `downsample` read the same input; the centre tap of the 3x3x3 conv touches exactly the (strided)
 * positions the shortcut reads): sc_y = sc_scale[co] * (sc_w (*) x) + sc_shift[co], no activation.
 * sc_w = v2ce_pack_weights_f16x2 of the [Cout][Cin][1] shortcut weights.  Requires precision F16X2,
 * ksize 3 and (stride 2 or Cout <= 32): the kernel variants whose waves own one 32-channel fragment
 * row and can hold the second accumulator set. */
int v2ce_conv3d_fwd_sc(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                       const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                       const float *scale, const float *shift, float *y,
                       const float *x0_absmax, const float *x1_absmax, float *y_absmax,
                       const void *sc_w, const float *sc_scale, const float *sc_shift, float *sc_y,
                       v2ce_stream_t stream);

/* v2ce_conv3d_fwd with a 1x1x1 convolution FOLDED into its K loop: y = act(scale[co] * (conv3x3x3(x) + conv1x1x1'(tx))[co] +
 * shift[co]) in ONE accumulator -- a whole residual block tail, scripts/submodules.py:249-264
 * relu(bn2(conv2(t)) + bn_d(conv_d(x))), with the caller folding the two affine maps: tail weights Wd' = Wd * sd[co] / s2[co],
 * scale = s2, shift = shift2 + shift_d.  No shortcut tensor is written or read, the shortcut's launch disappears.
 * tail_desc describes the 1x1x1 conv (ksize 1, stride 1 or 2, its own virtual input tx0 (++ tx1) with index maps, layout
 * C16; B, T, Cout, Hout, Wout equal to desc's); tail_w = v2ce_pack_weights_f16x2 of the [Cout][Cin_t][1] weights; the
 * tail inputs carry their own range slots (tx0_absmax / tx1_absmax, same batch stride as desc's).  The accumulators are
 * rescaled by an exact power of two between the two parts (the parts have different pre-scales); the range-guard value
 * is the SUM of the two parts' bounds.  Requires precision F16X2, ksize 3, stride 1, Cout >= 64. */
int v2ce_conv3d_fwd_tail(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                         const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                         const float *scale, const float *shift, float *y,
                         const float *x0_absmax, const float *x1_absmax, float *y_absmax,
                         const v2ce_conv3d_desc *tail_desc, const float *tx0, const float *tx1,
                         const int32_t *thmap, const int32_t *twmap, const void *tail_w,
                         const float *tx0_absmax, const float *tx1_absmax, v2ce_stream_t stream);
/* Round 6: the same tail behind the 32-channel convolution that carries the fused 1x1x1 head (v2ce_conv3d_fwd_pred: Cout = 32, ReLU,
 * pred_* as there; y may be NULL) -- conv2 + pred of the last decoder block (submodules.py:249-264 on unet_2layer.py:358-365,374) with
 * the block's shortcut split by source: the skip channels ride as the tail, the upsampled source's share bn_d-scale * Wd[:, :C0] * x0 is
 * computed at the source's resolution and arrives as `residual`, a LOW-resolution tensor [B][T][Cout/16][res_h = ceil(Hout / 2)]
 * [res_w_pitch][16] read at (h >> 1, w >> 1) and added before the activation (residual may be NULL; a full-resolution residual is not
 * taken).  One source x0 (channels-last-16), no index maps on the main conv. */
int v2ce_conv3d_fwd_tail_pred(const v2ce_conv3d_desc *desc, const float *x0, const float *w_packed, const float *scale,
                              const float *shift, float *y, const float *x0_absmax, float *y_absmax, const void *pred_w,
                              const float *pred_b, int pred_cout, float *pred_y, const v2ce_conv3d_desc *tail_desc,
                              const float *tx0, const float *tx1, const int32_t *thmap, const int32_t *twmap,
                              const void *tail_w, const float *tx0_absmax, const float *tx1_absmax, const float *residual,
                              int res_h, int res_w_pitch, v2ce_stream_t stream);

/* The UNet's first layer (scripts/unet_2layer.py:341: ConvLayer3D(2, 32, 3, padding 1) + LeakyReLU; scripts/submodules.py:96,
 * 115-124) in split-half arithmetic: K = 54 as four 16-wide k-steps of three fp16 MFMAs, the output -- 16x the input -- as
 * 1 KB contiguous stores of the channels-last-16 layout.  desc: C0 = 2, C1 = 0, Cout = 32, ksize 3, stride 1, layout C16 (the
 * layout of y; x is the planar network input [B][T][2][H][W0_pitch]).  w_table = v2ce_pack_head_weights_f16x2 of the
 * [32][2][27] f32 weights; bias [32].  x_absmax: per batch element (desc.absmax_batch_stride) max |x| -- v2ce_absmax_batch
 * computes it -- or NULL (|x| < 4094 required); y_absmax as for v2ce_conv3d_fwd (max |y|, range-guard value). */
size_t v2ce_pack_head_weights_f16x2_bytes(void);
int v2ce_pack_head_weights_f16x2(const float *w, void *table, v2ce_stream_t stream);
int v2ce_conv3d_head_f16x2(const v2ce_conv3d_desc *desc, const float *x, const void *w_table, const float *bias, float *y,
                           const float *x_absmax, float *y_absmax, v2ce_stream_t stream);
/* slots[b * stride] = max |x[b][0 .. n)| for b < B (slots zeroed by the caller): the range slot of a network INPUT. */
int v2ce_absmax_batch(const float *x, int B, long long n, float *slots, int stride, v2ce_stream_t stream);

/* conv1 of a decoder block (scripts/unet_2layer.py:358-365, scripts/submodules.py:249-264): the 3x3x3 conv whose input is the
 * virtual concat  nearest-upsample-2x(x0) ++ x1, with the upsampled channels PHASE-FOLDED.  ATen's nearest map for an exact
 * 2x size ratio (H0 = ceil(Hin / 2), W0 = ceil(Win / 2): every decoder size of the network) is src = dst >> 1, so an output
 * position of parity (ph, pw) sees only 2 x 2 distinct source pixels through its 3 x 3 (H, W) taps: 12 taps with pre-summed
 * weights instead of 27 on the C0 channels (5/9 of their multiplies never happen; 16.5 % of the whole network's).  Odd output
 * sizes (the last, even, row / column has the convolution's zero padding as its +1 neighbour) are exact through correction
 * lists in the tiles that hold that row / column (csrc/conv3d_up.hip).  Same result as v2ce_conv3d_fwd[_sc] on the same
 * tensors up to the summation order of the pre-summed weights (f32 sums of <= 4 quotients w / sigma).
 * desc: ksize 3, stride 1, precision F16X2, layout C16, H0 = ceil(Hin / 2), W0 = ceil(Win / 2), C0 % 16 == C1 % 16 == 0, C1 > 0,
 * Cout % 32 == 0; no index maps (the map is implied).  w_up = buffer of v2ce_pack_weights_f16x2_up: a v2ce_pack_weights_f16x2
 * buffer of the [Cout][C0 + C1][27] weights (valid as w_packed of v2ce_conv3d_fwd too) followed by the folded region of the
 * first C0 input channels (108 tap slots of [C0 / 16][Cout][16] per fp16 plane), one common power-of-two pre-scale.
 * sc_w .. sc_y: optional fused 1x1x1 shortcut exactly as v2ce_conv3d_fwd_sc (Cout <= 32); all NULL otherwise. */
size_t v2ce_pack_weights_f16x2_up_bytes(int Cout, int C0, int C1);
int v2ce_pack_weights_f16x2_up(const float *w, int Cout, int C0, int C1, const float *sigma, void *w_up, v2ce_stream_t stream);
int v2ce_conv3d_fwd_up2(const v2ce_conv3d_desc *desc, const float *x0, const float *x1, const void *w_up,
                        const float *scale, const float *shift, float *y, const float *x0_absmax,
                        const float *x1_absmax, float *y_absmax, const void *sc_w, const float *sc_scale,
                        const float *sc_shift, float *sc_y, v2ce_stream_t stream);
/* The upsampled channels' share of the same convolution alone: y = scale * conv(upsample(x0); W[:, :C0]) + shift, no activation (desc.act
 * is ignored), for a caller that computes the skip channels' share with another launch and adds this output as that launch's residual
 * (conv1 of the wide decoder blocks: the skip channels on the Winograd-T kernel, v2ce_conv3d_fwd_wt with a v2ce_pack_weights_f16x2_wt_slice
 * buffer of channels [C0, C0 + C1)).  desc and w_up exactly as for v2ce_conv3d_fwd_up2 (the buffer of the whole layer). */
int v2ce_conv3d_fwd_up2_part(const v2ce_conv3d_desc *desc, const float *x0, const void *w_up, const float *scale, const float *shift,
                             float *y, const float *x0_absmax, const float *x1_absmax, float *y_absmax, v2ce_stream_t stream);
/* Name of the kernel instantiation v2ce_conv3d_fwd_up2 would launch ("conv3d_up_kernel<WCO,CO_FR,PO_FR,FUSE>"). */
int v2ce_conv3d_up2_variant(const v2ce_conv3d_desc *desc, int with_shortcut, char *name, size_t cap);

/* A 3x3x3, stride-1, one-source convolution of a residual block (scripts/submodules.py:249-264: conv2 of every block, conv1 of the
 * two middle blocks) with the Winograd transform F(2,3) along T: per pair of time steps four transformed 3x3 (H, W) convolutions
 * instead of six direct tap rows -- 2/3 of the multiplies of v2ce_conv3d_fwd.  Transforms, products (split-half fp16 MFMA) and sums
 * in f32: the result differs from v2ce_conv3d_fwd's by rounding (1e-6 relative), inside the 1e-5 parity bar of the network
 * (csrc/conv3d_wt.hip; tools/winograd_t_sim.py).  desc: ksize 3, stride 1, precision F16X2, layout C16, C1 = 0 (x = x0 of C0
 * channels, H0 = Hin = Hout, W0 = Win = Wout), C0 % 16 == 0, Cout % 64 == 0.  w_wt = buffer of v2ce_pack_weights_f16x2_wt
 * (the transformed weights of W / sigma: 36 tap slots of [C0 / 16][Cout][16] per fp16 plane, then { 1.5 max |W / sigma| >= max |G|,
 * pre-scale, 0, 0 }).
 * residual (may be NULL): added before the activation, layout of y.  x_absmax / y_absmax as for v2ce_conv3d_fwd. */
size_t v2ce_pack_weights_f16x2_wt_bytes(int Cout, int Cin);
int v2ce_pack_weights_f16x2_wt(const float *w, int Cout, int Cin, const float *sigma, void *w_wt, v2ce_stream_t stream);
/* the same for input channels [ci0, ci0 + Cin) of a [Cout][Cin_total][27] tensor (ci0 % 16 == 0); the pre-scale bound is taken over the
 * whole tensor (what v2ce_sn_update_batch writes for v2ce_sn_layer.packed_skip) */
int v2ce_pack_weights_f16x2_wt_slice(const float *w, int Cout, int Cin_total, int ci0, int Cin, const float *sigma, void *w_wt,
                                     v2ce_stream_t stream);
int v2ce_conv3d_fwd_wt(const v2ce_conv3d_desc *desc, const float *x, const void *w_wt, const float *scale, const float *shift,
                       const float *residual, float *y, const float *x_absmax, float *y_absmax, v2ce_stream_t stream);
/* The same with the block's folded 1x1x1 shortcut in the launch's K loop -- the contract of v2ce_conv3d_fwd_tail (tail_desc,
 * tx0 / tx1 / maps, tail_w = v2ce_pack_weights_f16x2 of the [Cout][tC0 + tC1][1] weights, range slots of the tail inputs), on the
 * Winograd-T kernel: the tail's terms enter the transform domain split over the four slots (csrc/conv3d_wt.hip).  The tail's
 * channel count must be a multiple of 64. */
int v2ce_conv3d_fwd_wt_tail(const v2ce_conv3d_desc *desc, const float *x, const void *w_wt, const float *scale, const float *shift,
                            float *y, const float *x_absmax, float *y_absmax, const v2ce_conv3d_desc *tail_desc, const float *tx0,
                            const float *tx1, const int32_t *thmap, const int32_t *twmap, const void *tail_w,
                            const float *tx0_absmax, const float *tx1_absmax, const float *residual, int res_h, int res_w_pitch,
                            v2ce_stream_t stream);
/* residual (may be NULL): added before the activation like v2ce_conv3d_fwd_wt's.  res_h > 0: it is a LOW-resolution tensor
 * [B][T][Cout/16][res_h = ceil(Hout / 2)][res_w_pitch][16] read at (h >> 1, w >> 1) -- the 2x nearest upsample of a decoder block:
 * the upsampled source's share of the folded shortcut, bn_d-scale * Wd[:, :C0] * x0, is computed at the source's resolution (a quarter
 * of the positions) and only the skip channels ride as the tail. */
/* Name of the kernel instantiation v2ce_conv3d_fwd_wt[_tail] would launch ("conv3d_wt_kernel<CO_FR,PO_FR,RES,TAIL>");
 * with_residual: 0 | 1 | 2 = the tail form | 3 = the tail form with a residual. */
int v2ce_conv3d_wt_variant(const v2ce_conv3d_desc *desc, int with_residual, char *name, size_t cap);

/* Name of the kernel instantiation v2ce_conv3d_fwd would launch for desc ("conv3d_kernel<KS,S,
 * CO_FR,PO_FR,CK,EPT>", as it appears demangled in rocprofv3 traces); mapped != 0 means hmap/wmap
 * would be non-NULL.  Launches nothing.  Used by bench.py to attribute event timings. */
int v2ce_conv3d_variant(const v2ce_conv3d_desc *desc, int mapped, char *name, size_t cap);
/* Same for the fused entry points: fuse = 1 (v2ce_conv3d_fwd_pred), 2 (v2ce_conv3d_fwd_sc), 3 (v2ce_conv3d_fwd_tail), 0 (plain); + 4 when the
 * launch has a residual (the kernels are instantiated per case: the last template argument). */
int v2ce_conv3d_variant_fused(const v2ce_conv3d_desc *desc, int mapped, int fuse, char *name, size_t cap);

/* Weight re-layout [Cout][Cin][k^3] -> [Cin][k^3][Cout], optionally divided elementwise by
 * *sigma (spectral_norm.py:31 `w / sigma.expand_as(w)`; sigma NULL = plain re-layout). */
int v2ce_pack_weights(const float *w, int Cout, int Cin, int k3, const float *sigma,
                      float *w_packed, v2ce_stream_t stream);

/* Voxeliser (the inverse of LDATI): gen_discretized_event_volume of
 * train/scripts/utils/events_utils.py:118-175.  SoA events on the device -> volume [2*bins][H][W]
 * f32 (zeroed here): time rescaled to [0, bins-1] over the set's own [t_min, t_max] (computed here
 * into t_range [2] int64, device), each event split between its floor and ceil bin, polarity 1 in
 * planes [0,bins), polarity 0 in [bins,2*bins).  Same f32 arithmetic as the reference per event;
 * accumulation by float atomics (order differs).  n > 0 and t_max > t_min required (the reference
 * raises / produces NaN otherwise); events with x/y outside the volume are skipped (the reference
 * asserts). */
int v2ce_voxelize_events(const int64_t *ts, const int16_t *x, const int16_t *y, const int8_t *p, int64_t n,
                         int bins, int H, int W, float *volume, int64_t *t_range, v2ce_stream_t stream);

/* Ablation samplers of the reference's stage-2 study, csrc/sampler.hip:
 *   V2CE_SAMPLER_RANDOM / _EVEN  sample_voxel_baseline(random=True / even=True) of
 *                                train/scripts/stage2/sample_methods/random_even_sample.py:115-169
 *   V2CE_SAMPLER_PURE_SLOPE      sample_voxel_statistical of .../pure_slope_sample.py:57-149
 * vox [B][2][10][H][W] f32 (read only: the pure-slope reference folds bin 9 into bin 8 IN the caller's tensor,
 * here the fold happens in registers).  Per voxel: floor(y) events + one more with probability frac(y).  Output per
 * frame in the order of np.sort(order='timestamp') = lexicographic (timestamp, x, y, polarity); frames back to back.
 * Random draws: V2CE_RNG_REPLAY reads u_int [B][2][10][H][W][replay_M] (times of the floor(y) events; replay_M >=
 * max floor(y)), u_dec and u_bern [B][2][10][H][W] (time of / Bernoulli draw for the fractional event: it exists iff
 * u_bern < frac(y)); V2CE_RNG_PHILOX draws them from Philox4x32-10, counter (pixel, j >> 2, 32*kind + 10*P + c,
 * frame_base + b), kind 0 / 1 / 2 = u_int / u_dec / u_bern, key = seed.  _EVEN reads u_bern only.
 * Two-phase like LDATI: count -> the caller reads frame_counts [B] (int64, device) and max_int, allocates the SoA
 * outputs and the workspace -> emit.  status [1] (device int32) != 0: a timestamp fell outside the frame's key range
 * (NaN or inf from a degenerate slope; the reference's output is platform-defined there). */
#define V2CE_SAMPLER_RANDOM 0
#define V2CE_SAMPLER_EVEN 1
#define V2CE_SAMPLER_PURE_SLOPE 2
typedef struct {
    int mode;     /* V2CE_SAMPLER_* */
    int rng_mode; /* V2CE_RNG_* */
    double fps, t0;
    uint64_t seed;
    int frame_base;
    int replay_M;
    const float *u_int, *u_dec, *u_bern;
    const float *pooled; /* _PURE_SLOPE with pooling_type 'avg' / 'weighted' (pure_slope_sample.py:79-85): the output of
                          * v2ce_sampler_pool [B][2][10][H][W]; shapes only the slope parameters.  NULL = pooling 'none'. */
} v2ce_sampler_options;
/* y_pooled of pure_slope_sample.py:79-85 (V2CE_POOL_WEIGHTED: 3x3 [[1,2,1],[2,4,2],[1,2,1]]/16 conv, V2CE_POOL_AVG:
 * k x k mean; zero padding).  Sums of non-integer f32 values in row-major tap order: equal to the reference's up to
 * its backend's summation order (events within 1 us of the reference's: tests/test_gpu_samplers.py). */
int v2ce_sampler_pool(const float *vox, int B, int H, int W, int pooling_type, int pooling_kernel_size, float *pooled,
                      v2ce_stream_t stream);
int v2ce_sampler_count(const float *vox, int B, int H, int W, const v2ce_sampler_options *options,
                       int64_t *frame_counts, int32_t *max_int, v2ce_stream_t stream);
size_t v2ce_sampler_workspace_bytes(int64_t total_events);
int v2ce_sampler_emit(const float *vox, int B, int H, int W, const v2ce_sampler_options *options, int64_t total_events,
                      int64_t *ts, int16_t *x, int16_t *y, int8_t *p, void *workspace, size_t workspace_bytes,
                      int32_t *status, v2ce_stream_t stream);

/* One spectral-norm power iteration (spectral_norm.py:19-31), in place on u [rows], v [cols]:
 *   v = W^T u / (|W^T u| + 1e-12); u = W v / (|W v| + 1e-12); sigma = u . (W v)
 * w_bar [rows][cols] f32; sigma [1] f32 out; workspace >= v2ce_sn_workspace_bytes(rows, cols). */
size_t v2ce_sn_workspace_bytes(int rows, int cols);
int v2ce_sn_power_iter(float *u, float *v, const float *w_bar, int rows, int cols, float *sigma,
                       void *workspace, size_t workspace_bytes, v2ce_stream_t stream);

/* The same for ALL spectral-norm layers of a forward pass in six launches, with the split-half weight
 * re-pack (v2ce_pack_weights_f16x2 of w_bar / sigma, incl. its {max |w/sigma|, pre-scale} tail) fused behind it:
 * replaces 12 x (v2ce_sn_power_iter + v2ce_pack_weights_f16x2) = 84 launches.  Results are bit-identical to the
 * per-layer calls.  layers: HOST array (passed to the kernels by value), at most 16 entries; cols = Cin * k3,
 * Cin % 16 == 0; packed = buffer of v2ce_pack_weights_f16x2_bytes(rows, Cin, k3). */
typedef struct {
    const float *w_bar;   /* [rows][cols] */
    float *u, *v;         /* [rows], [cols]: updated in place */
    void *packed;         /* out */
    int32_t rows, cols, k3;
    int32_t up_c0;        /* 0, or: `packed` is a v2ce_pack_weights_f16x2_up buffer (decoder conv1, v2ce_conv3d_fwd_up2) whose first
                           * up_c0 input channels -- the nearest-upsampled source -- are also packed phase-folded */
    int32_t wt;           /* 1: `packed` is a v2ce_pack_weights_f16x2_wt buffer instead (Winograd-T planes of w_bar / sigma, k3 = 27,
                           * up_c0 = 0); 0: the plain planes */
    int32_t flags;        /* 0, or a sum of V2CE_SN_NO_PACK / V2CE_SN_NO_ITERATE (round 6, below) */
    void *packed_skip;    /* NULL, or (up_c0 > 0): also write the Winograd-T planes of input channels [up_c0, Cin) of w_bar / sigma --
                           * a v2ce_pack_weights_f16x2_wt_bytes(rows, Cin - up_c0) buffer, as v2ce_pack_weights_f16x2_wt_slice does --
                           * for the split launch of a decoder conv1 (v2ce_conv3d_fwd_up2_part + v2ce_conv3d_fwd_wt) */
    /* Round 6: the weights w_bar are constant, only the scalar sigma changes from call to call (spectral_norm.py:31 divides every
     * weight by it).  A caller may pack w_bar ONCE (sigma = 1) and carry 1 / sigma in the convolution's epilogue scale instead:
     *   V2CE_SN_NO_PACK     iterate only: u, v, sigma are updated, `packed` (may be NULL) is left alone;
     *                       scale_out [rows] (may be NULL) = bn_scale [rows] / sigma, inv_sigma_out [1] (may be NULL) = 1 / sigma
     *   V2CE_SN_NO_ITERATE  pack only: u, v (may be NULL) are left alone and `packed` = the plain planes (k3 = 1 or 27, up_c0 = wt = 0)
     *                       of w_bar / *sigma_src, its {max |w / sigma|, pre-scale} tail from wmax = max |w_bar| -- what a block's
     *                       folded 1x1x1 shortcut needs once its convolution carries 1 / sigma in the scale: Wd' * sigma, i.e.
     *                       sigma_src = that layer's inv_sigma_out of a PREVIOUS v2ce_sn_update_batch call on the stream. */
    const float *bn_scale;
    float *scale_out, *inv_sigma_out;
    const float *sigma_src;
    float wmax;
    int32_t reserved;     /* 0 */
} v2ce_sn_layer;
#define V2CE_SN_NO_PACK 1
#define V2CE_SN_NO_ITERATE 2
size_t v2ce_sn_batch_workspace_bytes(const v2ce_sn_layer *layers, int n);
int v2ce_sn_update_batch(const v2ce_sn_layer *layers, int n, void *workspace, size_t workspace_bytes,
                         v2ce_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* V2CE_HIP_H */
