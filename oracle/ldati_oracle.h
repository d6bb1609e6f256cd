/*
 * ldati_oracle.h -- CPU restatement of the reference LDATI sampler (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker for the HIP path, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  It restates, in plain scalar C, the algorithm of
 *   /root/reference/scripts/LDATI.py:13-51   (slope)          -> oracle_slope
 *   /root/reference/scripts/LDATI.py:80-106  (relocation)     -> oracle_relocate
 *   /root/reference/scripts/LDATI.py:126-214 (timestamps)     -> oracle_single_ts / oracle_multi_ts
 *   /root/reference/scripts/LDATI.py:217-310 (pick + sort)    -> v2ce_oracle_ldati_emit
 * Pinned against the reference itself (imported on CPU in the build container) through the golden
 * vectors under tests/golden/ (generator: oracle/make_goldens.py) and the notebook known-answer of
 * train/scripts/stage2/vis_stage2.ipynb.
 */
#ifndef V2CE_LDATI_ORACLE_H
#define V2CE_LDATI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2CE_ORACLE_STRATEGY_SLOPE 0 /* additional_events_strategy='slope' */
#define V2CE_ORACLE_STRATEGY_NONE 1  /* 'none': only single-event voxels emit (LDATI.py:206-207,241) */
#define V2CE_ORACLE_STRATEGY_RANDOM 2 /* 'random': the raw uniform IS the time offset in seconds (LDATI.py:173-174) */

#define V2CE_ORACLE_POOL_NONE 0
#define V2CE_ORACLE_POOL_AVG 1      /* nn.AvgPool2d(k, stride 1, padding k//2) of the counts (LDATI.py:181) */
#define V2CE_ORACLE_POOL_WEIGHTED 2 /* 3x3 [[1,2,1],[2,4,2],[1,2,1]]/16 conv, zero padding (LDATI.py:178-180) */

#define V2CE_ORACLE_RNG_REPLAY 0 /* uniforms read from a dense [B,2,9,H,W,max_n] tensor      */
#define V2CE_ORACLE_RNG_PHILOX 1 /* Philox4x32-10, counter (pixel, j>>2, p*9+c, frame), key seed */

/* Per-pixel relocation, LDATI.py:94-106.  y10 has stride `stride` floats between bins. */
void v2ce_oracle_relocate(const float *y10, int64_t stride, int64_t n[9], float debt[9]);

/* Phase 1: per-(frame, bin) event counts and the chunk-wide max count (LDATI.py:169).
 * vox: [B,2,10,H,W] f32.  seg_counts: [B*9] int64.  Returns 0. */
int v2ce_oracle_ldati_count(const float *vox, int B, int H, int W, int strategy, int64_t *seg_counts,
                            int32_t *max_n);

/* Phase 2: emit the events of every (frame, bin) segment in the reference's *stable* order
 * (neg singles, neg multis, pos singles, pos multis; then stable sort by timestamp), written at
 * seg_offsets[b*9+c] (exclusive prefix of seg_counts).  Outputs are SoA.
 * frame_base: global index of frame 0 (only used by the Philox counter).  Returns 0, or <0 on a
 * bad argument. */
int v2ce_oracle_ldati_emit(const float *vox, int B, int H, int W, double fps, double t0,
                           int strategy, int rng_mode, const float *uniforms, int replay_max_n, uint64_t seed,
                           int64_t frame_base, const int64_t *seg_offsets, int64_t *ts,
                           int16_t *x, int16_t *y, int8_t *p);

/* The same with the remaining options of sample_voxel_statistical (LDATI.py:126): bidirectional
 * relocation (LDATI.py:107-122), pooled counts for the slope (LDATI.py:177-182), 'random' strategy. */
void v2ce_oracle_relocate2(const float *y10, int64_t stride, int bidirectional, int64_t n[9], float tend[9]);
int v2ce_oracle_ldati_count2(const float *vox, int B, int H, int W, int strategy, int bidirectional,
                             int64_t *seg_counts, int32_t *max_n);
int v2ce_oracle_ldati_emit2(const float *vox, int B, int H, int W, double fps, double t0,
                            int strategy, int bidirectional, int pooling, int pooling_k,
                            int rng_mode, const float *uniforms, int replay_max_n, uint64_t seed,
                            int64_t frame_base, const int64_t *seg_offsets, int64_t *ts,
                            int16_t *x, int16_t *y, int8_t *p);

/* Materialise the Philox uniforms as the dense [B,2,9,H,W,max_n] tensor the reference would have
 * drawn with torch.rand (used only by oracle/make_goldens.py to feed the reference). */
void v2ce_oracle_philox_fill(float *out, int B, int H, int W, int max_n, uint64_t seed,
                             int64_t frame_base);

/* One Philox uniform (24-bit mantissa, [0,1)). */
float v2ce_oracle_philox_uniform(uint64_t seed, uint32_t pixel, uint32_t j, uint32_t pc,
                                 uint32_t frame);

#ifdef __cplusplus
}
#endif
#endif
