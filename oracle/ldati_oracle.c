/*
 * ldati_oracle.c -- CPU restatement of the reference LDATI sampler.  TEST INFRASTRUCTURE ONLY:
 * the product (v2ce-toolbox_amd/) never links, loads or calls this file.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math; x86-64 SSE2 arithmetic,
 * so every f32/f64 operation rounds once, exactly like the separate ATen CPU ops it restates).
 *
 * Reference: /root/reference/scripts/LDATI.py (CPU-torch semantics; every citation below is a
 * line range of that file).  Parity pin: tests/golden/ldati_*.npz, generated from the imported
 * reference by oracle/make_goldens.py, plus the notebook known-answer
 * (train/scripts/stage2/vis_stage2.ipynb cells 1-2).
 */
#include "ldati_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * Relocation of fractional voxel mass into integer counts -- LDATI.py:94-106
 *   for i in 0..8: _n = y[i] - debt (f32); n_i = ceil(_n - 1e-6) (f32 scalar); debt = n_i - _n
 *   n_8 += int32(y[9] - debt)
 * ------------------------------------------------------------------------------------------- */
void v2ce_oracle_relocate(const float *y10, int64_t stride, int64_t n[9], float debt[9]) {
    const float eps = (float)1e-6; /* python scalar 1e-6 is cast to f32 by the tensor op */
    float d = 0.0f;
    for (int i = 0; i < 9; ++i) {
        float r = y10[i * stride] - d;   /* LDATI.py:98 */
        float c = ceilf(r - eps);        /* LDATI.py:99 */
        d = c - r;                       /* LDATI.py:100 */
        n[i] = (int64_t)c;               /* LDATI.py:101 (f32 -> int64 store) */
        debt[i] = d;                     /* LDATI.py:103 (f32 value held in an f64 tensor) */
    }
    n[8] += (int64_t)(int32_t)(y10[9 * stride] - d); /* LDATI.py:106 (.int() truncates) */
}

/* ---------------------------------------------------------------------------------------------
 * bidirectional=True -- LDATI.py:107-122.  Bins 0..3 run the forward recurrence; bins 8,7,6 run
 * backwards from bless = y[9]: tendency = bless; n = floor((y + bless) + 1e-6);
 * bless = max(0, (y - n) + bless); bin 5 merges both: tendency = bless - debt, n = ceil((y + bless) - debt);
 * bin 4 is never written by the reference (range(4) / range(8, 5, -1) / 5) and stays 0.
 * f32 arithmetic operation by operation; tendency is stored into an f64 tensor (kept as f32 here).
 * ------------------------------------------------------------------------------------------- */
void v2ce_oracle_relocate2(const float *y10, int64_t stride, int bidirectional, int64_t n[9], float tend[9]) {
    if (!bidirectional) {
        v2ce_oracle_relocate(y10, stride, n, tend);
        return;
    }
    const float eps = (float)1e-6;
    float d = 0.0f;
    for (int i = 0; i < 9; ++i) { n[i] = 0; tend[i] = 0.0f; }
    for (int i = 0; i < 4; ++i) {                 /* LDATI.py:96-103 with from_left_until = 4 */
        float r = y10[i * stride] - d;
        float c = ceilf(r - eps);
        d = c - r;
        n[i] = (int64_t)c;
        tend[i] = d;
    }
    float bless = y10[9 * stride];                /* LDATI.py:108 */
    for (int i = 8; i > 5; --i) {                 /* LDATI.py:109-117 */
        const float ys = y10[i * stride];
        tend[i] = bless;
        float t = ys + bless;
        t = floorf(t + eps);
        bless = (ys - t) + bless;
        bless = bless < 0.0f ? 0.0f : bless;      /* torch.clamp(min=0) */
        n[i] = (int64_t)t;
    }
    {                                             /* LDATI.py:119-122, i = C//2 = 5 */
        const float ys = y10[5 * stride];
        tend[5] = bless - d;
        n[5] = (int64_t)ceilf((ys + bless) - d);
    }
}

/* ---------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11), counter = (pixel, j>>2, p*9+c, frame), key = seed.
 * The j&3-th output word gives the uniform (word >> 8) * 2^-24, the same 24-bit convention as
 * torch's CPU uniform_real_distribution<float>.
 * ------------------------------------------------------------------------------------------- */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

static void philox4x32_10(uint32_t ctr[4], uint64_t seed) {
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; ++r) {
        philox_round(ctr, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
}

float v2ce_oracle_philox_uniform(uint64_t seed, uint32_t pixel, uint32_t j, uint32_t pc,
                                 uint32_t frame) {
    uint32_t ctr[4] = {pixel, j >> 2, pc, frame};
    philox4x32_10(ctr, seed);
    return (float)(ctr[j & 3] >> 8) * (1.0f / 16777216.0f);
}

void v2ce_oracle_philox_fill(float *out, int B, int H, int W, int max_n, uint64_t seed,
                             int64_t frame_base) {
    int64_t idx = 0;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < 2; ++p)
            for (int c = 0; c < 9; ++c)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w)
                        for (int j = 0; j < max_n; ++j)
                            out[idx++] = v2ce_oracle_philox_uniform(
                                seed, (uint32_t)(h * W + w), (uint32_t)j, (uint32_t)(p * 9 + c),
                                (uint32_t)(frame_base + b));
}

/* --------------------------------------------------------------------------------------------- */
int v2ce_oracle_ldati_count2(const float *vox, int B, int H, int W, int strategy, int bidirectional,
                             int64_t *seg_counts, int32_t *max_n) {
    const int64_t HW = (int64_t)H * W;
    int64_t mx = 0;
    for (int b = 0; b < B; ++b) {
        int64_t *sc = seg_counts + (int64_t)b * 9;
        for (int c = 0; c < 9; ++c) sc[c] = 0;
        for (int p = 0; p < 2; ++p) {
            const float *base = vox + ((int64_t)(b * 2 + p) * 10) * HW;
            for (int64_t px = 0; px < HW; ++px) {
                int64_t n[9];
                float d[9];
                v2ce_oracle_relocate2(base + px, HW, bidirectional, n, d);
                for (int c = 0; c < 9; ++c) {
                    /* pick_elements keeps n==1 singles and the first n draws of n>=2 voxels
                     * (LDATI.py:228,236-239); n<=0 contributes nothing. */
                    if (strategy == V2CE_ORACLE_STRATEGY_NONE) sc[c] += (n[c] == 1); /* LDATI.py:241 */
                    else if (n[c] > 0) sc[c] += n[c];
                    if (n[c] > mx) mx = n[c]; /* LDATI.py:169 torch.max(y) over all counts */
                }
            }
        }
    }
    /* torch.max of the int64 count tensor; negative maxima cannot size a tensor */
    *max_n = (int32_t)(mx < 0 ? 0 : mx);
    return 0;
}

int v2ce_oracle_ldati_count(const float *vox, int B, int H, int W, int strategy, int64_t *seg_counts,
                            int32_t *max_n) {
    return v2ce_oracle_ldati_count2(vox, B, H, W, strategy, 0, seg_counts, max_n);
}

/* stable merge sort of an index permutation by key */
static void merge_sort_idx(const int64_t *key, int64_t *idx, int64_t *tmp, int64_t n) {
    for (int64_t width = 1; width < n; width *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * width) {
            int64_t mid = lo + width < n ? lo + width : n;
            int64_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (key[idx[j]] < key[idx[i]]) ? idx[j++] : idx[i++];
            while (i < mid) tmp[k++] = idx[i++];
            while (j < hi) tmp[k++] = idx[j++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
}

int v2ce_oracle_ldati_emit(const float *vox, int B, int H, int W, double fps, double t0,
                           int strategy, int rng_mode, const float *uniforms, int replay_max_n, uint64_t seed,
                           int64_t frame_base, const int64_t *seg_offsets, int64_t *ts_out,
                           int16_t *x_out, int16_t *y_out, int8_t *p_out) {
    return v2ce_oracle_ldati_emit2(vox, B, H, W, fps, t0, strategy, 0, V2CE_ORACLE_POOL_NONE, 3, rng_mode, uniforms,
                                   replay_max_n, seed, frame_base, seg_offsets, ts_out, x_out, y_out, p_out);
}

/* Pooled counts of one (frame, polarity) plane stack for the slope -- LDATI.py:177-182.
 * 'weighted': F.conv2d with [[1,2,1],[2,4,2],[1,2,1]]/16, padding 1 (zeros); 'avg': nn.AvgPool2d(k,
 * stride 1, padding k//2), count_include_pad: sum / (k*k).  The counts are integers and the weights
 * powers of two over 16, so every partial sum is exact in f32 whatever the summation order. */
static void pool_plane(const int64_t *nn /* [HW][9] */, int H, int W, int pooling, int k, float *yp /* [HW][9] */) {
    static const float w3[3][3] = {{1, 2, 1}, {2, 4, 2}, {1, 2, 1}};
    const int r = pooling == V2CE_ORACLE_POOL_WEIGHTED ? 1 : k / 2;
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
            for (int c = 0; c < 9; ++c) {
                float acc = 0.0f;
                for (int dh = -r; dh <= r; ++dh)
                    for (int dw = -r; dw <= r; ++dw) {
                        const int hh = h + dh, ww = w + dw;
                        if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                        const float v = (float)nn[((int64_t)hh * W + ww) * 9 + c];
                        acc += pooling == V2CE_ORACLE_POOL_WEIGHTED ? v * (w3[dh + 1][dw + 1] / 16.0f) : v;
                    }
                if (pooling == V2CE_ORACLE_POOL_AVG) acc = acc / (float)(k * k);
                yp[((int64_t)h * W + w) * 9 + c] = acc;
            }
}

int v2ce_oracle_ldati_emit2(const float *vox, int B, int H, int W, double fps, double t0,
                            int strategy, int bidirectional, int pooling, int pooling_k,
                            int rng_mode, const float *uniforms, int replay_max_n, uint64_t seed,
                            int64_t frame_base, const int64_t *seg_offsets, int64_t *ts_out,
                            int16_t *x_out, int16_t *y_out, int8_t *p_out) {
    if (rng_mode == V2CE_ORACLE_RNG_REPLAY && uniforms == NULL && replay_max_n > 0) return -1;
    const int64_t HW = (int64_t)H * W;

    /* scalars of LDATI.py:145-146 and their f32 casts (tensor op python-scalar => f32 scalar) */
    const double vs = 1.0 / fps / 9.0;            /* voxel_step */
    const float VS = (float)vs;
    const float VS2 = (float)(vs * vs);           /* voxel_step**2 */
    const float INV = (float)(1.0 / vs);          /* 1 / voxel_step */
    const float FPS = (float)fps;
    const float E6 = (float)1e6;
    float offt[9];                                /* arange(0, frame_step, voxel_step) + t0, f32 */
    for (int c = 0; c < 9; ++c) offt[c] = (float)(0.0 + (double)c * vs) + (float)t0;

    int64_t *nn = (int64_t *)malloc((size_t)(2 * HW * 9) * sizeof(int64_t));
    float *dd = (float *)malloc((size_t)(2 * HW * 9) * sizeof(float));
    float *yp = pooling != V2CE_ORACLE_POOL_NONE ? (float *)malloc((size_t)(2 * HW * 9) * sizeof(float)) : NULL;
    if (!nn || !dd || (pooling != V2CE_ORACLE_POOL_NONE && !yp)) { free(nn); free(dd); free(yp); return -2; }

    for (int b = 0; b < B; ++b) {
        for (int p = 0; p < 2; ++p) {
            const float *base = vox + ((int64_t)(b * 2 + p) * 10) * HW;
            for (int64_t px = 0; px < HW; ++px)
                v2ce_oracle_relocate2(base + px, HW, bidirectional, nn + (p * HW + px) * 9, dd + (p * HW + px) * 9);
            if (yp) pool_plane(nn + (int64_t)p * HW * 9, H, W, pooling, pooling_k, yp + (int64_t)p * HW * 9);
        }
        for (int c = 0; c < 9; ++c) {
            const int64_t seg_lo = seg_offsets[(int64_t)b * 9 + c];
            const int64_t seg_n = seg_offsets[(int64_t)b * 9 + c + 1] - seg_lo;
            if (seg_n <= 0) continue;
            int64_t *ets = (int64_t *)malloc((size_t)seg_n * sizeof(int64_t));
            int16_t *ex = (int16_t *)malloc((size_t)seg_n * sizeof(int16_t));
            int16_t *ey = (int16_t *)malloc((size_t)seg_n * sizeof(int16_t));
            int8_t *ep = (int8_t *)malloc((size_t)seg_n * sizeof(int8_t));
            int64_t *idx = (int64_t *)malloc((size_t)seg_n * sizeof(int64_t));
            int64_t *tmp = (int64_t *)malloc((size_t)seg_n * sizeof(int64_t));
            int64_t m = 0;
            /* pick_and_sort: negative (P index 1, polarity 0) first, then positive
             * (P index 0, polarity 1) -- LDATI.py:289-296,302-303 */
            for (int pass = 0; pass < 2; ++pass) {
                const int p = pass == 0 ? 1 : 0;
                const int8_t pol = pass == 0 ? 0 : 1;
                /* singles, row-major -- LDATI.py:228-234; time LDATI.py:156-165 */
                for (int64_t px = 0; px < HW; ++px) {
                    const int64_t *n = nn + (p * HW + px) * 9;
                    if (n[c] != 1) continue;
                    const float D = dd[(p * HW + px) * 9 + c];
                    double t = (double)D / fps / 9.0;   /* y_tendency / fps / C   (f64) */
                    t += (double)offt[c];               /* ts += arange + t0             */
                    t *= 1e6;                           /* ts *= 1e6                     */
                    if (m >= seg_n) goto overflow;
                    ets[m] = (int64_t)t;                /* .to(torch.long) truncates     */
                    ex[m] = (int16_t)(px % W);
                    ey[m] = (int16_t)(px / W);
                    ep[m] = pol;
                    ++m;
                }
                /* multis, row-major then draw index -- LDATI.py:236-244; time LDATI.py:188-212 */
                for (int64_t px = 0; px < HW; ++px) {
                    const int64_t *n = nn + (p * HW + px) * 9;
                    const int64_t nc = n[c];
                    if (nc < 2 || strategy == V2CE_ORACLE_STRATEGY_NONE) continue;
                    /* slope: reflect pad + [-1,0,1] conv -- LDATI.py:25,30,39; (3*sxy-0)/6 :45; on the
                     * pooled counts when pooling is on (LDATI.py:177-188) */
                    float fl, fr, fc;
                    if (yp) {
                        const float *q = yp + (p * HW + px) * 9;
                        fl = c == 0 ? q[1] : q[c - 1];
                        fr = c == 8 ? q[7] : q[c + 1];
                        fc = q[c];
                    } else {
                        fl = (float)(c == 0 ? n[1] : n[c - 1]);
                        fr = (float)(c == 8 ? n[7] : n[c + 1]);
                        fc = (float)nc;
                    }
                    const float sxy = fr - fl;
                    const float k0 = (3.0f * sxy) / 6.0f;
                    const float k = (k0 / VS2) / (fc + (float)1e-8);         /* LDATI.py:188 */
                    const float bb = INV - (VS * k) / 2.0f;                  /* LDATI.py:190 */
                    for (int64_t j = 0; j < nc; ++j) {
                        float u;
                        if (rng_mode == V2CE_ORACLE_RNG_REPLAY) {
                            if (j >= replay_max_n) goto overflow;
                            u = uniforms[((((int64_t)(b * 2 + p) * 9 + c) * HW + px) *
                                          replay_max_n) + j];
                        } else {
                            u = v2ce_oracle_philox_uniform(seed, (uint32_t)px, (uint32_t)j,
                                                           (uint32_t)(p * 9 + c),
                                                           (uint32_t)(frame_base + b));
                        }
                        float t;
                        if (strategy == V2CE_ORACLE_STRATEGY_RANDOM) {
                            t = u;                                          /* LDATI.py:173-174 */
                        } else if (k == 0.0f) {
                            t = (u / FPS) / 9.0f;                           /* LDATI.py:196 */
                        } else {
                            const float s = bb * bb + (2.0f * k) * u;       /* LDATI.py:195 */
                            t = (-bb + sqrtf(s)) / k;
                        }
                        t = t + offt[c];                                    /* LDATI.py:210 */
                        t = t * E6;                                         /* LDATI.py:211 */
                        if (m >= seg_n) goto overflow;
                        ets[m] = (int64_t)t;                                /* LDATI.py:212 */
                        ex[m] = (int16_t)(px % W);
                        ey[m] = (int16_t)(px / W);
                        ep[m] = pol;
                        ++m;
                    }
                }
            }
            if (m != seg_n) goto overflow;
            for (int64_t i = 0; i < seg_n; ++i) idx[i] = i;
            merge_sort_idx(ets, idx, tmp, seg_n); /* argsort, stable variant -- LDATI.py:297 */
            for (int64_t i = 0; i < seg_n; ++i) {
                ts_out[seg_lo + i] = ets[idx[i]];
                x_out[seg_lo + i] = ex[idx[i]];
                y_out[seg_lo + i] = ey[idx[i]];
                p_out[seg_lo + i] = ep[idx[i]];
            }
            free(ets); free(ex); free(ey); free(ep); free(idx); free(tmp);
            continue;
        overflow:
            free(ets); free(ex); free(ey); free(ep); free(idx); free(tmp);
            free(nn); free(dd); free(yp);
            return -3; /* seg_offsets inconsistent with the voxels */
        }
    }
    free(nn);
    free(dd);
    free(yp);
    return 0;
}
