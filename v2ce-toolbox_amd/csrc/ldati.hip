// ldati.hip -- LDATI (stage 2) for gfx950: voxel grid -> stably sorted (t, x, y, p) event list.
//
// Replaces /root/reference/scripts/LDATI.py:80-106 (y_relocate), :13-51 (slope), :126-214
// (sample_voxel_statistical) and :217-310 (pick_elements / pick_and_sort).
//
// Design (DESIGN.md "Stage 2"): nothing of size O(voxels x max_n) or O(events) is materialised
// except the final output.  Because every timestamp is a pure function of (voxel column, draw
// index) -- counter-based Philox, or a replayed uniform tensor -- it is RECOMPUTED wherever it is
// needed instead of being stored, sorted and gathered:
//
//   v2ce_ldati_count : thread per (frame, polarity, pixel); 9-step relocation recurrence in
//                      registers; per-(frame,bin) totals by wave reduce + one atomic per block.
//   v2ce_ldati_emit  : one workgroup per (frame, bin) SEGMENT = one stable counting sort with the
//                      whole key histogram in LDS (a bin spans ~1e6/fps/9 = 3704 distinct
//                      microsecond keys at 30 fps).  4 waves = the 4 tie-order categories
//                      [neg single, neg multi, pos single, pos multi]:
//                        A. histogram  cnt[cat][key] += 1          (LDS atomics, order free)
//                        B. exclusive scan in (key-major, category-minor) order = stable ranks
//                        C. replay the same events in pixel order; the rank of an event among the
//                           equal-key events of ITS wave batch comes from a ballot "match-any"
//                           (wave64), the running base from LDS; scatter straight to the final
//                           sorted position.  No global scratch, no sort passes over records.
//
// Arithmetic is bit-exact w.r.t. the CPU reference: every f32/f64 operation is a separate IEEE
// operation (-ffp-contract=off, correctly rounded '/' and sqrt), in the reference's order.
#include "common.h"

#include <cmath>

namespace v2ce {
namespace {

struct LdatiParams {
    const float *vox;
    int B, H, W, HW;
    // scalars of LDATI.py:145-146 cast the way CPU torch casts python scalars
    double fps;        // python number used in the f64 single-event path
    float VS, VS2, INV, FPS;
    float offt[9];     // f32(arange(0,1/fps,1/fps/9)[c]) + f32(t0)
    long long kbase[9];  // key = timestamp - kbase[c], clamped to [0, NK)
    int NK, nbits;
    int strategy;      // V2CE_STRATEGY_*: NONE drops every multi-event voxel (LDATI.py:206-207,241)
    int rng_mode;
    const float *uniforms;
    int replay_max_n;
    unsigned long long seed;
    long long frame_base;
    const long long *seg_offsets;
    const long long *frame_ts_add;
    long long *ts;
    short *x;
    short *y;
    signed char *p;
    // bucketed path (v2): coarse bucket = key >> key_shift; NB buckets per segment
    int key_shift, NB, PB;        // PB = bits of a pixel index
    unsigned *cbcount;            // [B*9][NB] events per bucket
    unsigned *cursor;             // [B*9][NB] append cursors
    unsigned *bofs;               // [B*9][NB] exclusive offsets inside the segment
    unsigned *wgtab;              // [B][wg_per_frame][9*NB] per-workgroup bucket counts -> offsets
    int wg_per_frame, blk_per_pol;
    unsigned *temp;               // [total events] unsorted 32-bit records (fine key | category | pixel)
    int *seg_flag;                // [B*9] 1 = a bucket exceeds the LDS sort capacity -> segment sweep path
};

constexpr int kSortCap = 8192;    // 32-bit records one workgroup sorts in LDS

// ---- Philox4x32-10, counter (pixel, j>>2, p*9+c, frame), key = seed ---------------------------
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned pixel, unsigned j,
                                                unsigned pc, unsigned frame) {
    unsigned c0 = pixel, c1 = j >> 2, c2 = pc, c3 = frame;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned sel = j & 3u;
    const unsigned w = sel == 0 ? c0 : sel == 1 ? c1 : sel == 2 ? c2 : c3;
    return (float)(w >> 8) * (1.0f / 16777216.0f);
}

// ---- relocation recurrence (LDATI.py:94-106) up to bin `last` ----------------------------------
// yv[i] holds voxel bin i of this lane's pixel (i <= last, plus yv[9] when last == 8).
// Returns the counts of bins c-1, c, c+1 and the debt of bin c.
__device__ __forceinline__ void relocate_bins(const float (&yv)[10], int c, int last, int &n_l,
                                              int &n_c, int &n_r, float &debt_c) {
    const float eps = 1e-6f;
    float d = 0.0f;
    n_l = n_c = n_r = 0;
    debt_c = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i <= last) {
            const float r = yv[i] - d;
            const float cc = ceilf(r - eps);
            d = cc - r;
            int ni = (int)cc;
            if (i == 8) ni += (int)(yv[9] - d);   // LDATI.py:106
            if (i == c - 1) n_l = ni;
            if (i == c) { n_c = ni; debt_c = d; }
            if (i == c + 1) n_r = ni;
        }
    }
}

__device__ __forceinline__ void load_bins(const float *plane0, long long HW, int px, bool valid,
                                          int last, float (&yv)[10]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const bool need = (i <= last) || (i == 9 && last == 8);
        yv[i] = (need && valid) ? plane0[(long long)i * HW + px] : 0.0f;
    }
}

// single-event timestamp, all f64 (LDATI.py:156-165)
__device__ __forceinline__ long long single_ts(float debt, double fps, float offt) {
    double t = (double)debt / fps / 9.0;
    t += (double)offt;
    t *= 1e6;
    return (long long)t;
}

// slope parameters of one multi-event voxel, f32 (LDATI.py:188-190 with :25-45 folded in)
__device__ __forceinline__ void slope_params(int n_l, int n_c, int n_r, int c, const LdatiParams &P,
                                             float &k, float &bb) {
    // reflect padding makes the central difference vanish at the first and last bin
    const float sxy = (c == 0 || c == 8) ? 0.0f : ((float)n_r - (float)n_l);
    const float k0 = (3.0f * sxy) / 6.0f;
    k = (k0 / P.VS2) / ((float)n_c + 1e-8f);
    bb = P.INV - (P.VS * k) / 2.0f;
}

// multi-event timestamp, all f32 (LDATI.py:195-196,210-212)
__device__ __forceinline__ long long multi_ts(float k, float bb, float u, float offt,
                                              const LdatiParams &P) {
    float t;
    if (k == 0.0f) {
        t = (u / P.FPS) / 9.0f;
    } else {
        const float s = bb * bb + (2.0f * k) * u;
        t = (-bb + __builtin_sqrtf(s)) / k;
    }
    t = t + offt;
    t = t * 1e6f;
    return (long long)t;
}

__device__ __forceinline__ int key_of(long long T, long long kbase, int NK) {
    long long k = T - kbase;
    k = k < 0 ? 0 : k;
    k = k >= NK ? NK - 1 : k;
    return (int)k;
}

// ballot match-any: lanes with `has` and equal `key` form a peer group
__device__ __forceinline__ unsigned long long match_key(bool has, int key, int nbits) {
    unsigned long long peers = __ballot(has);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (key >> b) & 1;
        const unsigned long long m = __ballot(has && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// ---------------------------------------------------------------------------------------------
// count kernel
// ---------------------------------------------------------------------------------------------
constexpr int kCountPixPerBlock = 4096;   // 16 pixels per thread: 16x fewer (contended) atomics

__global__ __launch_bounds__(256) void ldati_count_kernel(const float *__restrict__ vox, int HW,
                                                          unsigned long long *seg_counts,
                                                          int *max_n, int strategy) {
    // grid: (pixel blocks, 2*B)
    const int bp = blockIdx.y;           // b*2 + p
    const int b = bp >> 1;
    const float *plane0 = vox + (long long)bp * 10 * HW;
    int cnt[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cnt[i] = 0;
    int mx = 0;
    for (int it = 0; it < kCountPixPerBlock / 256; ++it) {
        const int px = blockIdx.x * kCountPixPerBlock + it * 256 + threadIdx.x;
        if (px >= HW) break;
        float yv[10];
        load_bins(plane0, HW, px, true, 8, yv);
        const float eps = 1e-6f;
        float d = 0.0f;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float r = yv[i] - d;
            const float cc = ceilf(r - eps);
            d = cc - r;
            int ni = (int)cc;
            if (i == 8) ni += (int)(yv[9] - d);
            cnt[i] += (strategy == V2CE_STRATEGY_NONE) ? (ni == 1) : (ni > 0 ? ni : 0);
            mx = ni > mx ? ni : mx;
        }
    }
    __shared__ int red[4][10];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int v = cnt[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wid][i] = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(mx, o);
        mx = t > mx ? t : mx;
    }
    if (lane == 0) red[wid][9] = mx;
    __syncthreads();
    if (threadIdx.x < 9) {
        const int i = threadIdx.x;
        const long long s = (long long)red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (s) atomicAdd(&seg_counts[(long long)b * 9 + i], (unsigned long long)s);
    } else if (threadIdx.x == 9) {
        int m = red[0][9];
        m = red[1][9] > m ? red[1][9] : m;
        m = red[2][9] > m ? red[2][9] : m;
        m = red[3][9] > m ? red[3][9] : m;
        if (m > 0) atomicMax(max_n, m);
    }
}

// exclusive scan of B*9 counts, single block (B*9 is small)
__global__ __launch_bounds__(256) void ldati_scan_kernel(const long long *counts, int n,
                                                         long long *offsets) {
    __shared__ long long part[256];
    const int t = threadIdx.x;
    const int per = (n + 255) / 256;
    const int lo = t * per, hi = (lo + per < n) ? lo + per : n;
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int i = 0; i < 256; ++i) {
            const long long v = part[i];
            part[i] = run;
            run += v;
        }
        offsets[n] = run;
    }
    __syncthreads();
    long long run = part[t];
    for (int i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
}

// ---------------------------------------------------------------------------------------------
// emit kernel: one workgroup (4 waves) per (frame, bin) segment
// ---------------------------------------------------------------------------------------------
template <bool RANK>
__device__ __forceinline__ void handle_event(bool has, long long T, int px, int c, int cat,
                                             signed char pol, int lane, long long seg_lo,
                                             long long ts_add, unsigned *cnt,
                                             const LdatiParams &P) {
    const int key = key_of(T, P.kbase[c], P.NK);
    unsigned *slot = cnt + cat * P.NK + key;
    if (!RANK) {
        if (has) atomicAdd(slot, 1u);
        return;
    }
    const unsigned long long peers = match_key(has, key, P.nbits);
    if (has) {
        const unsigned long long lt = (1ull << lane) - 1ull;
        const unsigned rank = (unsigned)__popcll(peers & lt);
        const unsigned npeer = (unsigned)__popcll(peers);
        const bool leader = lane == 63 - __clzll((long long)peers);
        volatile unsigned *vs = slot;
        const unsigned base = *vs;                 // every peer reads the running base ...
        __builtin_amdgcn_wave_barrier();
        if (leader) *vs = base + npeer;            // ... then the highest peer advances it
        const long long pos = seg_lo + (long long)base + rank;
        const int yy = px / P.W;
        P.ts[pos] = T + ts_add;
        P.x[pos] = (short)(px - yy * P.W);
        P.y[pos] = (short)yy;
        P.p[pos] = pol;
    }
    __builtin_amdgcn_wave_barrier();
}

template <bool RANK>
__device__ __forceinline__ void sweep_singles(int b, int c, int pidx, int cat, signed char pol,
                                              int lane, long long seg_lo, long long ts_add,
                                              unsigned *cnt, const LdatiParams &P) {
    const int last = c + 1 < 8 ? c + 1 : 8;
    const float *plane0 = P.vox + (long long)(b * 2 + pidx) * 10 * P.HW;
    float cur[10], nxt[10];
    load_bins(plane0, P.HW, lane, lane < P.HW, c, cur);   // singles only need bins 0..c
    for (int base = 0; base < P.HW; base += 64) {
        const int px = base + lane;
        const bool valid = px < P.HW;
        const int pxn = px + 64;
        load_bins(plane0, P.HW, pxn, pxn < P.HW, c, nxt);  // prefetch the next 64 pixels
        int n_l, n_c, n_r;
        float debt;
        relocate_bins(cur, c, c, n_l, n_c, n_r, debt);
        const bool has = valid && n_c == 1;
        const long long T = single_ts(debt, P.fps, P.offt[c]);
        handle_event<RANK>(has, T, px, c, cat, pol, lane, seg_lo, ts_add, cnt, P);
#pragma unroll
        for (int i = 0; i < 10; ++i) cur[i] = nxt[i];
    }
    (void)last;
}

template <bool RANK>
__device__ __forceinline__ void sweep_multis(int b, int c, int pidx, int cat, signed char pol,
                                             int lane, long long seg_lo, long long ts_add,
                                             unsigned *cnt, int *s_start, float *s_k, float *s_bb,
                                             const LdatiParams &P) {
    const int last = c + 1 < 8 ? c + 1 : 8;
    const float *plane0 = P.vox + (long long)(b * 2 + pidx) * 10 * P.HW;
    const unsigned pc = (unsigned)(pidx * 9 + c);
    const unsigned frame = (unsigned)(P.frame_base + b);
    float cur[10], nxt[10];
    load_bins(plane0, P.HW, lane, lane < P.HW, last, cur);
    for (int base = 0; base < P.HW; base += 64) {
        const int px = base + lane;
        const bool valid = px < P.HW;
        const int pxn = px + 64;
        load_bins(plane0, P.HW, pxn, pxn < P.HW, last, nxt);
        int n_l, n_c, n_r;
        float debt;
        relocate_bins(cur, c, last, n_l, n_c, n_r, debt);
        const int m = (valid && n_c >= 2) ? n_c : 0;
        float k, bb;
        slope_params(n_l, n_c, n_r, c, P, k, bb);
        // exclusive wave scan of m -> first event index of each pixel inside this batch
        int incl = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int E = __shfl(incl, 63);
        if (E > 0) {
            s_start[lane] = incl - m;
            s_k[lane] = k;
            s_bb[lane] = bb;
            __builtin_amdgcn_wave_barrier();
            for (int e0 = 0; e0 < E; e0 += 64) {
                const int e = e0 + lane;
                const bool act = e < E;
                // last lane l with s_start[l] <= e  (pixels with m == 0 share the next start)
                int lo = 0, hi = 63;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int mid = (lo + hi + 1) >> 1;
                    const bool le = s_start[mid] <= e;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid - 1;
                }
                const int src = act ? lo : 0;
                const int j = e - s_start[src];
                const int spx = base + src;
                float u = 0.0f;
                if (act) {
                    if (P.rng_mode == V2CE_RNG_REPLAY) {
                        if (j < P.replay_max_n)
                            u = P.uniforms[(((long long)(b * 2 + pidx) * 9 + c) * P.HW + spx) *
                                               P.replay_max_n + j];
                    } else {
                        u = philox_uniform(P.seed, (unsigned)spx, (unsigned)j, pc, frame);
                    }
                }
                const long long T = multi_ts(s_k[src], s_bb[src], u, P.offt[c], P);
                handle_event<RANK>(act, T, spx, c, cat, pol, lane, seg_lo, ts_add, cnt, P);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) cur[i] = nxt[i];
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char ldati_smem[];

__global__ __launch_bounds__(256) void ldati_emit_kernel(LdatiParams P) {
    const int seg = blockIdx.x;          // b*9 + c
    const int b = seg / 9, c = seg - b * 9;
    const long long seg_lo = P.seg_offsets[seg];
    const long long seg_n = P.seg_offsets[seg + 1] - seg_lo;
    if (seg_n <= 0) return;              // uniform per workgroup
    if (P.seg_flag && !P.seg_flag[seg]) return;   // bucketed path handled this segment
    const long long ts_add = P.frame_ts_add ? P.frame_ts_add[b] : 0;

    unsigned *cnt = reinterpret_cast<unsigned *>(ldati_smem);            // [4][NK]
    unsigned *part = cnt + 4 * P.NK;                                       // [256]
    int *s_start = reinterpret_cast<int *>(part + 256);                    // [2][64]
    float *s_k = reinterpret_cast<float *>(s_start + 128);                 // [2][64]
    float *s_bb = s_k + 128;                                               // [2][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int cat = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0 neg-single 1 neg-multi 2 pos-single 3 pos-multi
    const int pidx = cat < 2 ? 1 : 0;    // negative events live in P index 1 (LDATI.py:289)
    const signed char pol = cat < 2 ? 0 : 1;
    const bool multi = cat & 1;
    const bool skip = multi && P.strategy == V2CE_STRATEGY_NONE;   // wave-uniform
    int *my_start = s_start + (cat >> 1) * 64;
    float *my_k = s_k + (cat >> 1) * 64, *my_bb = s_bb + (cat >> 1) * 64;

    for (int i = tid; i < 4 * P.NK; i += 256) cnt[i] = 0;
    __syncthreads();

    // A. histogram
    if (skip) {
    } else if (multi)
        sweep_multis<false>(b, c, pidx, cat, pol, lane, seg_lo, ts_add, cnt, my_start, my_k, my_bb, P);
    else
        sweep_singles<false>(b, c, pidx, cat, pol, lane, seg_lo, ts_add, cnt, P);
    __syncthreads();

    // B. exclusive scan in (key-major, category-minor) order
    {
        const int kpt = (P.NK + 255) / 256;
        const int klo = tid * kpt, khi = (klo + kpt < P.NK) ? klo + kpt : P.NK;
        unsigned s = 0;
        for (int k = klo; k < khi; ++k)
            s += cnt[k] + cnt[P.NK + k] + cnt[2 * P.NK + k] + cnt[3 * P.NK + k];
        part[tid] = s;
        __syncthreads();
        // Hillis-Steele inclusive scan over 256 partials
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned v = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        unsigned run = part[tid] - s;
        for (int k = klo; k < khi; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned v = cnt[q * P.NK + k];
                cnt[q * P.NK + k] = run;
                run += v;
            }
        }
    }
    __syncthreads();

    // C. rank + scatter
    if (skip) {
    } else if (multi)
        sweep_multis<true>(b, c, pidx, cat, pol, lane, seg_lo, ts_add, cnt, my_start, my_k, my_bb, P);
    else
        sweep_singles<true>(b, c, pidx, cat, pol, lane, seg_lo, ts_add, cnt, P);
}

// ---------------------------------------------------------------------------------------------
// bucketed path (v2): pixel-parallel recompute -> coarse buckets -> LDS sort -> coalesced output
//
//   pass<false> : thread per (frame, polarity, pixel); ONE relocation recurrence over the 9 bins;
//                 every event's timestamp; histogram of coarse buckets (key >> key_shift) per segment
//   scan        : exclusive offsets of the buckets inside each segment; segments whose largest
//                 bucket exceeds kSortCap are flagged for the segment-sweep kernel above
//   pass<true>  : same recompute; append a 32-bit record (fine key | category | pixel) to its bucket
//   sort        : one workgroup per bucket: bitonic sort of the records in LDS, then the final SoA
//                 events are written as coalesced runs.  The record order (key, category, pixel) IS the
//                 reference's stable order: events that tie on all three are identical records.
// ---------------------------------------------------------------------------------------------
constexpr int kPixPerWg = 1024;   // pixels of one polarity plane per workgroup (4 per thread)

// One event of (frame b, bin c): count it in / append it through the workgroup's LDS table.
template <bool APPEND>
__device__ __forceinline__ void bucket_event(const LdatiParams &P, unsigned *lds, int c,
                                             long long seg_lo, long long T, unsigned cat,
                                             unsigned px) {
    const int key = key_of(T, P.kbase[c], P.NK);
    unsigned *slot = lds + c * P.NB + (key >> P.key_shift);
    if (!APPEND) {
        atomicAdd(slot, 1u);                               // LDS atomic, no return
    } else {
        const unsigned pos = atomicAdd(slot, 1u);          // LDS cursor (bucket base + rank)
        const unsigned fine = (unsigned)key & ((1u << P.key_shift) - 1u);
        P.temp[seg_lo + pos] = (fine << (2 + P.PB)) | (cat << P.PB) | px;
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char bucket_smem[];

// Pixel-parallel pass.  APPEND = false: per-workgroup histogram of coarse buckets (9 bins x NB) in
// LDS, stored to wgtab.  APPEND = true: the LDS table is preloaded with this workgroup's exclusive
// offsets (bucket offset inside the segment + events of earlier workgroups), and every event takes
// its slot with one LDS atomic: no global atomics anywhere.
template <bool APPEND>
__global__ __launch_bounds__(256) void ldati_bucket_pass_kernel(LdatiParams P) {
    unsigned *lds = reinterpret_cast<unsigned *>(bucket_smem);           // [9][NB]
    const int bp = blockIdx.y, b = bp >> 1, pidx = bp & 1;
    const int wg = pidx * P.blk_per_pol + blockIdx.x;
    const int ntab = 9 * P.NB;
    unsigned *tab = P.wgtab + ((long long)b * P.wg_per_frame + wg) * ntab;
    for (int i = threadIdx.x; i < ntab; i += 256)
        lds[i] = APPEND ? tab[i] + P.bofs[(long long)b * ntab + i] : 0u;
    __syncthreads();
    const float *plane0 = P.vox + (long long)bp * 10 * P.HW;
    const unsigned frame = (unsigned)(P.frame_base + b);
    const unsigned cat_single = pidx ? 0u : 2u;    // negative events live in P index 1
    for (int it = 0; it < kPixPerWg / 256; ++it) {
        const int px = blockIdx.x * kPixPerWg + it * 256 + threadIdx.x;
        if (px >= P.HW) break;
        float yv[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) yv[i] = plane0[(long long)i * P.HW + px];
        int n[9];
        float dbt[9];
        {
            const float eps = 1e-6f;
            float d = 0.0f;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const float r = yv[i] - d;
                const float cc = ceilf(r - eps);
                d = cc - r;
                int ni = (int)cc;
                if (i == 8) ni += (int)(yv[9] - d);
                n[i] = ni;
                dbt[i] = d;
            }
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int nc = n[c];
            if (nc < 1) continue;
            const long long seg_lo = APPEND ? P.seg_offsets[b * 9 + c] : 0;
            if (nc == 1) {
                bucket_event<APPEND>(P, lds, c, seg_lo, single_ts(dbt[c], P.fps, P.offt[c]), cat_single,
                                     (unsigned)px);
            } else if (P.strategy != V2CE_STRATEGY_NONE) {
                float k, bb;
                slope_params(c > 0 ? n[c - 1] : 0, nc, c < 8 ? n[c + 1] : 0, c, P, k, bb);
                const long long ubase = (((long long)bp * 9 + c) * P.HW + px) * P.replay_max_n;
                for (int j = 0; j < nc; ++j) {
                    float u = 0.0f;
                    if (P.rng_mode == V2CE_RNG_REPLAY) {
                        if (j < P.replay_max_n) u = P.uniforms[ubase + j];
                    } else {
                        u = philox_uniform(P.seed, (unsigned)px, (unsigned)j, (unsigned)(pidx * 9 + c), frame);
                    }
                    bucket_event<APPEND>(P, lds, c, seg_lo, multi_ts(k, bb, u, P.offt[c], P),
                                         cat_single + 1u, (unsigned)px);
                }
            }
        }
    }
    if (!APPEND) {
        __syncthreads();
        for (int i = threadIdx.x; i < ntab; i += 256) tab[i] = lds[i];
    }
}

// column scan over the workgroups of a frame: wgtab[b][w][i] -> exclusive prefix over w;
// cbcount[b][i] = total
__global__ __launch_bounds__(256) void ldati_wgtab_scan_kernel(LdatiParams P) {
    const int ntab = 9 * P.NB;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= ntab) return;
    unsigned *col = P.wgtab + (long long)b * P.wg_per_frame * ntab + i;
    unsigned run = 0;
    for (int w = 0; w < P.wg_per_frame; ++w) {
        const unsigned v = col[(long long)w * ntab];
        col[(long long)w * ntab] = run;
        run += v;
    }
    P.cbcount[(long long)b * ntab + i] = run;
}

__global__ __launch_bounds__(256) void ldati_bucket_scan_kernel(LdatiParams P) {
    __shared__ unsigned part[256];
    __shared__ unsigned big;
    const int seg = blockIdx.x, t = threadIdx.x;
    const unsigned *cnt = P.cbcount + (long long)seg * P.NB;
    unsigned *ofs = P.bofs + (long long)seg * P.NB;
    const int per = (P.NB + 255) / 256;
    const int lo = t * per, hi = (lo + per < P.NB) ? lo + per : P.NB;
    if (t == 0) big = 0;
    __syncthreads();
    unsigned s = 0, mx = 0;
    for (int i = lo; i < hi; ++i) {
        const unsigned v = cnt[i];
        s += v;
        mx = v > mx ? v : mx;
    }
    part[t] = s;
    if (mx > (unsigned)kSortCap) atomicOr(&big, 1u);
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned v = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - s;
    for (int i = lo; i < hi; ++i) {
        ofs[i] = run;
        run += cnt[i];
    }
    if (t == 0) P.seg_flag[seg] = (int)big;
}

__global__ __launch_bounds__(256) void ldati_bucket_sort_kernel(LdatiParams P) {
    __shared__ unsigned keys[kSortCap];
    const unsigned bucket = blockIdx.x;
    const int seg = bucket / P.NB, cb = bucket - seg * P.NB;
    const unsigned n = P.cbcount[bucket];
    if (n == 0 || P.seg_flag[seg]) return;          // uniform per workgroup
    const long long base = P.seg_offsets[seg] + P.bofs[bucket];
    unsigned np = 2;
    while (np < n) np <<= 1;
    const int tid = threadIdx.x;
    for (unsigned i = tid; i < np; i += 256) keys[i] = i < n ? P.temp[base + i] : 0xFFFFFFFFu;
    __syncthreads();
    for (unsigned k = 2; k <= np; k <<= 1) {
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t = tid; t < (np >> 1); t += 256) {
                const unsigned i = 2 * t - (t & (j - 1));
                const unsigned l = i + j;
                const unsigned a = keys[i], bq = keys[l];
                const bool up = (i & k) == 0;
                if ((a > bq) == up) { keys[i] = bq; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    const int b = seg / 9, c = seg - b * 9;
    const long long tbase = P.kbase[c] + ((long long)cb << P.key_shift) +
                            (P.frame_ts_add ? P.frame_ts_add[b] : 0);
    const unsigned pmask = (1u << P.PB) - 1u;
    for (unsigned i = tid; i < n; i += 256) {
        const unsigned r = keys[i];
        const unsigned px = r & pmask, cat = (r >> P.PB) & 3u, fine = r >> (P.PB + 2);
        const unsigned yy = px / (unsigned)P.W;
        P.ts[base + i] = tbase + fine;
        P.x[base + i] = (short)(px - yy * P.W);
        P.y[base + i] = (short)yy;
        P.p[base + i] = (signed char)(cat >> 1);
    }
}

// ---------------------------------------------------------------------------------------------
// pack kernel: SoA -> 13-byte records, staged through LDS so the global stores are whole dwords
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void events_pack_kernel(const long long *__restrict__ ts,
                                                          const short *__restrict__ x,
                                                          const short *__restrict__ y,
                                                          const signed char *__restrict__ p,
                                                          long long n, unsigned char *packed) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[256 * 13 + 12];
    const long long first = (long long)blockIdx.x * 256;
    const long long i = first + threadIdx.x;
    if (i < n) {
        const long long t = ts[i];
        const unsigned short xx = (unsigned short)x[i], yy = (unsigned short)y[i];
        unsigned char *d = stage + threadIdx.x * 13;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (unsigned char)((unsigned long long)t >> (8 * k));
        d[8] = (unsigned char)xx; d[9] = (unsigned char)(xx >> 8);
        d[10] = (unsigned char)yy; d[11] = (unsigned char)(yy >> 8);
        d[12] = (unsigned char)p[i];
    }
    __syncthreads();
    const long long remain = n - first;
    const int nev = remain < 256 ? (int)remain : 256;
    const int nbytes = nev * 13;
    unsigned char *dst = packed + first * 13;   // first*13 is a multiple of 4 (256*13 = 3328)
    const int nd = nbytes >> 2;
    for (int w = threadIdx.x; w < nd; w += 256)
        reinterpret_cast<unsigned *>(dst)[w] = reinterpret_cast<const unsigned *>(stage)[w];
    for (int r = (nd << 2) + threadIdx.x; r < nbytes; r += 256) dst[r] = stage[r];
}

// host-side scalars, computed exactly like CPU torch does
struct HostScalars {
    float VS, VS2, INV, FPS;
    float offt[9];
    long long kbase[9];
    int NK, nbits;
    size_t lds_bytes;
    bool ok;
};

HostScalars host_scalars(double fps, double t0) {
    HostScalars h{};
    const double vs = 1.0 / fps / 9.0;
    h.VS = (float)vs;
    h.VS2 = (float)(vs * vs);
    h.INV = (float)(1.0 / vs);
    h.FPS = (float)fps;
    for (int c = 0; c < 9; ++c) h.offt[c] = (float)(0.0 + (double)c * vs) + (float)t0;
    // f32 resolution of (t + offt)*1e6 near the last bin decides how far a multi-event timestamp
    // can round outside [offt, offt + vs]; size the slack from it.
    const double top = (double)fabsf(h.offt[8]) + vs;
    const double ulp_us = top * 1.1920929e-7 * 1e6;      // one f32 ulp of the largest time, in us
    const long long slack = 16 + (long long)(8.0 * ulp_us);
    const long long span = (long long)(vs * 1e6) + 2;
    const long long nk = span + 2 * slack;
    for (int c = 0; c < 9; ++c) h.kbase[c] = (long long)((double)h.offt[c] * 1e6) - slack;
    h.ok = nk > 0 && nk <= 9600;    // 4*NK*4 B must fit 160 KiB of LDS with the scratch beside it
    h.NK = (int)nk;
    int nb = 0;
    while ((1ll << nb) < nk) ++nb;
    h.nbits = nb;
    h.lds_bytes = (size_t)(4 * h.NK + 256) * 4 + 3 * 128 * 4;
    return h;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

extern "C" int v2ce_ldati_count(const float *vox, int B, int H, int W, int strategy,
                                int64_t *seg_counts, int32_t *max_n, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(vox && seg_counts && max_n, V2CE_ERR_BAD_ARG, "v2ce_ldati_count: null pointer");
    V2CE_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_count: bad shape B=%d H=%d W=%d", B, H, W);
    V2CE_REQUIRE(W <= 32767 && H <= 32767, V2CE_ERR_UNSUPPORTED,
                 "v2ce_ldati_count: x/y are int16 (LDATI.py:230-231)");
    V2CE_REQUIRE(2 * B <= 65535, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_count: B too large for one launch");
    V2CE_REQUIRE(strategy == V2CE_STRATEGY_SLOPE || strategy == V2CE_STRATEGY_NONE, V2CE_ERR_UNSUPPORTED,
                 "v2ce_ldati_count: strategy %d not implemented", strategy);
    hipStream_t s = as_stream(stream);
    V2CE_HIP_CHECK(hipMemsetAsync(seg_counts, 0, sizeof(int64_t) * 9 * (size_t)B, s));
    V2CE_HIP_CHECK(hipMemsetAsync(max_n, 0, sizeof(int32_t), s));
    const int HW = H * W;
    dim3 grid((HW + kCountPixPerBlock - 1) / kCountPixPerBlock, 2 * B);
    hipLaunchKernelGGL(ldati_count_kernel, grid, dim3(256), 0, s, vox, HW,
                       reinterpret_cast<unsigned long long *>(seg_counts), max_n, strategy);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" int v2ce_ldati_scan(const int64_t *seg_counts, int B, int64_t *seg_offsets,
                               v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(seg_counts && seg_offsets && B > 0, V2CE_ERR_BAD_ARG, "v2ce_ldati_scan: bad argument");
    hipLaunchKernelGGL(ldati_scan_kernel, dim3(1), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const long long *>(seg_counts), B * 9,
                       reinterpret_cast<long long *>(seg_offsets));
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" size_t v2ce_ldati_lds_bytes(double fps, double t0) {
    if (!(fps > 0)) return 0;
    const HostScalars h = host_scalars(fps, t0);
    return h.ok ? h.lds_bytes : 0;
}

namespace {
struct BucketPlan { int shift, NB, PB, blk_per_pol, wg_per_frame; size_t counters, tab, lds, bytes; };

// coarse-bucket width: average bucket of the LARGEST segment ~ kSortCap/4 records, but never so
// narrow that a workgroup's [9][NB] table exceeds its LDS budget
BucketPlan bucket_plan(const HostScalars &h, int B, int H, int W, int64_t total_events,
                       int64_t max_segment_events) {
    BucketPlan bp{};
    int shift = 0;
    while (shift < 13 && (double)max_segment_events * (double)(2 << shift) / (double)h.NK <= kSortCap / 4) ++shift;
    while (shift < 13 && (size_t)9 * ((h.NK + (1 << shift) - 1) >> shift) * 4 > 96 * 1024) ++shift;
    int pb = 1;
    while ((1ll << pb) < (long long)H * W) ++pb;
    if (shift + 2 + pb > 32) shift = -1;              // the record must fit 32 bits
    bp.shift = shift;
    bp.PB = pb;
    bp.NB = shift >= 0 ? (h.NK + (1 << shift) - 1) >> shift : 0;
    bp.blk_per_pol = (H * W + kPixPerWg - 1) / kPixPerWg;
    bp.wg_per_frame = 2 * bp.blk_per_pol;
    bp.counters = (size_t)B * 9 * (size_t)bp.NB;
    bp.tab = bp.counters * (size_t)bp.wg_per_frame;
    bp.lds = (size_t)9 * bp.NB * 4;
    bp.bytes = (2 * bp.counters + bp.tab + (size_t)B * 9 + (size_t)(total_events > 0 ? total_events : 0)) * 4;
    return bp;
}
}  // namespace

extern "C" size_t v2ce_ldati_workspace_bytes(int B, int H, int W, double fps, double t0,
                                             int64_t total_events, int64_t max_segment_events) {
    if (!(fps > 0) || B <= 0 || H <= 0 || W <= 0) return 0;
    const HostScalars h = host_scalars(fps, t0);
    if (!h.ok) return 0;
    const BucketPlan bp = bucket_plan(h, B, H, W, total_events, max_segment_events);
    return bp.shift >= 0 ? bp.bytes : 0;
}

extern "C" int v2ce_ldati_emit(const float *vox, int B, int H, int W, double fps, double t0,
                               int strategy, int rng_mode, const float *uniforms, int replay_max_n,
                               uint64_t seed, int64_t frame_base, const int64_t *seg_offsets,
                               const int64_t *frame_ts_add, int64_t *ts, int16_t *x, int16_t *y,
                               int8_t *p, int64_t total_events, int64_t max_segment_events,
                               void *workspace, size_t workspace_bytes, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(vox && seg_offsets, V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: null pointer");
    V2CE_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_emit: bad shape");
    V2CE_REQUIRE(W <= 32767 && H <= 32767, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_emit: x/y are int16");
    V2CE_REQUIRE(fps > 0, V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: fps must be positive");
    V2CE_REQUIRE(rng_mode == V2CE_RNG_REPLAY || rng_mode == V2CE_RNG_PHILOX, V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_emit: bad rng_mode %d", rng_mode);
    V2CE_REQUIRE(strategy == V2CE_STRATEGY_SLOPE || strategy == V2CE_STRATEGY_NONE, V2CE_ERR_UNSUPPORTED,
                 "v2ce_ldati_emit: strategy %d not implemented", strategy);
    V2CE_REQUIRE(rng_mode != V2CE_RNG_REPLAY || replay_max_n == 0 || uniforms != nullptr,
                 V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: REPLAY mode needs the uniform tensor");
    const HostScalars h = host_scalars(fps, t0);
    V2CE_REQUIRE(h.ok, V2CE_ERR_UNSUPPORTED,
                 "v2ce_ldati_emit: fps=%g t0=%g needs %d keys per bin; the LDS histogram holds 9600",
                 fps, t0, h.NK);
    LdatiParams P{};
    P.vox = vox; P.B = B; P.H = H; P.W = W; P.HW = H * W;
    P.fps = fps; P.VS = h.VS; P.VS2 = h.VS2; P.INV = h.INV; P.FPS = h.FPS;
    for (int c = 0; c < 9; ++c) { P.offt[c] = h.offt[c]; P.kbase[c] = h.kbase[c]; }
    P.NK = h.NK; P.nbits = h.nbits;
    P.strategy = strategy;
    P.rng_mode = rng_mode; P.uniforms = uniforms; P.replay_max_n = replay_max_n;
    P.seed = seed; P.frame_base = frame_base;
    P.seg_offsets = reinterpret_cast<const long long *>(seg_offsets);
    P.frame_ts_add = reinterpret_cast<const long long *>(frame_ts_add);
    P.ts = reinterpret_cast<long long *>(ts); P.x = x; P.y = y;
    P.p = reinterpret_cast<signed char *>(p);
    hipStream_t st = as_stream(stream);
    V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ldati_emit_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)h.lds_bytes));
    if (workspace != nullptr) {
        // ---- bucketed path; segments with an oversized bucket fall through to the sweep kernel
        V2CE_REQUIRE(total_events >= 0 && max_segment_events >= 0 && total_events < (1ll << 32),
                     V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: bad event counts");
        const BucketPlan bp = bucket_plan(h, B, H, W, total_events, max_segment_events);
        V2CE_REQUIRE(bp.shift >= 0, V2CE_ERR_UNSUPPORTED,
                     "v2ce_ldati_emit: %d x %d pixels do not fit the 32-bit bucket record; pass workspace = NULL", H, W);
        V2CE_REQUIRE(workspace_bytes >= bp.bytes, V2CE_ERR_WORKSPACE,
                     "v2ce_ldati_emit: workspace %zu < %zu", workspace_bytes, bp.bytes);
        V2CE_REQUIRE(bp.counters < (1ull << 31) && 2 * B <= 65535, V2CE_ERR_UNSUPPORTED,
                     "v2ce_ldati_emit: too many buckets for one launch");
        unsigned *w = static_cast<unsigned *>(workspace);
        P.key_shift = bp.shift; P.NB = bp.NB; P.PB = bp.PB;
        P.blk_per_pol = bp.blk_per_pol; P.wg_per_frame = bp.wg_per_frame;
        P.cbcount = w;
        P.bofs = w + bp.counters;
        P.wgtab = w + 2 * bp.counters;
        P.seg_flag = reinterpret_cast<int *>(w + 2 * bp.counters + bp.tab);
        P.temp = w + 2 * bp.counters + bp.tab + (size_t)B * 9;
        P.cursor = nullptr;
        V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ldati_bucket_pass_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp.lds));
        V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ldati_bucket_pass_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp.lds));
        dim3 grid(bp.blk_per_pol, 2 * B);
        hipLaunchKernelGGL(ldati_bucket_pass_kernel<false>, grid, dim3(256), bp.lds, st, P);
        hipLaunchKernelGGL(ldati_wgtab_scan_kernel, dim3((9 * bp.NB + 255) / 256, B), dim3(256), 0, st, P);
        hipLaunchKernelGGL(ldati_bucket_scan_kernel, dim3(B * 9), dim3(256), 0, st, P);
        hipLaunchKernelGGL(ldati_bucket_pass_kernel<true>, grid, dim3(256), bp.lds, st, P);
        hipLaunchKernelGGL(ldati_bucket_sort_kernel, dim3((unsigned)bp.counters), dim3(256), 0, st, P);
    }
    // segment-sweep kernel: every segment when there is no workspace, else only the flagged ones
    hipLaunchKernelGGL(ldati_emit_kernel, dim3(B * 9), dim3(256), h.lds_bytes, st, P);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" int v2ce_events_pack(const int64_t *ts, const int16_t *x, const int16_t *y,
                                const int8_t *p, int64_t n, uint8_t *packed, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(n >= 0, V2CE_ERR_BAD_ARG, "v2ce_events_pack: negative n");
    if (n == 0) return V2CE_OK;
    V2CE_REQUIRE(ts && x && y && p && packed, V2CE_ERR_BAD_ARG, "v2ce_events_pack: null pointer");
    V2CE_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 3) == 0, V2CE_ERR_BAD_ARG,
                 "v2ce_events_pack: packed must be 4-byte aligned");
    const long long blocks = (n + 255) / 256;
    V2CE_REQUIRE(blocks < (1ll << 31), V2CE_ERR_UNSUPPORTED, "v2ce_events_pack: too many events");
    hipLaunchKernelGGL(events_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const long long *>(ts), x, y,
                       reinterpret_cast<const signed char *>(p), (long long)n, packed);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
