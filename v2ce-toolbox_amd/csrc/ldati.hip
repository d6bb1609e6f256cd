// ldati.hip -- LDATI (stage 2) for gfx950: voxel grid -> stably sorted (t, x, y, p) event list.
//
// Replaces /root/reference/scripts/LDATI.py:80-106 (y_relocate), :13-51 (slope), :126-214
// (sample_voxel_statistical) and :217-310 (pick_elements / pick_and_sort).
//
// Design.  Every timestamp is a pure function of (voxel column, draw index) --
// counter-based Philox, or a replayed uniform tensor.  The reference's output order inside a
// (frame, bin) SEGMENT is the stable sort by timestamp of [neg singles, neg multis, pos singles,
// pos multis], each listed pixel-row-major; i.e. the order of the key (timestamp, category, pixel).
// Two-level counting sort with the pixel order kept BY CONSTRUCTION, so no pass ever sorts on the
// pixel bits:
//
//   count      : workgroup per (frame, polarity, 2048-pixel TILE): relocation recurrence in
//                registers -> events per (tile, bin); one tiny scan kernel -> per-tile record
//                offsets, segment offsets, statistics the host needs to allocate.
//   tile pass  : workgroup per tile, bin by bin.  Timestamps are computed ONCE: singles and
//                4-draw multi-event UNITS are compacted so that every lane of the f64 / Philox +
//                sqrt / divide code does useful work; the records land in LDS in pixel order.  A
//                stable counting sort on the COARSE key (timestamp >> shift; per-wave histograms +
//                ballot match-any ranks -- deterministic, no atomics decide an order) groups them
//                by bucket, and the tile's 4-byte records leave as ONE contiguous, coalesced run
//                per bin.
//   bucket scan: per segment, bucket totals over the tiles -> output offsets.
//   bucket sort: workgroup per (segment, bucket): gathers the tiles' runs in tile order (= pixel
//                order), one stable counting sort on (fine key, category) in LDS, and writes the
//                final 13-byte packed records (or SoA) as full coalesced lines.
//   sweep      : the v1 kernel (one workgroup per segment, whole key histogram in LDS) remains the
//                fallback for segments with a bucket beyond the LDS capacity (degenerate ties) and
//                for workspace == NULL; both paths are bit-identical.
//
// Arithmetic is bit-exact w.r.t. the CPU reference: every f32/f64 operation is a separate IEEE
// operation (-ffp-contract=off, correctly rounded '/' and sqrt), in the reference's order.
#include "common.h"

#include <atomic>
#include <mutex>
#include <type_traits>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cmath>
#include <cstdlib>

namespace v2ce {
namespace {

constexpr int kTilePix = 2048;        // pixels of one polarity plane per tile
constexpr int kLocalBits = 11;        // log2(kTilePix)
constexpr int kCountThreads = 512;
constexpr int kMaxTiles = 512;        // tiles per frame (both polarities) the bucket sort indexes
constexpr int kMaxNB = 512;           // coarse buckets per segment
constexpr int kMaxShift = 8;          // log2 of the widest coarse bucket
constexpr int kMaxSpanKeys = 128;     // timestamps a sort group spans at most (its histogram has 4x as many bins)
constexpr int kSmallGroupSpanKeys = 256;   // ... in the small-group regime (make_plan; measured 128 / 256 / 512: e2e sort 94 / 89 / 116 us)
constexpr int kCapTile = 15360;       // events of one (tile, bin) the tile pass can hold in LDS
#ifndef V2CE_SPARSE_CAP               // (diagnostic builds: tools/sparse_cap_ab.sh)
#define V2CE_SPARSE_CAP 8192
#endif
#ifndef V2CE_SPARSE_WAVES
#define V2CE_SPARSE_WAVES 1
#endif
constexpr int kSparseCap = V2CE_SPARSE_CAP;      // events of one tile over all nine bins the sparse tile kernel holds
constexpr int kSparseThreads = 512;
constexpr size_t kSparseLds = (size_t)(2 * kSparseCap + kSparseThreads * 5 + 34) * 4 + 9 * 8 + (kSparseThreads / 64) * 10 * 4;
constexpr int kSlopeM = 31;            // slope table (g_slope_tab): |count difference| <= kSlopeM, count <= kSlopeM; else computed
constexpr int kSlopeTab = (2 * kSlopeM + 1) * (kSlopeM + 1);
// sort workgroups: 256 threads (dense segments) or 128 (make_plan)

struct LdatiParams {
    const float *vox;
    int B, H, W, HW;
    // scalars of LDATI.py:145-146 cast the way CPU torch casts python scalars
    double fps;        // python number used in the f64 single-event path
    float VS, VS2, INV, FPS;
    float RFPS, R9;    // f32(1 / FPS), f32(1 / 9): reciprocals of the two constant divisors of the k == 0 time (k0_time)
    double RFPS64, R9_64;   // RN(1 / fps), RN(1 / 9) in f64: the single-event time's two constant divisors (single_key_fast)
    int fast_slot;     // slot of g_fastdiv that holds the exhaustive check of k0_time's fast form for this FPS, or -1
    float offt[9];     // f32(arange(0,1/fps,1/fps/9)[c]) + f32(t0)
    long long kbase[9];  // key = timestamp - kbase[c], clamped to [0, NK)
    int NK, nbits;
    int ts32;          // every timestamp and key base fits int32: the f32 -> int conversions use 32 bits
    int strategy;      // V2CE_STRATEGY_*: NONE drops every multi-event voxel (LDATI.py:206-207,241)
    int bidir;         // bidirectional relocation (LDATI.py:107-122)
    const float2 *kbb; // pooled slope parameters {k, b} [B][2][9][HW] (LDATI.py:177-190), or null
    unsigned long long *keys;     // generic path ('random'): one 64-bit sort key per event, or null
    int rng_mode;
    const float *uniforms;
    int replay_max_n;
    unsigned long long seed;
    long long frame_base;
    const long long *seg_offsets;
    const long long *frame_ts_add;
    long long *ts;                // SoA outputs (all four or none)
    short *x;
    short *y;
    signed char *p;
    unsigned char *packed;        // or 13-byte packed records
    // two-level path
    int shift, NB, nb1;           // coarse bucket = key >> shift; nb1 = bits of a bucket index
    int T, tpp;                   // tiles per frame (2*tpp), tiles per polarity plane
    int PB;                       // bits of a pixel index
    int capA, cap2;               // LDS capacities (records) of the tile pass / the bucket sort
    int capP;                     // > 0: ldati_tile_pair_kernel -- records of one PASS (one or two bins of a tile) its LDS holds (>= capA)
    unsigned *lists;              // ldati_tile_onepass_kernel: the tiles' work lists [B][T][list_stride] words, or null
    int list_stride;
    int tbits;                    // binary-search steps over the tiles of a frame
    const unsigned *tile_off;     // [B][T][9] exclusive prefix of the tile counts inside the segment
    const unsigned *tc;           // [B][T][9] the tile counts themselves
    int sparse_cap;               // tiles with at most this many events (all nine bins) go to the sparse tile kernel; 0 = none
    unsigned short *roff;         // [B*9][T][NB+1] per tile: exclusive prefix of its bucket counts (last = tile total <= kCapTile)
    unsigned *bofs;               // [B*9][NB+1] exclusive prefix of the bucket totals inside the segment
    unsigned *groups;             // [B*9][NB] sort groups: first bucket | (end bucket << 16)
    unsigned *ngroups;            // [B*9]
    unsigned *big_list;           // [B*9*NB] coarse buckets beyond cap2: (segment << 16) | bucket
    unsigned *nbig;               // [1] their number
    int span;                     // most coarse buckets a sort group may cover (key span <= kMaxSpanKeys)
    int hist_bins;                // bins reserved per wave in the sort's LDS histogram
    unsigned *temp;               // [total events] 4-byte records (fine | multi | local pixel)
    int *seg_flag;                // [B*9] 1 = a bucket exceeds cap2 -> segment goes to the sweep kernel
    int *status;                  // [1] != 0: a flagged segment could not be swept (NK too large)
    int sweep_ok;
    int ballot_ranks;             // 1 = ignore g_lds_order_ok and rank with the ballot match-any (V2CE_LDATI_NO_ATOMIC_ORDER=1: the
                                  // fallback a device that fails the probe would take, forced so that tests can run it)
    // fused count + sparse tile pass (v2ce_ldati_count_fused): every tile owns a slot of kSparseCap records
    unsigned *tc_w;               // [B][T][9] tile counts, written by the fused kernel
    unsigned long long *stats_w;  // [5] max voxel count | - | - | - | largest tile total (all nine bins)
    unsigned *tile_abs_w;         // [B*9][T] record index of the (tile, bin) run inside `temp`
    const unsigned *tile_abs;     // the same, read by the bucket sort (null: runs at seg_offsets + tile_off)
    const int *fused_status;      // status word of the fused kernel, folded into `status` by the bucket scan
    int slot_cap;                 // > 0: ldati_tile_dense_kernel is ALSO the count pass (v2ce_ldati_count_fused in the dense regime): every
                                  // (tile, bin) run goes to its own slot of slot_cap (= capA) records, counts to tc_w, maxima to stats_w
    int Tp;                       // T rounded up to a multiple of 8
    unsigned *gruns;              // [B*9][NB][Tp] per sort group and tile: run start inside the tile's (tile, bin) run | records << 16,
                                  // written by the bucket scan (which has the run table in L2 anyway) so that a sort workgroup's
                                  // setup is ONE contiguous row instead of two 938-byte-strided loads per tile
    const unsigned *tile_src;     // [B*9][Tp] record index of the (tile, bin) run relative to the sort's base, one contiguous row per
                                  // segment (two-pass: tile_off transposed by the tile scan; fused: the slot starts)
};

// ---- Philox4x32-10, counter (pixel, j>>2, p*9+c, frame), key = seed ---------------------------
__device__ __forceinline__ void philox4(unsigned long long seed, unsigned pixel, unsigned jb,
                                        unsigned pc, unsigned frame, unsigned (&out)[4]) {
    unsigned c0 = pixel, c1 = jb, c2 = pc, c3 = frame;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (unsigned)p1; c2 = n2; c3 = (unsigned)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u24(unsigned w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned pixel, unsigned j,
                                                unsigned pc, unsigned frame) {
    unsigned o[4];
    philox4(seed, pixel, j >> 2, pc, frame, o);
    const unsigned sel = j & 3u;
    return u24(sel == 0 ? o[0] : sel == 1 ? o[1] : sel == 2 ? o[2] : o[3]);
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

// all nine bins of one pixel at once: counts and tendencies (LDATI.py:94-106, or :107-122 when bidir)
__device__ __forceinline__ void relocate_all(const float (&yv)[10], bool bidir, int (&n)[9], float (&tend)[9]) {
    const float eps = 1e-6f;
    float d = 0.0f;
    if (!bidir) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float r = yv[i] - d;
            const float cc = ceilf(r - eps);
            d = cc - r;
            n[i] = (int)cc;
            tend[i] = d;
        }
        n[8] += (int)(yv[9] - d);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float r = yv[i] - d;
        const float cc = ceilf(r - eps);
        d = cc - r;
        n[i] = (int)cc;
        tend[i] = d;
    }
    n[4] = 0;                                  // never written by the reference's bidirectional branch
    tend[4] = 0.0f;
    float bless = yv[9];
#pragma unroll
    for (int i = 8; i > 5; --i) {
        tend[i] = bless;
        float t = yv[i] + bless;
        t = floorf(t + eps);
        bless = (yv[i] - t) + bless;
        bless = bless < 0.0f ? 0.0f : bless;
        n[i] = (int)t;
    }
    tend[5] = bless - d;
    n[5] = (int)ceilf((yv[5] + bless) - d);
}

// a[c] for a wave-uniform c without dynamic register indexing
template <typename T>
__device__ __forceinline__ T pick9(const T (&a)[9], int c) {
    T v = a[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) v = c == i ? a[i] : v;
    return v;
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
                                             float &k, float &bb, const float2 *tab = nullptr) {
    if (tab) {                                                 // the tabulated results of the expressions below
        const int d = (c == 0 || c == 8) ? 0 : n_r - n_l;
        if (d >= -kSlopeM && d <= kSlopeM && n_c >= 0 && n_c <= kSlopeM && n_l >= 0 && n_r >= 0 && n_l < (1 << 23) && n_r < (1 << 23)) {
            const float2 kb = tab[(d + kSlopeM) * (kSlopeM + 1) + n_c];
            k = kb.x; bb = kb.y;
            return;
        }
    }
    // reflect padding makes the central difference vanish at the first and last bin
    const float sxy = (c == 0 || c == 8) ? 0.0f : ((float)n_r - (float)n_l);
    const float k0 = (3.0f * sxy) / 6.0f;
    k = (k0 / P.VS2) / ((float)n_c + 1e-8f);
    bb = P.INV - (P.VS * k) / 2.0f;
}

// ---- the k == 0 time (u / fps) / 9 (LDATI.py:196) without the two IEEE division sequences ------------------------
// Both divisors are constants of the call.  x / y = fma(fma(-q, y, x), r, q) with q = x * r, r = RN(1 / y), is the
// correctly rounded quotient for all but rare (x, y); instead of proving which, the composition is checked against the
// IEEE divisions for EVERY uniform the Philox path can produce (u = m * 2^-24, m < 2^24) by a 16 M-thread kernel, once
// per device and FPS, enqueued in front of the first emit that needs it; the result lands in g_fastdiv[slot] and the
// kernels take the fast form only when it says "identical for all inputs" (replayed uniforms are arbitrary floats: they
// always take the divisions).  22 -> 6 VALU operations on a path every wave with a multi-event voxel executes.
struct FastDiv { unsigned fps_bits; int ok; int tab_ready; int ok64; };
__device__ FastDiv g_fastdiv[8];
__device__ unsigned g_fastdiv_bad[8];
__device__ unsigned g_fast64_bad[8];
// The slope parameters {k, b} of a multi-event voxel (LDATI.py:188-190) depend on two small integers only -- the central
// difference of the neighbouring counts and the voxel's own count -- and cost three IEEE divisions: tabulated once per
// device and FPS by the very expressions of slope_params (so the entries ARE its results), looked up afterwards.
__device__ float2 g_slope_tab[8][kSlopeTab];

__device__ __forceinline__ float k0_time_fast(float u, float FPS, float RFPS, float R9) {
    float q = u * RFPS;
    q = __builtin_fmaf(__builtin_fmaf(-q, FPS, u), RFPS, q);
    float t = q * R9;
    t = __builtin_fmaf(__builtin_fmaf(-t, 9.0f, q), R9, t);
    return t;
}

// ---- the single-event time (LDATI.py:156-165) without its two f64 division sequences ---------------------------------
// t = (double)debt / fps / 9 with two constant divisors: q = x r, q = fma(fma(-q, y, x), r, q) with r = RN(1 / y) in f64, twice.
// As for k0_time_fast the composition is CHECKED, not proven: a kernel compares it with the IEEE divisions for every f32 the
// tendency of a forward-relocated voxel can take -- all floats in [0, 1) and all negative ones down to -2^-18 (the
// recurrence leaves debt in (-1e-6 - ulp, 1)) -- once per device and fps (~2e9 values, a few ms); the kernels use it only when
// the verdict is "identical everywhere" AND the lane's value lies inside the checked range (anything else -- the
// bidirectional branch's tendencies reach 2 -- takes the divisions).  ~70 -> ~12 f64 operations per single event, which is
// most events of real UNet output.
__device__ __forceinline__ double single_time_fast(float debt, double fps, double rfps, double r9) {
    const double x = (double)debt;
    double q = x * rfps;
    q = __builtin_fma(__builtin_fma(-q, fps, x), rfps, q);
    double t = q * r9;
    t = __builtin_fma(__builtin_fma(-t, 9.0, q), r9, t);
    return t;
}
constexpr unsigned kFast64Neg = 0x36800000u;                 // bits of 2^-18: negative tendencies checked down to -2^-18
__device__ __forceinline__ bool single_fast_range(float debt) { return debt < 1.0f && debt > -0x1p-18f; }

__global__ __launch_bounds__(256) void ldati_fast64_check_kernel(double fps, double rfps, double r9, int slot) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    const unsigned npos = 0x3F800000u;                                    // floats in [0, 1)
    if (i >= (unsigned long long)npos + kFast64Neg) return;
    const unsigned bits = i < npos ? (unsigned)i : 0x80000000u + (unsigned)(i - npos);
    const float d = __uint_as_float(bits);
    const double want = (double)d / fps / 9.0;
    const double got = single_time_fast(d, fps, rfps, r9);
    if (!(want == got)) atomicAdd(&g_fast64_bad[slot], 1u);      // (numeric: -0 against +0 for debt = -0 is the same time)
}

__global__ __launch_bounds__(256) void ldati_fastdiv_check_kernel(float FPS, float RFPS, float R9, int slot) {
    const unsigned m = blockIdx.x * 256u + threadIdx.x;                  // < 2^24
    const float u = (float)m * (1.0f / 16777216.0f);
    const float want = (u / FPS) / 9.0f;
    const float got = k0_time_fast(u, FPS, RFPS, R9);
    if (__float_as_uint(want) != __float_as_uint(got)) atomicAdd(&g_fastdiv_bad[slot], 1u);
}
__global__ __launch_bounds__(256) void ldati_slope_tab_kernel(float VS, float VS2, float INV, int slot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kSlopeTab) return;
    const int d = i / (kSlopeM + 1) - kSlopeM, n = i % (kSlopeM + 1);
    const float sxy = (float)d;                                // = (float)n_r - (float)n_l: small integers, exact
    const float k0 = (3.0f * sxy) / 6.0f;
    const float k = (k0 / VS2) / ((float)n + 1e-8f);
    g_slope_tab[slot][i] = make_float2(k, INV - (VS * k) / 2.0f);
}
__global__ void ldati_fastdiv_commit_kernel(float FPS, int slot) {
    g_fastdiv[slot].fps_bits = __float_as_uint(FPS);
    g_fastdiv[slot].ok = g_fastdiv_bad[slot] == 0u ? 1 : 0;
    g_fastdiv[slot].ok64 = g_fast64_bad[slot] == 0u ? 1 : 0;
    g_fastdiv[slot].tab_ready = 1;
}

// multi-event timestamp, all f32 (LDATI.py:195-196,210-212)
__device__ __forceinline__ long long multi_ts(float k, float bb, float u, float offt,
                                              const LdatiParams &P) {
    float t;
    if (P.strategy == V2CE_STRATEGY_RANDOM) {
        t = u;                                    // LDATI.py:173-174: the raw uniform, in seconds
    } else if (k == 0.0f) {
        t = (u / P.FPS) / 9.0f;
    } else {
        const float s = bb * bb + (2.0f * k) * u;
        t = (-bb + __builtin_sqrtf(s)) / k;
    }
    t = t + offt;
    t = t * 1e6f;
    return (long long)t;
}

// the same, f32 -> i32 (bit-identical to the i64 conversion while |t| < 2^31: P.ts32) and the key
__device__ __forceinline__ unsigned multi_key(float k, float bb, float u, float offt, int kbase32, const LdatiParams &P,
                                              bool fast = false) {
    float t;
    if (P.strategy == V2CE_STRATEGY_RANDOM) {
        t = u;
    } else if (k == 0.0f) {
        t = fast ? k0_time_fast(u, P.FPS, P.RFPS, P.R9) : (u / P.FPS) / 9.0f;
    } else {
        const float s = bb * bb + (2.0f * k) * u;
        t = (-bb + __builtin_sqrtf(s)) / k;
    }
    t = t + offt;
    t = t * 1e6f;
    int key = (int)t - kbase32;
    key = key < 0 ? 0 : key;
    key = key >= P.NK ? P.NK - 1 : key;
    return (unsigned)key;
}

__device__ __forceinline__ int key_of(long long T, long long kbase, int NK) {
    long long k = T - kbase;
    k = k < 0 ? 0 : k;
    k = k >= NK ? NK - 1 : k;
    return (int)k;
}

// the key of a single event: fast form when the wave's tendencies all lie inside the checked range (`fast`: the device
// verdict, 32-bit times), else the divisions.  Must be called by whole waves (the range test is a wave vote).
__device__ __forceinline__ unsigned single_key(bool has, float debt, float offt, long long kbase, bool fast, const LdatiParams &P) {
    const bool in = !has || single_fast_range(debt);
    if (fast && __ballot(!in) == 0ull) {
        double t = single_time_fast(debt, P.fps, P.RFPS64, P.R9_64);
        t += (double)offt;
        t *= 1e6;
        int k = (int)t - (int)kbase;                      // (int)t == (long long)t while |t| < 2^31 (P.ts32)
        k = k < 0 ? 0 : k;
        return (unsigned)(k >= P.NK ? P.NK - 1 : k);
    }
    return (unsigned)key_of(single_ts(debt, P.fps, offt), kbase, P.NK);
}


// ballot match-any: lanes of `has_mask` with equal `key` form a peer group.  Returns the rank of
// this lane inside its group (peers on lower lanes) and the group size.  ~5 VALU per key bit.
__device__ __forceinline__ unsigned match_rank(unsigned key, int nbits, unsigned long long has_mask,
                                               unsigned &npeers) {
    unsigned mlo = 0, mhi = 0;                           // lanes that differ from this lane in some bit
    for (int b = 0; b < nbits; ++b) {
        const int sel = __builtin_amdgcn_sbfe((int)key, b, 1);          // 0 or -1
        const unsigned long long m = __ballot(sel != 0);
        mlo |= (unsigned)m ^ (unsigned)sel;
        mhi |= (unsigned)(m >> 32) ^ (unsigned)sel;
    }
    const unsigned plo = (unsigned)has_mask & ~mlo, phi = (unsigned)(has_mask >> 32) & ~mhi;
    npeers = (unsigned)__popc(plo) + (unsigned)__popc(phi);
    return __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
}

// One 64-record batch of a stable counting sort: `slot` = this wave's running base of the lane's
// bin (LDS, owned by the wave).  Returns base + rank; the last peer advances the base.
__device__ __forceinline__ unsigned take_slots(bool has, unsigned key, int nbits, unsigned *slot) {
    unsigned npeers;
    const unsigned rank = match_rank(key, nbits, __ballot(has), npeers);
    unsigned pos = 0;
    if (has) {
        const unsigned base = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_wave_barrier();
        if (rank + 1 == npeers) __hip_atomic_store(slot, base + npeers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        pos = base + rank;
    }
    __builtin_amdgcn_wave_barrier();
    return pos;
}

// The same on a histogram that packs the counters of TWO waves into one word (16 bits each, `sh` = 0 or 16: the tile
// pass): the word is shared with the neighbouring wave, so the group's slots are taken with one atomic add by its first
// lane and handed to the peers through the LDS crossbar.
__device__ __forceinline__ unsigned take_slots_packed(bool has, unsigned key, int nbits, unsigned *slot, unsigned sh) {
    unsigned mlo = 0, mhi = 0;
    const unsigned long long has_mask = __ballot(has);
    for (int b = 0; b < nbits; ++b) {
        const int sel = __builtin_amdgcn_sbfe((int)key, b, 1);
        const unsigned long long m = __ballot(sel != 0);
        mlo |= (unsigned)m ^ (unsigned)sel;
        mhi |= (unsigned)(m >> 32) ^ (unsigned)sel;
    }
    const unsigned plo = (unsigned)has_mask & ~mlo, phi = (unsigned)(has_mask >> 32) & ~mhi;
    const unsigned npeers = (unsigned)__popc(plo) + (unsigned)__popc(phi);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    const int leader = plo ? __builtin_ctz(plo) : 32 + __builtin_ctz(phi | 0x80000000u);
    unsigned old = 0;
    if (has && rank == 0) old = atomicAdd(slot, npeers << sh);
    old = (unsigned)__shfl((int)old, leader);
    return ((old >> sh) & 0xFFFFu) + rank;
}

// ---- ranks straight from LDS atomics ------------------------------------------------------------
// On gfx950 one wave-instruction of ds_add_rtn_u32 serves the lanes that hit the same LDS word in
// ascending lane order (tools/micro/lds_atomic_order.hip: 0 exceptions in 5.4e9 returned values),
// so the returned value IS the stable rank and the ballot match-any (~4 VALU per key bit and batch)
// is not needed.  That order is not an architectural promise: a probe kernel checks it on every
// device the library runs on (enqueued once, in front of the first count call) and only then sets
// g_lds_order_ok; until / unless it does, the kernels use the ballot ranks (identical results).
__device__ int g_lds_order_ok = 0;
__device__ unsigned g_lds_probe_bad = 0, g_lds_probe_done = 0;

// ---- in-kernel phase stamps (diagnostic build only: make STAMP=1) ---------------------------------
#ifdef V2CE_STAMP
__device__ unsigned long long g_stamp[32];
#define STAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[12] = {0}
#define STAMP(i) do { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[i] += st_now - st_last; st_last = st_now; } while (0)
#define STAMP_FLUSH(base, n) do { if (threadIdx.x == 0) for (int st_i = 0; st_i < (n); ++st_i) atomicAdd(&g_stamp[(base) + st_i], st_acc[st_i]); } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(base, n)
#endif

__global__ __launch_bounds__(256) void ldati_lds_order_probe_kernel(int iters) {
    __shared__ unsigned tab[4][512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = lane; i < 512; i += 64) tab[wid][i] = 7u * i;
        __builtin_amdgcn_wave_barrier();
        s = s * 1664525u + 1013904223u;
        const unsigned range = 1u << (it % 10);
        const unsigned key = (s >> 9) & (range - 1u);
        const bool act = ((s >> 5) & 7u) != 0u || (it & 1);
        unsigned got = 0;
        if (act) got = atomicAdd(&tab[wid][key], 1u);
        unsigned np;
        const unsigned want = 7u * key + match_rank(key, 9, __ballot(act), np);
        if (act && got != want) ++nbad;
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(&g_lds_probe_bad, nbad);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned done = atomicAdd(&g_lds_probe_done, 1u);
        if (done == gridDim.x - 1) {
            __threadfence();
            g_lds_order_ok = atomicAdd(&g_lds_probe_bad, 0u) == 0u ? 1 : 0;
        }
    }
}

// slot of one record in a stable counting sort batch: LDS-atomic rank when the device passed the
// probe, ballot rank otherwise
__device__ __forceinline__ unsigned take_slot(bool atomic_order, bool has, unsigned key, int nbits, unsigned *slot) {
    if (atomic_order) return has ? atomicAdd(slot, 1u) : 0u;
    return take_slots(has, key, nbits, slot);
}

// px / W for px + 0.5 < 2^22 in three operations: (px + 0.5) / W lies at least 0.5 / W away from every integer, and the
// two roundings (1 / W, the product) move it by less than (px + 0.5) / W * 2^-23 < 0.5 / W, so the truncation is exact
__device__ __forceinline__ unsigned div_tiny(unsigned px, float rcpW) {
    return (unsigned)(((float)px + 0.5f) * rcpW);
}

// px / W for px < 2^24 (exact in f32) without an integer division
__device__ __forceinline__ unsigned div_small(unsigned px, unsigned W, float rcpW) {
    unsigned q = (unsigned)((float)px * rcpW);
    const int r = (int)(px - q * W);
    q += (r >= (int)W) ? 1u : 0u;
    q -= (r < 0) ? 1u : 0u;
    return q;
}

__device__ __forceinline__ void store_packed_bytes(unsigned char *dst, long long t, unsigned xx,
                                                   unsigned yy, unsigned pp) {
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = (unsigned char)((unsigned long long)t >> (8 * k));
    dst[8] = (unsigned char)xx; dst[9] = (unsigned char)(xx >> 8);
    dst[10] = (unsigned char)yy; dst[11] = (unsigned char)(yy >> 8);
    dst[12] = (unsigned char)pp;
}

// inclusive scan over the 64 lanes of a wave: DPP row shifts inside the rows of 16 lanes, then the two row
// broadcasts of gfx9 (lane 15 of a row into the next row; lane 31 into rows 2-3) -- six VALU operations instead
// of six dependent ds_bpermute round trips through the LDS crossbar (~100 cycles each)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
    (void)lane;
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return (unsigned)x;
}

// exclusive scan of one value per thread over a workgroup of NW <= 64 waves; `part` = NW LDS words.
// Returns the exclusive prefix; *total = sum over the workgroup.  ONE barrier: every wave scans the NW wave totals itself
// (a 64-lane DPP scan costs less than a second barrier and a serial loop on one thread).  The caller separates two scans
// that share `part` by a barrier of its own (every call site has one: the totals are read right behind the barrier here).
template <int NW>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *part, unsigned *total) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned incl = wave_incl_scan(v, lane);
    if (lane == 63) part[wid] = incl;
    __syncthreads();
    const unsigned pin = wave_incl_scan(lane < NW ? part[lane] : 0u, lane);
    *total = (unsigned)__builtin_amdgcn_readlane((int)pin, NW - 1);
    const unsigned base = wid ? (unsigned)__builtin_amdgcn_readlane((int)pin, wid - 1) : 0u;
    return base + incl - v;
}

// the same for two values per thread with one barrier; `part` = 2 * NW LDS words
template <int NW>
__device__ __forceinline__ void block_excl_scan2(unsigned a, unsigned e, unsigned *part, unsigned &a_ex,
                                                 unsigned &e_ex, unsigned &a_tot, unsigned &e_tot) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned ia = wave_incl_scan(a, lane), ie = wave_incl_scan(e, lane);
    if (lane == 63) { part[wid] = ia; part[NW + wid] = ie; }
    __syncthreads();
    const unsigned pa = wave_incl_scan(lane < NW ? part[lane] : 0u, lane);
    const unsigned pe = wave_incl_scan(lane < NW ? part[NW + lane] : 0u, lane);
    a_tot = (unsigned)__builtin_amdgcn_readlane((int)pa, NW - 1);
    e_tot = (unsigned)__builtin_amdgcn_readlane((int)pe, NW - 1);
    a_ex = (wid ? (unsigned)__builtin_amdgcn_readlane((int)pa, wid - 1) : 0u) + ia - a;
    e_ex = (wid ? (unsigned)__builtin_amdgcn_readlane((int)pe, wid - 1) : 0u) + ie - e;
}

// `want` consecutive slots of an LDS counter for every lane with ONE atomic per wave (all 64 lanes must be active):
// a per-lane atomicAdd on one address is served lane by lane -- up to 64 LDS cycles per wave instruction.
__device__ __forceinline__ unsigned wave_alloc(unsigned *counter, unsigned want, int lane) {
    const unsigned incl = wave_incl_scan(want, lane);
    const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    unsigned base = 0;
    if (tot) {                                              // wave-uniform
        if (lane == 63) base = atomicAdd(counter, tot);
        base = (unsigned)__builtin_amdgcn_readlane((int)base, 63);
    }
    return base + incl - want;
}

// ---------------------------------------------------------------------------------------------
// count: workgroup per (frame, tile); events per (tile, bin); max count per voxel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCountThreads) void ldati_count_tiles_kernel(
    const float *__restrict__ vox, int HW, int tpp, int strategy, int bidir, unsigned *__restrict__ tc,
    unsigned long long *stats) {
    const int t = blockIdx.x, b = blockIdx.y, T = 2 * tpp;
    const int pidx = t < tpp ? 1 : 0;                 // negative tiles first (LDATI.py:289)
    const int x0 = (t < tpp ? t : t - tpp) * kTilePix;
    const float *plane0 = vox + (long long)(b * 2 + pidx) * 10 * HW;
    int cnt[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cnt[i] = 0;
    int mx = 0;
    constexpr int PPT = kTilePix / kCountThreads;
    // four consecutive pixels per thread: one 16-byte load per plane, all ten in flight at once (any pixel order
    // is right: only sums are formed)
    float yq[PPT][10];
    {
        const int px = x0 + (int)threadIdx.x * PPT;
        if ((HW & 3) == 0 && px + PPT <= HW) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const float4 v = *reinterpret_cast<const float4 *>(plane0 + (long long)i * HW + px);
                yq[0][i] = v.x; yq[1][i] = v.y; yq[2][i] = v.z; yq[3][i] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q)
#pragma unroll
                for (int i = 0; i < 10; ++i) yq[q][i] = (px + q < HW) ? plane0[(long long)i * HW + px + q] : 0.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        int nn[9];
        float td[9];
        relocate_all(yq[q], bidir != 0, nn, td);                     // (an all-zero pixel past the image counts nothing)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int ni = nn[i];
            cnt[i] += (strategy == V2CE_STRATEGY_NONE) ? (ni == 1) : (ni > 0 ? ni : 0);
            mx = ni > mx ? ni : mx;
        }
    }
    __shared__ int red[kCountThreads / 64][10];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const unsigned v = wave_incl_scan((unsigned)cnt[i], lane);       // (DPP: no LDS crossbar traffic)
        if (lane == 63) red[wid][i] = (int)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int m = __shfl_xor(mx, o);
        mx = m > mx ? m : mx;
    }
    if (lane == 0) red[wid][9] = mx;
    __syncthreads();
    if (threadIdx.x < 9) {
        unsigned s = 0;
#pragma unroll
        for (int w = 0; w < kCountThreads / 64; ++w) s += (unsigned)red[w][threadIdx.x];
        tc[((long long)b * T + t) * 9 + threadIdx.x] = s;
    } else if (threadIdx.x == 9) {
        int m = 0;
#pragma unroll
        for (int w = 0; w < kCountThreads / 64; ++w) m = red[w][9] > m ? red[w][9] : m;
        // (a plain read first: after the first few tiles the maximum rarely grows, and 2 000 same-address atomics serialise)
        if (m > 0 && (unsigned long long)m > *reinterpret_cast<volatile unsigned long long *>(&stats[0])) atomicMax(&stats[0], (unsigned long long)m);
    }
}

// Tile counts -> exclusive tile offsets inside each segment, one wave per segment (the segment count
// lands in seg_offsets[seg]; the largest (tile, bin) and segment counts in stats[1], stats[2]).
__global__ __launch_bounds__(256) void ldati_tile_scan_kernel(const unsigned *__restrict__ tc, int B,
                                                              int T, unsigned *__restrict__ tile_off,
                                                              unsigned *__restrict__ tile_src, int Tp,
                                                              long long *seg_offsets,
                                                              unsigned long long *stats) {
    // one wave per FRAME, lane = tile, all nine bins of the tile in the lane: the nine counts of a tile are 36 contiguous bytes,
    // so the wave reads and writes whole blocks (a wave per (frame, bin) read and wrote every ninth word of the same cache
    // lines as its eight siblings: 16 us for the 576 segments of an e2e call, now 4)
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    unsigned run[9], mx = 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) run[c] = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int tt = t0 + lane;
        const long long i0 = ((long long)b * T + tt) * 9;
        unsigned v[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c] = tt < T ? tc[i0 + c] : 0u;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const unsigned incl = wave_incl_scan(v[c], lane);
            const unsigned ex = run[c] + incl - v[c];
            if (tt < T) {
                tile_off[i0 + c] = ex;
                tile_src[(long long)(b * 9 + c) * Tp + tt] = ex;      // the same, [segment][tile] for the bucket sort's setup
            }
            run[c] += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            mx = v[c] > mx ? v[c] : mx;
        }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned m = __shfl_xor(mx, o);
        mx = m > mx ? m : mx;
    }
    if (lane < 9) {
        unsigned r = run[0];
#pragma unroll
        for (int c = 1; c < 9; ++c) r = lane == c ? run[c] : r;
        seg_offsets[b * 9 + lane] = r;
        atomicMax(&stats[2], (unsigned long long)r);
    }
    if (lane == 0) atomicMax(&stats[1], (unsigned long long)mx);
}

// One workgroup: segment counts -> exclusive segment offsets (seg_offsets[n] = stats[3] = total).
__global__ __launch_bounds__(256) void ldati_seg_scan_kernel(int n, long long *seg_offsets, unsigned long long *stats) {
    __shared__ long long part[256];
    const int t = threadIdx.x;
    const int per = (n + 255) / 256;
    const int lo = t * per, hi = (lo + per < n) ? lo + per : n;
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += seg_offsets[i];
    part[t] = s;
    __syncthreads();
    if (t < 64) {                                   // 256 partials: four per lane of one wave
        long long v[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = part[4 * t + j]; sum += v[j]; }
        long long incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long u = __shfl_up(incl, o);
            if (t >= o) incl += u;
        }
        long long run = incl - sum;
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[4 * t + j] = run; run += v[j]; }
        if (t == 63) {
            seg_offsets[n] = incl;
            stats[3] = (unsigned long long)incl;
        }
    }
    __syncthreads();
    long long run = part[t];
    for (int i = lo; i < hi; ++i) {
        const long long v = seg_offsets[i];
        seg_offsets[i] = run;
        run += v;
    }
}

// ---------------------------------------------------------------------------------------------
// sweep kernel (v1): one workgroup (4 waves = the 4 tie-order categories) per (frame, bin) segment:
//   A. histogram cnt[cat][key]  B. exclusive scan in (key, category) order = stable ranks
//   C. replay the same events in pixel order; rank among the equal-key events of the wave batch
//      from a ballot match-any, running base from LDS; scatter straight to the final position.
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
    const unsigned slot_pos = take_slots(has, (unsigned)key, P.nbits, slot);
    if (has) {
        const long long pos = seg_lo + (long long)slot_pos;
        const int yy = px / P.W;
        if (P.packed) {
            store_packed_bytes(P.packed + pos * 13, T + ts_add, (unsigned)(px - yy * P.W) & 0xFFFFu,
                               (unsigned)yy & 0xFFFFu, (unsigned)pol);
        } else {
            P.ts[pos] = T + ts_add;
            P.x[pos] = (short)(px - yy * P.W);
            P.y[pos] = (short)yy;
            P.p[pos] = pol;
        }
    }
}

template <bool RANK>
__device__ __forceinline__ void sweep_singles(int b, int c, int pidx, int cat, signed char pol,
                                              int lane, long long seg_lo, long long ts_add,
                                              unsigned *cnt, const LdatiParams &P) {
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
        if (P.kbb) {
            const float2 kq = valid ? P.kbb[((long long)(b * 2 + pidx) * 9 + c) * P.HW + px] : make_float2(0.0f, 0.0f);
            k = kq.x; bb = kq.y;
        } else {
            slope_params(n_l, n_c, n_r, c, P, k, bb);
        }
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
    if (P.seg_flag && !P.seg_flag[seg]) return;   // the two-level path handled this segment
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
// tile pass: workgroup per (frame, tile), NT threads, PPT consecutive pixels per thread, bin by bin
// ---------------------------------------------------------------------------------------------
// LDS map (dynamic): S [capA] u32 | O [capA + 2048] u32 (aliased by the unit tables while the
// timestamps are computed) | PT [2048] {k, bb} | hist [NW/2][NB] u32 (two 16-bit chunk counters per word) | misc
extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];

template <int NT, int PPT, bool BIDIR>
__global__ __launch_bounds__(NT) void ldati_tile_pass_kernel(LdatiParams P) {
    static_assert(NT * PPT == kTilePix, "tile geometry");
    const int t = blockIdx.x, b = blockIdx.y;
    if (P.sparse_cap) {                                 // the sparse tile kernel owns the lightly populated tiles
        const unsigned *tcr = P.tc + ((long long)b * P.T + t) * 9;
        unsigned ntot = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) ntot += tcr[c];
        if (ntot <= (unsigned)P.sparse_cap) return;
    }
    const int pidx = t < P.tpp ? 1 : 0;
    const int x0 = (t < P.tpp ? t : t - P.tpp) * kTilePix;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = NT / 64;
    const bool atomic_order = !P.ballot_ranks && __builtin_amdgcn_readfirstlane(g_lds_order_ok) != 0;
    // the checked fast form of the k == 0 time: Philox uniforms only, and only if this FPS passed the exhaustive check
    const bool slot_ok = P.fast_slot >= 0 &&
                         __builtin_amdgcn_readfirstlane((int)g_fastdiv[P.fast_slot >= 0 ? P.fast_slot : 0].fps_bits) == (int)__float_as_uint(P.FPS);
    const bool fast_k0 = slot_ok && P.rng_mode == V2CE_RNG_PHILOX && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].ok) != 0;
    const float2 *stab = (slot_ok && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].tab_ready) != 0) ? g_slope_tab[P.fast_slot] : nullptr;

    unsigned *S = reinterpret_cast<unsigned *>(tile_smem);
    unsigned *O = S + P.capA;
    float2 *PT = reinterpret_cast<float2 *>(O + P.capA + 2048);
    unsigned *hist = reinterpret_cast<unsigned *>(PT + kTilePix);
    unsigned *part = hist + (NW / 2) * P.NB;           // [NW] x 2 (hist: two 16-bit counters per word)
    // unit tables alias O: singles {debt bits, local} then multi units {info, event offset}
    uint2 *SL = reinterpret_cast<uint2 *>(O);

    const float *plane0 = P.vox + (long long)(b * 2 + pidx) * 10 * P.HW;
    const unsigned frame = (unsigned)(P.frame_base + b);
    const int lpx0 = tid * PPT;                        // first local pixel of this thread
    const float eps = 1e-6f;

    // relocation state per pixel: counts of bins c-1, c, c+1; tendency ("debt") of bin c.  Forward
    // relocation rolls along the bins (the voxels of bin c+2 are fetched while bin c is processed);
    // the bidirectional variant (LDATI.py:107-122) needs all ten voxels first and keeps all nine bins.
    int nprev[PPT], ncur[PPT], nnext[PPT];
    float dcur[PPT], dnext[PPT];
    bool valid[PPT];
    int nA[BIDIR ? PPT : 1][9];
    float tA[BIDIR ? PPT : 1][9];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int px = x0 + lpx0 + q;
        valid[q] = px < P.HW;
        if (BIDIR) {
            float yv[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) yv[i] = valid[q] ? plane0[(long long)i * P.HW + px] : 0.0f;
            relocate_all(yv, true, nA[q], tA[q]);
            nprev[q] = ncur[q] = nnext[q] = 0;
            dcur[q] = dnext[q] = 0.0f;
        } else {
            const float y0 = valid[q] ? plane0[px] : 0.0f;
            const float y1 = valid[q] ? plane0[(long long)P.HW + px] : 0.0f;
            float r = y0 - 0.0f;
            float cc = ceilf(r - eps);
            dcur[q] = cc - r;
            ncur[q] = (int)cc;
            r = y1 - dcur[q];
            cc = ceilf(r - eps);
            dnext[q] = cc - r;
            nnext[q] = (int)cc;
            nprev[q] = 0;
        }
    }

    STAMP_DECL;
    for (int c = 0; c < 9; ++c) {
        STAMP(0);
        // prefetch the voxels of bin c+2 (and bin 9 with it when c+2 == 8)
        float ynn[PPT], y9[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int px = x0 + lpx0 + q;
            ynn[q] = (!BIDIR && c + 2 <= 8 && valid[q]) ? plane0[(long long)(c + 2) * P.HW + px] : 0.0f;
            y9[q] = (!BIDIR && c + 2 == 8 && valid[q]) ? plane0[(long long)9 * P.HW + px] : 0.0f;
            if (BIDIR) {
                nprev[q] = c > 0 ? pick9(nA[q], c - 1) : 0;
                ncur[q] = pick9(nA[q], c);
                nnext[q] = c < 8 ? pick9(nA[q], c + 1) : 0;
                dcur[q] = pick9(tA[q], c);
            }
        }
        // ---- P1: classify ------------------------------------------------------------------
        // Multi-event voxels come in two classes: slope k == 0 (time = (u / fps) / 9: two constant divisions) and k != 0
        // (a square root and a division).  Their 4-draw units go to two separate regions of the unit table (k == 0
        // units first), so that a wave of the timestamp phase executes ONE of the two code paths, not both.
        unsigned a_tot = 0, e_tot = 0;                  // per thread: singles | k==0 units << 12;  multi events | k!=0 units << 14
        unsigned a_q[PPT], e_q[PPT];
        bool kz_q[PPT];
        float kk_q[PPT], bb_q[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int n = valid[q] ? ncur[q] : 0;
            const bool single = n == 1;
            const bool multi = n >= 2 && P.strategy != V2CE_STRATEGY_NONE;
            kk_q[q] = bb_q[q] = 0.0f;
            if (multi) {
                if (P.kbb) {                         // pooled counts (LDATI.py:177-190): from the pre-pass
                    const float2 kq = P.kbb[((long long)(b * 2 + pidx) * 9 + c) * P.HW + (x0 + lpx0 + q)];
                    kk_q[q] = kq.x; bb_q[q] = kq.y;
                } else {
                    slope_params(nprev[q], n, nnext[q], c, P, kk_q[q], bb_q[q], stab);
                }
            }
            kz_q[q] = kk_q[q] == 0.0f;
            const unsigned units = multi ? (unsigned)(n + 3) >> 2 : 0u;
            a_q[q] = a_tot;
            e_q[q] = e_tot;
            a_tot += (single ? 1u : 0u) + (kz_q[q] ? units << 12 : 0u);
            e_tot += (multi ? (unsigned)n : 0u) + (kz_q[q] ? 0u : units << 14);
        }
        // ---- P2: workgroup scans in pixel order (thread-major, pixel-minor).  Their first barrier
        // also separates the previous bin's reads of O (P7) from this bin's unit tables.
        unsigned a_base, e_base, A_all, E_all;
        block_excl_scan2<NW>(a_tot, e_tot, part, a_base, e_base, A_all, E_all);
        STAMP(1);
        const unsigned Ns = A_all & 0xFFFu, U0 = A_all >> 12, Nm = E_all & 0x3FFFu, Um = U0 + (E_all >> 14);
        const unsigned N = Ns + Nm;
        uint2 *MU = SL + Ns;
        // the records are ranked in NC contiguous chunks of L records, one wave each (a sparse tile
        // needs few: the histogram work below is proportional to NC)
        const unsigned NC = N > 256u * NW ? (unsigned)NW : (N + 255u) / 256u;
        const unsigned L = NC ? ((N + 64u * NC - 1u) / (64u * NC)) * 64u : 64u;
        for (unsigned i = tid; i < ((NC + 1u) >> 1) * P.NB; i += NT) hist[i] = 0;   // (chunk pair, bucket): two 16-bit counters per word
        // ---- P3: unit tables ----------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int n = valid[q] ? ncur[q] : 0;
            const unsigned local = (unsigned)(lpx0 + q);
            const unsigned ab = a_base + a_q[q], eb = e_base + e_q[q];
            if (n == 1) {
                SL[ab & 0xFFFu] = make_uint2(__float_as_uint(dcur[q]), local);
            } else if (n >= 2 && P.strategy != V2CE_STRATEGY_NONE) {
                PT[local] = make_float2(kk_q[q], bb_q[q]);
                const unsigned u0 = kz_q[q] ? ab >> 12 : U0 + (eb >> 14), ev0 = eb & 0x3FFFu;
                const unsigned units = (unsigned)(n + 3) >> 2;
                for (unsigned jb = 0; jb < units; ++jb) {
                    const unsigned left = (unsigned)n - 4u * jb;
                    MU[u0 + jb] = make_uint2(local | (jb << kLocalBits) | ((left < 4u ? left : 4u) << 28),
                                             ev0 + 4u * jb);
                }
            }
        }
        __syncthreads();
        STAMP(2);
        // ---- P4: timestamps, once, every lane busy; each record also counts in the histogram of
        // the wave that will rank it (contiguous chunks of S, L records per wave) -----------------
        const float invL = 1.0f / (float)L;
        for (unsigned q = tid; q < Ns; q += NT) {
            const uint2 e = SL[q];
            const long long Tq = single_ts(__uint_as_float(e.x), P.fps, P.offt[c]);
            const unsigned key = (unsigned)key_of(Tq, P.kbase[c], P.NK);
            S[q] = (key << 12) | e.y;
            const unsigned w = (unsigned)(((float)q + 0.5f) * invL);
            if (!P.keys) atomicAdd(&hist[__umul24(w >> 1, (unsigned)P.NB) + (key >> P.shift)], 1u << ((w & 1u) * 16u));
        }
        STAMP(3);
        // The common case -- Philox draws, 32-bit timestamps, the 'slope' strategy, the bucket machinery -- has its own copy of
        // the loop without the per-event uniform tests (the pass is bound by instruction issue); COMMON 1 = with the checked
        // fast k == 0 time, 2 = with the divisions.  The slope class is tested once per unit (waves are class-homogeneous).
        auto multis = [&](auto common_c) {
            constexpr int COMMON = decltype(common_c)::value;
            const float offt_c = P.offt[c];
            const int kbase_c = (int)P.kbase[c];
            for (unsigned q = tid; q < Um; q += NT) {
                const uint2 e = MU[q];
                const unsigned local = e.x & (kTilePix - 1), jb = (e.x >> kLocalBits) & 0x1FFFFu, cnt = e.x >> 28;
                const float2 kb = PT[local];
                const unsigned px = (unsigned)x0 + local;
                float u[4];
                if (!COMMON && P.rng_mode == V2CE_RNG_REPLAY) {
                    const long long ub = (((long long)(b * 2 + pidx) * 9 + c) * P.HW + px) * P.replay_max_n;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int j = (int)(4u * jb) + s;
                        u[s] = ((unsigned)s < cnt && j < P.replay_max_n) ? P.uniforms[ub + j] : 0.0f;
                    }
                } else {
                    unsigned o[4];
                    philox4(P.seed, px, jb, (unsigned)(pidx * 9 + c), frame, o);
#pragma unroll
                    for (int s = 0; s < 4; ++s) u[s] = u24(o[s]);
                }
                unsigned key[4];
                if (COMMON) {
                    float t[4];
                    if (kb.x == 0.0f) {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            t[s] = COMMON == 1 ? k0_time_fast(u[s], P.FPS, P.RFPS, P.R9) : (u[s] / P.FPS) / 9.0f;
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const float sq = kb.y * kb.y + (2.0f * kb.x) * u[s];
                            t[s] = (-kb.y + __builtin_sqrtf(sq)) / kb.x;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {                   // multi_key's tail: (t + offt) * 1e6 -> key, clamped
                        float tt = t[s] + offt_c;
                        tt = tt * 1e6f;
                        int k = (int)tt - kbase_c;
                        k = k < 0 ? 0 : k;
                        key[s] = (unsigned)(k >= P.NK ? P.NK - 1 : k);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        key[s] = P.ts32 ? multi_key(kb.x, kb.y, u[s], P.offt[c], (int)P.kbase[c], P, fast_k0)
                                        : (unsigned)key_of(multi_ts(kb.x, kb.y, u[s], P.offt[c], P), P.kbase[c], P.NK);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if ((unsigned)s < cnt) {
                        const unsigned pos = Ns + e.y + s;
                        S[pos] = (key[s] << 12) | (1u << kLocalBits) | local;
                        const unsigned w = (unsigned)(((float)pos + 0.5f) * invL);
                        if (COMMON || !P.keys) atomicAdd(&hist[__umul24(w >> 1, (unsigned)P.NB) + (key[s] >> P.shift)], 1u << ((w & 1u) * 16u));
                    }
                }
            }
        };
        if (P.rng_mode == V2CE_RNG_PHILOX && P.ts32 && !P.keys && P.strategy == V2CE_STRATEGY_SLOPE) {
            if (fast_k0) multis(std::integral_constant<int, 1>{});
            else multis(std::integral_constant<int, 2>{});
        } else {
            multis(std::integral_constant<int, 0>{});
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);
        if (P.keys) {
            // generic path (key range beyond the bucket machinery: 'random'): one 64-bit key per event,
            // (segment | timestamp - bin start | category | pixel), sorted by a library radix sort
            unsigned long long *dst = P.keys + P.seg_offsets[b * 9 + c] + P.tile_off[((long long)b * P.T + t) * 9 + c];
            const unsigned long long seg_bits = (unsigned long long)(b * 9 + c) << 44;
            for (unsigned i = tid; i < N; i += NT) {
                const unsigned r = S[i];
                const unsigned cat = (pidx ? 0u : 2u) + ((r >> kLocalBits) & 1u);
                dst[i] = seg_bits | ((unsigned long long)(r >> 12) << 24) | ((unsigned long long)cat << 22) |
                         (unsigned long long)((unsigned)x0 + (r & (kTilePix - 1)));
            }
        } else {
        // ---- P5: bucket-major, wave-minor exclusive scan; the tile's bucket counts and run offsets
        {
            // (counts and offsets are below 2^16: a (tile, bin) holds at most kCapTile = 15360 records)
            constexpr int NWP = NW / 2;                          // chunk pairs
            const unsigned ncp = (NC + 1u) >> 1;
            unsigned v[NWP];
            unsigned run = 0;
            if (tid < P.NB) {
#pragma unroll
                for (int w = 0; w < NWP; ++w) v[w] = (unsigned)w < ncp ? hist[w * P.NB + tid] : 0u;   // independent reads
#pragma unroll
                for (int w = 0; w < NWP; ++w) {
                    const unsigned lo = v[w] & 0xFFFFu, hi = v[w] >> 16;
                    v[w] = run | ((run + lo) << 16);                // exclusive starts of the even and the odd chunk
                    run += lo + hi;
                }
            }
            unsigned tot;
            const unsigned boff = block_excl_scan<NW>(run, part, &tot);
            if (tid < P.NB) {
                const unsigned b2 = boff | (boff << 16);
#pragma unroll
                for (int w = 0; w < NWP; ++w)
                    if ((unsigned)w < ncp) hist[w * P.NB + tid] = v[w] + b2;
            }
            // the tile's row of the run table: one contiguous, coalesced store
            unsigned short *row = P.roff + ((long long)(b * 9 + c) * P.T + t) * (P.NB + 1);
            if (tid < P.NB) row[tid] = (unsigned short)boff;
            if (tid == 0) row[P.NB] = (unsigned short)N;
        }
        __syncthreads();
        STAMP(6);
        // ---- P6: stable ranks: ballot match-any inside a 64-record batch, running base in LDS ----
        const unsigned lo = wid * L, hi = (lo + L < N) ? lo + L : N;
        if (atomic_order) {                              // the rank IS what the LDS atomic returns (g_lds_order_ok)
            unsigned *myhist = hist + (wid >> 1) * P.NB;
            const unsigned sh = 12 + P.shift, hs = (wid & 1) * 16;
            for (unsigned i0 = lo; i0 < hi; i0 += 64) {
                const unsigned i = i0 + lane;
                if (i < hi) {
                    const unsigned rec = S[i];
                    O[(atomicAdd(&myhist[rec >> sh], 1u << hs) >> hs) & 0xFFFFu] = rec;
                }
            }
        } else {
            for (unsigned i0 = lo; i0 < hi; i0 += 64) {
                const unsigned i = i0 + lane;
                const bool has = i < hi;
                const unsigned rec = has ? S[i] : 0u;
                const unsigned bucket = rec >> (12 + P.shift);
                const unsigned pos = take_slots_packed(has, bucket, P.nb1, &hist[(wid >> 1) * P.NB + bucket], (wid & 1) * 16);
                if (has) O[pos] = rec;
            }
        }
        STAMP(7);
        __syncthreads();
        STAMP(8);
        // ---- P7: the tile's records of this bin leave as one contiguous run ---------------------
        {
            unsigned *dst = P.temp + P.seg_offsets[b * 9 + c] + P.tile_off[((long long)b * P.T + t) * 9 + c];
            for (unsigned i = tid; i < N; i += NT) dst[i] = O[i];
        }
        STAMP(9);
        }
        // ---- advance the relocation recurrence to bin c+2 ---------------------------------------
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (BIDIR) continue;
            nprev[q] = ncur[q];
            ncur[q] = nnext[q];
            dcur[q] = dnext[q];
            if (c + 2 <= 8) {
                const float r = ynn[q] - dnext[q];
                const float cc = ceilf(r - eps);
                dnext[q] = cc - r;
                int ni = (int)cc;
                if (c + 2 == 8) ni += (int)(y9[q] - dnext[q]);   // LDATI.py:106
                nnext[q] = ni;
            }
        }
        // no barrier here: the next bin touches O / hist / S only behind the barriers of its scans
    }
    STAMP(0);
    STAMP_FLUSH(0, 10);
}

// ---------------------------------------------------------------------------------------------
// dense tile pass, round 4: the common call (forward relocation, 'slope' / 'none', no pooling, 32-bit times).
//
// What the SQ counters said about ldati_tile_pass_kernel on the stress chunk (profiles/r04_a_ldati_sq_counters.txt): its
// waves are PARKED half of their cycles (six workgroup barriers and two workgroup scans per bin, one 1024-thread
// workgroup per CU with nobody to run in the gaps), 182 VALU lane-slots per event at 78 % active lanes.  This kernel
// keeps the algorithm (per bin: every timestamp once, stable counting sort by coarse bucket, one coalesced run) and
// changes what that costs:
//   * a wave OWNS 64 * PPT consecutive pixels: classification, the unit compaction (so that every lane of the
//     timestamp code is busy) and the record positions come from 64-lane DPP scans, the unit tables are private to the
//     wave, and the wave's own row of the histogram counts the records it will later rank -- no workgroup scan, no
//     chunk arithmetic per record; the waves meet at four barriers per bin (wave totals, histogram complete, bucket
//     offsets ready, run complete);
//   * the slope parameters {k, b} of a multi-event voxel travel as an 11-bit index into g_slope_tab inside the unit
//     entry (a voxel outside the table is generated by its owner lane in place), so no per-pixel table occupies LDS:
//     records in, records out, NW histogram rows -- two 512-thread workgroups fit a CU and cover each other's barriers;
//   * Philox rounds on v_bitop3_b32 (three-input xor), and for Philox draws the square root and the division of
//     LDATI.py:195 as the compiler's own correctly rounded sequences minus their range scaling, with the refined
//     reciprocal of the voxel's slope shared by its four draws (sqrt_rn_nr / div_rn_nr below: identical results inside
//     the ranges the slope table spans; tests/test_gpu_ldati.py::test_exact_math_helpers checks them exhaustively).
// Everything else (bidirectional, pooled slope, 'random', 64-bit times, a table that is not ready) stays on
// ldati_tile_pass_kernel; both produce the same bytes (test_dense_tile_kernel_equals_per_bin_kernel).
// LDS map (dynamic): S [capA] u32 | O [capA + 2] u32 (the workgroup's work lists alias it) | hist [NW][NB] | misc
// ---------------------------------------------------------------------------------------------
// correctly rounded sqrt for a == 0, a >= 2^-96, negative or NaN a: v_sqrt_f32 (1 ulp) and the two residual tests of the
// compiler's expansion (which additionally rescales a < 2^-96: never the case for b^2 + 2 k u of a table entry)
__device__ __forceinline__ float sqrt_rn_nr(float a) {
    const float r = __builtin_amdgcn_sqrtf(a);
    const float rdn = __uint_as_float(__float_as_uint(r) - 1u), rup = __uint_as_float(__float_as_uint(r) + 1u);
    const float edn = __builtin_fmaf(-rdn, r, a), eup = __builtin_fmaf(-rup, r, a);
    float o = (0.0f >= edn) ? rdn : r;
    o = (0.0f < eup) ? rup : o;
    return o;
}
// the refined reciprocal of the compiler's f32 division (v_rcp_f32 + one Newton step)
__device__ __forceinline__ float rcp_refined(float k) {
    const float r0 = __builtin_amdgcn_rcpf(k);
    const float e = __builtin_fmaf(-k, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
// x / k, correctly rounded, by the compiler's sequence without v_div_scale / v_div_fixup: valid while neither rescales,
// i.e. k and 1/k normal, x zero, NaN or >= 2^-103 in magnitude, x / k normal (LDATI: |k| in [1e2, 1e8], x = sqrt(.) - b)
__device__ __forceinline__ float div_rn_nr(float x, float k, float r1) {
    float q = x * r1;
    float rem = __builtin_fmaf(-k, q, x);
    q = __builtin_fmaf(rem, r1, q);
    rem = __builtin_fmaf(-k, q, x);
    return __builtin_fmaf(rem, r1, q);
}
// Philox4x32-10 as philox4(), the two three-input xors of a round as one v_bitop3_b32 each.  `seed` must be wave-uniform (it
// is the call's seed everywhere): the key schedule lives in scalar registers
__device__ __forceinline__ void philox4_b3(unsigned long long seed, unsigned pixel, unsigned jb, unsigned pc, unsigned frame,
                                           unsigned (&out)[4]) {
    unsigned c0 = pixel, c1 = jb, c2 = pc, c3 = frame;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    // (the key schedule is recomputed here, two scalar adds per round: hoisted out of the caller's loops its twenty values
    // live in SGPRs the kernel does not have -- they were spilled to VGPR lanes and read back with v_readlane every round)
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c1, k0, 0x96);
        const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c3, k1, 0x96);
        c0 = n0; c1 = (unsigned)p1; c2 = n2; c3 = (unsigned)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// slope-table index of a multi-event voxel, or -1 when it lies outside the table (slope_params' own test)
__device__ __forceinline__ int slope_index(int n_l, int n_c, int n_r, int c) {
    const int d = (c == 0 || c == 8) ? 0 : n_r - n_l;
    if (d >= -kSlopeM && d <= kSlopeM && n_c >= 0 && n_c <= kSlopeM && n_l >= 0 && n_r >= 0 && n_l < (1 << 23) && n_r < (1 << 23))
        return (d + kSlopeM) * (kSlopeM + 1) + n_c;
    return -1;
}

// FAST: every run-time switch of the common call is known to be on (Philox draws, the checked fast divisions, the slope table,
// LDS-atomic ranks, 16-byte aligned planes, 'slope'): the switches cost SGPRs the kernel does not have (they were spilled to VGPR
// lanes and read back with v_readlane), and their untaken sides cost code
template <int NW, bool FAST>
__device__ __forceinline__ void dense_tile_body(const LdatiParams &P) {
    constexpr int NT = 64 * NW, PPT = kTilePix / NT, WPX = 64 * PPT;
    static_assert(PPT == 4 || PPT == 2, "a lane owns 2 or 4 consecutive pixels");
    const int t = blockIdx.x, b = blockIdx.y;
    if (P.sparse_cap) {                                 // the sparse tile kernel owns the lightly populated tiles
        const unsigned *tcr = P.tc + ((long long)b * P.T + t) * 9;
        unsigned ntot = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) ntot += tcr[c];
        if (ntot <= (unsigned)P.sparse_cap) return;
    }
    const int pidx = t < P.tpp ? 1 : 0;
    const int x0 = (t < P.tpp ? t : t - P.tpp) * kTilePix;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool atomic_order = FAST || (!P.ballot_ranks && __builtin_amdgcn_readfirstlane(g_lds_order_ok) != 0);
    const bool slot_ok = FAST || (P.fast_slot >= 0 &&
                         __builtin_amdgcn_readfirstlane((int)g_fastdiv[P.fast_slot >= 0 ? P.fast_slot : 0].fps_bits) == (int)__float_as_uint(P.FPS));
    const bool philox = FAST || P.rng_mode == V2CE_RNG_PHILOX;
    const bool fast_k0 = FAST || (slot_ok && philox && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].ok) != 0);
    // (a table that another stream is still filling: every multi-event voxel takes the owner-lane path once)
    const bool tab_ok = FAST || (slot_ok && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].tab_ready) != 0);
    const bool fast_s = FAST || (slot_ok && P.ts32 && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].ok64) != 0);   // single_key's fast form
    const float2 *stab = g_slope_tab[P.fast_slot >= 0 ? P.fast_slot : 0];

    unsigned *S = reinterpret_cast<unsigned *>(tile_smem);
    unsigned *O = S + P.capA;
    unsigned *hist = O + P.capA + 8;                    // [NW][NB]: row r counts the records at positions [r L, (r + 1) L)
    unsigned *part = hist + NW * P.NB;                  // [3][NW] wave totals: events | singles << 16, k == 0 units | k != 0 units << 16, slot mode: events unpacked
    unsigned *spart = part + 3 * NW;                    // [NW + 1] scan partials
    unsigned *bctr = spart + NW + 1;                    // the next batch of the timestamp phase
    unsigned *dsto = bctr + 2;                          // [9][2]: where the tile's run of bin c starts in records[] (loaded once: a
                                                        // global load per bin would stall all sixteen waves for its whole latency)
    unsigned *nbin = dsto + 18;                         // [9] slot mode: the bins' record counts
    unsigned *myhist = hist + wid * P.NB;

    const float *plane0 = P.vox + (long long)(b * 2 + pidx) * 10 * P.HW;
    const unsigned frame = (unsigned)(P.frame_base + b);
    const int lpx0 = wid * WPX + lane * PPT;           // the lane's PPT consecutive local pixels: lpx0 + q
    const int gpx0 = x0 + lpx0;
    const bool vec = FAST || (P.HW & 3) == 0;          // every plane 16-byte aligned: one load per plane and lane
    const float eps = 1e-6f;

    // one plane's voxels of the lane's pixels (zero past the image)
    auto load_plane = [&](int plane, float (&y)[PPT]) {
        const float *src = plane0 + (long long)plane * P.HW + gpx0;
        if (vec) {
            if (PPT == 4) {
                const float4 v = gpx0 < P.HW ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
                y[0] = v.x; y[1] = v.y; y[PPT - 2] = v.z; y[PPT - 1] = v.w;
            } else {
                const float2 v = gpx0 < P.HW ? *reinterpret_cast<const float2 *>(src) : make_float2(0.f, 0.f);
                y[0] = v.x; y[1] = v.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q) y[q] = gpx0 + q < P.HW ? src[q] : 0.0f;
        }
    };

    int nprev[PPT], ncur[PPT], nnext[PPT];
    float dcur[PPT], dnext[PPT];
    {
        float y0[PPT], y1[PPT];
        load_plane(0, y0);
        load_plane(1, y1);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            float r = y0[q] - 0.0f;
            float cc = ceilf(r - eps);
            dcur[q] = cc - r;
            ncur[q] = (int)cc;
            r = y1[q] - dcur[q];
            cc = ceilf(r - eps);
            dnext[q] = cc - r;
            nnext[q] = (int)cc;
            nprev[q] = 0;
        }
    }
    for (int i = lane; i < P.NB; i += 64) myhist[i] = 0;
    if (tid == 0) *bctr = 0;
    if (tid < 9) {
        // slot mode (the kernel is the count pass too: no offsets exist yet): the (tile, bin) run goes to its own slot
        const long long d = P.slot_cap ? (((long long)b * P.T + t) * 9 + tid) * (long long)P.slot_cap
                                       : P.seg_offsets[b * 9 + tid] + (long long)P.tile_off[((long long)b * P.T + t) * 9 + tid];
        dsto[2 * tid] = (unsigned)d;
        dsto[2 * tid + 1] = (unsigned)((unsigned long long)d >> 32);
        if (P.slot_cap) P.tile_abs_w[(long long)(b * 9 + tid) * P.Tp + t] = (unsigned)d;
    }
    int vmax_l = 0;                                     // slot mode: the lane's largest voxel count (the bins' counts wait in nbin)
    __syncthreads();

    STAMP_DECL;
    for (int c = 0; c < 9; ++c) {
        STAMP(0);
        float ynn[PPT], y9[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) ynn[q] = y9[q] = 0.0f;
        if (c + 2 <= 8) load_plane(c + 2, ynn);
        if (c + 2 == 8) load_plane(9, y9);
        // ---- D1: classify; positions in pixel order (lane-major, slot-minor): a running sum per lane, ONE wave scan --
        // cls: 0 nothing, 1 single, 2 multi with k == 0, 3 multi with k != 0 (both from the table), 4 multi outside the table;
        // the slope-table index rides in bits 3..13
        unsigned cls[PPT], aex[PPT], uex[PPT];
        unsigned At = 0, Ut = 0;
        // slot mode: the lane's events once more, UNPACKED.  The packed totals wrap at 2^16; the two-pass path bounds a (tile, bin)
        // by capA before this kernel runs, but as the count pass it must count ANY grid right (an unphysical one behind a dense call
        // that armed the dense hint: a wrapped total would pass the `N > capA` test below and index S past the slot)
        unsigned Ae = 0;
        const bool multi_on = FAST || P.strategy != V2CE_STRATEGY_NONE, edge = c == 0 || c == 8;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int n = ncur[q];                                          // (a pixel past the image holds zeros: no event)
            const bool multi = n >= 2 && multi_on;
            // slope-table entry (slope_index's test in three unsigned compares): |difference| <= 31, count <= 31, neighbours in [0, 2^23)
            const int dd = edge ? 0 : nnext[q] - nprev[q];
            const bool intab = tab_ok && (unsigned)(dd + kSlopeM) <= 2u * kSlopeM && (unsigned)n <= (unsigned)kSlopeM &&
                               (unsigned)(nprev[q] | nnext[q]) < (1u << 23);
            const unsigned si = (unsigned)((dd + kSlopeM) * (kSlopeM + 1) + n);
            const unsigned cl = n == 1 ? 1u : !multi ? 0u : !intab ? 4u : dd == 0 ? 2u : 3u;      // table entries: k == 0 exactly when the difference is
            cls[q] = cl | ((si & 0x7FFu) << 3);
            const unsigned units = (unsigned)(n + 3) >> 2;
            aex[q] = At;
            uex[q] = Ut;
            At += (cl ? (unsigned)n : 0u) + (cl == 1u ? 0x10000u : 0u);
            Ut += cl == 2u ? units : cl == 3u ? units << 16 : 0u;
            Ae += cl ? ((unsigned)n < (1u << 20) ? (unsigned)n : 1u << 20) : 0u;     // (saturating per voxel: the sum of a tile stays below 2^32)
        }
        const unsigned iA = wave_incl_scan(At, lane), iU = wave_incl_scan(Ut, lane);
        const unsigned baseA = iA - At, baseU = iU - Ut;
        const unsigned runA = (unsigned)__builtin_amdgcn_readlane((int)iA, 63), runU = (unsigned)__builtin_amdgcn_readlane((int)iU, 63);
        if (lane == 0) { part[wid] = runA; part[NW + wid] = runU; }      // (events | singles << 16), (k == 0 units | k != 0 units << 16)
        if (P.slot_cap) {                                                   // uniform
            const unsigned runE = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(Ae, lane), 63);
            if (lane == 0) part[2 * NW + wid] = runE;
        }
        __syncthreads();                                 // A: wave totals; O (the previous bin's run) is free again
        STAMP(1);
        // Work lists of the WORKGROUP in O (2 (U1 + U0) + Ns <= N words): k != 0 units | k == 0 units | singles, each in
        // pixel order; the timestamp phase below hands them out in batches of 64 to whichever wave is free, so a lane is
        // idle only in the last batch of a class (per-wave lists left 38 % of the VALU lanes idle: r04_j counters)
        unsigned sbase, N, sS, sU0, sU1, Ns, U0, U1;
        {
            // two packed scans (every total is below 2^16: a (tile, bin) holds at most kCapTile records)
            const unsigned v0 = lane < NW ? part[lane] : 0u, v1 = lane < NW ? part[NW + lane] : 0u;
            const unsigned p0 = wave_incl_scan(v0, lane), p1 = wave_incl_scan(v1, lane);
            const unsigned tA = (unsigned)__builtin_amdgcn_readlane((int)p0, NW - 1), tU = (unsigned)__builtin_amdgcn_readlane((int)p1, NW - 1);
            const unsigned bA = wid ? (unsigned)__builtin_amdgcn_readlane((int)p0, wid - 1) : 0u;
            const unsigned bU = wid ? (unsigned)__builtin_amdgcn_readlane((int)p1, wid - 1) : 0u;
            N = tA & 0xFFFFu; Ns = tA >> 16; U0 = tU & 0xFFFFu; U1 = tU >> 16;
            sbase = bA & 0xFFFFu; sS = bA >> 16; sU0 = bU & 0xFFFFu; sU1 = bU >> 16;
        }
        if (P.slot_cap) {
            // the count pass's outputs; a run beyond the slot (and the LDS) is only counted -- the host sees the largest
            // (tile, bin) count in the statistics and repeats the call on the two-pass path.  The count is the UNPACKED sum:
            // the packed N above is only meaningful once this one says the run fits (capA < 2^16)
            const unsigned N32 = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(lane < NW ? part[2 * NW + lane] : 0u, lane), NW - 1);
            if (tid == 0) nbin[c] = N32;                 // (written out at the end: no pointer or sum lives across the bins)
#pragma unroll
            for (int q = 0; q < PPT; ++q) vmax_l = ncur[q] > vmax_l ? ncur[q] : vmax_l;
            if (N32 > (unsigned)P.capA) {                // uniform
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    nprev[q] = ncur[q];
                    ncur[q] = nnext[q];
                    dcur[q] = dnext[q];
                    if (c + 2 <= 8) {
                        const float r = ynn[q] - dnext[q];
                        const float cc = ceilf(r - eps);
                        dnext[q] = cc - r;
                        int ni = (int)cc;
                        if (c + 2 == 8) ni += (int)(y9[q] - dnext[q]);   // LDATI.py:106
                        nnext[q] = ni;
                    }
                }
                continue;
            }
        }
        // where the tile's run of this bin goes; its records are ranked into O at the same phase modulo four records, so that the
        // copy-out moves whole 16-byte pieces
        const unsigned dshift = dsto[2 * c] & 3u;
        uint2 *UL1 = reinterpret_cast<uint2 *>(O), *UL0 = UL1 + U1;
        unsigned *SLs = O + 2u * (U1 + U0);
        // ranks are taken in NW chunks of L = 2^lgL consecutive positions (one wave each); a record counts in its chunk's row
        int lgL = 6;
        while (((unsigned)NW << lgL) < N) ++lgL;
        const float offt_c = P.offt[c];
        const int kbase_c = (int)P.kbase[c];
        const unsigned pc = (unsigned)(pidx * 9 + c);
        // ---- D2: unit tables (and the rare voxel outside the slope table, generated in place by its owner) ---------
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const unsigned local = (unsigned)(lpx0 + q), pos = sbase + ((baseA + aex[q]) & 0xFFFFu);     // position in S
            const int n = ncur[q];
            const unsigned cl = cls[q] & 7u;
            if (cl == 1u) {
                SLs[sS + ((baseA + aex[q]) >> 16)] = pos | (local << 14);
                S[pos] = __float_as_uint(dcur[q]);               // the single's tendency waits in its record slot
            } else if (cl == 2u || cl == 3u) {
                const unsigned ue = baseU + uex[q];
                uint2 *dstu = cl == 2u ? UL0 + sU0 + (ue & 0xFFFFu) : UL1 + sU1 + (ue >> 16);
                const unsigned units = (unsigned)(n + 3) >> 2, hi = (cls[q] >> 3) << 14;
                dstu[0] = make_uint2(local | (((unsigned)n < 4u ? (unsigned)n : 4u) << 29), pos | hi);      // (most voxels: one or two units)
                if (units > 1u) {
                    const unsigned l1 = (unsigned)n - 4u;
                    dstu[1] = make_uint2(local | (1u << 11) | ((l1 < 4u ? l1 : 4u) << 29), (pos + 4u) | hi);
                    for (unsigned jb = 2; jb < units; ++jb) {
                        const unsigned left = (unsigned)n - 4u * jb;
                        dstu[jb] = make_uint2(local | (jb << 11) | ((left < 4u ? left : 4u) << 29), (pos + 4u * jb) | hi);
                    }
                }
            } else if (cl == 4u) {
                float k, bb;
                slope_params(nprev[q], n, nnext[q], c, P, k, bb);
                const unsigned px = (unsigned)x0 + local;
                for (int j = 0; j < n; ++j) {
                    float u = 0.0f;
                    if (philox) u = philox_uniform(P.seed, px, (unsigned)j, pc, frame);
                    else if (j < P.replay_max_n) u = P.uniforms[(((long long)(b * 2 + pidx) * 9 + c) * P.HW + px) * P.replay_max_n + j];
                    const unsigned key = multi_key(k, bb, u, offt_c, kbase_c, P, fast_k0);
                    const unsigned sp = pos + (unsigned)j;
                    S[sp] = (key << 12) | (1u << kLocalBits) | local;
                    atomicAdd(&hist[__umul24(sp >> lgL, (unsigned)P.NB) + (key >> P.shift)], 1u);
                }
            }
        }
        __syncthreads();                                 // A2: the work lists are complete
        STAMP(2);
        // ---- D3: timestamps, once; batches of 64 list entries, handed out through an LDS counter --------------------
        auto put = [&](unsigned sp, unsigned rec, unsigned key) {
            S[sp] = rec;
            atomicAdd(&hist[__umul24(sp >> lgL, (unsigned)P.NB) + (key >> P.shift)], 1u);
        };
        auto single_batch = [&](unsigned i) {                // (called by whole waves: single_key votes)
            const bool has = i < Ns;
            const unsigned e = has ? SLs[i] : 0u, sp = e & 0x3FFFu;
            const unsigned key = single_key(has, has ? __uint_as_float(S[sp]) : 0.0f, offt_c, P.kbase[c], fast_s, P);
            if (has) put(sp, (key << 12) | (e >> 14), key);
        };
        auto unit_batch = [&](auto mode_c, const uint2 *list, unsigned i, unsigned count) {
            // MODE 0: replayed uniforms (IEEE operations as the compiler expands them); 1 / 2: Philox, k == 0 units with /
            // without the checked fast constant divisions; 3: Philox, k != 0 units (sqrt_rn_nr / div_rn_nr)
            constexpr int MODE = decltype(mode_c)::value;
            if (i < count) {
                const uint2 e = list[i];
                const unsigned local = e.x & (kTilePix - 1), jb = (e.x >> kLocalBits) & 0x3FFFFu, cnt = e.x >> 29;
                const unsigned sp = e.y & 0x3FFFu;
                float2 kb = make_float2(0.0f, 0.0f);
                if (MODE == 0 || MODE == 3) kb = stab[e.y >> 14];
                const unsigned px = (unsigned)x0 + local;
                float u[4];
                unsigned key[4];
                if (MODE == 0 && !philox) {
                    const long long ub = (((long long)(b * 2 + pidx) * 9 + c) * P.HW + px) * P.replay_max_n;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int j = (int)(4u * jb) + s;
                        u[s] = ((unsigned)s < cnt && j < P.replay_max_n) ? P.uniforms[ub + j] : 0.0f;
                    }
                } else {
                    unsigned o[4];
                    philox4_b3(P.seed, px, jb, pc, frame, o);
#pragma unroll
                    for (int s = 0; s < 4; ++s) u[s] = u24(o[s]);
                }
                if (MODE == 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) key[s] = multi_key(kb.x, kb.y, u[s], offt_c, kbase_c, P, fast_k0);
                } else {
                    float tq[4];
                    if (MODE == 3) {
                        const float r1 = rcp_refined(kb.x), bb2 = kb.y * kb.y, k2 = 2.0f * kb.x;
#pragma unroll
                        for (int s = 0; s < 4; ++s) tq[s] = div_rn_nr(-kb.y + sqrt_rn_nr(bb2 + k2 * u[s]), kb.x, r1);
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) tq[s] = MODE == 1 ? k0_time_fast(u[s], P.FPS, P.RFPS, P.R9) : (u[s] / P.FPS) / 9.0f;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        float tt = tq[s] + offt_c;
                        tt = tt * 1e6f;
                        int kk = (int)tt - kbase_c;
                        kk = kk < 0 ? 0 : kk;
                        key[s] = (unsigned)(kk >= P.NK ? P.NK - 1 : kk);
                    }
                }
                const unsigned tag = (1u << kLocalBits) | local;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if ((unsigned)s < cnt) put(sp + s, (key[s] << 12) | tag, key[s]);
            }
        };
        {
            const unsigned nK = (U1 + 63u) >> 6, nZ = (U0 + 63u) >> 6, nS = (Ns + 63u) >> 6, nb = nK + nZ + nS;
            unsigned nxt = 0;
            if (lane == 0) nxt = atomicAdd(bctr, 1u);
            for (;;) {
                const unsigned bid = (unsigned)__builtin_amdgcn_readfirstlane((int)nxt);
                if (bid >= nb) break;
                // far from the end the next batch is reserved now (its LDS round trip runs under this batch); near the end a
                // wave takes its next batch only when it is free, so that nobody sits on a batch an idle wave could run
                const bool early = bid + 2u * NW < nb;
                if (early && lane == 0) nxt = atomicAdd(bctr, 1u);
                if (bid < nK) {                          // the long batches first
                    if (philox) unit_batch(std::integral_constant<int, 3>{}, UL1, bid * 64u + lane, U1);
                    else unit_batch(std::integral_constant<int, 0>{}, UL1, bid * 64u + lane, U1);
                } else if (bid < nK + nZ) {
                    const unsigned i = (bid - nK) * 64u + lane;
                    if (!philox) unit_batch(std::integral_constant<int, 0>{}, UL0, i, U0);
                    else if (fast_k0) unit_batch(std::integral_constant<int, 1>{}, UL0, i, U0);
                    else unit_batch(std::integral_constant<int, 2>{}, UL0, i, U0);
                } else {
                    single_batch((bid - nK - nZ) * 64u + lane);
                }
                if (!early && lane == 0) nxt = atomicAdd(bctr, 1u);
            }
        }
        STAMP(4);
        __syncthreads();                                 // B: every row of the histogram is complete
        STAMP(5);
        // ---- D4: bucket-major, wave-minor exclusive scan; the tile's row of the run table ------------------------
        {
            // thread 2 b + h takes rows [h NW/2, (h + 1) NW/2) of bucket b when the workgroup has 2 NB threads (the workgroup
            // scan then runs over (bucket, half) in exactly the order the offsets need), else one thread takes the bucket
            const bool two = 2 * P.NB <= NT;                  // uniform
            constexpr int RH = NW / 2;
            const int bkt = two ? tid >> 1 : tid, r0 = two ? (tid & 1) * RH : 0, nr = two ? RH : NW;
            unsigned v[NW];
            unsigned run = 0;
            if (bkt < P.NB) {
#pragma unroll
                for (int w = 0; w < NW; ++w) v[w] = w < nr ? hist[__umul24((unsigned)(r0 + w), (unsigned)P.NB) + bkt] : 0u;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const unsigned x = v[w];
                    v[w] = run;
                    run += x;
                }
            }
            unsigned tot;
            const unsigned boff = block_excl_scan<NW>(run, spart, &tot);
            if (bkt < P.NB) {
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    if (w < nr) hist[__umul24((unsigned)(r0 + w), (unsigned)P.NB) + bkt] = v[w] + boff + dshift;
            }
            unsigned short *row = P.roff + ((long long)(b * 9 + c) * P.T + t) * (P.NB + 1);
            if (bkt < P.NB && r0 == 0) row[bkt] = (unsigned short)boff;
            if (tid == 0) row[P.NB] = (unsigned short)N;
        }
        __syncthreads();                                 // C: bucket offsets per wave
        STAMP(6);
        // ---- D5: stable ranks: the wave walks its records in pixel order -------------------------------------------
        {
            const unsigned sh = 12 + P.shift;
            const unsigned lo = (unsigned)wid << lgL, hi = min(N, lo + (1u << lgL));
            if (atomic_order) {                          // the rank IS what the LDS atomic returns (g_lds_order_ok)
                // (four read -> atomic -> write chains in flight; a wave's LDS atomics execute in program order, so the ranks
                // are still taken in position order)
                unsigned i = lo + lane, i0 = lo;
                for (; i0 + 256u <= hi; i0 += 256u, i += 256u) {      // (a wave-uniform bound: every lane takes the same trips)
                    unsigned rec[4], at[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) rec[j] = S[i + 64u * j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) at[j] = atomicAdd(&myhist[rec[j] >> sh], 1u);
#pragma unroll
                    for (int j = 0; j < 4; ++j) O[at[j]] = rec[j];
                }
                for (; i < hi; i += 64u) {
                    const unsigned rec = S[i];
                    O[atomicAdd(&myhist[rec >> sh], 1u)] = rec;
                }
            } else {
                for (unsigned i0 = lo; i0 < hi; i0 += 64) {
                    const unsigned i = i0 + lane;
                    const bool has = i < hi;
                    const unsigned rec = has ? S[i] : 0u;
                    const unsigned bucket = rec >> sh;
                    const unsigned pos = take_slots(has, bucket, P.nb1, &myhist[bucket]);
                    if (has) O[pos] = rec;
                }
            }
        }
        STAMP(7);
        __syncthreads();                                 // D: the run is complete in O
        STAMP(8);
        for (int i = lane; i < P.NB; i += 64) myhist[i] = 0;    // for the next bin (first touched behind its barrier A)
        if (tid == 0) *bctr = 0;
        {
            // O[dshift + i] -> dst[i]: 16-byte pieces where the piece lies inside the run, single records at its two ends
            const long long dst0 = (long long)(((unsigned long long)dsto[2 * c + 1] << 32) | dsto[2 * c]);
            unsigned *dstq = P.temp + (dst0 - (long long)dshift);             // 16-byte aligned (temp is; dst0 - dshift is a multiple of 4 records)
            const unsigned nq = (dshift + N + 3u) >> 2;
            for (unsigned q = tid; q < nq; q += NT) {
                const uint4 v = reinterpret_cast<const uint4 *>(O)[q];
                if (4u * q >= dshift && 4u * q + 4u <= dshift + N) {
                    reinterpret_cast<uint4 *>(dstq)[q] = v;
                } else {
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4u * q + j >= dshift && 4u * q + j < dshift + N) dstq[4u * q + j] = w[j];
                }
            }
        }
        STAMP(9);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            nprev[q] = ncur[q];
            ncur[q] = nnext[q];
            dcur[q] = dnext[q];
            if (c + 2 <= 8) {
                const float r = ynn[q] - dnext[q];
                const float cc = ceilf(r - eps);
                dnext[q] = cc - r;
                int ni = (int)cc;
                if (c + 2 == 8) ni += (int)(y9[q] - dnext[q]);   // LDATI.py:106
                nnext[q] = ni;
            }
        }
    }
    if (P.slot_cap) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int m = __shfl_xor(vmax_l, o);
            vmax_l = m > vmax_l ? m : vmax_l;
        }
        // (plain reads first: after the first few tiles these maxima rarely grow, and same-address atomics serialise)
        if (lane == 0 && vmax_l > 0 && (unsigned long long)vmax_l > *reinterpret_cast<volatile unsigned long long *>(&P.stats_w[0]))
            atomicMax(&P.stats_w[0], (unsigned long long)vmax_l);
        __syncthreads();
        if (tid < 9) P.tc_w[((long long)b * P.T + t) * 9 + tid] = nbin[tid];
        if (tid == 0) {
            unsigned tile_total = 0;
#pragma unroll
            for (int c = 0; c < 9; ++c) tile_total += nbin[c];
            if (tile_total > 0 && (unsigned long long)tile_total > *reinterpret_cast<volatile unsigned long long *>(&P.stats_w[4]))
                atomicMax(&P.stats_w[4], (unsigned long long)tile_total);
        }
    }
    STAMP(0);
    STAMP_FLUSH(0, 10);
}

// ---------------------------------------------------------------------------------------------
// Round 6: TWO bins of a dense tile per pass (`dense_pair_body`, the common call only: dense_tile_body<NW, true>'s switches).
// What the per-bin kernel pays per (tile, BIN) whatever the bin holds -- the classification scans, six workgroup barriers, the
// scan of the (row, bucket) histogram, the list and batch hand-out, the copy-out prologue: 4.9 us of a workgroup's life per
// bin (the round-5 density sweep, tools/ldati_density_probe.py: the tile pass of 24 frame-pairs takes 363 us + 2.25 us per million
// events, i.e. more than half of the stress chunk's 650 us does not depend on the events) -- is paid once per PASS; a pass holds bins c and c + 1 when their records fit the LDS together (capP), else bin c alone.  The two
// bins' records share S (bin c's positions first), the work lists (an entry carries its bin), the batches of the timestamp
// phase, one histogram whose cell index is (bin in pass, bucket), one scan, one rank phase and one copy-out loop per run.
// To make room for two bins' cells the histogram counts in 16 bits, two cells per word: a cell never exceeds the pass's record
// count (<= capP < 2^16), offsets included, so a half never carries into its neighbour, and the LDS atomic still returns the
// stable rank (lanes of one wave-instruction that hit the same WORD are served in lane order whichever half they add to).
// LDS map (dynamic): S [capP] | O [capP + 16] (lists alias it) | hist [NW][2^(12 - shift)] words | wave totals, partials, ... | voxel 9 [2048]
// Requires 12-bit keys (NK <= 4096: the bin of a record rides in bit 24 of its S word) and 2^(12 - shift) <= threads.
// ---------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void dense_pair_body(const LdatiParams &P) {
    constexpr int NT = 64 * NW, PPT = kTilePix / NT, WPX = 64 * PPT;
    static_assert(PPT == 4 || PPT == 2, "a lane owns 2 or 4 consecutive pixels");
    const int t = blockIdx.x, b = blockIdx.y;
    if (P.sparse_cap) {                                 // the sparse tile kernel owns the lightly populated tiles
        const unsigned *tcr = P.tc + ((long long)b * P.T + t) * 9;
        unsigned ntot = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) ntot += tcr[c];
        if (ntot <= (unsigned)P.sparse_cap) return;
    }
    const int pidx = t < P.tpp ? 1 : 0;
    const int x0 = (t < P.tpp ? t : t - P.tpp) * kTilePix;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float2 *stab = g_slope_tab[P.fast_slot];
    const unsigned hsl = 12u - (unsigned)P.shift;       // log2 of a bin's cell stride = log2 of the words of a histogram row
    const unsigned HS = 1u << hsl;

    unsigned *S = reinterpret_cast<unsigned *>(tile_smem);
    unsigned *O = S + P.capP;
    unsigned *hist = O + P.capP + 16;                   // [NW][HS] words, two 16-bit cells each: cell g = bin in pass << hsl | bucket
    unsigned *part = hist + NW * HS;                    // [6][NW] wave totals (see D1)
    unsigned *spart = part + 6 * NW;                    // [NW + 1] scan partials
    unsigned *bctr = spart + NW + 1;                    // the next batch of the timestamp phase
    unsigned *dsto = bctr + 2;                          // [9][2] where the tile's run of bin c starts in records[]
    unsigned *nbin = dsto + 18;                         // [9] slot mode: the bins' record counts
    float *y9s = reinterpret_cast<float *>(nbin + 10);  // [PPT][NT] voxel 9 of the tile's pixels
    unsigned *myhist = hist + wid * HS;

    const float *plane0 = P.vox + (long long)(b * 2 + pidx) * 10 * P.HW;
    const unsigned frame = (unsigned)(P.frame_base + b);
    const int lpx0 = wid * WPX + lane * PPT;
    const int gpx0 = x0 + lpx0;
    const float eps = 1e-6f;

    auto load_plane = [&](int plane, float (&y)[PPT]) {          // (16-byte aligned planes: the common call)
        const float *src = plane0 + (long long)plane * P.HW + gpx0;
        if (PPT == 4) {
            const float4 v = gpx0 < P.HW ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            y[0] = v.x; y[1] = v.y; y[PPT - 2] = v.z; y[PPT - 1] = v.w;
        } else {
            const float2 v = gpx0 < P.HW ? *reinterpret_cast<const float2 *>(src) : make_float2(0.f, 0.f);
            y[0] = v.x; y[1] = v.y;
        }
    };

    // the relocation recurrence as a window over the bins: counts of bins c - 1 .. c + 2, tendencies of c .. c + 2.  The planes the
    // window needs to move on (c + 3, c + 4) are loaded at the head of every pass and consumed at its end -- each into its own
    // registers (a register move of a load in flight would wait for it); voxel 9, needed once beside plane 8, waits in LDS
    int nm1[PPT], n0[PPT], n1[PPT], n2[PPT];
    float d0[PPT], d1[PPT], d2[PPT], ya[PPT], yb[PPT];
    {
        float p0[PPT], p1[PPT], p2[PPT], p9[PPT];
        load_plane(0, p0);
        load_plane(1, p1);
        load_plane(2, p2);
        load_plane(9, p9);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            float r = p0[q] - 0.0f;
            float cc = ceilf(r - eps);
            d0[q] = cc - r;
            n0[q] = (int)cc;
            r = p1[q] - d0[q];
            cc = ceilf(r - eps);
            d1[q] = cc - r;
            n1[q] = (int)cc;
            r = p2[q] - d1[q];
            cc = ceilf(r - eps);
            d2[q] = cc - r;
            n2[q] = (int)cc;
            nm1[q] = 0;
            ya[q] = yb[q] = 0.0f;
            y9s[q * NT + tid] = p9[q];                  // (read back by the same thread: no barrier)
        }
    }
    auto advance = [&](int pn, const float (&yp)[PPT]) {  // the window's first bin moves on by one; yp = plane pn = its first bin + 3
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            nm1[q] = n0[q]; n0[q] = n1[q]; n1[q] = n2[q];
            d0[q] = d1[q]; d1[q] = d2[q];
            int ni = 0;
            if (pn <= 8) {
                const float r = yp[q] - d2[q];
                const float cc = ceilf(r - eps);
                d2[q] = cc - r;
                ni = (int)cc;
                if (pn == 8) ni += (int)(y9s[q * NT + tid] - d2[q]);        // LDATI.py:106
            }
            n2[q] = ni;
        }
    };

    for (unsigned i = tid; i < NW * HS; i += NT) hist[i] = 0;
    if (tid == 0) *bctr = 0;
    if (tid < 9) {
        const long long d = P.slot_cap ? (((long long)b * P.T + t) * 9 + tid) * (long long)P.slot_cap
                                       : P.seg_offsets[b * 9 + tid] + (long long)P.tile_off[((long long)b * P.T + t) * 9 + tid];
        dsto[2 * tid] = (unsigned)d;
        dsto[2 * tid + 1] = (unsigned)((unsigned long long)d >> 32);
        if (P.slot_cap) P.tile_abs_w[(long long)(b * 9 + tid) * P.Tp + t] = (unsigned)d;
    }
    int vmax_l = 0;
    __syncthreads();

    // one voxel's class (dense_tile_body's D1): 0 nothing, 1 single, 2 / 3 multi from the slope table with k == 0 / k != 0, 4 multi
    // outside the table; the table index in bits 3..13.  The pass computes it twice -- for the sums in front of barrier A and again
    // behind it for the lists -- instead of keeping six values per pixel alive across the barrier (the kernel has no registers left)
    auto voxel_class = [&](int np, int n, int nn, bool edge) -> unsigned {
        const int dd = edge ? 0 : nn - np;
        const bool intab = (unsigned)(dd + kSlopeM) <= 2u * kSlopeM && (unsigned)n <= (unsigned)kSlopeM && (unsigned)(np | nn) < (1u << 23);
        const unsigned si = (unsigned)((dd + kSlopeM) * (kSlopeM + 1) + n);
        const unsigned cl = n == 1 ? 1u : n < 2 ? 0u : !intab ? 4u : dd == 0 ? 2u : 3u;
        return cl | ((si & 0x7FFu) << 3);
    };
    // a voxel's contribution to the lane's running sums: At = events | singles << 16, Ut = k == 0 units | k != 0 units << 16
    auto voxel_sums = [&](unsigned cls, int n, unsigned &At, unsigned &Ut) {
        const unsigned cl = cls & 7u, units = (unsigned)(n + 3) >> 2;
        At += (cl ? (unsigned)n : 0u) + (cl == 1u ? 0x10000u : 0u);
        Ut += cl == 2u ? units : cl == 3u ? units << 16 : 0u;
    };
    auto classify = [&](const int (&np)[PPT], const int (&nc)[PPT], const int (&nn)[PPT], bool edge, unsigned &At, unsigned &Ut, unsigned &Ae) {
        At = Ut = Ae = 0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const unsigned cls = voxel_class(np[q], nc[q], nn[q], edge);
            voxel_sums(cls, nc[q], At, Ut);
            Ae += (cls & 7u) ? ((unsigned)nc[q] < (1u << 20) ? (unsigned)nc[q] : 1u << 20) : 0u;
        }
    };
    // totals and this wave's exclusive prefix of NW per-wave values written to LDS in front of a barrier
    auto wave_totals = [&](const unsigned *src, unsigned &total, unsigned &base) {
        const unsigned p = wave_incl_scan(lane < NW ? src[lane] : 0u, lane);
        total = (unsigned)__builtin_amdgcn_readlane((int)p, NW - 1);
        base = wid ? (unsigned)__builtin_amdgcn_readlane((int)p, wid - 1) : 0u;
    };

    STAMP_DECL;
    for (int c = 0; c < 9;) {
        STAMP(0);
        const bool hasB = c + 1 < 9;
        if (c + 3 <= 8) load_plane(c + 3, ya);           // consumed when the window moves on, behind the pass
        if (c + 4 <= 8) load_plane(c + 4, yb);
        // ---- D1: classes and positions of bins c (A) and c + 1 (B) -------------------------------------------------
        unsigned AtA, UtA, AeA, AtB, UtB, AeB;
        classify(nm1, n0, n1, c == 0 || c == 8, AtA, UtA, AeA);
        classify(n0, n1, n2, c + 1 == 8, AtB, UtB, AeB);
        if (!hasB) { AtB = UtB = AeB = 0; }
        const unsigned iA = wave_incl_scan(AtA, lane), iUA = wave_incl_scan(UtA, lane);
        const unsigned iB = wave_incl_scan(AtB, lane), iUB = wave_incl_scan(UtB, lane);
        const unsigned rA = (unsigned)__builtin_amdgcn_readlane((int)iA, 63), rUA = (unsigned)__builtin_amdgcn_readlane((int)iUA, 63);
        const unsigned rB = (unsigned)__builtin_amdgcn_readlane((int)iB, 63), rUB = (unsigned)__builtin_amdgcn_readlane((int)iUB, 63);
        if (lane == 0) { part[wid] = rA; part[NW + wid] = rUA; part[2 * NW + wid] = rB; part[3 * NW + wid] = rUB; }
        if (P.slot_cap) {                                // (uniform) the unpacked event sums: see dense_tile_body
            const unsigned eA = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(AeA, lane), 63);
            const unsigned eB = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(AeB, lane), 63);
            if (lane == 0) { part[4 * NW + wid] = eA; part[5 * NW + wid] = eB; }
        }
        __syncthreads();                                 // A: wave totals; O (the previous pass's runs) is free again
        STAMP(1);
        unsigned tA, tUA, tB, tUB, bsA, bsUA, bsB, bsUB;
        wave_totals(part, tA, bsA);
        wave_totals(part + NW, tUA, bsUA);
        wave_totals(part + 2 * NW, tB, bsB);
        wave_totals(part + 3 * NW, tUB, bsUB);
        bool doB = hasB;
        if (P.slot_cap) {
            unsigned N32A, N32B, dummy;
            wave_totals(part + 4 * NW, N32A, dummy);
            wave_totals(part + 5 * NW, N32B, dummy);
            if (tid == 0) {
                nbin[c] = N32A;
                if (hasB) nbin[c + 1] = N32B;
            }
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                vmax_l = n0[q] > vmax_l ? n0[q] : vmax_l;
                if (hasB) vmax_l = n1[q] > vmax_l ? n1[q] : vmax_l;
            }
            if (N32A > (unsigned)P.capA) {               // uniform: a run beyond its slot is only counted
                advance(c + 3, ya);
                ++c;
                continue;
            }
            if (N32B > (unsigned)P.capA) doB = false;
        }
        const unsigned NA = tA & 0xFFFFu;
        if (doB && NA + (tB & 0xFFFFu) > (unsigned)P.capP) doB = false;
        if (!doB) { tB = tUB = 0; }
        const unsigned NB_ = tB & 0xFFFFu, Ntot = NA + NB_;
        const unsigned NsA = tA >> 16, NsB = tB >> 16, Ns = NsA + NsB;
        const unsigned U0 = (tUA & 0xFFFFu) + (tUB & 0xFFFFu), U1 = (tUA >> 16) + (tUB >> 16);
        // the two runs in O: A at its destination's phase modulo four records, B behind it at its own
        const unsigned dshA = dsto[2 * c] & 3u, dshB = doB ? dsto[2 * (c + 1)] & 3u : 0u;
        const unsigned obB = ((dshA + NA + 3u) & ~3u) + dshB;
        uint2 *UL1 = reinterpret_cast<uint2 *>(O), *UL0 = UL1 + U1;
        unsigned *SLs = O + 2u * (U1 + U0);
        int lgL = 6;
        while (((unsigned)NW << lgL) < Ntot) ++lgL;
        const float offtA = P.offt[c], offtB = P.offt[hasB ? c + 1 : c];
        const int kbA = (int)P.kbase[c], kbB = (int)P.kbase[hasB ? c + 1 : c];
        const unsigned pcA = (unsigned)(pidx * 9 + c);
        auto put = [&](unsigned sp, unsigned rec, unsigned g) {
            S[sp] = rec;
            atomicAdd(&hist[((sp >> lgL) << hsl) + (g >> 1)], 1u << ((g & 1u) << 4));
        };
        // ---- D2: work lists of the pass: k != 0 units | k == 0 units | singles, each wave's entries of A in front of its entries of B
        {
            // this wave's first entries: the waves in front (both bins), then for B this wave's entries of A
            const unsigned wU = (bsUA & 0xFFFFu) + (doB ? bsUB & 0xFFFFu : 0u), wK = (bsUA >> 16) + (doB ? bsUB >> 16 : 0u);
            const unsigned wS = (bsA >> 16) + (doB ? bsB >> 16 : 0u);
            auto lists = [&](unsigned bsel, const int (&np)[PPT], const int (&nc)[PPT], const int (&nn)[PPT], const float (&dc)[PPT], bool edge,
                             unsigned lanA, unsigned lanU, unsigned pos0, unsigned u0at, unsigned u1at, unsigned sat, int cb) {
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    int n = nc[q];
                    asm volatile("" : "+v"(n));           // (opaque: the class is recomputed here, not carried across the barrier)
                    const unsigned cls = voxel_class(np[q], n, nn[q], edge);
                    const unsigned local = (unsigned)(lpx0 + q), pos = pos0 + (lanA & 0xFFFFu);
                    const unsigned cl = cls & 7u;
                    if (cl == 1u) {
                        SLs[sat + (lanA >> 16)] = pos | (local << 14) | (bsel << 25);
                        S[pos] = __float_as_uint(dc[q]);                 // the single's tendency waits in its record slot
                    } else if (cl == 2u || cl == 3u) {
                        const unsigned ue = lanU;
                        uint2 *dstu = cl == 2u ? UL0 + u0at + (ue & 0xFFFFu) : UL1 + u1at + (ue >> 16);
                        const unsigned units = (unsigned)(n + 3) >> 2, hi = (cls >> 3) << 14, lb = local | (bsel << 28);
                        dstu[0] = make_uint2(lb | (((unsigned)n < 4u ? (unsigned)n : 4u) << 29), pos | hi);
                        for (unsigned jb = 1; jb < units; ++jb) {
                            const unsigned left = (unsigned)n - 4u * jb;
                            dstu[jb] = make_uint2(lb | (jb << 11) | ((left < 4u ? left : 4u) << 29), (pos + 4u * jb) | hi);
                        }
                    } else if (cl == 4u) {               // outside the slope table: generated in place by its owner
                        float k, bb;
                        slope_params(np[q], n, nn[q], cb, P, k, bb);
                        const unsigned px = (unsigned)x0 + local;
                        for (int j = 0; j < n; ++j) {
                            const float u = philox_uniform(P.seed, px, (unsigned)j, (unsigned)(pidx * 9 + cb), frame);
                            const unsigned key = multi_key(k, bb, u, P.offt[cb], (int)P.kbase[cb], P, true);
                            put(pos + (unsigned)j, (bsel << 24) | (key << 12) | (1u << kLocalBits) | local, (bsel << hsl) | (key >> P.shift));
                        }
                    }
                    voxel_sums(cls, n, lanA, lanU);      // the lane's running sums: the next pixel's positions
                }
            };
            lists(0u, nm1, n0, n1, d0, c == 0 || c == 8, iA - AtA, iUA - UtA, bsA & 0xFFFFu, wU, wK, wS, c);
            if (doB)
                lists(1u, n0, n1, n2, d1, c + 1 == 8, iB - AtB, iUB - UtB, NA + (bsB & 0xFFFFu), wU + (rUA & 0xFFFFu), wK + (rUA >> 16),
                      wS + (rA >> 16), c + 1);
        }
        __syncthreads();                                 // A2: the work lists are complete
        STAMP(2);
        // ---- D3: timestamps, once; batches of 64 list entries handed out through an LDS counter ----------------------
        auto single_batch = [&](unsigned i) {            // (called by whole waves: single_key votes)
            const bool has = i < Ns;
            const unsigned e = has ? SLs[i] : 0u, sp = e & 0x3FFFu, bsel = e >> 25;
            const unsigned key = single_key(has, has ? __uint_as_float(S[sp]) : 0.0f, bsel ? offtB : offtA, (long long)(bsel ? kbB : kbA), true, P);
            if (has) put(sp, (bsel << 24) | (key << 12) | ((e >> 14) & (kTilePix - 1)), (bsel << hsl) | (key >> P.shift));
        };
        auto unit_batch = [&](auto mode_c, const uint2 *list, unsigned i, unsigned count) {
            constexpr int MODE = decltype(mode_c)::value;        // 1: k == 0 units (checked fast constant divisions), 3: k != 0 units
            if (i < count) {
                const uint2 e = list[i];
                const unsigned local = e.x & (kTilePix - 1), jb = (e.x >> kLocalBits) & 0x1FFFFu, bsel = (e.x >> 28) & 1u, cnt = e.x >> 29;
                const unsigned sp = e.y & 0x3FFFu;
                float2 kb = make_float2(0.0f, 0.0f);
                if (MODE == 3) kb = stab[e.y >> 14];
                const unsigned px = (unsigned)x0 + local;
                const float offt_c = bsel ? offtB : offtA;
                const int kbase_c = bsel ? kbB : kbA;
                unsigned o[4];
                philox4_b3(P.seed, px, jb, pcA + bsel, frame, o);
                float tq[4];
                if (MODE == 3) {
                    const float r1 = rcp_refined(kb.x), bb2 = kb.y * kb.y, k2 = 2.0f * kb.x;
#pragma unroll
                    for (int s = 0; s < 4; ++s) tq[s] = div_rn_nr(-kb.y + sqrt_rn_nr(bb2 + k2 * u24(o[s])), kb.x, r1);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) tq[s] = k0_time_fast(u24(o[s]), P.FPS, P.RFPS, P.R9);
                }
                const unsigned tag = (bsel << 24) | (1u << kLocalBits) | local, gsel = bsel << hsl;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float tt = tq[s] + offt_c;
                    tt = tt * 1e6f;
                    int kk = (int)tt - kbase_c;
                    kk = kk < 0 ? 0 : kk;
                    const unsigned key = (unsigned)(kk >= P.NK ? P.NK - 1 : kk);
                    if ((unsigned)s < cnt) put(sp + s, (key << 12) | tag, gsel | (key >> P.shift));
                }
            }
        };
        {
            const unsigned nK = (U1 + 63u) >> 6, nZ = (U0 + 63u) >> 6, nS = (Ns + 63u) >> 6, nb = nK + nZ + nS;
            unsigned nxt = 0;
            if (lane == 0) nxt = atomicAdd(bctr, 1u);
            for (;;) {
                const unsigned bid = (unsigned)__builtin_amdgcn_readfirstlane((int)nxt);
                if (bid >= nb) break;
                const bool early = bid + 2u * NW < nb;
                if (early && lane == 0) nxt = atomicAdd(bctr, 1u);
                if (bid < nK) unit_batch(std::integral_constant<int, 3>{}, UL1, bid * 64u + lane, U1);
                else if (bid < nK + nZ) unit_batch(std::integral_constant<int, 1>{}, UL0, (bid - nK) * 64u + lane, U0);
                else single_batch((bid - nK - nZ) * 64u + lane);
                if (!early && lane == 0) nxt = atomicAdd(bctr, 1u);
            }
        }
        STAMP(4);
        __syncthreads();                                 // B: every row of the histogram is complete
        STAMP(5);
        // ---- D4: cell-major, row-minor exclusive scan; the tile's rows of the run table -----------------------------
        {
            // thread = one word of the rows = two neighbouring cells of one bin, all NW rows (HS <= NT: the host's condition)
            unsigned v[NW];
            unsigned run0 = 0, run1 = 0;
            const bool mine = (unsigned)tid < HS;
            if (mine) {
#pragma unroll
                for (int w = 0; w < NW; ++w) v[w] = hist[(unsigned)w * HS + tid];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const unsigned c0 = v[w] & 0xFFFFu, c1 = v[w] >> 16;
                    v[w] = run0 | (run1 << 16);
                    run0 += c0;
                    run1 += c1;
                }
            }
            unsigned tot;
            const unsigned boff = block_excl_scan<NW>(run0 + run1, spart, &tot);
            if (mine) {
                const unsigned bsel = (unsigned)tid >> (hsl - 1u);           // cells [0, HS) are bin A's, [HS, 2 HS) bin B's
                const unsigned adj = bsel ? obB - NA : dshA;                  // prefix inside the pass -> index in O
                const unsigned packed = (boff + adj) | ((boff + run0 + adj) << 16);
#pragma unroll
                for (int w = 0; w < NW; ++w) hist[(unsigned)w * HS + tid] = v[w] + packed;
                const unsigned bkt = (2u * (unsigned)tid) & (HS - 1u);
                const int cb = c + (int)bsel;
                if (bsel == 0u || doB) {
                    unsigned short *row = P.roff + ((long long)(b * 9 + cb) * P.T + t) * (P.NB + 1);
                    const unsigned rb = boff - (bsel ? NA : 0u);
                    if (bkt < (unsigned)P.NB) row[bkt] = (unsigned short)rb;
                    if (bkt + 1u < (unsigned)P.NB) row[bkt + 1u] = (unsigned short)(rb + run0);
                }
            }
            if (tid == 0) {
                P.roff[((long long)(b * 9 + c) * P.T + t) * (P.NB + 1) + P.NB] = (unsigned short)NA;
                if (doB) P.roff[((long long)(b * 9 + c + 1) * P.T + t) * (P.NB + 1) + P.NB] = (unsigned short)NB_;
            }
        }
        __syncthreads();                                 // C: cell offsets per row
        STAMP(6);
        // ---- D5: stable ranks: the wave walks its records in position order; the rank is what the LDS atomic returns ------
        {
            const unsigned sh = 12u + (unsigned)P.shift;
            const unsigned lo = (unsigned)wid << lgL, hi = min(Ntot, lo + (1u << lgL));
            unsigned i = lo + lane, i0 = lo;
            for (; i0 + 256u <= hi; i0 += 256u, i += 256u) {
                unsigned rec[4], at[4], hs[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) rec[j] = S[i + 64u * j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned g = rec[j] >> sh;
                    hs[j] = (g & 1u) << 4;
                    at[j] = atomicAdd(&myhist[g >> 1], 1u << hs[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) O[(at[j] >> hs[j]) & 0xFFFFu] = rec[j] & 0xFFFFFFu;
            }
            for (; i < hi; i += 64u) {
                const unsigned rec = S[i], g = rec >> sh, hs = (g & 1u) << 4;
                O[(atomicAdd(&myhist[g >> 1], 1u << hs) >> hs) & 0xFFFFu] = rec & 0xFFFFFFu;
            }
        }
        STAMP(7);
        __syncthreads();                                 // D: the runs are complete in O
        STAMP(8);
        for (unsigned i = tid; i < NW * HS; i += NT) hist[i] = 0;      // for the next pass (first touched behind its barrier A)
        if (tid == 0) *bctr = 0;
        {
            // O[ob + i] -> dst[i]: 16-byte pieces where the piece lies inside the run, single records at its two ends
            auto copy_run = [&](int cb, unsigned ob, unsigned dsh, unsigned N) {
                const long long dst0 = (long long)(((unsigned long long)dsto[2 * cb + 1] << 32) | dsto[2 * cb]);
                unsigned *dstq = P.temp + (dst0 - (long long)dsh);            // 16-byte aligned
                const uint4 *src = reinterpret_cast<const uint4 *>(O + (ob - dsh));
                const unsigned nq = (dsh + N + 3u) >> 2;
                for (unsigned q = tid; q < nq; q += NT) {
                    const uint4 v = src[q];
                    if (4u * q >= dsh && 4u * q + 4u <= dsh + N) {
                        reinterpret_cast<uint4 *>(dstq)[q] = v;
                    } else {
                        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4u * q + j >= dsh && 4u * q + j < dsh + N) dstq[4u * q + j] = w[j];
                    }
                }
            };
            copy_run(c, dshA, dshA, NA);
            if (doB) copy_run(c + 1, obB, dshB, NB_);
        }
        STAMP(9);
        advance(c + 3, ya);
        if (doB) advance(c + 4, yb);
        c += doB ? 2 : 1;
    }
    if (P.slot_cap) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int m = __shfl_xor(vmax_l, o);
            vmax_l = m > vmax_l ? m : vmax_l;
        }
        if (lane == 0 && vmax_l > 0 && (unsigned long long)vmax_l > *reinterpret_cast<volatile unsigned long long *>(&P.stats_w[0]))
            atomicMax(&P.stats_w[0], (unsigned long long)vmax_l);
        __syncthreads();
        if (tid < 9) P.tc_w[((long long)b * P.T + t) * 9 + tid] = nbin[tid];
        if (tid == 0) {
            unsigned tile_total = 0;
#pragma unroll
            for (int c = 0; c < 9; ++c) tile_total += nbin[c];
            if (tile_total > 0 && (unsigned long long)tile_total > *reinterpret_cast<volatile unsigned long long *>(&P.stats_w[4]))
                atomicMax(&P.stats_w[4], (unsigned long long)tile_total);
        }
    }
    STAMP(0);
    STAMP_FLUSH(0, 10);
}

// ---------------------------------------------------------------------------------------------
// Round 6, second form: ONE pass over the tile for the classification of all nine bins (`dense_onepass_body`).
// The pair passes above showed where the per-bin kernel's fixed cost sits: not in latency that a second bin could share, but in
// the classification (D1) and the work lists (D2) themselves -- every bin re-runs ~200 wave-instructions on all sixteen waves
// for two voxels per lane, 27 % of the kernel (profiles/r06_a_ldati_pair_stamps_stress.txt).  The sparse kernel does the same
// work for all nine bins of four pixels in one sweep at an eighth of that per voxel.  So:
//   pre-phase (once per tile): ten plane loads, the relocation recurrence, the classes of all nine bins in registers; per bin three
//       packed wave scans (events | singles, k == 0 units | k != 0 units, events unpacked | voxels outside the slope table); ONE
//       exchange of the wave totals; wave 0 turns them into every wave's bases and the PASS PLAN (bins paired where their records
//       fit the LDS together, as in dense_pair_body); then every lane writes its voxels' list entries -- 8 bytes per unit of four
//       draws or single event, 16 per voxel outside the table -- to the tile's scratch in GLOBAL memory (the lists of nine bins
//       are 200 KB on the stress chunk: they do not fit beside the records), grouped by (pass, class), positions already final;
//   per pass: timestamps straight from the lists (a batch's 512 bytes are requested one batch ahead), then the pair body's
//       histogram scan, ranks and copy-out.  No classification, no list building, no window over the planes inside the loop, and
//       four workgroup barriers per pass instead of six per bin: a wave clears its own histogram row behind its rank loop and the
//       batch counter alternates between two words, so nothing separates a pass's copy-out from the next pass's timestamps.
// Requirements as dense_pair_body's (12-bit keys, the common call); scratch: P.lists, P.list_stride words per tile.
// ---------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void dense_onepass_body(const LdatiParams &P) {
    constexpr int NT = 64 * NW, PPT = kTilePix / NT, WPX = 64 * PPT;
    static_assert(PPT == 4 || PPT == 2, "a lane owns 2 or 4 consecutive pixels");
    const int t = blockIdx.x, b = blockIdx.y;
    if (P.sparse_cap) {                                 // the sparse tile kernel owns the lightly populated tiles
        const unsigned *tcr = P.tc + ((long long)b * P.T + t) * 9;
        unsigned ntot = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) ntot += tcr[c];
        if (ntot <= (unsigned)P.sparse_cap) return;
    }
    const int pidx = t < P.tpp ? 1 : 0;
    const int x0 = (t < P.tpp ? t : t - P.tpp) * kTilePix;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float2 *stab = g_slope_tab[P.fast_slot];
    const unsigned hsl = 12u - (unsigned)P.shift;
    const unsigned HS = 1u << hsl;

    unsigned *S = reinterpret_cast<unsigned *>(tile_smem);
    unsigned *O = S + P.capP;
    unsigned *hist = O + P.capP + 16;                   // [NW][HS] words, two 16-bit cells each
    unsigned *part = hist + NW * HS;                    // [9][3][NW]: wave totals, then (wave 0) every wave's exclusive prefix
    unsigned *btot = part + 27 * NW;                    // [9][3] the bins' totals
    unsigned *bplan = btot + 27;                        // [9] pass | half << 4 | skipped << 5 of every bin
    unsigned *pinfo = bplan + 9;                        // [9][12] per pass: see the plan below
    unsigned *spart = pinfo + 9 * 12;                   // [NW + 1] scan partials
    unsigned *bctr = spart + NW + 1;                    // [2] the next batch of the timestamp phase (alternating by pass)
    unsigned *dsto = bctr + 2;                          // [9][2] where the tile's run of bin c starts in records[]
    unsigned *misc = dsto + 18;                         // [0] passes
    unsigned *myhist = hist + wid * HS;

    const float *plane0 = P.vox + (long long)(b * 2 + pidx) * 10 * P.HW;
    const unsigned frame = (unsigned)(P.frame_base + b);
    const int lpx0 = wid * WPX + lane * PPT;
    const int gpx0 = x0 + lpx0;
    unsigned *lists = P.lists + ((long long)b * P.T + t) * (long long)P.list_stride;

    // ---- pre-phase 1: the ten planes, the recurrence, per-lane sums of every bin ------------------------------------------
    int nn[PPT][9];
    float td[PPT][9];
    {
        float yv[PPT][10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const float *src = plane0 + (long long)i * P.HW + gpx0;
            if (PPT == 4) {
                const float4 v = gpx0 < P.HW ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
                yv[0][i] = v.x; yv[1][i] = v.y; yv[PPT - 2][i] = v.z; yv[PPT - 1][i] = v.w;
            } else {
                const float2 v = gpx0 < P.HW ? *reinterpret_cast<const float2 *>(src) : make_float2(0.f, 0.f);
                yv[0][i] = v.x; yv[1][i] = v.y;
            }
        }
        for (unsigned i = tid; i < NW * HS; i += NT) hist[i] = 0;
        if (tid < 2) bctr[tid] = 0;
        if (tid < 9) {
            const long long d = P.slot_cap ? (((long long)b * P.T + t) * 9 + tid) * (long long)P.slot_cap
                                           : P.seg_offsets[b * 9 + tid] + (long long)P.tile_off[((long long)b * P.T + t) * 9 + tid];
            dsto[2 * tid] = (unsigned)d;
            dsto[2 * tid + 1] = (unsigned)((unsigned long long)d >> 32);
            if (P.slot_cap) P.tile_abs_w[(long long)(b * 9 + tid) * P.Tp + t] = (unsigned)d;
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) relocate_all(yv[q], false, nn[q], td[q]);
    }
    // a voxel's class: 0 nothing, 1 single, 2 / 3 multi from the slope table with k == 0 / k != 0, 4 multi outside the table; the
    // table index in bits 3..13
    auto voxel_class = [&](int np, int n, int nx, bool edge) -> unsigned {
        const int dd = edge ? 0 : nx - np;
        const bool intab = (unsigned)(dd + kSlopeM) <= 2u * kSlopeM && (unsigned)n <= (unsigned)kSlopeM && (unsigned)(np | nx) < (1u << 23);
        const unsigned si = (unsigned)((dd + kSlopeM) * (kSlopeM + 1) + n);
        const unsigned cl = n == 1 ? 1u : n < 2 ? 0u : !intab ? 4u : dd == 0 ? 2u : 3u;
        return cl | ((si & 0x7FFu) << 3);
    };
    // its contribution to the lane's running sums: A = events (32 bits, saturating per voxel: a tile's sum cannot wrap, so the
    // count is right for ANY grid), U = k == 0 units | k != 0 units << 16, R = singles | voxels outside the table << 16 (the packed
    // fields are meaningful for the bins that fit: at most capA < 2^16 events)
    auto voxel_sums = [&](unsigned cls, int n, unsigned &A, unsigned &U, unsigned &R) {
        const unsigned cl = cls & 7u, units = (unsigned)(n + 3) >> 2;
        A += cl ? ((unsigned)n < (1u << 20) ? (unsigned)n : 1u << 20) : 0u;
        U += cl == 2u ? units : cl == 3u ? units << 16 : 0u;
        R += (cl == 1u ? 1u : 0u) + (cl == 4u ? 0x10000u : 0u);
    };
    int vmax_l = 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        unsigned A = 0, U = 0, R = 0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const unsigned cls = voxel_class(c > 0 ? nn[q][c - 1] : 0, nn[q][c], c < 8 ? nn[q][c + 1] : 0, c == 0 || c == 8);
            voxel_sums(cls, nn[q][c], A, U, R);
            vmax_l = nn[q][c] > vmax_l ? nn[q][c] : vmax_l;
        }
        const unsigned iA = wave_incl_scan(A, lane), iU = wave_incl_scan(U, lane), iR = wave_incl_scan(R, lane);
        if (lane == 63) { part[(c * 3 + 0) * NW + wid] = iA; part[(c * 3 + 1) * NW + wid] = iU; part[(c * 3 + 2) * NW + wid] = iR; }
    }
    // (the lane's own prefixes are formed again in phase 3 -- 27 scans once per tile -- instead of living in 27 registers across
    // the two barriers; the counts are made opaque so that the compiler does not carry the first results over)
#pragma unroll
    for (int q = 0; q < PPT; ++q)
#pragma unroll
        for (int c = 0; c < 9; ++c) asm volatile("" : "+v"(nn[q][c]));
    __syncthreads();                                    // 1: wave totals of all nine bins
    // ---- pre-phase 2 (wave 0): every wave's exclusive prefix, the bins' totals, the pass plan ------------------------------
    if (wid == 0) {
        if (lane < 27) {
            unsigned v[NW], run = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) v[w] = part[lane * NW + w];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                part[lane * NW + w] = run;
                run += v[w];
            }
            btot[lane] = run;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // bins are walked in order; bin c opens a pass and takes c + 1 along when both fit the LDS; a bin beyond its slot (slot
            // mode) is only counted.  pinfo[p]: 0 first bin | bins << 8, 1 N_A, 2 N_B, 3 k != 0 units, 4 k == 0 units, 5 singles,
            // 6 voxels outside the table, 7 first word of the pass's lists
            unsigned np = 0, off = 0;
            int c = 0;
            while (c < 9) {
                const unsigned NA32 = btot[c * 3];
                if (P.slot_cap && NA32 > (unsigned)P.capA) { bplan[c] = 1u << 5; ++c; continue; }
                unsigned nbins = 1, NB32 = 0;
                if (c + 1 < 9) {
                    NB32 = btot[(c + 1) * 3];
                    if (!(P.slot_cap && NB32 > (unsigned)P.capA) && NA32 + NB32 <= (unsigned)P.capP) nbins = 2;
                }
                unsigned u1 = btot[c * 3 + 1] >> 16, u0 = btot[c * 3 + 1] & 0xFFFFu, ns = btot[c * 3 + 2] & 0xFFFFu, nr = btot[c * 3 + 2] >> 16;
                bplan[c] = np;
                if (nbins == 2) {
                    u1 += btot[(c + 1) * 3 + 1] >> 16; u0 += btot[(c + 1) * 3 + 1] & 0xFFFFu;
                    ns += btot[(c + 1) * 3 + 2] & 0xFFFFu; nr += btot[(c + 1) * 3 + 2] >> 16;
                    bplan[c + 1] = np | (1u << 4);
                }
                unsigned *pi = pinfo + np * 12;
                pi[0] = (unsigned)c | (nbins << 8);
                pi[1] = NA32; pi[2] = nbins == 2 ? NB32 : 0u;
                pi[3] = u1; pi[4] = u0; pi[5] = ns; pi[6] = nr; pi[7] = off;
                pi[8] = off + ((2u * (u1 + u0 + ns) + 3u) & ~3u);          // the 16-byte entries of the voxels outside the table
                off = pi[8] + 4u * nr;
                ++np;
                c += (int)nbins;
            }
            misc[0] = np;
            if (off > (unsigned)P.list_stride) misc[0] = 0xFFFFFFFFu;      // cannot happen (the host sizes the scratch from the slot capacity): checked below
        }
    }
    __syncthreads();                                    // 2: prefixes, totals, plan
    const unsigned npass = misc[0];
    if (npass == 0xFFFFFFFFu) {                          // uniform
        if (tid == 0) atomicExch(reinterpret_cast<unsigned *>(P.status), 4u);
        return;
    }
    // ---- pre-phase 3: the list entries of every voxel, straight to their final places in the tile's scratch ------------------
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const unsigned bp = bplan[c];
        if (bp & (1u << 5)) continue;                    // uniform: only counted
        const unsigned *pi = pinfo + (bp & 15u) * 12;
        const unsigned bsel = (bp >> 4) & 1u;
        const unsigned cA = pi[0] & 0xFFu, twoBins = (pi[0] >> 8) == 2u;
        const unsigned U1 = pi[3], U0 = pi[4], Ns = pi[5];
        uint2 *UL1 = reinterpret_cast<uint2 *>(lists + pi[7]), *UL0 = UL1 + U1, *SL = UL0 + U0;
        uint4 *RL = reinterpret_cast<uint4 *>(lists + pi[8]);
        (void)Ns;
        // entries of the pass's first bin precede the second bin's, wave by wave inside each
        const unsigned pA = part[(c * 3 + 0) * NW + wid], pU = part[(c * 3 + 1) * NW + wid], pR = part[(c * 3 + 2) * NW + wid];
        unsigned lanA, lanU, lanR;                       // the lane's exclusive prefixes inside its wave
        {
            unsigned A = 0, U = 0, R = 0;
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                voxel_sums(voxel_class(c > 0 ? nn[q][c - 1] : 0, nn[q][c], c < 8 ? nn[q][c + 1] : 0, c == 0 || c == 8), nn[q][c], A, U, R);
            lanA = wave_incl_scan(A, lane) - A; lanU = wave_incl_scan(U, lane) - U; lanR = wave_incl_scan(R, lane) - R;
        }
        unsigned a = pA + lanA + (bsel ? pi[1] : 0u);                                              // position in the pass's S
        unsigned s = (pR & 0xFFFFu) + (lanR & 0xFFFFu) + (bsel ? btot[cA * 3 + 2] & 0xFFFFu : 0u);
        unsigned u0 = (pU & 0xFFFFu) + (lanU & 0xFFFFu) + (bsel ? btot[cA * 3 + 1] & 0xFFFFu : 0u);
        unsigned u1 = (pU >> 16) + (lanU >> 16) + (bsel ? btot[cA * 3 + 1] >> 16 : 0u);
        unsigned r = (pR >> 16) + (lanR >> 16) + (bsel ? btot[cA * 3 + 2] >> 16 : 0u);
        (void)twoBins;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int n = nn[q][c];
            const unsigned cls = voxel_class(c > 0 ? nn[q][c - 1] : 0, n, c < 8 ? nn[q][c + 1] : 0, c == 0 || c == 8);
            const unsigned cl = cls & 7u, local = (unsigned)(lpx0 + q);
            if (cl == 1u) {
                SL[s] = make_uint2(a | (local << 14) | (bsel << 25), __float_as_uint(td[q][c]));
                ++s;
            } else if (cl == 2u || cl == 3u) {
                uint2 *dstu = cl == 2u ? UL0 + u0 : UL1 + u1;
                const unsigned units = (unsigned)(n + 3) >> 2, hi = (cls >> 3) << 14, lb = local | (bsel << 28);
                dstu[0] = make_uint2(lb | (((unsigned)n < 4u ? (unsigned)n : 4u) << 29), a | hi);
                for (unsigned jb = 1; jb < units; ++jb) {
                    const unsigned left = (unsigned)n - 4u * jb;
                    dstu[jb] = make_uint2(lb | (jb << 11) | ((left < 4u ? left : 4u) << 29), (a + 4u * jb) | hi);
                }
                if (cl == 2u) u0 += units; else u1 += units;
            } else if (cl == 4u) {
                float k, bb;
                slope_params(c > 0 ? nn[q][c - 1] : 0, n, c < 8 ? nn[q][c + 1] : 0, c, P, k, bb);
                RL[r] = make_uint4(local | (bsel << 11) | ((unsigned)n << 12), a, __float_as_uint(k), __float_as_uint(bb));
                ++r;
            }
            a += cl ? (unsigned)n : 0u;
        }
    }
    if (P.slot_cap) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int m = __shfl_xor(vmax_l, o);
            vmax_l = m > vmax_l ? m : vmax_l;
        }
        if (lane == 0 && vmax_l > 0 && (unsigned long long)vmax_l > *reinterpret_cast<volatile unsigned long long *>(&P.stats_w[0]))
            atomicMax(&P.stats_w[0], (unsigned long long)vmax_l);
        if (tid < 9) P.tc_w[((long long)b * P.T + t) * 9 + tid] = btot[tid * 3];
        if (tid == 0) {
            unsigned tile_total = 0;
#pragma unroll
            for (int c = 0; c < 9; ++c) tile_total += btot[c * 3];
            if (tile_total > 0 && (unsigned long long)tile_total > *reinterpret_cast<volatile unsigned long long *>(&P.stats_w[4]))
                atomicMax(&P.stats_w[4], (unsigned long long)tile_total);
        }
    }
    __syncthreads();                                    // 3: the lists are complete (and visible: one CU, one vector cache)

    STAMP_DECL;
    for (unsigned p = 0; p < npass; ++p) {
        STAMP(0);
        const unsigned *pi = pinfo + p * 12;
        const int c = (int)(pi[0] & 0xFFu);
        const bool doB = (pi[0] >> 8) == 2u;
        const unsigned NA = pi[1], NB_ = pi[2], Ntot = NA + NB_, U1 = pi[3], U0 = pi[4], Ns = pi[5], Nr = pi[6];
        const uint2 *UL1 = reinterpret_cast<const uint2 *>(lists + pi[7]), *UL0 = UL1 + U1, *SL = UL0 + U0;
        const uint4 *RL = reinterpret_cast<const uint4 *>(lists + pi[8]);
        const unsigned dshA = dsto[2 * c] & 3u, dshB = doB ? dsto[2 * (c + 1)] & 3u : 0u;
        const unsigned obB = ((dshA + NA + 3u) & ~3u) + dshB;
        int lgL = 6;
        while (((unsigned)NW << lgL) < Ntot) ++lgL;
        const float offtA = P.offt[c], offtB = P.offt[doB ? c + 1 : c];
        const int kbA = (int)P.kbase[c], kbB = (int)P.kbase[doB ? c + 1 : c];
        const unsigned pcA = (unsigned)(pidx * 9 + c);
        unsigned *bc = bctr + (p & 1u);
        auto put = [&](unsigned sp, unsigned rec, unsigned g) {
            S[sp] = rec;
            atomicAdd(&hist[((sp >> lgL) << hsl) + (g >> 1)], 1u << ((g & 1u) << 4));
        };
        // ---- D3: timestamps, once, from the lists; batches of 64 entries handed out through an LDS counter ------------------
        auto single_batch = [&](uint2 e, bool has) {     // (called by whole waves: single_key votes)
            const unsigned sp = e.x & 0x3FFFu, bsel = (e.x >> 25) & 1u;
            const unsigned key = single_key(has, has ? __uint_as_float(e.y) : 0.0f, bsel ? offtB : offtA, (long long)(bsel ? kbB : kbA), true, P);
            if (has) put(sp, (bsel << 24) | (key << 12) | ((e.x >> 14) & (kTilePix - 1)), (bsel << hsl) | (key >> P.shift));
        };
        auto unit_batch = [&](auto mode_c, uint2 e, bool has) {
            constexpr int MODE = decltype(mode_c)::value;        // 1: k == 0 units (checked fast constant divisions), 3: k != 0 units
            if (has) {
                const unsigned local = e.x & (kTilePix - 1), jb = (e.x >> kLocalBits) & 0x1FFFFu, bsel = (e.x >> 28) & 1u, cnt = e.x >> 29;
                const unsigned sp = e.y & 0x3FFFu;
                float2 kb = make_float2(0.0f, 0.0f);
                if (MODE == 3) kb = stab[e.y >> 14];
                const unsigned px = (unsigned)x0 + local;
                const float offt_c = bsel ? offtB : offtA;
                const int kbase_c = bsel ? kbB : kbA;
                unsigned o[4];
                philox4_b3(P.seed, px, jb, pcA + bsel, frame, o);
                float tq[4];
                if (MODE == 3) {
                    const float r1 = rcp_refined(kb.x), bb2 = kb.y * kb.y, k2 = 2.0f * kb.x;
#pragma unroll
                    for (int s = 0; s < 4; ++s) tq[s] = div_rn_nr(-kb.y + sqrt_rn_nr(bb2 + k2 * u24(o[s])), kb.x, r1);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) tq[s] = k0_time_fast(u24(o[s]), P.FPS, P.RFPS, P.R9);
                }
                const unsigned tag = (bsel << 24) | (1u << kLocalBits) | local, gsel = bsel << hsl;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float tt = tq[s] + offt_c;
                    tt = tt * 1e6f;
                    int kk = (int)tt - kbase_c;
                    kk = kk < 0 ? 0 : kk;
                    const unsigned key = (unsigned)(kk >= P.NK ? P.NK - 1 : kk);
                    if ((unsigned)s < cnt) put(sp + s, (key << 12) | tag, gsel | (key >> P.shift));
                }
            }
        };
        auto rare_batch = [&](unsigned i) {              // voxels outside the slope table: the lane walks its voxel's draws
            if (i < Nr) {
                const uint4 e = RL[i];
                const unsigned local = e.x & (kTilePix - 1), bsel = (e.x >> 11) & 1u, n = e.x >> 12;
                const float k = __uint_as_float(e.z), bb = __uint_as_float(e.w);
                const unsigned px = (unsigned)x0 + local;
                for (unsigned j = 0; j < n; ++j) {
                    const float u = philox_uniform(P.seed, px, j, pcA + bsel, frame);
                    const unsigned key = multi_key(k, bb, u, bsel ? offtB : offtA, bsel ? kbB : kbA, P, true);
                    put(e.y + j, (bsel << 24) | (key << 12) | (1u << kLocalBits) | local, (bsel << hsl) | (key >> P.shift));
                }
            }
        };
        {
            const unsigned nK = (U1 + 63u) >> 6, nZ = (U0 + 63u) >> 6, nS = (Ns + 63u) >> 6, nR = (Nr + 63u) >> 6, nb = nK + nZ + nS + nR;
            // the list entry of batch `bid` for this lane (k != 0 units | k == 0 units | singles are all 8-byte entries of one array)
            auto entry_of = [&](unsigned bid, bool &has) -> uint2 {
                unsigned i, cnt;
                const uint2 *lst;
                if (bid < nK) { i = bid * 64u + lane; cnt = U1; lst = UL1; }
                else if (bid < nK + nZ) { i = (bid - nK) * 64u + lane; cnt = U0; lst = UL0; }
                else { i = (bid - nK - nZ) * 64u + lane; cnt = bid < nK + nZ + nS ? Ns : 0u; lst = SL; }
                has = i < cnt;
                return has ? lst[i] : make_uint2(0u, 0u);
            };
            unsigned nxt = 0;
            if (lane == 0) nxt = atomicAdd(bc, 1u);
            unsigned bid = (unsigned)__builtin_amdgcn_readfirstlane((int)nxt);
            bool has = false;
            uint2 e = bid < nb ? entry_of(bid, has) : make_uint2(0u, 0u);
            while (bid < nb) {
                // the next batch is reserved now and its entries requested: both round trips run under this batch
                if (lane == 0) nxt = atomicAdd(bc, 1u);
                const unsigned nbid = (unsigned)__builtin_amdgcn_readfirstlane((int)nxt);
                bool nhas = false;
                const uint2 ne = nbid < nb ? entry_of(nbid, nhas) : make_uint2(0u, 0u);
                if (bid < nK) unit_batch(std::integral_constant<int, 3>{}, e, has);
                else if (bid < nK + nZ) unit_batch(std::integral_constant<int, 1>{}, e, has);
                else if (bid < nK + nZ + nS) single_batch(e, has);
                else rare_batch((bid - nK - nZ - nS) * 64u + lane);
                bid = nbid; e = ne; has = nhas;
            }
        }
        STAMP(4);
        __syncthreads();                                 // B: every row of the histogram is complete
        STAMP(5);
        if (tid == 0) bctr[(p + 1u) & 1u] = 0;          // the next pass's counter (this pass's is not read again)
        // ---- D4: cell-major, row-minor exclusive scan; the tile's rows of the run table (dense_pair_body's) ----------------
        {
            unsigned v[NW];
            unsigned run0 = 0, run1 = 0;
            const bool mine = (unsigned)tid < HS;
            if (mine) {
#pragma unroll
                for (int w = 0; w < NW; ++w) v[w] = hist[(unsigned)w * HS + tid];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const unsigned c0 = v[w] & 0xFFFFu, c1 = v[w] >> 16;
                    v[w] = run0 | (run1 << 16);
                    run0 += c0;
                    run1 += c1;
                }
            }
            unsigned tot;
            const unsigned boff = block_excl_scan<NW>(run0 + run1, spart, &tot);
            if (mine) {
                const unsigned bsel = (unsigned)tid >> (hsl - 1u);
                const unsigned adj = bsel ? obB - NA : dshA;
                const unsigned packed = (boff + adj) | ((boff + run0 + adj) << 16);
#pragma unroll
                for (int w = 0; w < NW; ++w) hist[(unsigned)w * HS + tid] = v[w] + packed;
                const unsigned bkt = (2u * (unsigned)tid) & (HS - 1u);
                const int cb = c + (int)bsel;
                if (bsel == 0u || doB) {
                    unsigned short *row = P.roff + ((long long)(b * 9 + cb) * P.T + t) * (P.NB + 1);
                    const unsigned rb = boff - (bsel ? NA : 0u);
                    if (bkt < (unsigned)P.NB) row[bkt] = (unsigned short)rb;
                    if (bkt + 1u < (unsigned)P.NB) row[bkt + 1u] = (unsigned short)(rb + run0);
                }
            }
            if (tid == 0) {
                P.roff[((long long)(b * 9 + c) * P.T + t) * (P.NB + 1) + P.NB] = (unsigned short)NA;
                if (doB) P.roff[((long long)(b * 9 + c + 1) * P.T + t) * (P.NB + 1) + P.NB] = (unsigned short)NB_;
            }
        }
        __syncthreads();                                 // C: cell offsets per row
        STAMP(6);
        // ---- D5: stable ranks; the wave then clears its own row for the next pass (nobody else touches it before barrier D) ----
        {
            const unsigned sh = 12u + (unsigned)P.shift;
            const unsigned lo = (unsigned)wid << lgL, hi = min(Ntot, lo + (1u << lgL));
            unsigned i = lo + lane, i0 = lo;
            for (; i0 + 256u <= hi; i0 += 256u, i += 256u) {
                unsigned rec[4], at[4], hs[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) rec[j] = S[i + 64u * j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned g = rec[j] >> sh;
                    hs[j] = (g & 1u) << 4;
                    at[j] = atomicAdd(&myhist[g >> 1], 1u << hs[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) O[(at[j] >> hs[j]) & 0xFFFFu] = rec[j] & 0xFFFFFFu;
            }
            for (; i < hi; i += 64u) {
                const unsigned rec = S[i], g = rec >> sh, hs = (g & 1u) << 4;
                O[(atomicAdd(&myhist[g >> 1], 1u << hs) >> hs) & 0xFFFFu] = rec & 0xFFFFFFu;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();             // (one wave's LDS operations complete in order)
            for (unsigned j = lane; j < HS; j += 64u) myhist[j] = 0;
        }
        STAMP(7);
        __syncthreads();                                 // D: the runs are complete in O, the histogram is clean
        STAMP(8);
        {
            auto copy_run = [&](int cb, unsigned ob, unsigned dsh, unsigned N) {
                const long long dst0 = (long long)(((unsigned long long)dsto[2 * cb + 1] << 32) | dsto[2 * cb]);
                unsigned *dstq = P.temp + (dst0 - (long long)dsh);            // 16-byte aligned
                const uint4 *src = reinterpret_cast<const uint4 *>(O + (ob - dsh));
                const unsigned nq = (dsh + N + 3u) >> 2;
                for (unsigned q = tid; q < nq; q += NT) {
                    const uint4 v = src[q];
                    if (4u * q >= dsh && 4u * q + 4u <= dsh + N) {
                        reinterpret_cast<uint4 *>(dstq)[q] = v;
                    } else {
                        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4u * q + j >= dsh && 4u * q + j < dsh + N) dstq[4u * q + j] = w[j];
                    }
                }
            };
            copy_run(c, dshA, dshA, NA);
            if (doB) copy_run(c + 1, obB, dshB, NB_);
        }
        STAMP(9);
        // (the next pass's timestamps write S and the histogram, the copy-out above reads O: no barrier between them; O is
        // written again only behind the next pass's barrier C, which every thread reaches after its copy-out)
    }
    STAMP(0);
    STAMP_FLUSH(0, 10);
}

// true when every run-time switch of the common call is on (dense_tile_body<NW, true>'s condition)
__device__ __forceinline__ bool dense_fast_call(const LdatiParams &P) {
    bool fast = P.fast_slot >= 0 && P.rng_mode == V2CE_RNG_PHILOX && !P.ballot_ranks && P.ts32 && (P.HW & 3) == 0 &&
                P.strategy != V2CE_STRATEGY_NONE;
    if (fast) {
        const FastDiv &f = g_fastdiv[P.fast_slot];
        fast = __builtin_amdgcn_readfirstlane((int)f.fps_bits) == (int)__float_as_uint(P.FPS) && __builtin_amdgcn_readfirstlane(f.ok) != 0 &&
               __builtin_amdgcn_readfirstlane(f.tab_ready) != 0 && __builtin_amdgcn_readfirstlane(f.ok64) != 0 &&
               __builtin_amdgcn_readfirstlane(g_lds_order_ok) != 0;
    }
    return fast;
}

// the pair-pass form for the common call; any other call (replayed uniforms, 'none', ballot ranks, a device table not ready)
// runs the per-bin body in the same launch (the LDS of the launch covers both maps)
template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void ldati_tile_pair_kernel(LdatiParams P) {
    if (dense_fast_call(P)) dense_pair_body<NW>(P);
    else dense_tile_body<NW, false>(P);
}

// the one-pass form for the common call (P.lists != NULL); any other call runs the per-bin body in the same launch
template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void ldati_tile_onepass_kernel(LdatiParams P) {
    if (dense_fast_call(P)) dense_onepass_body<NW>(P);
    else dense_tile_body<NW, false>(P);
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void ldati_tile_dense_kernel(LdatiParams P) {   // 4 waves per SIMD: two 512-thread workgroups per CU
    bool fast = P.fast_slot >= 0 && P.rng_mode == V2CE_RNG_PHILOX && !P.ballot_ranks && P.ts32 && (P.HW & 3) == 0 &&
                P.strategy != V2CE_STRATEGY_NONE;
    if (fast) {
        const FastDiv &f = g_fastdiv[P.fast_slot];
        fast = __builtin_amdgcn_readfirstlane((int)f.fps_bits) == (int)__float_as_uint(P.FPS) && __builtin_amdgcn_readfirstlane(f.ok) != 0 &&
               __builtin_amdgcn_readfirstlane(f.tab_ready) != 0 && __builtin_amdgcn_readfirstlane(f.ok64) != 0 &&
               __builtin_amdgcn_readfirstlane(g_lds_order_ok) != 0;
    }
    if (fast) dense_tile_body<NW, true>(P);
    else dense_tile_body<NW, false>(P);
}

// Exhaustive check of the dense kernel's exact-math helpers (v2ce_ldati_selfcheck; a test, never on the product path):
// for EVERY slope-table entry with k != 0 and EVERY uniform the Philox path can produce (m * 2^-24, m < 2^24) the time
// (-b + sqrt(b^2 + 2 k u)) / k by sqrt_rn_nr / div_rn_nr against the compiler's IEEE square root and division; and
// philox4_b3 against philox4 on the same counters.  bad[0] counts time mismatches, bad[1] Philox mismatches.
__global__ __launch_bounds__(256) void ldati_selfcheck_kernel(float VS, float VS2, float INV, int d0, unsigned long long *bad) {
    const unsigned m = blockIdx.x * 256u + threadIdx.x;                  // < 2^24
    const float u = (float)m * (1.0f / 16777216.0f);
    const int d = d0 + (int)blockIdx.y;
    unsigned nb = 0, np = 0;
    for (int n = 2; n <= kSlopeM; ++n) {
        const float sxy = (float)d;
        const float k0 = (3.0f * sxy) / 6.0f;
        const float k = (k0 / VS2) / ((float)n + 1e-8f);
        const float bb = INV - (VS * k) / 2.0f;
        if (k == 0.0f) continue;
        const float want = (-bb + __builtin_sqrtf(bb * bb + (2.0f * k) * u)) / k;
        const float r1 = rcp_refined(k), bb2 = bb * bb, k2 = 2.0f * k;
        const float got = div_rn_nr(-bb + sqrt_rn_nr(bb2 + k2 * u), k, r1);
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
        nb += same ? 0u : 1u;
    }
    {
        unsigned a[4], c[4];
        philox4(0x123456789ABCDEFull + (unsigned long long)d, m, m >> 7, (unsigned)(d & 15), m ^ 0x5bd1e995u, a);
        philox4_b3(0x123456789ABCDEFull + (unsigned long long)d, m, m >> 7, (unsigned)(d & 15), m ^ 0x5bd1e995u, c);
        np += (a[0] != c[0] || a[1] != c[1] || a[2] != c[2] || a[3] != c[3]) ? 1u : 0u;
    }
    if (nb) atomicAdd(&bad[0], (unsigned long long)nb);
    if (np) atomicAdd(&bad[1], (unsigned long long)np);
}

// ---------------------------------------------------------------------------------------------
// sparse tile pass: the same job as ldati_tile_pass_kernel for tiles with at most kSparseCap events
// over ALL NINE bins (real UNet output: ~2000 per 2048-pixel tile; the per-bin kernel spends its
// time in ~60 barriers and 18 workgroup scans for ~230 records per bin there).  One pass, 8 barriers:
//   1. all ten voxels of the thread's four pixels (one float4 per plane), relocation in registers
//   2. single events and multi-event (pixel, bin) pairs appended to two LDS lists (from both ends of
//      one 32 KB area), re-read densely: every lane computes a timestamp
//   3. records (bin-major combined key | multi | local pixel) appended unordered; u16 histogram
//      over the 9 * NB (bin, coarse bucket) cells by packed LDS atomics
//   4. in-place exclusive scan of the histogram; placement by a second packed atomic
//   5. inside each cell (mostly 0-2 records) the records are ordered by local pixel: the order the
//      bucket sort's stable ranks need (equal (timestamp, category) => ascending pixel)
//   6. nine coalesced runs + nine rows of the run table, exactly what the per-bin kernel writes.
// LDS: list area / records S + O (32 KB) | packed histogram (10 KB) | scalars.
// ---------------------------------------------------------------------------------------------
// FUSED (round 4, v2ce_ldati_count_fused): the kernel also IS the count pass -- it forms the tile's counts from the voxels it
// has just loaded (what ldati_count_tiles_kernel computes), so the voxel grid is read once per call instead of twice; the
// record offsets a count pass would have provided are not needed because every tile writes into its own slot of kSparseCap
// records (`temp` + slot; the (tile, bin) starts go to tile_abs for the bucket sort).  A tile beyond kSparseCap only reports
// its counts; the host then takes the two-pass path for the call (v2ce_ldati_emit_fused).
template <bool FUSED>
__global__ __launch_bounds__(kSparseThreads, V2CE_SPARSE_WAVES) void ldati_tile_sparse_kernel(LdatiParams P) {
    constexpr int NT = kSparseThreads, NW = NT / 64, PPT = kTilePix / NT;
    constexpr int HWORDS = NT * 5;                       // 10 cells per thread >= 9 * kMaxNB + 1
    const int t = blockIdx.x, b = blockIdx.y;
    unsigned ntot = 0;
    if (!FUSED) {
        const unsigned *tcr = P.tc + ((long long)b * P.T + t) * 9;
#pragma unroll
        for (int c = 0; c < 9; ++c) ntot += tcr[c];
        if (ntot > (unsigned)P.sparse_cap) return;          // dense tile: ldati_tile_pass_kernel handles it
    }
    const int pidx = t < P.tpp ? 1 : 0;
    const int x0 = (t < P.tpp ? t : t - P.tpp) * kTilePix;
    const int tid = threadIdx.x;
    const bool slot_ok = P.fast_slot >= 0 &&
                         __builtin_amdgcn_readfirstlane((int)g_fastdiv[P.fast_slot >= 0 ? P.fast_slot : 0].fps_bits) == (int)__float_as_uint(P.FPS);
    const bool fast_k0 = slot_ok && P.rng_mode == V2CE_RNG_PHILOX && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].ok) != 0;
    const float2 *stab = (slot_ok && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].tab_ready) != 0) ? g_slope_tab[P.fast_slot] : nullptr;

    unsigned *S = reinterpret_cast<unsigned *>(tile_smem);          // [kSparseCap] records (unordered, later final)
    unsigned *O = S + kSparseCap;                                   // [kSparseCap] records placed by cell
    uint2 *SL = reinterpret_cast<uint2 *>(S);                       // singles list, grows up: {debt bits, local | c << 11}
    unsigned *MPtop = S + 2 * kSparseCap;                           // multi list, grows down: {info, k, bb}
    unsigned *hist = S + 2 * kSparseCap;                            // [HWORDS] two u16 cells per word
    unsigned *misc = hist + HWORDS;                                 // cursors, scan partials, bin starts, per-bin scalars
    unsigned *cur = misc;                                           // [0] singles, [1] multi pairs, [2] records
    unsigned *part = misc + 4;                                      // [NW + 1]
    unsigned *binstart = part + 10;                                 // [10]
    float *offt_s = reinterpret_cast<float *>(binstart + 10);       // [9]
    long long *kbase_s = reinterpret_cast<long long *>(offt_s + 10);   // [9], 8-byte aligned (kSparseLds)
    unsigned *red = reinterpret_cast<unsigned *>(kbase_s + 9);         // [NW][10] the fused count's wave totals
    static_assert(NW + 1 <= 10 && ((2 * kSparseCap + HWORDS + 34) & 1) == 0, "sparse LDS map");
    long long dst_off[9];                                           // where the tile's nine runs go
#pragma unroll
    for (int c = 0; c < 9; ++c)
        dst_off[c] = FUSED ? 0ll : P.seg_offsets[b * 9 + c] + (long long)P.tile_off[((long long)b * P.T + t) * 9 + c];

    const float *plane0 = P.vox + (long long)(b * 2 + pidx) * 10 * P.HW;
    const unsigned frame = (unsigned)(P.frame_base + b);
    const int lpx0 = tid * PPT;
    const unsigned NKS = (unsigned)P.NB << P.shift;                 // key stride between bins (>= NK)

    STAMP_DECL;
    // ---- 1: loads (all in flight at once), then the LDS tables are cleared under them
    float yv[PPT][10];
    {
        const int px = x0 + lpx0;
        if ((P.HW & 3) == 0 && px + PPT <= P.HW) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const float4 v = *reinterpret_cast<const float4 *>(plane0 + (long long)i * P.HW + px);
                yv[0][i] = v.x; yv[1][i] = v.y; yv[2][i] = v.z; yv[3][i] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q)
#pragma unroll
                for (int i = 0; i < 10; ++i) yv[q][i] = (px + q < P.HW) ? plane0[(long long)i * P.HW + px + q] : 0.0f;
        }
    }
    for (int i = tid; i < HWORDS; i += NT) hist[i] = 0;
    if (tid < 4) cur[tid] = 0;
    if (tid < 9) { offt_s[tid] = P.offt[tid]; kbase_s[tid] = P.kbase[tid]; }
    __syncthreads();

    STAMP(0);
    // ---- 2a: classify; append to the lists (slots from wave_alloc: one LDS atomic per wave, pixel and list)
    const int lane_s = tid & 63;
    // FUSED: the count pass rides along (what ldati_count_tiles_kernel computes: events per bin, largest voxel count).  The
    // lists are written before the tile's total is known, so their indices are bounded: a tile whose lists would collide
    // has more than kSparseCap events and is discarded below anyway.
    int cnt9[9], vmax = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) cnt9[i] = 0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        int nn[9];
        float td[9];
        relocate_all(yv[q], P.bidir != 0, nn, td);
        const unsigned local = (unsigned)(lpx0 + q);
        const bool valid = x0 + lpx0 + q < P.HW;
        unsigned ns = 0, nm = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int n = valid ? nn[c] : 0;
            ns += n == 1 ? 1u : 0u;
            nm += (n >= 2 && P.strategy != V2CE_STRATEGY_NONE) ? 1u : 0u;
            if (FUSED) {
                cnt9[c] += (P.strategy == V2CE_STRATEGY_NONE) ? (n == 1) : (n > 0 ? n : 0);
                vmax = n > vmax ? n : vmax;
            }
        }
        unsigned si = wave_alloc(&cur[0], ns, lane_s), mi0 = wave_alloc(&cur[1], nm, lane_s);
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int n = valid ? nn[c] : 0;
            if (n == 1) {
                if (!FUSED || si < (unsigned)kSparseCap) SL[si] = make_uint2(__float_as_uint(td[c]), local | ((unsigned)c << kLocalBits));
                ++si;
            } else if (n >= 2 && P.strategy != V2CE_STRATEGY_NONE && (!FUSED || 3u * (mi0 + 1u) <= 2u * (unsigned)kSparseCap)) {
                float k, bb;
                if (P.kbb) {
                    const float2 kq = P.kbb[((long long)(b * 2 + pidx) * 9 + c) * P.HW + (x0 + lpx0 + q)];
                    k = kq.x; bb = kq.y;
                } else {
                    slope_params(c > 0 ? nn[c - 1] : 0, n, c < 8 ? nn[c + 1] : 0, c, P, k, bb, stab);
                }
                unsigned *e = MPtop - 3 * (mi0 + 1);
                ++mi0;
                e[0] = local | ((unsigned)c << kLocalBits) | ((unsigned)n << 15);
                e[1] = __float_as_uint(k);
                e[2] = __float_as_uint(bb);
            } else if (FUSED && n >= 2 && P.strategy != V2CE_STRATEGY_NONE) {
                ++mi0;                                       // (beyond the list: the tile is dense)
            }
        }
    }
    STAMP(1);
    if (FUSED) {
        // the tile's total and its largest voxel count: one wave scan and one wave maximum.  The per-bin counts of a tile that
        // fits come out of the cell scan for free (binstart, step 6); only a tile beyond kSparseCap -- which this kernel cannot
        // place -- reduces its nine per-bin sums here (what ldati_count_tiles_kernel reports)
        unsigned tl = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) tl += (unsigned)cnt9[i];
        const unsigned wt = wave_incl_scan(tl, lane_s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int m = __shfl_xor(vmax, o);
            vmax = m > vmax ? m : vmax;
        }
        if (lane_s == 63) red[(tid >> 6) * 10 + 8] = wt;
        if (lane_s == 0) red[(tid >> 6) * 10 + 9] = (unsigned)vmax;
    }
    __syncthreads();
    if (FUSED) {
        unsigned mx = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            ntot += red[w * 10 + 8];
            const unsigned v = red[w * 10 + 9];
            mx = v > mx ? v : mx;
        }
        // (plain reads first: after the first few tiles these maxima rarely grow, and same-address atomics serialise)
        if (tid == 9 && mx > 0 && (unsigned long long)mx > P.stats_w[0]) atomicMax(&P.stats_w[0], (unsigned long long)mx);
        if (tid == 10 && ntot > 0 && (unsigned long long)ntot > P.stats_w[4]) atomicMax(&P.stats_w[4], (unsigned long long)ntot);
        if (ntot > (unsigned)P.sparse_cap) {             // uniform: the call falls back to the two-pass path
            __syncthreads();                             // (red is rewritten)
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const unsigned tot = wave_incl_scan((unsigned)cnt9[i], lane_s);
                if (lane_s == 63) red[(tid >> 6) * 10 + i] = tot;
            }
            __syncthreads();
            if (tid < 9) {
                unsigned mine = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) mine += red[w * 10 + tid];
                P.tc_w[((long long)b * P.T + t) * 9 + tid] = mine;
            }
            return;
        }
    }
    STAMP(2);
    // ---- 2b: the lists move to registers (they share their LDS with the records)
    const unsigned Ns = cur[0], Nmp = cur[1];
    constexpr int SPT = kSparseCap / NT, MPT = kSparseCap / 2 / NT;    // most list entries per thread
    uint2 sl[SPT];
    unsigned mi[MPT];
    float mk[MPT], mb[MPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) sl[j] = (unsigned)(tid + j * NT) < Ns ? SL[tid + j * NT] : make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < MPT; ++j) {
        const unsigned m = (unsigned)(tid + j * NT);
        const unsigned *e = MPtop - 3 * (m + 1);
        mi[j] = m < Nmp ? e[0] : 0u;
        mk[j] = m < Nmp ? __uint_as_float(e[1]) : 0.0f;
        mb[j] = m < Nmp ? __uint_as_float(e[2]) : 0.0f;
    }
    __syncthreads();
    STAMP(3);
    // ---- 3: timestamps -> records, cell histogram
    auto put = [&](unsigned idx, unsigned c, unsigned key, unsigned multi, unsigned local) {
        const unsigned ck = __umul24(c, NKS) + key;       // (c < 9, NKS < 2^16: the 24-bit multiply issues at full rate)
        const unsigned g = ck >> P.shift;
        atomicAdd(&hist[g >> 1], 1u << ((g & 1u) * 16u));
        S[idx] = (ck << 12) | (multi << kLocalBits) | local;
    };
    const bool fast_s = slot_ok && P.ts32 && __builtin_amdgcn_readfirstlane(g_fastdiv[P.fast_slot].ok64) != 0;      // single_key's fast form
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        if ((unsigned)(j * NT) >= Ns) break;                 // uniform: no single left for anybody
        const bool has = (unsigned)(tid + j * NT) < Ns;
        const unsigned c = has ? sl[j].y >> kLocalBits : 0u;
        const unsigned key = single_key(has, __uint_as_float(sl[j].x), offt_s[c], kbase_s[c], fast_s, P);
        if (has) put((unsigned)(tid + j * NT), c, key, 0u, sl[j].y & (kTilePix - 1));   // a single's record index = its list index
    }
    STAMP(4);
#pragma unroll
    for (int j = 0; j < MPT; ++j) {
        if ((unsigned)(j * NT) >= Nmp) break;                // uniform: no multi-event pair left for anybody
        const bool have = (unsigned)(tid + j * NT) < Nmp;
        // the multi-event records follow the singles; cur[2] counts them (one atomic per wave)
        const unsigned i0 = Ns + wave_alloc(&cur[2], have ? mi[j] >> 15 : 0u, lane_s);
        if (have) {
            const unsigned local = mi[j] & (kTilePix - 1), c = (mi[j] >> kLocalBits) & 15u, n = mi[j] >> 15;
            const unsigned px = (unsigned)x0 + local;
            const float offt = offt_s[c];
            const long long kb64 = kbase_s[c];
            for (unsigned jb = 0; 4u * jb < n; ++jb) {
                const unsigned left = n - 4u * jb, cnt = left < 4u ? left : 4u;
                float u[4];
                if (P.rng_mode == V2CE_RNG_REPLAY) {
                    const long long ub = (((long long)(b * 2 + pidx) * 9 + c) * P.HW + px) * P.replay_max_n;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int jj = (int)(4u * jb) + s;
                        u[s] = ((unsigned)s < cnt && jj < P.replay_max_n) ? P.uniforms[ub + jj] : 0.0f;
                    }
                } else {
                    unsigned o[4];
                    philox4(P.seed, px, jb, (unsigned)pidx * 9u + c, frame, o);
#pragma unroll
                    for (int s = 0; s < 4; ++s) u[s] = u24(o[s]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if ((unsigned)s < cnt) {
                        const unsigned key = P.ts32 ? multi_key(mk[j], mb[j], u[s], offt, (int)kb64, P, fast_k0)
                                                    : (unsigned)key_of(multi_ts(mk[j], mb[j], u[s], offt, P), kb64, P.NK);
                        put(i0 + 4u * jb + s, c, key, 1u, local);
                    }
            }
        }
    }
    __syncthreads();
    STAMP(5);
    const unsigned N = Ns + cur[2];
    if (N != ntot) {                                     // cannot happen: the count kernel saw the same voxels
        if (tid == 0) atomicExch(reinterpret_cast<unsigned *>(P.status), 3u);
        return;
    }
    // ---- 4: exclusive scan of the cells, in place (thread: cells 10 tid .. 10 tid + 9)
    {
        unsigned w[5], cnt[10], run = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] = hist[tid * 5 + j];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            cnt[j] = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
            const unsigned v = cnt[j];
            cnt[j] = run;
            run += v;
        }
        unsigned tot;
        const unsigned ex = block_excl_scan<NW>(run, part, &tot);
#pragma unroll
        for (int j = 0; j < 5; ++j) hist[tid * 5 + j] = (cnt[2 * j] + ex) | ((cnt[2 * j + 1] + ex) << 16);
    }
    __syncthreads();
    STAMP(6);
    auto cell = [&](unsigned g) { return (hist[g >> 1] >> ((g & 1u) * 16u)) & 0xFFFFu; };
    // placement: the cell's word advances from its start to its end
    for (unsigned i = tid; i < N; i += NT) {
        const unsigned rec = S[i];
        const unsigned g = rec >> (12 + P.shift);
        const unsigned old = atomicAdd(&hist[g >> 1], 1u << ((g & 1u) * 16u));
        O[(old >> ((g & 1u) * 16u)) & 0xFFFFu] = rec;
    }
    __syncthreads();
    STAMP(7);
    if (tid < 10) binstart[tid] = tid == 9 ? N : (tid ? cell((unsigned)tid * (unsigned)P.NB - 1u) : 0u);
    // ---- 5: order inside the cells: by local pixel (ties: by arrival, any fixed order is right); four records per thread at a
    // time, so that their dependent LDS reads (record, the cell's two bounds, the cell's members) overlap
    for (unsigned i0 = tid; i0 < N; i0 += 4 * NT) {
        unsigned rec[4], lo[4], hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rec[j] = i0 + j * NT < N ? O[i0 + j * NT] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned g = rec[j] >> (12 + P.shift);
            lo[j] = g ? cell(g - 1) : 0u;
            hi[j] = cell(g);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = i0 + j * NT;
            if (i < N) {
                unsigned r = lo[j];
                if (hi[j] - lo[j] > 1u) {
                    const unsigned me = rec[j] & (kTilePix - 1);
                    for (unsigned q = lo[j]; q < hi[j]; ++q) {
                        const unsigned o = O[q] & (kTilePix - 1);
                        r += (o < me || (o == me && q < i)) ? 1u : 0u;
                    }
                }
                S[r] = rec[j];
            }
        }
    }
    __syncthreads();
    STAMP(8);
    // ---- 6: nine runs of records, nine rows of the run table
    unsigned bst[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) bst[c] = binstart[c];
    if (FUSED) {
        // the nine runs are contiguous in the tile's slot: one loop, the bin of a record from its position
        const long long slot = ((long long)b * P.T + t) * kSparseCap;
        if (tid < 9) {
            P.tile_abs_w[(long long)(b * 9 + tid) * P.Tp + t] = (unsigned)(slot + bst[tid]);
            P.tc_w[((long long)b * P.T + t) * 9 + tid] = bst[tid + 1] - bst[tid];       // the count pass's output
        }
        unsigned *dst = P.temp + slot;
        for (unsigned i = tid; i < N; i += NT) {
            unsigned c = 0;
#pragma unroll
            for (int j = 1; j < 9; ++j) c += i >= bst[j] ? 1u : 0u;
            dst[i] = S[i] - (__umul24(c, NKS) << 12);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const unsigned bs = bst[c], Nc = bst[c + 1] - bs;
            const unsigned strip = ((unsigned)c * NKS) << 12;
            unsigned *dst = P.temp + dst_off[c];
            for (unsigned i = tid; i < Nc; i += NT) dst[i] = S[bs + i] - strip;
        }
    }
    {
        unsigned short *row0 = P.roff + ((long long)(b * 9) * P.T + t) * (P.NB + 1);
        const long long rstep = (long long)P.T * (P.NB + 1);
        for (int k0 = 0; k0 < P.NB; k0 += NT) {          // (one trip: NB <= kSparseThreads unless the key range is huge)
            const int kk = k0 + tid;
            if (kk < P.NB) {
                unsigned cv[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    const unsigned g = (unsigned)c * (unsigned)P.NB + (unsigned)kk;
                    cv[c] = g ? cell(g - 1) : 0u;
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) row0[c * rstep + kk] = (unsigned short)(cv[c] - bst[c]);
            }
        }
        if (tid < 9) row0[tid * rstep + P.NB] = (unsigned short)(bst[tid + 1] - bst[tid]);
    }
    STAMP(9);
    STAMP_FLUSH(0, 10);
}

// per segment: bucket totals over the tiles, their exclusive prefix, and the SORT GROUPS: maximal
// runs of consecutive coarse buckets with at most cap2 records and at most `span` buckets (so that the
// group's (key, category) histogram fits the sort workgroup).  Timestamps crowd towards the end of a
// bin on real UNet output (small voxel values: the single event falls where the accumulated mass
// crosses 1), so equal-width buckets differ by 20x in a segment; the groups even that out.  A single
// coarse bucket beyond cap2 (degenerate ties: constant images; bidirectional relocation puts every
// single of bin 8 with y[9] = 0 at the same microsecond) goes to the big-bucket kernel's list.
__global__ __launch_bounds__(512) void ldati_bucket_scan_kernel(LdatiParams P) {
    __shared__ unsigned part[9];
    __shared__ unsigned pre[kMaxNB + 2];               // sums of the tiles' run starts, then the bucket prefix
    const int seg = blockIdx.x, t = threadIdx.x;
    const unsigned short *tab = P.roff + (long long)seg * P.T * (P.NB + 1);
    unsigned *bofs = P.bofs + (long long)seg * (P.NB + 1);
    // a bucket's total = sum over the tiles of (start of the next bucket - its start) = difference of the
    // column sums: one load per tile and thread (NB + 1 <= 513 columns, 512 threads + one straggler).
    // (Round 4 tried handing the table to the sort TRANSPOSED -- [bucket][tile], through LDS here -- so that its setup reads
    // two contiguous columns instead of one 938-byte-strided load per tile: sort 520 -> 483 us on the stress chunk, but this
    // kernel 34 -> 48 us there and 20 -> 50 us in the e2e regime (576 segments, latency-bound): removed.)
    for (int i = t; i <= P.NB; i += 512) {
        unsigned s = 0;
#pragma unroll 22                                           // (latency-bound: a quarter of the 88 tiles' loads in flight at a time)
        for (int tt = 0; tt < P.T; ++tt) s += (unsigned)tab[(long long)tt * (P.NB + 1) + i];
        pre[i] = s;
    }
    __syncthreads();
    const unsigned tot_i = t < P.NB ? pre[t + 1] - pre[t] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan<8>(tot_i, part, &total);      // barriers inside: pre[] is free afterwards
    if (t < P.NB) bofs[t] = ex;
    if (t == 0) bofs[P.NB] = total;
    __syncthreads();
    if (t < P.NB) pre[t] = ex;
    if (t == 0) pre[P.NB] = total;
    __syncthreads();
    // greedy groups: from bucket i, the farthest end with <= cap2 records and <= span buckets; a bucket beyond cap2 on its
    // own goes to the big-bucket list; runs of empty buckets are skipped.  The step from EVERY bucket is computed in
    // parallel (binary searches on the prefix: one thread per bucket), then one thread follows the chain from bucket 0
    // -- one LDS read per group instead of ~9 dependent ones (the serial version was 60 of this kernel's 75 us).
    __shared__ unsigned step_to[kMaxNB + 1];           // next bucket | kind << 16  (kind 0 skip, 1 group, 2 big, 3 stop)
    if (t < P.NB) {
        const int i = t;
        const unsigned base = pre[i];
        unsigned st;
        if (base == total) {
            st = (3u << 16);                                           // nothing but empty buckets left
        } else if (pre[i + 1] == base) {                               // a run of empty buckets: to the first non-empty one
            int lo = i + 1, hi = P.NB;                                 // first j > i with pre[j] > base
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (pre[mid] > base) hi = mid; else lo = mid + 1;
            }
            st = (unsigned)(lo - 1);
        } else if (pre[i + 1] - base > (unsigned)P.cap2) {
            st = (unsigned)(i + 1) | (2u << 16);
        } else {
            int lo = i + 1, hi = i + P.span < P.NB ? i + P.span : P.NB;   // largest e in [lo, hi] with pre[e] - base <= cap2
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pre[mid] - base <= (unsigned)P.cap2) lo = mid; else hi = mid - 1;
            }
            st = (unsigned)lo | (1u << 16);
        }
        step_to[i] = st;
    }
    if (t == 0) step_to[P.NB] = 3u << 16;
    __syncthreads();
    __shared__ unsigned sgrp[kMaxNB];
    __shared__ unsigned sng;
    if (t == 0) {
        unsigned *grp = P.groups + (long long)seg * P.NB;
        unsigned ng = 0;
        int i = 0;
        while (i < P.NB) {
            const unsigned st = step_to[i];
            const unsigned kind = st >> 16, nx = st & 0xFFFFu;
            if (kind == 3u) break;
            if (kind == 1u) { sgrp[ng] = (unsigned)i | (nx << 16); grp[ng++] = (unsigned)i | (nx << 16); }
            else if (kind == 2u) P.big_list[atomicAdd(P.nbig, 1u)] = ((unsigned)seg << 16) | (unsigned)i;
            i = (int)nx;
        }
        P.ngroups[seg] = ng;
        sng = ng;
        P.seg_flag[seg] = 0;
        if (seg == 0 && P.fused_status && *P.fused_status) atomicExch(reinterpret_cast<unsigned *>(P.status), (unsigned)*P.fused_status);
    }
    __syncthreads();
    // every sort group's run in every tile (the table was read a moment ago: these loads hit L2), one contiguous row per group
    {
        const unsigned ng = sng;
        unsigned *out = P.gruns + (long long)seg * P.NB * P.Tp;
        // thread = (tile, group lane): no division, the groups' loads independent of each other
        int lgT = 3;
        while ((1 << lgT) < P.T) ++lgT;
        if (lgT <= 9) {
            const unsigned tt = (unsigned)t & ((1u << lgT) - 1u), gstep = 512u >> lgT;
            if (tt < (unsigned)P.T) {
                const unsigned short *row = tab + (long long)tt * (P.NB + 1);
#pragma unroll 4
                for (unsigned g = (unsigned)t >> lgT; g < ng; g += gstep) {
                    const unsigned gr = sgrp[g];
                    const unsigned r0 = row[gr & 0xFFFFu];
                    out[(long long)g * P.Tp + tt] = r0 | (((unsigned)row[gr >> 16] - r0) << 16);
                }
            }
        } else {
            for (unsigned idx = t; idx < ng * (unsigned)P.T; idx += 512) {
                const unsigned g = idx / (unsigned)P.T, tt = idx - g * (unsigned)P.T;
                const unsigned gr = sgrp[g];
                const unsigned short *row = tab + (long long)tt * (P.NB + 1);
                const unsigned r0 = row[gr & 0xFFFFu];
                out[(long long)g * P.Tp + tt] = r0 | (((unsigned)row[gr >> 16] - r0) << 16);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bucket sort: workgroup (256 threads) per (segment, bucket), at most K records per thread, kept in
// registers between the histogram and the rank phase.  LDS map (dynamic):
//   Out [256*K] u32 | hist [4][bins] u32 | tcnt/toff [T] | tpre [T+1] | part | stage [256*13] u32
// ---------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char sort_smem[];

template <bool PACKED, int K, int kSortThreads>
__global__ __launch_bounds__(kSortThreads, 4) void ldati_bucket_sort_kernel(LdatiParams P) {
    constexpr int kSortWaves = kSortThreads / 64;
    const int seg = blockIdx.y;
    // (the group word is requested together with the group count -- an entry beyond the count holds whatever the workspace held and
    // is not used --: one dependent global-load latency less in front of every working workgroup's setup)
    const unsigned ngr = P.ngroups[seg];
    const unsigned grp = P.groups[(long long)seg * P.NB + blockIdx.x];
    if (blockIdx.x >= ngr) return;                       // uniform per workgroup (flagged segments have no groups; empty
                                                         // workgroups cost nothing measurable: a compact group list changed nothing)
    const int bk0 = (int)(grp & 0xFFFFu), bk1 = (int)(grp >> 16);           // coarse buckets [bk0, bk1)
    const unsigned *bofs = P.bofs + (long long)seg * (P.NB + 1);
    const unsigned N = bofs[bk1] - bofs[bk0];
    const unsigned key0 = (unsigned)bk0 << P.shift;      // first key of the group
    const int b = seg / 9, c = seg - b * 9;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int bins = ((bk1 - bk0) << P.shift) * 4;       // (relative key, category); <= 4 * kMaxSpanKeys
    int nb2 = 2;
    while ((1 << nb2) < bins) ++nb2;
    const bool atomic_order = !P.ballot_ranks && __builtin_amdgcn_readfirstlane(g_lds_order_ok) != 0;

    unsigned *Out = reinterpret_cast<unsigned *>(sort_smem);
    unsigned *hist = Out + kSortThreads * K + 20;        // [kSortWaves][bins]  (Out: 20 more words, the slot phase of the emit loop)
    unsigned *ne_src = hist + kSortWaves * P.hist_bins;  // [T] per non-empty tile: source index - flat index
    unsigned *ne_info = ne_src + P.T;                    // [T] (polarity category << PB) | first pixel of the tile
    unsigned *bits = ne_info + P.T;                      // [K*8] bit i = a tile's run starts at flat index i
    constexpr int kWords = kSortThreads * K / 32;        // words of the run-start bit table
    unsigned *wpre = bits + kWords;                      // [kWords] run starts below each 32-index word
    unsigned *part = wpre + kWords;                      // [kSortWaves + 1]
    unsigned *stage = hist;                              // [256 * 13], 16-byte aligned; aliases the tables (dead in S5)

    STAMP_DECL;
    // S0 (wave 0; the other waves clear the histograms meanwhile): this bucket's run in every tile,
    // TPL consecutive tiles per lane; exclusive prefix of (records | non-empty << 16) over the tiles
    // by a wave scan; list of the non-empty tiles; bit table of the run starts and its word prefix.
    if (wid == 0) {
        auto setup = [&](auto tpl_c) {
        constexpr int TPL = decltype(tpl_c)::value;
        for (int i = lane; i < kWords; i += 64) bits[i] = 0;
        unsigned cv[TPL], ov[TPL];
        unsigned sum = 0;
        {
            // branch-free (tiles past the last one read the last one's entries and are masked): all 2 TPL loads are issued
            // back to back and waited for once -- behind a per-tile `if` each pair was waited for before the next was issued
            const unsigned *gr_row = P.gruns + ((long long)seg * P.NB + blockIdx.x) * P.Tp;
            const unsigned *ts_row = P.tile_src + (long long)seg * P.Tp;
            unsigned gr[TPL], ts[TPL];
#pragma unroll
            for (int q = 0; q < TPL; ++q) {
                const int tt = min(lane * TPL + q, P.T - 1);
                gr[q] = gr_row[tt];
                ts[q] = ts_row[tt];
            }
#pragma unroll
            for (int q = 0; q < TPL; ++q) {
                const bool in = lane * TPL + q < P.T;
                cv[q] = in ? gr[q] >> 16 : 0u;
                ov[q] = (gr[q] & 0xFFFFu) + ts[q];
                sum += cv[q] | (cv[q] ? 0x10000u : 0u);
            }
        }
        unsigned ex = wave_incl_scan(sum, lane) - sum;
#pragma unroll
        for (int q = 0; q < TPL; ++q) {
            if (cv[q]) {
                const unsigned tt = (unsigned)(lane * TPL + q), pre = ex & 0xFFFFu, j = ex >> 16;
                const bool neg = tt < (unsigned)P.tpp;
                ne_src[j] = ov[q] - pre;
                ne_info[j] = ((neg ? 0u : 2u) << P.PB) | ((neg ? tt : tt - (unsigned)P.tpp) * kTilePix);
                atomicOr(&bits[pre >> 5], 1u << (pre & 31u));
                ex += cv[q] | 0x10000u;
            }
        }
        // word prefix of the run starts: WPL consecutive words per lane
        constexpr int WPL = (kWords + 63) / 64;
        unsigned pc[WPL];
        unsigned ps = 0;
#pragma unroll
        for (int q = 0; q < WPL; ++q) {
            const int wd = lane * WPL + q;
            pc[q] = wd < kWords ? (unsigned)__popc(bits[wd]) : 0u;
            ps += pc[q];
        }
        unsigned pex = wave_incl_scan(ps, lane) - ps;
#pragma unroll
        for (int q = 0; q < WPL; ++q) {
            const int wd = lane * WPL + q;
            if (wd < kWords) wpre[wd] = pex;
            pex += pc[q];
        }
        };
        if (P.T <= 128) setup(std::integral_constant<int, 2>{}); else setup(std::integral_constant<int, kMaxTiles / 64>{});
    } else if (kSortWaves > 1) {
        for (int i = tid - 64; i < kSortWaves * bins; i += kSortThreads - 64) hist[i] = 0;
    }
    if (kSortWaves == 1)
        for (int i = tid; i < bins; i += 64) hist[i] = 0;
    __syncthreads();
    STAMP(0);
    // S1: gather.  The bucket's records are the tiles' runs in tile order (negative tiles first,
    // each in pixel order); wave w owns the contiguous flat range [lo, hi).  The tile of a flat
    // index = number of run starts at or below it (bit table + word prefix); every lane issues all
    // its loads before the first use.
    const unsigned L = ((N + kSortThreads - 1) / kSortThreads) * 64;
    const unsigned lo = wid * L, hi = (lo + L < N) ? lo + L : N;
    const unsigned *seg_temp = P.tile_abs ? P.temp : P.temp + P.seg_offsets[seg];
    unsigned rec[K];
    unsigned tinfo[K];
    {
        // branch-free and staged, so that every stage's LDS reads / global loads are issued back to
        // back and waited for once: indices past the wave's range are clamped (their loads are valid
        // and ignored)
        unsigned bw[K], pw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned i = min(lo + 64u * k + lane, N - 1u);
            bw[k] = bits[i >> 5];
            pw[k] = wpre[i >> 5];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned i = min(lo + 64u * k + lane, N - 1u);
            const unsigned j = pw[k] + (unsigned)__popc(bw[k] & (0xFFFFFFFFu >> (31u - (i & 31u)))) - 1u;
            tinfo[k] = ne_info[j];
            bw[k] = ne_src[j];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned i = min(lo + 64u * k + lane, N - 1u);
            rec[k] = seg_temp[bw[k] + i];
        }
    }
    STAMP(1);
    // S2: widen to (fine | category | global pixel); per-wave histograms of (fine, category).  A wave's range is whole
    // 64-record batches plus at most one partial one: the full batches run without per-lane guards (a scalar branch per
    // batch instead of an exec-mask sequence per record; the kernel is bound by instruction issue, not by memory)
    const int nfull = __builtin_amdgcn_readfirstlane((int)((hi > lo ? hi - lo : 0u) >> 6));
    const int ntail = __builtin_amdgcn_readfirstlane((int)((hi > lo ? hi - lo : 0u) & 63u));
    unsigned *myhist = hist + wid * bins;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k < nfull || (k == nfull && lane < ntail)) {     // first test: wave-uniform
            const unsigned r = rec[k];
            rec[k] = (((r >> 12) - key0) << (2 + P.PB)) | (tinfo[k] + (((r >> kLocalBits) & 1u) << P.PB) + (r & (kTilePix - 1)));
            atomicAdd(&myhist[rec[k] >> P.PB], 1u);
        }
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    // The packed emit loop (S5) walks SLOTS: four records per lane starting at a global record index that is a multiple of
    // 16; record i of the group is ranked into Out[i + oshift], its slot, so that a lane's four records are one aligned
    // 16-byte LDS read (they used to be four reads at a four-word lane stride: 8-way bank conflicts, and an index clamp each)
    const long long g0 = P.seg_offsets[seg] + bofs[bk0];                  // first global record
    const unsigned oshift = PACKED ? (unsigned)(g0 - (((g0 >> 2) & ~3ll) << 2)) : 0u;      // in [0, 15]
    // S3: bin-major, wave-minor exclusive scan
    {
        const int per = (bins + kSortThreads - 1) / kSortThreads;
        const int b0 = tid * per, b1 = (b0 + per < bins) ? b0 + per : bins;
        unsigned s = 0;
        for (int q = b0; q < b1; ++q)
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) s += hist[w * bins + q];
        unsigned tot;
        unsigned run = block_excl_scan<kSortWaves>(s, part, &tot) + oshift;
        for (int q = b0; q < b1; ++q) {
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) {
                const unsigned v = hist[w * bins + q];
                hist[w * bins + q] = run;
                run += v;
            }
        }
    }
    __syncthreads();
    STAMP(4);
    // S4: stable ranks
    if (atomic_order) {                                  // the rank IS the value the LDS atomic returns (g_lds_order_ok)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (k < nfull) {                             // wave-uniform
                const unsigned r = rec[k];
                Out[atomicAdd(&myhist[r >> P.PB], 1u)] = r;
            } else if (k == nfull && lane < ntail) {
                const unsigned r = rec[k];
                Out[atomicAdd(&myhist[r >> P.PB], 1u)] = r;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (lo + 64u * k < hi) {                     // wave-uniform
                const unsigned i = lo + 64u * k + lane;
                const bool has = i < hi;
                const unsigned r = rec[k];
                const unsigned bin = r >> P.PB;
                const unsigned pos = take_slots(has, bin, nb2, &myhist[bin]);
                if (has) Out[pos] = r;
            }
        }
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
    // S5: decode and write the final records
    const long long tbase = P.kbase[c] + (long long)key0 +
                            (P.frame_ts_add ? P.frame_ts_add[b] : 0);
    const unsigned pmask = (1u << P.PB) - 1u;
    const unsigned W = (unsigned)P.W;
    const float rcpW = 1.0f / (float)W;
    const bool small = P.HW < (1 << 24);                 // pixel indices exact in f32
    const bool tiny = P.HW < (1 << 22) - 1;              // ... and far enough from 2^22 for div_tiny
    if (!PACKED) {
        for (unsigned i = tid; i < N; i += kSortThreads) {
            const unsigned r = Out[i];
            const unsigned px = r & pmask, cat = (r >> P.PB) & 3u, fine = r >> (P.PB + 2);
            const unsigned yy = tiny ? div_tiny(px, rcpW) : small ? div_small(px, W, rcpW) : px / W;
            P.ts[g0 + i] = tbase + fine;
            P.x[g0 + i] = (short)(px - yy * W);
            P.y[g0 + i] = (short)yy;
            P.p[g0 + i] = (signed char)(cat >> 1);
        }
        return;
    }
    // Four records starting at a global record index that is a multiple of 4 are 13 whole dwords
    // (52 bytes).  Per iteration the workgroup assembles 256 such groups (13 312 bytes, starting at a
    // 16-byte aligned global address) in LDS and copies the image out 16 bytes per lane; only the
    // two 16-byte pieces at the ends of the bucket are written byte by byte.
    const long long gN = g0 + N;
    const long long G0 = (g0 >> 2) & ~3ll, G1 = (gN + 3) >> 2;         // G0 % 4 == 0: 52*G0 % 16 == 0
    const long long B0 = 13 * g0, B1 = 13 * gN;                         // this bucket's bytes
    const int nG_all = (int)(G1 - G0);                                 // groups of four records this bucket touches
    // one copy of the loop per division form (the kernel is bound by instruction issue: a uniform three-way branch per
    // record costs more than the division it selects)
    auto emit_loop = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;                  // 0: div_tiny, 1: div_small, 2: integer division
        // Every WAVE assembles and copies out its own 64 groups (3328 bytes = 208 16-byte pieces, 16-byte aligned in global
        // memory like the workgroup's image): no workgroup barrier inside the loop (round 4; the loop used to pay two per
        // 1024 records and was 45 % of the kernel's time)
        for (int ga = 0; ga < nG_all; ga += kSortThreads) {
            const int gw0 = ga + 64 * wid;
            const int g = gw0 + lane;
            if (g < nG_all) {
                unsigned A[4], Bh[4], C[4], D[4];
                // (slots outside the bucket hold whatever the LDS held: their bytes are never copied out)
                const uint4 r4 = reinterpret_cast<const uint4 *>(Out)[g];
                const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned r = rr[q];
                    const unsigned px = r & pmask, cat = (r >> P.PB) & 3u, fine = r >> (P.PB + 2);
                    const unsigned yy = MODE == 0 ? div_tiny(px, rcpW) : MODE == 1 ? div_small(px, W, rcpW) : px / W;
                    const long long tq = tbase + fine;
                    A[q] = (unsigned)tq;
                    Bh[q] = (unsigned)((unsigned long long)tq >> 32);
                    // (v_mul_lo_u32 issues at a quarter of the rate of v_mul_u32_u24; px < 2^24 in the two float-division modes)
                    // yy * W: only its low 16 bits are used, so the compiler drops __umul24's masks and falls back to the
                    // quarter-rate v_mul_lo_u32 -- hence the instruction by name (px < 2^24 in the two float-division modes)
                    unsigned yw;
                    if (MODE == 2) yw = yy * W;
                    else asm("v_mul_u32_u24 %0, %1, %2" : "=v"(yw) : "v"(yy), "v"(W));
                    C[q] = ((px - yw) & 0xFFFFu) | (yy << 16);
                    D[q] = cat >> 1;
                }
                unsigned *d = stage + tid * 13;
                d[0] = A[0]; d[1] = Bh[0]; d[2] = C[0];
                d[3] = D[0] | (A[1] << 8);
                d[4] = (A[1] >> 24) | (Bh[1] << 8);
                d[5] = (Bh[1] >> 24) | (C[1] << 8);
                d[6] = (C[1] >> 24) | (D[1] << 8) | (A[2] << 16);
                d[7] = (A[2] >> 16) | (Bh[2] << 16);
                d[8] = (Bh[2] >> 16) | (C[2] << 16);
                d[9] = (C[2] >> 16) | (D[2] << 16) | (A[3] << 24);
                d[10] = (A[3] >> 8) | (Bh[3] << 24);
                d[11] = (Bh[3] >> 8) | (C[3] << 24);
                d[12] = (C[3] >> 8) | (D[3] << 24);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                               // (one wave's LDS operations complete in order)
            const int left = nG_all - gw0;
            const int nG = left < 64 ? (left > 0 ? left : 0) : 64;
            const int nPieces = (nG * 52 + 15) >> 4;
            const long long img = 52 * (G0 + gw0);                         // global byte address of the wave's image
            // pieces [qa, qb) lie wholly inside this bucket's bytes (all but the first and the last of the bucket)
            const long long r0l = B0 - img, r1l = B1 - img;
            const int rel0 = (int)(r0l < -(1 << 20) ? -(1 << 20) : r0l), rel1 = (int)r1l;     // rel1 <= 13 * cap2 + 256
            const int qa = rel0 > 0 ? (rel0 + 15) >> 4 : 0, qb = rel1 >> 4;
            unsigned char *dst = P.packed + img;
            const unsigned *stage_w = stage + wid * (64 * 13);
            for (int q = lane; q < nPieces; q += 64) {
                if (q >= qa && q < qb) {
                    reinterpret_cast<uint4 *>(dst)[q] = reinterpret_cast<const uint4 *>(stage_w)[q];
                } else {
                    const unsigned char *sb = reinterpret_cast<const unsigned char *>(stage_w) + 16 * q;
                    for (int k = 0; k < 16; ++k)
                        if (16 * q + k >= rel0 && 16 * q + k < rel1) dst[16 * q + k] = sb[k];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (tiny) emit_loop(std::integral_constant<int, 0>{});
    else if (small) emit_loop(std::integral_constant<int, 1>{});
    else emit_loop(std::integral_constant<int, 2>{});
    STAMP(7);
    STAMP_FLUSH(16, 8);
}

// ---------------------------------------------------------------------------------------------
// big buckets: a coarse bucket with more records than a sort workgroup holds in LDS (degenerate ties).
// One workgroup per listed bucket: histogram of (fine key, category) over all its records (any
// order), exclusive scan, then ONE wave walks the records in input order (tiles ascending, pixel order
// inside a run) and takes stable slots 64 at a time; the records go straight to their final place.
// Rare and slow on purpose; keeps the two-level path complete for any input.
// ---------------------------------------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(256) void ldati_big_bucket_kernel(LdatiParams P) {
    __shared__ unsigned hist[4 << kMaxShift];
    __shared__ unsigned part[5];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool atomic_order = !P.ballot_ranks && __builtin_amdgcn_readfirstlane(g_lds_order_ok) != 0;
    const int bins = 4 << P.shift, nb2 = P.shift + 2;
    const unsigned nbig = *P.nbig;
    for (unsigned idx = blockIdx.x; idx < nbig; idx += gridDim.x) {
        const unsigned e = P.big_list[idx];
        const int seg = (int)(e >> 16), bucket = (int)(e & 0xFFFFu);
        const int b = seg / 9, c = seg - b * 9;
        const unsigned *bofs = P.bofs + (long long)seg * (P.NB + 1);
        const long long g0 = P.seg_offsets[seg] + bofs[bucket];
        const unsigned key0 = (unsigned)bucket << P.shift;
        const long long tbase = P.kbase[c] + (long long)key0 + (P.frame_ts_add ? P.frame_ts_add[b] : 0);
        const unsigned *seg_temp = P.tile_abs ? P.temp : P.temp + P.seg_offsets[seg];
        for (int i = tid; i < bins; i += 256) hist[i] = 0;
        __syncthreads();
        for (int tt = 0; tt < P.T; ++tt) {
            const unsigned short *row = P.roff + ((long long)seg * P.T + tt) * (P.NB + 1);
            const unsigned r0 = row[bucket], len = (unsigned)row[bucket + 1] - r0;
            const unsigned *src = seg_temp + P.tile_src[(long long)seg * P.Tp + tt] + r0;
            const unsigned catb = tt < P.tpp ? 0u : 2u;
            for (unsigned j = tid; j < len; j += 256) {
                const unsigned r = src[j];
                atomicAdd(&hist[(((r >> 12) - key0) << 2) | (catb + ((r >> kLocalBits) & 1u))], 1u);
            }
        }
        __syncthreads();
        {
            const int per = (bins + 255) / 256;
            const int b0 = tid * per, b1 = (b0 + per < bins) ? b0 + per : bins;
            unsigned s = 0;
            for (int q = b0; q < b1; ++q) s += hist[q];
            unsigned tot;
            unsigned run = block_excl_scan<4>(s, part, &tot);
            for (int q = b0; q < b1; ++q) {
                const unsigned v = hist[q];
                hist[q] = run;
                run += v;
            }
        }
        __syncthreads();
        if (tid < 64) {
            for (int tt = 0; tt < P.T; ++tt) {
                const unsigned short *row = P.roff + ((long long)seg * P.T + tt) * (P.NB + 1);
                const unsigned r0 = row[bucket], len = (unsigned)row[bucket + 1] - r0;
                const unsigned *src = seg_temp + P.tile_src[(long long)seg * P.Tp + tt] + r0;
                const unsigned catb = tt < P.tpp ? 0u : 2u;
                const unsigned pxb = (unsigned)(tt < P.tpp ? tt : tt - P.tpp) * kTilePix;
                for (unsigned j0 = 0; j0 < len; j0 += 64) {
                    const unsigned j = j0 + lane;
                    const bool has = j < len;
                    const unsigned r = has ? src[j] : 0u;
                    const unsigned cat = catb + ((r >> kLocalBits) & 1u);
                    const unsigned fine = has ? (r >> 12) - key0 : 0u;
                    const unsigned bin = (fine << 2) | cat;
                    const unsigned pos = take_slot(atomic_order, has, bin, nb2, &hist[bin]);
                    if (has) {
                        const unsigned px = pxb + (r & (kTilePix - 1));
                        const unsigned yy = px / (unsigned)P.W;
                        const long long g = g0 + pos;
                        if (PACKED) {
                            store_packed_bytes(P.packed + g * 13, tbase + fine, (px - yy * P.W) & 0xFFFFu, yy & 0xFFFFu, cat >> 1);
                        } else {
                            P.ts[g] = tbase + fine;
                            P.x[g] = (short)(px - yy * P.W);
                            P.y[g] = (short)yy;
                            P.p[g] = (signed char)(cat >> 1);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// pooled slope parameters (LDATI.py:177-190): thread per (frame, polarity, pixel).  The counts of the
// k x k neighbourhood are recomputed from the voxels (zero padding outside the image), pooled --
// 'weighted': [[1,2,1],[2,4,2],[1,2,1]]/16; 'avg': sum / k^2 -- and turned into {k, b} per bin.
// Counts are integers and the weights powers of two over 16: every partial sum is exact in f32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ldati_pool_slope_kernel(LdatiParams P, int pooling, int pool_k, float2 *kbb) {
    const int px = blockIdx.x * 256 + threadIdx.x, bp = blockIdx.y;
    if (px >= P.HW) return;
    const int h = px / P.W, w = px - h * P.W;
    const int r = pooling == V2CE_POOL_WEIGHTED ? 1 : pool_k / 2;
    const float *plane0 = P.vox + (long long)bp * 10 * P.HW;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.0f;
    for (int dh = -r; dh <= r; ++dh)
        for (int dw = -r; dw <= r; ++dw) {
            const int hh = h + dh, ww = w + dw;
            if (hh < 0 || hh >= P.H || ww < 0 || ww >= P.W) continue;
            float yv[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) yv[i] = plane0[(long long)i * P.HW + hh * P.W + ww];
            int nn[9];
            float td[9];
            relocate_all(yv, P.bidir != 0, nn, td);
            const float wt = pooling == V2CE_POOL_WEIGHTED ? (float)((2 - (dh < 0 ? -dh : dh)) * (2 - (dw < 0 ? -dw : dw))) / 16.0f : 1.0f;
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[i] += (float)nn[i] * wt;
        }
    if (pooling == V2CE_POOL_AVG) {
        const float div = (float)(pool_k * pool_k);
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = acc[i] / div;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const float fl = c == 0 ? acc[1] : acc[c - 1], fr = c == 8 ? acc[7] : acc[c + 1];
        const float sxy = fr - fl;
        const float k0 = (3.0f * sxy) / 6.0f;
        const float k = (k0 / P.VS2) / (acc[c] + 1e-8f);
        const float bb = P.INV - (P.VS * k) / 2.0f;
        kbb[((long long)bp * 9 + c) * P.HW + px] = make_float2(k, bb);
    }
}

// generic path: sorted 64-bit keys -> SoA events
__global__ __launch_bounds__(256) void ldati_keys_decode_kernel(LdatiParams P, const unsigned long long *keys, long long n,
                                                                long long *ts, short *x, short *y, signed char *pp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const int seg = (int)(k >> 44), b = seg / 9, c = seg - b * 9;
    const unsigned rel = (unsigned)(k >> 24) & 0xFFFFFu, cat = (unsigned)(k >> 22) & 3u, px = (unsigned)k & 0x3FFFFFu;
    const unsigned yy = px / (unsigned)P.W;
    ts[i] = P.kbase[c] + (long long)rel + (P.frame_ts_add ? P.frame_ts_add[b] : 0);
    x[i] = (short)(px - yy * P.W);
    y[i] = (short)yy;
    pp[i] = (signed char)(cat >> 1);
}

// ---------------------------------------------------------------------------------------------
// pack / unpack kernels: SoA <-> 13-byte records
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void events_pack_kernel(const long long *__restrict__ ts,
                                                          const short *__restrict__ x,
                                                          const short *__restrict__ y,
                                                          const signed char *__restrict__ p,
                                                          long long n, unsigned char *packed) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[256 * 13 + 12];
    const long long first = (long long)blockIdx.x * 256;
    const long long i = first + threadIdx.x;
    if (i < n)
        store_packed_bytes(stage + threadIdx.x * 13, ts[i], (unsigned short)x[i], (unsigned short)y[i],
                           (unsigned char)p[i]);
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

__global__ __launch_bounds__(256) void events_unpack_kernel(const unsigned char *__restrict__ packed,
                                                            long long n, long long *ts, short *x,
                                                            short *y, signed char *p) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[256 * 13 + 12];
    const long long first = (long long)blockIdx.x * 256;
    const long long remain = n - first;
    const int nev = remain < 256 ? (int)remain : 256;
    const int nbytes = nev * 13;
    const unsigned char *src = packed + first * 13;
    const int nd = nbytes >> 2;
    for (int w = threadIdx.x; w < nd; w += 256)
        reinterpret_cast<unsigned *>(stage)[w] = reinterpret_cast<const unsigned *>(src)[w];
    for (int r = (nd << 2) + threadIdx.x; r < nbytes; r += 256) stage[r] = src[r];
    __syncthreads();
    const long long i = first + threadIdx.x;
    if (i < n) {
        const unsigned char *s = stage + threadIdx.x * 13;
        unsigned long long t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t |= (unsigned long long)s[k] << (8 * k);
        ts[i] = (long long)t;
        x[i] = (short)(s[8] | (s[9] << 8));
        y[i] = (short)(s[10] | (s[11] << 8));
        p[i] = (signed char)s[12];
    }
}

// the few words a call needs zeroed before its first kernel (statistics, status words): one launch instead of the two to four
// fill kernels that hipMemsetAsync makes of two small, 8-byte-aligned ranges (~5 us each plus their gaps: 4 % of an e2e call)
// (not named ldati_*: the profile summaries tell a call's steady-state kernels from first-call-only ones by their launch counts, and this
// one runs twice per call)
__global__ void zero_words_kernel(unsigned *a, int na, unsigned *b, int nb) {
    const int i = threadIdx.x;
    if (i < na) a[i] = 0u;
    if (b && i < nb) b[i] = 0u;
}

// host-side scalars, computed exactly like CPU torch does
struct HostScalars {
    float VS, VS2, INV, FPS;
    float offt[9];
    long long kbase[9];
    long long NK;
    int nbits;
    size_t lds_bytes;     // of the sweep kernel
    bool sweep_ok;        // 4*NK counters fit the LDS
    bool ok;
};

HostScalars host_scalars(double fps, double t0, bool bidir = false, bool random = false) {
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
    // forward relocation: every timestamp lies in its bin.  Bidirectional (LDATI.py:107-122): a single
    // event's tendency lies in (-1, 2) bin widths (bin 5: bless - debt; bin 8: y[9] < 2 when n == 1).
    // 'random' (LDATI.py:173-174): the multi-event offsets are raw uniforms in SECONDS.
    const long long before = bidir ? span : 0;
    const long long after = (random ? 1000000 : 0) + (bidir ? span : 0);
    const long long nk = before + span + after + 2 * slack;
    for (int c = 0; c < 9; ++c) h.kbase[c] = (long long)((double)h.offt[c] * 1e6) - slack - before;
    h.NK = nk;
    // the two-level path's key range; the generic ('random') path only needs 20-bit keys
    h.ok = nk > 0 && (random ? nk < (1ll << 20) : nk <= ((long long)kMaxNB << kMaxShift));
    // sweep kernel: 4*NK*4 B must fit 160 KiB of LDS with the scratch beside it; forward relocation only
    h.sweep_ok = nk > 0 && nk <= 9600 && !bidir && !random;
    int nb = 0;
    while ((1ll << nb) < nk) ++nb;
    h.nbits = nb;
    h.lds_bytes = (size_t)(4 * nk + 256) * 4 + 3 * 128 * 4;
    return h;
}

// workgroup size of the tile pass: 1024 threads (2 pixels each) for dense tiles, whose LDS footprint
// allows one workgroup per CU anyway; 512 threads (4 pixels each, half the barrier traffic and
// histogram rows) for sparse ones.  V2CE_LDATI_TILE_THREADS = 512 | 1024 overrides (kernel A/B runs).
int tile_threads_choice(int64_t max_tile_events) {
    static const int v = [] {
        const char *e = getenv("V2CE_LDATI_TILE_THREADS");
        const int n = e ? atoi(e) : 0;
        return n == 512 || n == 1024 ? n : 0;
    }();
    return v ? v : (max_tile_events > 4096 ? 1024 : 512);
}

// dynamic LDS of ldati_tile_dense_kernel<NW>: S [capA] | O [capA + 2] | hist [NW][NB] | wave totals, scan partials, batch counter
size_t dense_tile_lds(int capA, int NB, int NW) {
    return ((size_t)2 * capA + 8 + (size_t)NW * NB + 3 * NW + NW + 1 + 2 + 18 + 10 + 2) * 4;
}

// words of one tile's work lists (ldati_tile_onepass_kernel): two per unit of four draws or single event -- at most one per event
// of a multi-event voxel, two per single, i.e. N + 2048 per bin --, four per voxel outside the slope table (>= 32 events each), and
// the passes' alignment; capA = the most events a (tile, bin) that is not merely counted can hold
size_t onepass_list_words(int capA) { return ((size_t)9 * ((size_t)capA + capA / 8 + kTilePix) + 64 + 3) & ~(size_t)3; }
// Opt-in (V2CE_LDATI_ONEPASS=1).  Measured in round 6 (profiles/r06_d_ldati_onepass_*): inside the pass loop the form does what it
// was built for -- classification and list building are gone (103 Mcycles per stress call), the histogram scan, copy-out and loop
// head shrink with the pair passes (94 -> 46) -- but the sweep in front of the loop costs ~150 Mcycles (54 wave scans, 36
// classifications and ~22 000 scattered 8-byte stores per tile, on a workgroup that is alone on its CU) and the timestamp phase
// 127 -> 210: a batch's list entries now come from L2 / the memory-side cache, one batch of prefetch does not cover that.  Stress
// chunk 698 -> 880 us, half density 480 -> 640, quarter density 400 -> 530.  The lists have to stay in LDS; there they only fit per bin.
bool onepass_enabled() { const char *e = getenv("V2CE_LDATI_ONEPASS"); return e && e[0] == '1'; }
// geometry and capacities of the two-level path
struct Plan {
    int tpp, T, Tp, shift, NB, nb1, PB, capA, cap2, tbits, tile_threads, span, sort_threads;
    size_t n_tab, n_bkt;                 // entries of roff; of bofs
    size_t lds_tile, lds_sort;
    size_t bytes;                        // workspace
    size_t list_words, off_lists;        // ldati_tile_onepass_kernel's work lists: words per tile, byte offset inside the workspace (0 words: off)
    bool ok;
};

Plan make_plan(const HostScalars &h, int B, int H, int W, int64_t total_events,
               int64_t max_segment_events, int64_t max_tile_events) {
    Plan p{};
    const long long HW = (long long)H * W;
    p.tpp = (int)((HW + kTilePix - 1) / kTilePix);
    p.T = 2 * p.tpp;
    p.Tp = (p.T + 7) & ~7;
    int pb = 1;
    while ((1ll << pb) < HW) ++pb;
    p.PB = pb;
    // coarse (level 1) bucket width 2^shift us: at most 16 us, finer when the densest segment would
    // put more than cap2/20 records into an AVERAGE bucket (on real UNet output the fullest bucket of a
    // segment holds ~20x the average: timestamps crowd at the end of a bin), never finer than kMaxNB
    // buckets allow.  The sort groups (bucket scan kernel) merge consecutive buckets up to cap2 records.
    // sort workgroups of 128 threads (3072 records) for segments of real UNet output, 256 (6144) for dense ones: measured on the
    // e2e step (densest segment 85 K events) 151 -> 98 us, sparse bench 93 -> 73 us, on the stress chunk 432 -> 490 us, pano sort
    // -66 us but bucket scan +80 us (V2CE_LDATI_SORT_THREADS overrides; kernel A/B runs)
    {
        const char *e = getenv("V2CE_LDATI_SORT_THREADS");            // (read per plan: tests switch it inside one process)
        const int ev = e ? atoi(e) : 0, forced = ev == 64 || ev == 128 || ev == 256 ? ev : 0;
        // (the densest segment spread evenly over its keys: a group of kMaxSpanKeys keys then holds at most 1.5 x 3072 records --
        // groups of such segments are closed by their key span, not by their record count)
        p.sort_threads = forced ? forced : (max_segment_events * kMaxSpanKeys > 4608 * h.NK ? 256 : 128);
    }
    const int kSortThreads = max_segment_events > 2048 ? p.sort_threads : 256, kSortWaves = kSortThreads / 64;
    p.sort_threads = kSortThreads;
    p.cap2 = max_segment_events > 2048 ? kSortThreads * 24 : kSortThreads * 8;
    int shift = 4;
    while (shift > 0 && (double)max_segment_events * (double)(1 << shift) / (double)h.NK > p.cap2 / 20.0) --shift;
    while (shift < kMaxShift && ((h.NK + (1ll << shift) - 1) >> shift) > kMaxNB) ++shift;
    // Dense segments (the rule above asks for the finest buckets) gain nothing from more than ~256 buckets: the sort groups
    // merge consecutive buckets up to cap2 records anyway, while the tile pass pays per (wave, bucket) cell and the gather per
    // run (V2CE_LDATI_NB_SOFT overrides the soft limit; kernel A/B runs)
    {
        static const int soft = [] { const char *e = getenv("V2CE_LDATI_NB_SOFT"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= kMaxNB ? v : kMaxNB; }();   // (measured: 256 = tile pass -36 us, sort +31 us, bucket scan +10 us on the stress chunk: off)
        while (shift < 4 && ((h.NK + (1ll << shift) - 1) >> shift) > soft) ++shift;
    }
    p.shift = shift;
    p.NB = (int)((h.NK + (1ll << shift) - 1) >> shift);
    int nb1 = 0;
    while ((1 << nb1) < p.NB) ++nb1;
    p.nb1 = nb1;
    // key span of a sort group: 128 keys for dense segments; for the small-group regime (128-thread workgroups) the groups are
    // closed by their span, not by their records, so a wider span means fewer, fuller groups (V2CE_LDATI_SPAN_KEYS: A/B runs)
    int span_keys = kMaxSpanKeys;
    {
        const char *e = getenv("V2CE_LDATI_SPAN_KEYS");
        const int ev = e ? atoi(e) : 0;
        if (ev == 128 || ev == 256 || ev == 512) span_keys = ev;
        else if (p.sort_threads == 128 && max_segment_events > 2048) span_keys = kSmallGroupSpanKeys;
    }
    p.span = (span_keys >> shift) > 0 ? (span_keys >> shift) : 1;
    p.tbits = 0;
    while ((1 << p.tbits) < p.T) ++p.tbits;
    p.capA = (int)((max_tile_events + 255) / 256 * 256);
    if (p.capA < 256) p.capA = 256;
    p.ok = h.ok && p.T <= kMaxTiles && p.NB <= kMaxNB && p.capA <= kCapTile &&
           pb <= 22 && total_events < (1ll << 32) && B * 9 <= 65535;
    p.n_bkt = (size_t)B * 9 * (size_t)(p.NB + 1);
    p.n_tab = p.n_bkt * (size_t)p.T;
    p.tile_threads = tile_threads_choice(max_tile_events);
    p.lds_tile = (size_t)(2 * p.capA + 2048) * 4 + (size_t)kTilePix * 8 +
                 (size_t)(p.tile_threads / 128) * p.NB * 4 + 2 * (p.tile_threads / 64 + 1) * 4;
    const size_t bins = (size_t)4 * (size_t)(p.span << shift);          // <= 4 * max(kMaxSpanKeys, 2^shift)
    const size_t hist_bins = bins > 4 * (size_t)kMaxSpanKeys ? bins : 4 * (size_t)kMaxSpanKeys;
    const size_t tables = kSortWaves * hist_bins * 4 + (size_t)(2 * p.T) * 4 + (size_t)(2 * (p.cap2 / 32)) * 4 + (kSortWaves + 1) * 4;
    const size_t stage = (size_t)kSortThreads * 13 * 4;
    p.lds_sort = (size_t)(p.cap2 + 20) * 4 + (tables > stage ? tables : stage);
    // bofs | groups [B*9*NB] | big_list [B*9*NB] | ngroups [B*9] | seg_flag [B*9] | status [4] (status, nbig) |
    // records (u32) | roff (u16) | gruns (u32 [B*9][NB][Tp])
    p.bytes = (p.n_bkt + 2 * (size_t)B * 9 * p.NB + 2 * (size_t)B * 9 + 4 +
               (size_t)(total_events > 0 ? total_events : 0)) * 4 + ((p.n_tab * 2 + 3) / 4) * 4 + (size_t)B * 9 * p.NB * p.Tp * 4;
    if (p.lds_tile > 160 * 1024 || p.lds_sort > 160 * 1024) p.ok = false;
    p.list_words = 0;
    p.off_lists = (p.bytes + 15) / 16 * 16;
    if (onepass_enabled() && p.capA <= kCapTile) {
        p.list_words = onepass_list_words(p.capA);
        p.bytes = p.off_lists + (size_t)B * p.T * p.list_words * 4;
    }
    return p;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

namespace {
struct Opts { int strategy, bidir, pooling, pool_k; };

int probe_lds_order(hipStream_t s) {   // once per device: check that LDS atomics return ranks in lane order (see g_lds_order_ok)
    static std::atomic<unsigned long long> probed{0};
    int dev = 0;
    V2CE_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(probed.fetch_or(bit) & bit))
        hipLaunchKernelGGL(ldati_lds_order_probe_kernel, dim3(64), dim3(256), 0, s, 200);
    return V2CE_OK;
}

// NULL options = the CLI's call (v2ce.py:356): 'slope', no pooling, forward relocation
int read_options(const v2ce_ldati_options *o, Opts &out, const char *who) {
    out = Opts{V2CE_STRATEGY_SLOPE, 0, V2CE_POOL_NONE, 3};
    if (!o) return V2CE_OK;
    out = Opts{o->strategy, o->bidirectional != 0, o->pooling_type, o->pooling_kernel_size};
    V2CE_REQUIRE(out.strategy == V2CE_STRATEGY_SLOPE || out.strategy == V2CE_STRATEGY_NONE || out.strategy == V2CE_STRATEGY_RANDOM,
                 V2CE_ERR_BAD_ARG, "%s: bad strategy %d", who, out.strategy);
    V2CE_REQUIRE(out.pooling == V2CE_POOL_NONE || out.pooling == V2CE_POOL_AVG || out.pooling == V2CE_POOL_WEIGHTED,
                 V2CE_ERR_BAD_ARG, "%s: bad pooling_type %d", who, out.pooling);
    V2CE_REQUIRE(out.pooling != V2CE_POOL_AVG || (out.pool_k >= 1 && out.pool_k <= 15 && (out.pool_k & 1)), V2CE_ERR_UNSUPPORTED,
                 "%s: pooling_kernel_size %d (odd sizes 1..15: nn.AvgPool2d with padding k//2 keeps H x W only for odd k)",
                 who, out.pool_k);
    if (out.strategy != V2CE_STRATEGY_SLOPE) out.pooling = V2CE_POOL_NONE;      // pooling only shapes the slope (LDATI.py:175)
    return V2CE_OK;
}

// workspace of one emit call: the two-level part (or the generic part for 'random'), then the pooled
// slope parameters
struct Layout {
    Plan plan;
    bool generic;
    size_t main_bytes, keys_bytes, sort_temp_bytes, soa_bytes, kbb_bytes, bytes;
    bool ok;
};

Layout make_layout(const HostScalars &h, const Opts &o, int B, int H, int W, int64_t total, int64_t max_seg,
                   int64_t max_tile, bool packed_out) {
    Layout L{};
    L.generic = o.strategy == V2CE_STRATEGY_RANDOM;
    L.plan = make_plan(h, B, H, W, total, max_seg, max_tile);
    const size_t n = (size_t)(total > 0 ? total : 0);
    if (L.generic) {
        // the tile pass still runs (in its key-writing mode): it needs the tile geometry and LDS plan
        L.ok = h.ok && L.plan.T <= kMaxTiles && L.plan.capA <= kCapTile && L.plan.PB <= 22 && B * 9 <= 65535 &&
               total < (1ll << 32) && L.plan.lds_tile <= 160 * 1024;
        L.main_bytes = 16;                                            // status words
        L.keys_bytes = 2 * ((n * 8 + 15) / 16) * 16;
        size_t tb = 0;
        if (n) (void)rocprim::radix_sort_keys(nullptr, tb, (unsigned long long *)nullptr, (unsigned long long *)nullptr, n, 0, 60);
        L.sort_temp_bytes = (tb + 15) / 16 * 16;
        L.soa_bytes = packed_out ? ((n * 13 + 15) / 16) * 16 + 64 : 0;
    } else {
        L.ok = L.plan.ok;
        L.main_bytes = (L.plan.bytes + 15) / 16 * 16;
    }
    L.kbb_bytes = o.pooling != V2CE_POOL_NONE ? (size_t)B * 2 * 9 * (size_t)H * W * 8 : 0;
    L.bytes = L.main_bytes + L.keys_bytes + L.sort_temp_bytes + L.soa_bytes + L.kbb_bytes;
    return L;
}
}  // namespace

extern "C" size_t v2ce_ldati_tile_ws_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const long long tpp = ((long long)H * W + kTilePix - 1) / kTilePix;
    const long long Tp = (2 * tpp + 7) & ~7ll;
    return (size_t)2 * (size_t)B * (size_t)(2 * tpp) * 9 * 4 + (size_t)B * 9 * (size_t)Tp * 4;     // tile counts | tile offsets | the offsets [segment][tile]
}

extern "C" int v2ce_ldati_count(const float *vox, int B, int H, int W, const v2ce_ldati_options *options, void *tile_ws,
                                size_t tile_ws_bytes, int64_t *seg_offsets, int64_t *stats,
                                v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(vox && tile_ws && seg_offsets && stats, V2CE_ERR_BAD_ARG, "v2ce_ldati_count: null pointer");
    V2CE_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_count: bad shape B=%d H=%d W=%d", B, H, W);
    V2CE_REQUIRE(W <= 32767 && H <= 32767, V2CE_ERR_UNSUPPORTED,
                 "v2ce_ldati_count: x/y are int16 (LDATI.py:230-231)");
    V2CE_REQUIRE(B <= 65535, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_count: B too large for one launch");
    Opts o;
    if (int rc = read_options(options, o, "v2ce_ldati_count")) return rc;
    V2CE_REQUIRE(tile_ws_bytes >= v2ce_ldati_tile_ws_bytes(B, H, W), V2CE_ERR_WORKSPACE,
                 "v2ce_ldati_count: tile workspace %zu < %zu", tile_ws_bytes, v2ce_ldati_tile_ws_bytes(B, H, W));
    hipStream_t s = as_stream(stream);
    const int HW = H * W;
    const int tpp = (HW + kTilePix - 1) / kTilePix, T = 2 * tpp;
    unsigned *tc = static_cast<unsigned *>(tile_ws);
    unsigned *tile_off = tc + (size_t)B * T * 9;
    if (int rc = probe_lds_order(s)) return rc;
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned *>(stats), 8, static_cast<unsigned *>(nullptr), 0);
    // 'random' emits like 'slope' (every draw of a multi-event voxel); 'none' only the singles
    const int count_strategy = o.strategy == V2CE_STRATEGY_NONE ? V2CE_STRATEGY_NONE : V2CE_STRATEGY_SLOPE;
    hipLaunchKernelGGL(ldati_count_tiles_kernel, dim3(T, B), dim3(kCountThreads), 0, s, vox, HW, tpp, count_strategy, o.bidir, tc,
                       reinterpret_cast<unsigned long long *>(stats));
    hipLaunchKernelGGL(ldati_tile_scan_kernel, dim3((B + 3) / 4), dim3(256), 0, s, tc, B, T, tile_off, tile_off + (size_t)B * T * 9, (T + 7) & ~7,
                       reinterpret_cast<long long *>(seg_offsets), reinterpret_cast<unsigned long long *>(stats));
    hipLaunchKernelGGL(ldati_seg_scan_kernel, dim3(1), dim3(256), 0, s, B * 9, reinterpret_cast<long long *>(seg_offsets),
                       reinterpret_cast<unsigned long long *>(stats));
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" size_t v2ce_ldati_lds_bytes(double fps, double t0) {
    if (!(fps > 0)) return 0;
    const HostScalars h = host_scalars(fps, t0);
    return h.sweep_ok ? h.lds_bytes : 0;
}

extern "C" size_t v2ce_ldati_workspace_bytes(int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                                             int64_t total_events, int64_t max_segment_events,
                                             int64_t max_tile_events, int packed_output) {
    if (!(fps > 0) || B <= 0 || H <= 0 || W <= 0) return 0;
    Opts o;
    if (read_options(options, o, "v2ce_ldati_workspace_bytes")) return 0;
    const HostScalars h = host_scalars(fps, t0, o.bidir, o.strategy == V2CE_STRATEGY_RANDOM);
    const Layout L = make_layout(h, o, B, H, W, total_events, max_segment_events, max_tile_events, packed_output != 0);
    return L.ok ? L.bytes : 0;
}

namespace {
// ---- fused count + sparse tile pass: geometry the kernel assumes BEFORE the counts exist, and its workspace ----------
// The coarse-bucket geometry of a call follows from its densest segment (make_plan).  The fused kernel runs before that is
// known, with the geometry of the caller's HINT (the previous call's max_segment_events: consecutive batches of a clip
// agree); v2ce_ldati_emit_fused uses its records only if the plan made from the real counts has the same geometry and no
// tile exceeded its slot.
struct FusedLayout {
    Plan p0;
    size_t off_abs, off_rec, off_roff, off_lists, bytes;
    int slot_cap;                        // dense mode: records per (tile, bin) slot (= the tile pass's LDS capacity); 0 = sparse mode
    bool ok;
};
// tile_bin_hint = 0: the sparse kernel's fused form (a slot of kSparseCap records per tile);  > 0: the dense kernel's (a slot per
// (tile, bin), sized from the caller's expectation of the densest one)
FusedLayout make_fused_layout(const HostScalars &h, const Opts &o, int B, int H, int W, int64_t seg_hint, int64_t tile_bin_hint = 0) {
    FusedLayout F{};
    F.p0 = make_plan(h, B, H, W, 0, seg_hint > 0 ? seg_hint : 0, tile_bin_hint > 0 ? tile_bin_hint : 0);
    const Plan &p = F.p0;
    F.slot_cap = tile_bin_hint > 0 ? p.capA : 0;
    const size_t n_abs = (size_t)B * 9 * p.Tp, n_rec = F.slot_cap ? (size_t)B * p.T * 9 * (size_t)F.slot_cap : (size_t)B * p.T * kSparseCap;
    F.ok = h.ok && p.T <= kMaxTiles && p.NB <= kMaxNB && p.PB <= 22 && B * 9 <= 65535 && n_rec < (1ull << 32) &&
           (F.slot_cap ? (p.capA <= kCapTile && !o.bidir && dense_tile_lds(p.capA, p.NB, 16) <= 160 * 1024 && !getenv("V2CE_LDATI_OLD_TILE") &&
                          n_rec <= (1ull << 30))                 // (at most 4 GiB of slots: beyond that the count pass is the cheaper price)
                       : 9ll * ((long long)p.NB << p.shift) < (1ll << 20) && !getenv("V2CE_LDATI_NO_SPARSE")) &&
           (o.strategy == V2CE_STRATEGY_SLOPE || o.strategy == V2CE_STRATEGY_NONE) && o.pooling == V2CE_POOL_NONE &&
           !getenv("V2CE_LDATI_NO_FUSED");
    F.off_abs = 16;
    F.off_rec = (F.off_abs + n_abs * 4 + 15) / 16 * 16;
    F.off_roff = (F.off_rec + n_rec * 4 + 15) / 16 * 16;
    F.off_lists = (F.off_roff + p.n_tab * 2 + 15) / 16 * 16;
    F.bytes = F.off_lists + (F.slot_cap && p.list_words ? (size_t)B * p.T * p.list_words * 4 : 0);
    return F;
}

int fill_params(LdatiParams &P, const HostScalars &h, const Opts &o, const float *vox, int B, int H, int W, double fps,
                int rng_mode, const float *uniforms, int replay_max_n, uint64_t seed, int64_t frame_base, hipStream_t st) {
    P.vox = vox; P.B = B; P.H = H; P.W = W; P.HW = H * W;
    P.fps = fps; P.VS = h.VS; P.VS2 = h.VS2; P.INV = h.INV; P.FPS = h.FPS;
    for (int c = 0; c < 9; ++c) { P.offt[c] = h.offt[c]; P.kbase[c] = h.kbase[c]; }
    P.NK = (int)h.NK; P.nbits = h.nbits;
    P.ts32 = (fabs((double)h.offt[8]) + 1.0 + 1.0 / fps) * 1e6 < 2.0e9 ? 1 : 0;
    P.strategy = o.strategy; P.bidir = o.bidir;
    P.rng_mode = rng_mode; P.uniforms = uniforms; P.replay_max_n = replay_max_n;
    P.seed = seed; P.frame_base = frame_base;
    P.sweep_ok = h.sweep_ok ? 1 : 0;
    P.RFPS = (float)(1.0 / (double)h.FPS); P.R9 = (float)(1.0 / 9.0);
    P.RFPS64 = 1.0 / fps; P.R9_64 = 1.0 / 9.0;
    P.fast_slot = -1;
    if (o.strategy == V2CE_STRATEGY_SLOPE && !getenv("V2CE_LDATI_NO_FASTDIV")) {
        // once per device and FPS: the exhaustive check of k0_time_fast against the IEEE divisions (see g_fastdiv) and the
        // table of slope parameters (g_slope_tab)
        static std::mutex mu;
        static unsigned seen[64][8];
        static int n_seen[64];
        int dev = 0;
        V2CE_HIP_CHECK(hipGetDevice(&dev));
        unsigned bits;
        memcpy(&bits, &h.FPS, 4);
        std::lock_guard<std::mutex> g(mu);
        if (dev >= 0 && dev < 64) {
            int slot = -1;
            for (int i = 0; i < n_seen[dev]; ++i)
                if (seen[dev][i] == bits) slot = i;
            if (slot < 0 && n_seen[dev] < 7) {                   // (slot 7: v2ce_ldati_selfcheck's scratch)
                slot = n_seen[dev]++;
                seen[dev][slot] = bits;
                hipLaunchKernelGGL(ldati_fastdiv_check_kernel, dim3(65536), dim3(256), 0, st, h.FPS, P.RFPS, P.R9, slot);
                hipLaunchKernelGGL(ldati_fast64_check_kernel, dim3((0x3F800000u + kFast64Neg + 255u) / 256u), dim3(256), 0, st, fps, P.RFPS64,
                                   P.R9_64, slot);
                hipLaunchKernelGGL(ldati_slope_tab_kernel, dim3((kSlopeTab + 255) / 256), dim3(256), 0, st, h.VS, h.VS2, h.INV, slot);
                hipLaunchKernelGGL(ldati_fastdiv_commit_kernel, dim3(1), dim3(1), 0, st, h.FPS, slot);
            }
            P.fast_slot = slot;
        }
    }
    { const char *e = getenv("V2CE_LDATI_NO_ATOMIC_ORDER"); P.ballot_ranks = (e && e[0] == '1') ? 1 : 0; }
    return V2CE_OK;
}

}  // namespace

namespace {
// the round-4 dense tile kernel serves the common call (forward relocation, 'slope' with the device tables of this fps or 'none',
// no pooling, 32-bit times); everything else stays on the per-bin kernel
bool dense_kernel_serves(const LdatiParams &P, const Opts &o) {
    return !o.bidir && !P.kbb && P.ts32 && !getenv("V2CE_LDATI_OLD_TILE") &&
           (o.strategy == V2CE_STRATEGY_NONE || (o.strategy == V2CE_STRATEGY_SLOPE && P.fast_slot >= 0));
}
// the pair-pass kernel's host-side conditions (its device-side ones -- the tables of this fps, the LDS order probe -- are read
// in the kernel, which runs the per-bin body when one fails)
bool dense_kernel_serves_pair(const LdatiParams &P) {
    return P.rng_mode == V2CE_RNG_PHILOX && P.strategy == V2CE_STRATEGY_SLOPE && !P.ballot_ranks && P.ts32 && (P.HW & 3) == 0 &&
           P.fast_slot >= 0;
}
// dynamic LDS of ldati_tile_pair_kernel<NW>'s pair body: S [capP] | O [capP + 16] | hist [NW][2^(12 - shift)] | wave totals, partials, ...
size_t dense_pair_lds(int capP, int shift, int NW) {
    return ((size_t)2 * capP + 16 + ((size_t)NW << (12 - shift)) + 6 * NW + NW + 1 + 2 + 18 + 10 + kTilePix + 3) * 4;
}
// records of one pass the pair body can hold in `budget` bytes of LDS (a multiple of 256), 0 when the call has no pair form:
// 12-bit keys (the bin of a record rides above them), one histogram word per thread in the scan, 14-bit positions
int pair_capacity(const LdatiParams &P, const Plan &pl, int NW, size_t budget) {
    // Opt-in (V2CE_LDATI_PAIR=1; V2CE_LDATI_NO_PAIR=1 wins): measured in round 6 (profiles/r06_a_ldati_pair_*), the pair passes
    // save what they were built to save -- the scan of the histogram, the barriers, the copy-out prologue: 49 -> 30, 36 -> 49 (!),
    // 26 -> 19 Mcycles per stress call -- and lose more than that in the two phases that turned out NOT to be latency: the
    // classification and the work lists are bound by VALU issue (all sixteen waves run the same ~100 + ~100 instructions at
    // once, four per SIMD), their work doubles with the second bin and what is shared -- one barrier -- is small against it
    // (D1 45 -> 90, D2 58 -> 126 Mcycles).  Stress chunk 691 -> 731 us; half density 481 -> 480; quarter density 402 -> 374.
    const char *on = getenv("V2CE_LDATI_PAIR");
    if (!(on && on[0] == '1') || getenv("V2CE_LDATI_NO_PAIR")) return 0;
    if (P.NK > 4096 || pl.shift > 8 || (1 << (12 - pl.shift)) > 64 * NW) return 0;
    const size_t fixed = dense_pair_lds(0, pl.shift, NW);
    if (fixed + 2048 > budget) return 0;
    long long cap = (long long)((budget - fixed) / 8) & ~255ll;
    if (cap > kCapTile) cap = kCapTile;
    return cap >= pl.capA ? (int)cap : 0;
}
// dynamic LDS of ldati_tile_onepass_kernel<NW>'s body: S [capP] | O [capP + 16] | hist [NW][2^(12 - shift)] | wave prefixes [27][NW] | totals,
// plan, pass table, partials, ...
size_t dense_onepass_lds(int capP, int shift, int NW) {
    return ((size_t)2 * capP + 16 + ((size_t)NW << (12 - shift)) + 27 * NW + 27 + 9 + 108 + NW + 1 + 2 + 18 + 4 + 3) * 4;
}
int onepass_capacity(const LdatiParams &P, const Plan &pl, int NW, size_t budget) {
    if (P.NK > 4096 || pl.shift > 8 || (1 << (12 - pl.shift)) > 64 * NW) return 0;
    const size_t fixed = dense_onepass_lds(0, pl.shift, NW);
    if (fixed + 2048 > budget) return 0;
    long long cap = (long long)((budget - fixed) / 8) & ~255ll;
    if (cap > kCapTile) cap = kCapTile;
    return cap >= pl.capA ? (int)cap : 0;
}
int launch_dense_kernel(const LdatiParams &P, const Plan &pl, int B, hipStream_t st) {
    const size_t lds8 = dense_tile_lds(pl.capA, pl.NB, 8), lds16 = dense_tile_lds(pl.capA, pl.NB, 16);
    if (P.lists && dense_kernel_serves_pair(P)) {           // the one-pass classification (round 6, second form)
        static const int force = [] { const char *e = getenv("V2CE_LDATI_DENSE_NW"); return e ? atoi(e) : 0; }();
        const int cap8 = onepass_capacity(P, pl, 8, 80 * 1024), cap16 = onepass_capacity(P, pl, 16, 160 * 1024);
        const bool p8 = force == 16 ? false : cap8 > 0;
        const int capP = p8 ? cap8 : cap16;
        if (capP > 0) {
            LdatiParams Q = P;
            Q.capP = capP;
            const int nw = p8 ? 8 : 16;
            size_t lds = dense_onepass_lds(capP, pl.shift, nw);
            const size_t single = dense_tile_lds(pl.capA, pl.NB, nw);
            if (single > lds) lds = single;
            if (lds <= 160 * 1024) {
                auto ok = p8 ? ldati_tile_onepass_kernel<8> : ldati_tile_onepass_kernel<16>;
                V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ok), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(ok, dim3(pl.T, B), dim3(64 * nw), lds, st, Q);
                return V2CE_OK;
            }
        }
    }
    static const int force_nw = [] { const char *e = getenv("V2CE_LDATI_DENSE_NW"); return e ? atoi(e) : 0; }();   // kernel A/B runs
    // round 6: two bins per pass where their records fit the LDS together (dense_pair_body); 512 threads when the largest single
    // (tile, bin) fits two workgroups per CU in that form too
    if (dense_kernel_serves_pair(P)) {
        const int cap8 = pair_capacity(P, pl, 8, 80 * 1024), cap16 = pair_capacity(P, pl, 16, 160 * 1024);
        const bool p8 = force_nw == 16 ? false : cap8 > 0;
        const int capP = p8 ? cap8 : cap16;
        if (capP > 0) {
            LdatiParams Q = P;
            Q.capP = capP;
            const int nw = p8 ? 8 : 16;
            size_t lds = dense_pair_lds(capP, pl.shift, nw);
            const size_t single = dense_tile_lds(pl.capA, pl.NB, nw);          // (a call that is not the common one runs the per-bin body)
            if (single > lds) lds = single;
            if (lds <= 160 * 1024) {
                auto pk = p8 ? ldati_tile_pair_kernel<8> : ldati_tile_pair_kernel<16>;
                V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(pk, dim3(pl.T, B), dim3(64 * nw), lds, st, Q);
                return V2CE_OK;
            }
        }
    }
    const bool w8 = force_nw == 16 ? false : (force_nw == 8 && lds8 <= 160 * 1024) ? true : lds8 <= 80 * 1024;       // two workgroups per CU
    auto dk = w8 ? ldati_tile_dense_kernel<8> : ldati_tile_dense_kernel<16>;
    const size_t lds = w8 ? lds8 : lds16;
    V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(dk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(dk, dim3(pl.T, B), dim3(w8 ? 512 : 1024), lds, st, P);
    return V2CE_OK;
}
}  // namespace

extern "C" size_t v2ce_ldati_fused_ws_bytes(int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                                            int64_t expected_max_segment_events, int64_t expected_max_tile_bin_events) {
    if (!(fps > 0) || B <= 0 || H <= 0 || W <= 0 || (long long)H * W >= (1ll << 30)) return 0;
    Opts o;
    if (read_options(options, o, "v2ce_ldati_fused_ws_bytes")) return 0;
    const HostScalars h = host_scalars(fps, t0, o.bidir, o.strategy == V2CE_STRATEGY_RANDOM);
    const FusedLayout F = make_fused_layout(h, o, B, H, W, expected_max_segment_events, expected_max_tile_bin_events);
    return F.ok ? F.bytes : 0;
}

extern "C" int v2ce_ldati_count_fused(const float *vox, int B, int H, int W, double fps, double t0,
                                      const v2ce_ldati_options *options, int rng_mode, const float *uniforms, int replay_max_n,
                                      uint64_t seed, int64_t frame_base, int64_t expected_max_segment_events,
                                      int64_t expected_max_tile_bin_events, void *tile_ws,
                                      size_t tile_ws_bytes, void *fused_ws, size_t fused_ws_bytes, int64_t *seg_offsets, int64_t *stats,
                                      v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(vox && tile_ws && fused_ws && seg_offsets && stats, V2CE_ERR_BAD_ARG, "v2ce_ldati_count_fused: null pointer");
    V2CE_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_count_fused: bad shape B=%d H=%d W=%d", B, H, W);
    V2CE_REQUIRE(W <= 32767 && H <= 32767, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_count_fused: x/y are int16 (LDATI.py:230-231)");
    V2CE_REQUIRE(B <= 65535, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_count_fused: B too large for one launch");
    V2CE_REQUIRE(fps > 0, V2CE_ERR_BAD_ARG, "v2ce_ldati_count_fused: fps must be positive");
    V2CE_REQUIRE(rng_mode == V2CE_RNG_REPLAY || rng_mode == V2CE_RNG_PHILOX, V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_count_fused: bad rng_mode %d", rng_mode);
    V2CE_REQUIRE(rng_mode != V2CE_RNG_REPLAY || replay_max_n == 0 || uniforms != nullptr, V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_count_fused: REPLAY mode needs the uniform tensor");
    Opts o;
    if (int rc = read_options(options, o, "v2ce_ldati_count_fused")) return rc;
    const HostScalars h = host_scalars(fps, t0, o.bidir, false);
    const FusedLayout F = make_fused_layout(h, o, B, H, W, expected_max_segment_events, expected_max_tile_bin_events);
    V2CE_REQUIRE(F.ok, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_count_fused: these arguments have no fused path (v2ce_ldati_fused_ws_bytes = 0)");
    V2CE_REQUIRE(tile_ws_bytes >= v2ce_ldati_tile_ws_bytes(B, H, W), V2CE_ERR_WORKSPACE,
                 "v2ce_ldati_count_fused: tile workspace %zu < %zu", tile_ws_bytes, v2ce_ldati_tile_ws_bytes(B, H, W));
    V2CE_REQUIRE(fused_ws_bytes >= F.bytes, V2CE_ERR_WORKSPACE, "v2ce_ldati_count_fused: workspace %zu < %zu", fused_ws_bytes, F.bytes);
    V2CE_REQUIRE((reinterpret_cast<uintptr_t>(fused_ws) & 15) == 0, V2CE_ERR_BAD_ARG, "v2ce_ldati_count_fused: workspace must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    const Plan &pl = F.p0;
    unsigned *tc = static_cast<unsigned *>(tile_ws);
    unsigned *tile_off = tc + (size_t)B * pl.T * 9;
    if (int rc = probe_lds_order(s)) return rc;
    unsigned char *fb = static_cast<unsigned char *>(fused_ws);
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned *>(stats), 16, reinterpret_cast<unsigned *>(fb), 4);
    LdatiParams P{};
    if (int rc = fill_params(P, h, o, vox, B, H, W, fps, rng_mode, uniforms, replay_max_n, seed, frame_base, s)) return rc;
    P.shift = pl.shift; P.NB = pl.NB; P.nb1 = pl.nb1; P.T = pl.T; P.Tp = pl.Tp; P.tpp = pl.tpp; P.PB = pl.PB;
    P.sparse_cap = kSparseCap;
    P.status = reinterpret_cast<int *>(fb);
    P.tc_w = tc;
    P.stats_w = reinterpret_cast<unsigned long long *>(stats);
    P.tile_abs_w = reinterpret_cast<unsigned *>(fb + F.off_abs);
    P.temp = reinterpret_cast<unsigned *>(fb + F.off_rec);
    P.roff = reinterpret_cast<unsigned short *>(fb + F.off_roff);
    if (F.slot_cap && pl.list_words) {
        P.lists = reinterpret_cast<unsigned *>(fb + F.off_lists);
        P.list_stride = (int)pl.list_words;
    }
    if (F.slot_cap && dense_kernel_serves(P, o)) {
        // dense regime: ldati_tile_dense_kernel is the count pass and the tile pass at once (slot mode)
        P.sparse_cap = 0;
        P.slot_cap = F.slot_cap;
        P.capA = pl.capA;
        launch_dense_kernel(P, pl, B, s);
    } else if (F.slot_cap) {
        // (a call the dense kernel does not serve -- more than seven fps values on this device, 64-bit times: the plain count
        // pass, so that the emit phase finds valid counts and takes the two-pass path)
        hipLaunchKernelGGL(ldati_count_tiles_kernel, dim3(pl.T, B), dim3(kCountThreads), 0, s, vox, H * W, pl.tpp, o.strategy, o.bidir, tc,
                           reinterpret_cast<unsigned long long *>(stats));
    } else {
    V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ldati_tile_sparse_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSparseLds));
    hipLaunchKernelGGL(ldati_tile_sparse_kernel<true>, dim3(pl.T, B), dim3(kSparseThreads), kSparseLds, s, P);
    }
    hipLaunchKernelGGL(ldati_tile_scan_kernel, dim3((B + 3) / 4), dim3(256), 0, s, tc, B, pl.T, tile_off, tile_off + (size_t)B * pl.T * 9, pl.Tp,
                       reinterpret_cast<long long *>(seg_offsets), reinterpret_cast<unsigned long long *>(stats));
    hipLaunchKernelGGL(ldati_seg_scan_kernel, dim3(1), dim3(256), 0, s, B * 9, reinterpret_cast<long long *>(seg_offsets),
                       reinterpret_cast<unsigned long long *>(stats));
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

namespace {
int emit_impl(const float *vox, int B, int H, int W, double fps, double t0,
                               const v2ce_ldati_options *options, int rng_mode, const float *uniforms, int replay_max_n,
                               uint64_t seed, int64_t frame_base, const int64_t *seg_offsets,
                               const int64_t *frame_ts_add, int64_t *ts, int16_t *x, int16_t *y,
                               int8_t *p, uint8_t *packed, int64_t total_events,
                               int64_t max_segment_events, int64_t max_tile_events, const void *tile_ws,
                               void *workspace, size_t workspace_bytes, v2ce_stream_t stream,
                               const void *fused_ws, size_t fused_ws_bytes, int64_t fused_tile_max, int64_t fused_seg_hint,
                               int64_t fused_tile_bin_hint) {
    clear_error();
    V2CE_REQUIRE(vox && seg_offsets, V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: null pointer");
    V2CE_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_emit: bad shape");
    V2CE_REQUIRE(W <= 32767 && H <= 32767, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_emit: x/y are int16");
    V2CE_REQUIRE(fps > 0, V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: fps must be positive");
    V2CE_REQUIRE(rng_mode == V2CE_RNG_REPLAY || rng_mode == V2CE_RNG_PHILOX, V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_emit: bad rng_mode %d", rng_mode);
    Opts o;
    if (int rc = read_options(options, o, "v2ce_ldati_emit")) return rc;
    V2CE_REQUIRE(rng_mode != V2CE_RNG_REPLAY || replay_max_n == 0 || uniforms != nullptr,
                 V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: REPLAY mode needs the uniform tensor");
    const bool soa = ts && x && y && p;
    V2CE_REQUIRE(soa != (packed != nullptr) && (soa || !(ts || x || y || p)), V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_emit: give either the four SoA arrays or the packed buffer");
    V2CE_REQUIRE(!packed || (reinterpret_cast<uintptr_t>(packed) & 3) == 0, V2CE_ERR_BAD_ARG,
                 "v2ce_ldati_emit: packed must be 4-byte aligned");
    const bool random = o.strategy == V2CE_STRATEGY_RANDOM;
    const HostScalars h = host_scalars(fps, t0, o.bidir, random);
    V2CE_REQUIRE(h.ok, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_emit: fps=%g t0=%g needs %lld keys per bin (max %d)",
                 fps, t0, h.NK, kMaxNB << kMaxShift);
    LdatiParams P{};
    hipStream_t st = as_stream(stream);
    if (int rc = fill_params(P, h, o, vox, B, H, W, fps, rng_mode, uniforms, replay_max_n, seed, frame_base, st)) return rc;
    P.seg_offsets = reinterpret_cast<const long long *>(seg_offsets);
    P.frame_ts_add = reinterpret_cast<const long long *>(frame_ts_add);
    P.ts = reinterpret_cast<long long *>(ts); P.x = x; P.y = y;
    P.p = reinterpret_cast<signed char *>(p);
    P.packed = packed;
    if (h.sweep_ok)
        V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ldati_emit_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_bytes));
    if (workspace != nullptr) {
        V2CE_REQUIRE(tile_ws, V2CE_ERR_BAD_ARG, "v2ce_ldati_emit: the two-level path needs v2ce_ldati_count's tile workspace");
        V2CE_REQUIRE(total_events >= 0 && max_segment_events >= 0 && max_tile_events >= 0, V2CE_ERR_BAD_ARG,
                     "v2ce_ldati_emit: bad event counts");
        const Layout L = make_layout(h, o, B, H, W, total_events, max_segment_events, max_tile_events, packed != nullptr);
        const Plan &pl = L.plan;
        V2CE_REQUIRE(L.ok, V2CE_ERR_UNSUPPORTED,
                     "v2ce_ldati_emit: shape / density outside the two-level path (tiles %d, buckets %d, "
                     "largest tile-bin %lld events); pass workspace = NULL", pl.T, pl.NB, (long long)max_tile_events);
        V2CE_REQUIRE(workspace_bytes >= L.bytes, V2CE_ERR_WORKSPACE, "v2ce_ldati_emit: workspace %zu < %zu",
                     workspace_bytes, L.bytes);
        unsigned char *wb = static_cast<unsigned char *>(workspace);
        unsigned *w = static_cast<unsigned *>(workspace);
        P.shift = pl.shift; P.NB = pl.NB; P.nb1 = pl.nb1; P.T = pl.T; P.Tp = pl.Tp; P.tpp = pl.tpp; P.PB = pl.PB;
        P.capA = pl.capA; P.cap2 = pl.cap2; P.tbits = pl.tbits;
        P.tile_off = static_cast<const unsigned *>(tile_ws) + (size_t)B * pl.T * 9;
        P.tile_src = P.tile_off + (size_t)B * pl.T * 9;
        P.tc = static_cast<const unsigned *>(tile_ws);
        // lightly populated tiles (all nine bins <= kSparseCap events) take the one-pass sparse kernel; it needs
        // the nine bins' keys side by side in 20 bits
        P.sparse_cap = (!L.generic && 9ll * ((long long)pl.NB << pl.shift) < (1ll << 20) && !getenv("V2CE_LDATI_NO_SPARSE")) ? kSparseCap : 0;
        P.span = pl.span;
        P.hist_bins = (4 * (pl.span << pl.shift)) > 4 * kMaxSpanKeys ? (4 * (pl.span << pl.shift)) : 4 * kMaxSpanKeys;
        if (L.kbb_bytes) {
            // pooled slope parameters first (LDATI.py:177-190)
            float2 *kbb = reinterpret_cast<float2 *>(wb + L.bytes - L.kbb_bytes);
            hipLaunchKernelGGL(ldati_pool_slope_kernel, dim3((P.HW + 255) / 256, 2 * B), dim3(256), 0, st, P, o.pooling,
                               o.pool_k, kbb);
            P.kbb = kbb;
        }
        auto launch_tile_pass = [&]() -> int {
            if (!L.generic && dense_kernel_serves(P, o) && dense_tile_lds(pl.capA, pl.NB, 16) <= 160 * 1024) return launch_dense_kernel(P, pl, B, st);
            auto tile_kernel = o.bidir ? (pl.tile_threads == 512 ? ldati_tile_pass_kernel<512, 4, true> : ldati_tile_pass_kernel<1024, 2, true>)
                                       : (pl.tile_threads == 512 ? ldati_tile_pass_kernel<512, 4, false> : ldati_tile_pass_kernel<1024, 2, false>);
            V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(tile_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_tile));
            hipLaunchKernelGGL(tile_kernel, dim3(pl.T, B), dim3(pl.tile_threads), pl.lds_tile, st, P);
            return V2CE_OK;
        };
        if (L.generic) {
            // ---- generic path ('random': timestamps spread over a second): tile pass in key mode, one
            // library radix sort of the 60-bit keys, decode (+ pack)
            P.status = reinterpret_cast<int *>(wb);
            hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned *>(P.status), 4, static_cast<unsigned *>(nullptr), 0);
            unsigned long long *kA = reinterpret_cast<unsigned long long *>(wb + L.main_bytes);
            unsigned long long *kB = reinterpret_cast<unsigned long long *>(wb + L.main_bytes + L.keys_bytes / 2);
            void *tmp = wb + L.main_bytes + L.keys_bytes;
            P.keys = kA;
            if (int rc = launch_tile_pass()) return rc;
            if (total_events > 0) {
                size_t tb = L.sort_temp_bytes;
                V2CE_HIP_CHECK(rocprim::radix_sort_keys(tmp, tb, kA, kB, (size_t)total_events, 0, 60, st));
                long long *ts2 = P.ts;
                short *x2 = P.x, *y2 = P.y;
                signed char *p2 = P.p;
                if (packed) {
                    unsigned char *sb = wb + L.main_bytes + L.keys_bytes + L.sort_temp_bytes;
                    ts2 = reinterpret_cast<long long *>(sb);
                    x2 = reinterpret_cast<short *>(sb + 8 * (size_t)total_events);
                    y2 = x2 + total_events;
                    p2 = reinterpret_cast<signed char *>(y2 + total_events);
                }
                const long long blocks = (total_events + 255) / 256;
                hipLaunchKernelGGL(ldati_keys_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, st, P, kB,
                                   (long long)total_events, ts2, x2, y2, p2);
                if (packed)
                    hipLaunchKernelGGL(events_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ts2, x2, y2, p2,
                                       (long long)total_events, packed);
            }
            V2CE_HIP_CHECK(hipGetLastError());
            return V2CE_OK;
        }
        // ---- two-level path; segments with an oversized bucket fall through to the sweep kernel
        P.bofs = w;
        P.groups = P.bofs + pl.n_bkt;
        P.big_list = P.groups + (size_t)B * 9 * pl.NB;
        P.ngroups = P.big_list + (size_t)B * 9 * pl.NB;
        P.seg_flag = reinterpret_cast<int *>(P.ngroups + (size_t)B * 9);
        P.status = P.seg_flag + (size_t)B * 9;
        P.nbig = reinterpret_cast<unsigned *>(P.status + 1);
        P.temp = reinterpret_cast<unsigned *>(P.status + 4);
        P.roff = reinterpret_cast<unsigned short *>(P.temp + (size_t)total_events);
        P.gruns = reinterpret_cast<unsigned *>(P.roff + ((pl.n_tab + 1) & ~(size_t)1));
        if (pl.list_words) {
            P.lists = reinterpret_cast<unsigned *>(wb + pl.off_lists);
            P.list_stride = (int)pl.list_words;
        }
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned *>(P.status), 4, static_cast<unsigned *>(nullptr), 0);
        // the fused count already ran the sparse tile pass: usable when its assumed geometry is the plan's and every tile fitted
        bool fused = false;
        if (fused_ws && !P.kbb && (fused_tile_bin_hint > 0 || P.sparse_cap)) {
            const FusedLayout F = make_fused_layout(h, o, B, H, W, fused_seg_hint, fused_tile_bin_hint);
            // sparse form: every tile fitted its slot; dense form: the dense kernel ran (count_fused's own test) and every (tile, bin) run fitted
            fused = F.ok && fused_ws_bytes >= F.bytes && F.p0.shift == pl.shift && F.p0.NB == pl.NB && F.p0.T == pl.T &&
                    (F.slot_cap ? dense_kernel_serves(P, o) && max_tile_events <= F.slot_cap : fused_tile_max <= kSparseCap);
            if (getenv("V2CE_LDATI_DEBUG"))
                fprintf(stderr, "v2ce_ldati_emit_fused: fused=%d ok=%d bytes %zu/%zu tile_max=%lld shift %d/%d NB %d/%d T %d/%d max_segment %lld sort threads %d cap2 %d\n", (int)fused, (int)F.ok,
                        fused_ws_bytes, F.bytes, (long long)fused_tile_max, F.p0.shift, pl.shift, F.p0.NB, pl.NB, F.p0.T, pl.T, (long long)max_segment_events, pl.sort_threads, pl.cap2);
            if (getenv("V2CE_LDATI_DEBUG") && F.slot_cap) fprintf(stderr, "   dense slots of %d records, largest (tile, bin) %lld\n", F.slot_cap, (long long)max_tile_events);
            if (fused) {
                const unsigned char *fb = static_cast<const unsigned char *>(fused_ws);
                P.fused_status = reinterpret_cast<const int *>(fb);
                P.tile_abs = reinterpret_cast<const unsigned *>(fb + F.off_abs);
                P.tile_src = P.tile_abs;
                P.temp = const_cast<unsigned *>(reinterpret_cast<const unsigned *>(fb + F.off_rec));
                P.roff = const_cast<unsigned short *>(reinterpret_cast<const unsigned short *>(fb + F.off_roff));
            }
        }
        // (Tried in round 3 and removed: walking the frames in groups so that the HBM-bound bucket sort of group g runs on a
        // low-priority side stream under the VALU-bound tile pass of group g + 1.  24 stress frame-pairs: 1 group 1.74 ms,
        // 2 groups 1.86, 4 groups 1.83, 8 groups 2.21 -- the tile pass needs whole CUs (1024 threads, ~100 KB of LDS), the sort
        // workgroups that slip in between delay its rounds, and each group adds a partial last round.)
        if (P.sparse_cap && !fused) {
            V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ldati_tile_sparse_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSparseLds));
            hipLaunchKernelGGL(ldati_tile_sparse_kernel<false>, dim3(pl.T, B), dim3(kSparseThreads), kSparseLds, st, P);
        }
        if (!fused)
            if (int rc = launch_tile_pass()) return rc;
        hipLaunchKernelGGL(ldati_bucket_scan_kernel, dim3(B * 9), dim3(512), 0, st, P);
        {
            const bool k24 = pl.cap2 > pl.sort_threads * 8;
            const int sth = pl.sort_threads;
            auto sort_kernel = packed ? (k24 ? (sth == 64 ? ldati_bucket_sort_kernel<true, 24, 64> : sth == 128 ? ldati_bucket_sort_kernel<true, 24, 128> : ldati_bucket_sort_kernel<true, 24, 256>)
                                             : ldati_bucket_sort_kernel<true, 8, 256>)
                                      : (k24 ? (sth == 64 ? ldati_bucket_sort_kernel<false, 24, 64> : sth == 128 ? ldati_bucket_sort_kernel<false, 24, 128> : ldati_bucket_sort_kernel<false, 24, 256>)
                                             : ldati_bucket_sort_kernel<false, 8, 256>);
            V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(sort_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_sort));
            hipLaunchKernelGGL(sort_kernel, dim3(pl.NB, B * 9), dim3(pl.sort_threads), pl.lds_sort, st, P);
        }
        if (packed) hipLaunchKernelGGL(ldati_big_bucket_kernel<true>, dim3(256), dim3(256), 0, st, P);
        else hipLaunchKernelGGL(ldati_big_bucket_kernel<false>, dim3(256), dim3(256), 0, st, P);
        V2CE_HIP_CHECK(hipGetLastError());
        return V2CE_OK;
    } else {
        V2CE_REQUIRE(h.sweep_ok && o.pooling == V2CE_POOL_NONE, V2CE_ERR_UNSUPPORTED,
                     "v2ce_ldati_emit: the sweep kernel (workspace = NULL) covers forward relocation without pooling at "
                     "key ranges <= 9600 (fps=%g t0=%g needs %lld keys per bin): pass a workspace", fps, t0, h.NK);
    }
    // segment-sweep kernel (workspace = NULL): every segment
    hipLaunchKernelGGL(ldati_emit_kernel, dim3(B * 9), dim3(256), h.lds_bytes, st, P);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

}  // namespace

extern "C" int v2ce_ldati_emit(const float *vox, int B, int H, int W, double fps, double t0,
                               const v2ce_ldati_options *options, int rng_mode, const float *uniforms, int replay_max_n,
                               uint64_t seed, int64_t frame_base, const int64_t *seg_offsets,
                               const int64_t *frame_ts_add, int64_t *ts, int16_t *x, int16_t *y,
                               int8_t *p, uint8_t *packed, int64_t total_events,
                               int64_t max_segment_events, int64_t max_tile_events, const void *tile_ws,
                               void *workspace, size_t workspace_bytes, v2ce_stream_t stream) {
    return emit_impl(vox, B, H, W, fps, t0, options, rng_mode, uniforms, replay_max_n, seed, frame_base, seg_offsets, frame_ts_add, ts, x,
                     y, p, packed, total_events, max_segment_events, max_tile_events, tile_ws, workspace, workspace_bytes, stream,
                     nullptr, 0, 0, 0, 0);
}

extern "C" int v2ce_ldati_emit_fused(const float *vox, int B, int H, int W, double fps, double t0,
                                     const v2ce_ldati_options *options, int rng_mode, const float *uniforms, int replay_max_n,
                                     uint64_t seed, int64_t frame_base, const int64_t *seg_offsets,
                                     const int64_t *frame_ts_add, int64_t *ts, int16_t *x, int16_t *y,
                                     int8_t *p, uint8_t *packed, int64_t total_events,
                                     int64_t max_segment_events, int64_t max_tile_events, const void *tile_ws,
                                     void *workspace, size_t workspace_bytes, const void *fused_ws, size_t fused_ws_bytes,
                                     int64_t largest_tile_events, int64_t expected_max_segment_events,
                                     int64_t expected_max_tile_bin_events, v2ce_stream_t stream) {
    return emit_impl(vox, B, H, W, fps, t0, options, rng_mode, uniforms, replay_max_n, seed, frame_base, seg_offsets, frame_ts_add, ts, x,
                     y, p, packed, total_events, max_segment_events, max_tile_events, tile_ws, workspace, workspace_bytes, stream,
                     fused_ws, fused_ws_bytes, largest_tile_events, expected_max_segment_events, expected_max_tile_bin_events);
}

extern "C" int v2ce_ldati_plan_info(int B, int H, int W, double fps, double t0, const v2ce_ldati_options *options,
                                    int64_t total_events, int64_t max_segment_events, int64_t max_tile_events, int64_t *info) {
    clear_error();
    V2CE_REQUIRE(info && fps > 0 && B > 0 && H > 0 && W > 0, V2CE_ERR_BAD_ARG, "v2ce_ldati_plan_info: bad argument");
    Opts o;
    if (int rc = read_options(options, o, "v2ce_ldati_plan_info")) return rc;
    const HostScalars h = host_scalars(fps, t0, o.bidir, o.strategy == V2CE_STRATEGY_RANDOM);
    const Plan pl = make_plan(h, B, H, W, total_events, max_segment_events, max_tile_events);
    const int64_t v[10] = {pl.ok, pl.shift, pl.NB, pl.T, pl.capA, pl.cap2, (int64_t)pl.n_tab, (int64_t)pl.n_bkt,
                           (int64_t)pl.lds_tile, (int64_t)pl.lds_sort};   // workspace layout: see v2ce_hip.h
    for (int i = 0; i < 10; ++i) info[i] = v[i];
    return V2CE_OK;
}

extern "C" int v2ce_ldati_status(const void *workspace, int B, int H, int W, double fps, double t0,
                                 const v2ce_ldati_options *options, int64_t total_events, int64_t max_segment_events,
                                 int64_t max_tile_events, const int32_t **status_dev) {
    clear_error();
    V2CE_REQUIRE(workspace && status_dev, V2CE_ERR_BAD_ARG, "v2ce_ldati_status: null pointer");
    Opts o;
    if (int rc = read_options(options, o, "v2ce_ldati_status")) return rc;
    const bool random = o.strategy == V2CE_STRATEGY_RANDOM;
    const HostScalars h = host_scalars(fps, t0, o.bidir, random);
    const Plan pl = make_plan(h, B, H, W, total_events, max_segment_events, max_tile_events);
    const unsigned *w = static_cast<const unsigned *>(workspace);
    if (random) {
        *status_dev = reinterpret_cast<const int32_t *>(w);
        return V2CE_OK;
    }
    V2CE_REQUIRE(pl.ok, V2CE_ERR_UNSUPPORTED, "v2ce_ldati_status: no two-level plan for these arguments");
    *status_dev = reinterpret_cast<const int32_t *>(w + pl.n_bkt + 2 * (size_t)B * 9 * pl.NB + 2 * (size_t)B * 9);
    return V2CE_OK;
}

extern "C" int v2ce_ldati_rank_mode(int32_t *mode) {
    clear_error();
    V2CE_REQUIRE(mode, V2CE_ERR_BAD_ARG, "v2ce_ldati_rank_mode: null pointer");
    int ok = 0;
    V2CE_HIP_CHECK(hipMemcpyFromSymbol(&ok, HIP_SYMBOL(g_lds_order_ok), sizeof(int)));
    const char *e = getenv("V2CE_LDATI_NO_ATOMIC_ORDER");
    *mode = (ok && !(e && e[0] == '1')) ? 1 : 0;
    return V2CE_OK;
}

extern "C" int v2ce_ldati_selfcheck(double fps, int64_t *mismatches) {
    clear_error();
    V2CE_REQUIRE(mismatches && fps > 0, V2CE_ERR_BAD_ARG, "v2ce_ldati_selfcheck: bad argument");
    const HostScalars h = host_scalars(fps, 0.0);
    unsigned long long *bad = nullptr;
    V2CE_HIP_CHECK(hipMalloc(&bad, 2 * sizeof(unsigned long long)));
    V2CE_HIP_CHECK(hipMemset(bad, 0, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(ldati_selfcheck_kernel, dim3(65536, 2 * kSlopeM + 1), dim3(256), 0, nullptr, h.VS, h.VS2, h.INV, -kSlopeM, bad);
    unsigned long long host[2] = {0, 0};
    const hipError_t e = hipMemcpy(host, bad, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipFree(bad);
    V2CE_HIP_CHECK(e);
    mismatches[0] = (int64_t)host[0];
    mismatches[1] = (int64_t)host[1];
    // the single-event time's fast form (single_time_fast): the product's own one-time check, run here in a spare slot
    const unsigned zero = 0;
    V2CE_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fast64_bad), &zero, sizeof(unsigned), 7 * sizeof(unsigned)));
    hipLaunchKernelGGL(ldati_fast64_check_kernel, dim3((0x3F800000u + kFast64Neg + 255u) / 256u), dim3(256), 0, nullptr, fps, 1.0 / fps, 1.0 / 9.0, 7);
    unsigned bad64 = 0;
    V2CE_HIP_CHECK(hipMemcpyFromSymbol(&bad64, HIP_SYMBOL(g_fast64_bad), sizeof(unsigned), 7 * sizeof(unsigned)));
    mismatches[2] = (int64_t)bad64;
    return V2CE_OK;
}

#ifdef V2CE_STAMP
extern "C" int v2ce_debug_stamps(unsigned long long *out32_host) {
    unsigned long long zero[32] = {0};
    if (hipMemcpyFromSymbol(out32_host, HIP_SYMBOL(g_stamp), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

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

extern "C" int v2ce_events_unpack(const uint8_t *packed, int64_t n, int64_t *ts, int16_t *x, int16_t *y,
                                  int8_t *p, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(n >= 0, V2CE_ERR_BAD_ARG, "v2ce_events_unpack: negative n");
    if (n == 0) return V2CE_OK;
    V2CE_REQUIRE(ts && x && y && p && packed, V2CE_ERR_BAD_ARG, "v2ce_events_unpack: null pointer");
    V2CE_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 3) == 0, V2CE_ERR_BAD_ARG,
                 "v2ce_events_unpack: packed must be 4-byte aligned");
    const long long blocks = (n + 255) / 256;
    V2CE_REQUIRE(blocks < (1ll << 31), V2CE_ERR_UNSUPPORTED, "v2ce_events_unpack: too many events");
    hipLaunchKernelGGL(events_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), packed,
                       (long long)n, reinterpret_cast<long long *>(ts), x, y, reinterpret_cast<signed char *>(p));
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
