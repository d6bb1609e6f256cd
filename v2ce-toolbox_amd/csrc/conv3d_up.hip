// conv3d_up.hip -- conv1 of the UNet's decoder blocks on gfx950: a 3x3x3 convolution whose input is the
// VIRTUAL concatenation  nearest-upsample-2x(x0) ++ x1  (/root/reference/scripts/unet_2layer.py:358-365,
// /root/reference/scripts/submodules.py:249-264), with the upsampled part PHASE-FOLDED.
//
// ATen's nearest map for every decoder size of this network is src = dst >> 1 (H0 = ceil(Hin / 2)), so an
// output position of parity (ph, pw) sees, through the 3x3 (H, W) taps on an upsampled channel, only 2x2
// distinct source pixels:
//     even row h = 2i:   w[-1] x[i-1] + (w[0] + w[+1]) x[i]          ("E": 2 taps)
//     odd  row h = 2i+1: (w[-1] + w[0]) x[i] + w[+1] x[i+1]          ("O": 2 taps)
// and the same along W: 12 taps with pre-summed weights instead of 27 on the C0 channels -- 5/9 of their
// multiplies (16.5 % of the whole network's) never happen, and the low-resolution halo box staged per chunk
// is a third of the mapped one.  The skip channels (x1) keep their 27 taps.
// One exception: when Hout (Wout) is odd, the LAST row (column) is an even one whose +1 neighbour is the
// convolution's zero padding, not x[i] again: the folded tap (w[0] + w[+1]) x[i] carries a phantom w[+1] x[i].
// Tiles that contain that row (column) run, behind the folded list of an even phase, a CORRECTION list of the same
// shape -- 12 taps with the weights -w[+1] (folded along the other axis) on the taps that read x[i], zero elsewhere --
// in which only the lanes of the last row (column) read the halo box; every other lane reads a zeroed region of
// LDS.  (Corner tiles: one list per axis and a third, +w[+1][+1], for the corner lane: inclusion-exclusion.)
//
// Same workgroup as conv3d_f16x2_ws_kernel (conv3d.hip): 8 waves, producers 4-7 gather + split a 16-channel
// chunk of the halo into fp16 hi / lo pieces one chunk ahead, consumers 0-3 run three fp16 MFMAs per k-step
// with A fragments (weights) from L2 through a three-slot ring, one barrier per chunk, persistent over tiles.
// What differs:
//   * every 32-position MFMA fragment is PHASE-HOMOGENEOUS (its positions share (ph, pw)), all fragments of
//     a wave's phase share the A fragment of a tap: a tile is a (TT, TH, TW) box of full-resolution outputs
//     (TH, TW even) whose four phase sub-boxes (TT, TH/2, TW/2) each fill NFG fragments;
//   * a chunk's taps are a LIST of consecutive tap slots of one weight region -- a folded list of 12 per phase
//     (+ correction lists in edge tiles) on the upsampled chunks, the 27 plain taps on the skip chunks -- all
//     multiples of three, so the ring's three slots stay static across lists, chunks and tiles;
//   * the skip channels' halo is staged with its columns DE-INTERLEAVED (even columns, then odd ones), so the
//     stride-2 positions of a phase read consecutive 16-byte pieces.
#include "conv3d_dev.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace v2ce {
namespace {

// Tap slots of the folded region, 12 per list in (dt, a, b) order [plane][slot][cg][co][16]:
//   slot 12 p            phase p = 2 ph + pw: fold_h(ph) x fold_w(pw), fold(0) = E, fold(1) = O
//   slot 48 + 12 pw      H correction of phase (0, pw): -w[.][+1][fold_w(pw)] at a = 1
//   slot 72 + 12 ph      W correction of phase (ph, 0): -w[.][fold_h(ph)][+1] at b = 1
//   slot 96              corner correction of phase (0, 0): +w[.][+1][+1] at (a, b) = (1, 1)
constexpr int kUpSlots = 108;
constexpr int kUpListTaps = 12;

// source taps of folded tap t (0, 1) of an axis: bit k set = w[k - 1] takes part.  odd = 0: E = w[-1] | w[0] + w[+1];
// odd = 1: O = w[-1] + w[0] | w[+1]
__host__ __device__ inline int up_axis_mask(int odd, int t) { return odd ? (t == 0 ? 3 : 4) : (t == 0 ? 1 : 6); }
// weight of slot list `li` (0 .. 8), tap (a, b), as masks over (dh, dw) and a sign; 0 masks = a zero weight
__host__ __device__ inline void up_list_masks(int li, int a, int b, int &mh, int &mw, int &sign) {
    sign = 1;
    if (li < 4) { mh = up_axis_mask(li >> 1, a); mw = up_axis_mask(li & 1, b); return; }
    sign = -1;
    if (li < 6) { mh = a == 1 ? 4 : 0; mw = up_axis_mask(li - 4, b); return; }          // H correction: the +1 row
    if (li < 8) { mh = up_axis_mask(li - 6, a); mw = b == 1 ? 4 : 0; return; }          // W correction: the +1 column
    sign = 1; mh = a == 1 ? 4 : 0; mw = b == 1 ? 4 : 0;                                  // corner
}

struct UpParams {
    ConvParams C;
    int SH, SW, n_sub;            // phase sub-box (TH / 2, TW / 2), its positions TT * SH * SW
    int HH0, HW0, plane0;         // low-resolution halo box (TT + 2) x (SH + 2) x (SW + 2); plane0 = HT * Q0 pieces in LDS
    int P0, Q0, P1, Q1;           // LDS pitches (pieces) between rows / time steps of the low-resolution box and of the skip box
                                  // (>= the dense ones: chosen with the lane order so that the 16-lane groups of a ds_read_b128 hit
                                  // sixteen different 16-byte slots of the 256-byte bank row)
    int lbit[5];                  // lane order: bit b of a lane's rank in its ds_read_b128 group pair becomes bit lbit[b] of its position
    int zero0;                    // first piece of the zeroed region of an upsampled chunk's planes (edge tiles; >= plane0)
    int chs;                      // pieces per plane of the LDS buffers (>= the skip halo, >= zero0 + plane0)
    int stage_off;                // bytes from the start of LDS to the consumers' epilogue staging (4 x 4 KB)
    int dbg;                      // timing experiments (V2CE_UP_DBG; WRONG results): 1 = stores dropped by the range check, 2 = no epilogue
    int HWh;                      // HWd / 2: first odd column of the de-interleaved skip halo
    int CG0;                      // C0 / 16
    int part;                     // 1: only the upsampled channels' chunks run (v2ce_conv3d_fwd_up2_part): the skip channels' share of the
                                  // convolution comes from another launch that adds this one's output as its residual
    int odd_h, odd_w;
    int fold_off;                 // bytes from wq to the folded region's hi plane
    int fold_plane;               // bytes between its hi and lo planes
};

#if defined(__HIP_DEVICE_COMPILE__)

// halo offsets of the two sources (channels-last-16: byte offset of the element's 64-byte group in channel group 0
// of its time step; kOOB = zero padding)
//   low:  element r of the (HT, HH0, HW0) box = x0 at (t0 - 1 + ht, i0 - 1 + hr, j0 - 1 + hc)
//   skip: element r of the (HT, HH, HWd) box, columns de-interleaved: x = r % HWd < HWh ? column 2x : column 2 (x - HWh) + 1
template <int EPT>
__device__ __forceinline__ void up_offsets(const UpParams &U, bool skip, int t0, int h0, int w0, int tid, unsigned (&goff)[EPT]) {
    const ConvParams &P = U.C;
    // slot r of the PITCHED box: time step r / Q, row (r % Q) / P, column r % P; slots in the padding read as zero
    const int hh_n = skip ? P.HH : U.HH0, hw_n = skip ? P.HWd : U.HW0, pr = skip ? U.P1 : U.P0, qr = skip ? U.Q1 : U.Q0;
    const int plane = skip ? P.plane : U.plane0;
    const int Cs = skip ? P.C1 : P.C0, Hs = skip ? P.Hin : P.H0, Ws = skip ? P.Win : P.W0, Wp = skip ? P.Winp : P.W0p;
    const int hb = skip ? h0 - 1 : (h0 >> 1) - 1, wb = skip ? w0 - 1 : (w0 >> 1) - 1;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int r = tid + 256 * i;
        unsigned off = kOOB;
        if (r < plane) {
            const int ht = r / qr;
            const int rem = r - ht * qr;
            const int hr = rem / pr;
            int hc = rem - hr * pr;
            if (hr < hh_n && hc < hw_n) {
                if (skip) hc = hc < U.HWh ? 2 * hc : 2 * (hc - U.HWh) + 1;
                const int t = t0 - 1 + ht, h = hb + hr, w = wb + hc;
                if (t >= 0 && t < P.T && h >= 0 && h < Hs && w >= 0 && w < Ws)
                    off = 4u * (unsigned)((t * Cs) * (Hs * Wp)) + 64u * (unsigned)(h * Wp + w);
            }
        }
        goff[i] = off;
    }
}

// rank of lane l32 inside its pair of ds_read_b128 lane groups ({0-3, 12-15, 20-27} = ranks 0-15, {4-11, 16-19, 28-31} = 16-31;
// MI355X_MICROARCH.md, LDS), permuted bitwise into the lane's position inside its fragment
__device__ __forceinline__ int up_lane_position(const UpParams &U, int l32) {
    int rank;
    if (l32 < 4) rank = l32;
    else if (l32 < 12) rank = 16 + (l32 - 4);
    else if (l32 < 16) rank = 4 + (l32 - 12);
    else if (l32 < 20) rank = 24 + (l32 - 16);
    else if (l32 < 28) rank = 8 + (l32 - 20);
    else rank = 28 + (l32 - 28);
    int pos = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) pos |= ((rank >> b) & 1) << U.lbit[b];
    return pos;
}

#endif  // __HIP_DEVICE_COMPILE__

#if defined(__HIP_DEVICE_COMPILE__)
// Epilogue of the phase-folded kernel: y = act(acc * scale + shift), max |y| tracking -- conv_epilogue's arithmetic (RES 0) --
// with the stores re-shaped through a wave-private 4 KB of LDS.  A fragment's 32 positions share a phase: consecutive lanes
// are positions TWO columns apart, so conv_epilogue's direct stores (lane = position, 32 bytes of its 64-byte channel group
// per instruction) would touch 32 half-empty 128-byte lines per instruction -- measured: 35 k cycles for the two epilogues of
// a dec3.conv1 tile against 16 k on the contiguous layout.  Here every lane writes its four channel quads position-major into
// LDS ([group][position][quarter] x 16 B) and reads them back as lane = (position, quarter): a store instruction then carries
// the WHOLE 64-byte groups of 16 positions (16 lines per instruction, four instructions per fragment and 16-channel group pair).
// stage: this wave's 256 x 16 B.  Positions outside the tensor / the box: poff < 0.
template <int CO_FR, int PO_FR>
__device__ __forceinline__ void up_epilogue(const ConvParams &P, f32x16 (&acc)[CO_FR][PO_FR], const int (&poff)[PO_FR], int co0,
                                            int lane, int b, float inv_scale, float *stage_f, float *y, const float *scale_p,
                                            const float *shift_p, int act, float *y_absmax) {
    typedef float f32x4q __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
    f32x4q *stage = reinterpret_cast<f32x4q *>(stage_f);
    const int l32 = lane & 31, half = lane >> 5;
    const long long seq = (long long)P.T * P.Cout * (P.Hout * P.Woutp);
    const int gstride = P.Hout * P.Woutp * 64;              // bytes between 16-channel groups
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(y + b * seq, 0, (int)(seq * 4), 0x00020000);
    const float slope = act_slope(act);
    const int cbase = co0 + 4 * half;
    // the position this lane STORES for fragment f, half k: lane (16 k + lane / 4) of the fragment (lanes 0-31 and 32-63 hold the same poff)
    unsigned vo[PO_FR][2], vmask[PO_FR];
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) {
        vmask[f] = poff[f] >= 0 ? 0x7fffffffu : 0u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pn = __builtin_amdgcn_ds_bpermute((16 * k + (lane >> 2)) * 4, poff[f]);
            vo[f][k] = pn >= 0 ? (unsigned)(pn + 16 * (lane & 3)) : kOOB;
        }
    }
    unsigned ymax = 0u;
#pragma unroll
    for (int q = 0; q < CO_FR; ++q) {
        const bool cok = co0 + q * 32 < P.Cout;              // uniform (Cout need not fill the last channel tile)
        float sc[16], sh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int co = cbase + q * 32 + (r & 3) + 8 * (r >> 2);
            co = co < P.Cout ? co : P.Cout - 1;
            sc[r] = scale_p[co] * inv_scale;
            sh[r] = shift_p[co];
        }
#pragma unroll
        for (int f = 0; f < PO_FR; ++f) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4q out;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * r4 + k;
                    float v = acc[q][f][r] * sc[r] + sh[r];
                    v = apply_act(v, slope);
                    out[k] = v;
                    const unsigned av = __builtin_bit_cast(unsigned, v) & (cok ? vmask[f] : 0u);
                    ymax = av > ymax ? av : ymax;
                }
                // channel quad r4 of this lane = group r4 / 2, quarter (r4 & 1) * 2 + half
                stage[((r4 >> 1) * 32 + l32) * 4 + (r4 & 1) * 2 + half] = out;
            }
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x4q v = stage[(gq * 32 + 16 * k) * 4 + lane];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, v), rs_y, cok ? vo[f][k] : kOOB,
                                                           (co0 / 16 + 2 * q + gq) * gstride, 0);
                    asm volatile("s_nop 1" : "+v"(v));       // store-data hazard of 16-byte stores, see conv_epilogue
                }
        }
    }
    if (y_absmax) absmax_commit(__builtin_bit_cast(float, ymax), y_absmax + b * P.amax_bs);
}
#endif  // __HIP_DEVICE_COMPILE__

// WCO: consumer waves across the channel tile (1: every wave one phase; 2: two position-waves of two phases each)
// FUSE: 0 = plain, 2 = fused 1x1x1 shortcut (second accumulator set at the centre tap; CO_FR == 1)
#ifdef V2CE_STAMP
// diagnostic build: per workgroup and role (consumer wave 0, producer wave 4) the cycles of the whole kernel [0], inside the
// chunk barriers [1], (consumer) inside the epilogues [2] and the tile setup [3]
#define TICK() __builtin_amdgcn_s_memtime()
#define ACC_T(var_, t0_) var_ += TICK() - (t0_)
#else
#define TICK() 0ull
#define ACC_T(var_, t0_) do {} while (0)
#endif
template <int WCO, int CO_FR, int PO_FR, int FUSE>
__global__ __launch_bounds__(512, 1) void conv3d_up_kernel(UpParams U) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvParams &P = U.C;
    [[maybe_unused]] unsigned long long t_all = TICK(), t_bar = 0, t_epi = 0, t_set = 0;
    constexpr int CK = 16, EPT = 5, K3 = 27;
    constexpr int NG = WCO;                       // phase groups per consumer wave
    constexpr int NFG = PO_FR / NG;               // fragments per phase
    static_assert(WCO == 1 || WCO == 2, "one or two phases per wave");
    static_assert(PO_FR % NG == 0, "fragments split evenly over the phases of a wave");
    constexpr int CO_TILE = WCO * CO_FR * 32;
    const int chs = U.chs;
    f16x8 *pieces = reinterpret_cast<f16x8 *>(conv_smem);                      // [2][4][chs] x 16 B

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CG = P.Cin / CK, CG0 = U.CG0;
    const int CGE = U.part ? CG0 : CG;                                       // chunks a tile runs (the weight buffer's strides keep CG)
    const long long wplane = (long long)K3 * CG * P.Cout * 16;               // halves per plane of the plain region
    const int tap_stride = CG * P.Cout * 32, tap_stride0 = CG0 * P.Cout * 32, cg_stride = P.Cout * 32;


    struct TileId { int b, co_t, t0, h0, w0; };
    auto decode = [&](int vb, TileId &T) -> bool {
        const int xcd = vb & 7, q = vb >> 3;
        T.co_t = q % P.n_co_tiles;
        const int sp = q / P.n_co_tiles;
        int bid = xcd * P.per_xcd + sp;
        if (sp >= P.per_xcd || bid >= P.n_spatial) return false;
        const int iw = bid % P.nW;            bid /= P.nW;
        const int ih = bid % P.nH;            bid /= P.nH;
        const int it = bid % P.nT;            bid /= P.nT;
        T.b = bid;
        T.t0 = it * P.TT; T.h0 = ih * P.TH; T.w0 = iw * P.TW;
        return true;
    };
    auto next_tile = [&](int &vb, TileId &T) -> bool {
        for (; vb < P.total_blocks; vb += (int)gridDim.x)
            if (decode(vb, T)) return true;
        return false;
    };

    const float w_scale = reinterpret_cast<const float *>(P.wq + 2 * wplane)[1];
    auto amax_of = [&](int b) -> float {
        if (!P.x0_absmax) return 4094.0f;
        float am = P.x0_absmax[b * P.amax_bs];
        if (P.x1_absmax) am = fmaxf(am, P.x1_absmax[b * P.amax_bs]);
        return am;
    };
    auto scale_of = [&](int b) -> float { return P.x0_absmax ? pow2_prescale(amax_of(b)) : kActScale; };
    // range guard: as conv3d_f16x2_ws_kernel (tail[0] covers the folded weights too; K = Cin * 27 bounds the folded sums)
    if (P.guard && blockIdx.x == 0 && wave == 0) {
        auto smax = [&](const float *scale) {
            float sm = 0.0f;
            for (int co = lane; co < P.Cout; co += 64) sm = fmaxf(sm, fabsf(scale[co]));
#pragma unroll
            for (int o = 32; o; o >>= 1) sm = fmaxf(sm, __shfl_xor(sm, o));
            return sm;
        };
        const float sm = smax(P.scale), smd = FUSE == 2 ? smax(P.sc_scale) : 0.0f;
        const float *tail = reinterpret_cast<const float *>(P.wq + 2 * wplane);
        const float *taild = reinterpret_cast<const float *>(P.sc_w + 2 * (long long)CG * P.Cout * 16);
        const int nb = P.amax_bs ? P.B : 1;
        for (int b = lane; b < nb; b += 64) {
            const float am = amax_of(b), xs = scale_of(b);
            float E = sm * (float)(P.Cin * K3) * 0x1p-25f * (tail[0] / xs + am / tail[1]);
            if (FUSE == 2) E = fmaxf(E, smd * (float)P.Cin * 0x1p-25f * (taild[0] / xs + am / taild[1]));
            P.guard[b * P.amax_bs] = E;
        }
    }

    int vb = blockIdx.x;
    TileId T;
    if (!next_tile(vb, T)) return;
    int gc = 0;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers (see conv3d_f16x2_ws_kernel)
        const int ptid = tid - 256;
        float x_scale = scale_of(T.b);
        unsigned goff[EPT];
        float R0[CK][EPT], R1[CK][EPT];
        __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0), 0, 0, 0x00020000);
        int cur_src = -1, src_cstride4 = 0;
        auto load_chunk = [&](const TileId &L, int cidx, float (&R)[CK][EPT]) {
            const int want_src = cidx < CG0 ? 0 : 1;
            if (want_src != cur_src) {   // uniform; twice per tile
                cur_src = want_src;
                up_offsets<EPT>(U, want_src == 1, L.t0, L.h0, L.w0, ptid, goff);
                if (want_src == 0) {
                    const long long seq = (long long)P.T * P.C0 * (P.H0 * P.W0p);
                    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0 + L.b * seq), 0, (int)(seq * 4), 0x00020000);
                    src_cstride4 = P.H0 * P.W0p * 4;
                } else {
                    const long long seq = (long long)P.T * P.C1 * (P.Hin * P.Winp);
                    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x1 + L.b * seq), 0, (int)(seq * 4), 0x00020000);
                    src_cstride4 = P.Hin * P.Winp * 4;
                }
            }
            const int lim = want_src ? P.plane : U.plane0;
            const int grp = want_src ? cidx - CG0 : cidx;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                if ((wave - 4) * 64 + 256 * i < lim) {                // wave-uniform
                    typedef float f32x4g __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int k4 = 0; k4 < CK / 4; ++k4) {
                        const f32x4g v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(
                            rs_in, goff[i], grp * (src_cstride4 * 16) + 16 * k4, 0));
                        R[4 * k4][i] = v[0]; R[4 * k4 + 1][i] = v[1]; R[4 * k4 + 2][i] = v[2]; R[4 * k4 + 3][i] = v[3];
                    }
                }
            }
        };
        int vbL = vb, cgL = 0;
        TileId TL = T;
        bool moreL = true;
        auto load_next = [&](float (&R)[CK][EPT]) {
            if (!moreL) return;
            load_chunk(TL, cgL, R);
            if (++cgL == CGE) {
                cgL = 0;
                vbL += (int)gridDim.x;
                moreL = next_tile(vbL, TL);
                cur_src = -1;
            }
        };
        int cgC = 0;
        auto convert = [&](const float (&R)[CK][EPT]) {
            f16x8 *qb = pieces + (gc & 1) * 4 * chs;
            const int lim = cgC < CG0 ? U.plane0 : P.plane;
            // edge tiles: the region the correction lists' other lanes read (the skip chunks overwrite it every time)
            if (cgC < CG0 && ((U.odd_h && T.h0 + P.TH >= P.Hout) || (U.odd_w && T.w0 + P.TW >= P.Wout))) {     // uniform
                const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int r = ptid; r < U.plane0; r += 256) {
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl) qb[pl * chs + U.zero0 + r] = z;
                }
            }
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int r = ptid + 256 * i;
                if ((wave - 4) * 64 + 256 * i < lim) {                // wave-uniform (lanes past the box write padding)
#pragma unroll
                    for (int hg = 0; hg < 2; ++hg) {
                        typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
                        u32x4c ph, pl;
#pragma unroll
                        for (int c2 = 0; c2 < 4; ++c2) {
                            const float xa = R[8 * hg + 2 * c2][i], xb = R[8 * hg + 2 * c2 + 1][i];
                            unsigned h, l;
                            asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                                "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                                "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
                                "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                : "=&v"(h), "=&v"(l) : "v"(xa), "v"(xb), "v"(x_scale));
                            ph[c2] = h;
                            pl[c2] = l;
                        }
                        qb[hg * chs + r] = __builtin_bit_cast(f16x8, ph);
                        qb[(2 + hg) * chs + r] = __builtin_bit_cast(f16x8, pl);
                    }
                }
            }
        };
        bool moreC = true;
        auto advance = [&]() {
            ++gc;
            if (++cgC == CGE) {
                cgC = 0;
                vb += (int)gridDim.x;
                moreC = next_tile(vb, T);
                if (moreC) x_scale = scale_of(T.b);
            }
        };
        load_next(R0);
        load_next(R1);
        while (moreC) {
            convert(R0);
            load_next(R0);
            { [[maybe_unused]] const unsigned long long tb = TICK();
            __syncthreads();                                          // barrier gc: pieces[gc & 1] ready
            ACC_T(t_bar, tb); }
            advance();
            if (!moreC) break;
            convert(R1);
            load_next(R1);
            { [[maybe_unused]] const unsigned long long tb = TICK();
            __syncthreads();
            ACC_T(t_bar, tb); }
            advance();
        }
#ifdef V2CE_STAMP
        if (lane == 0 && wave == 4) {
            P.stamps[((long long)blockIdx.x * 2 + 1) * 8 + 0] = TICK() - t_all;
            P.stamps[((long long)blockIdx.x * 2 + 1) * 8 + 1] = t_bar;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    __builtin_amdgcn_s_setprio(2);
    const int wco = wave % WCO, wpo = wave / WCO;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(P.wq), 0, U.fold_off + 2 * U.fold_plane, 0x00020000);
    const int lo_off = (int)(2 * wplane);                  // plain region: bytes from the hi plane to the lo plane
    constexpr bool SC = FUSE == 2;
    static_assert(!SC || CO_FR == 1, "the fused shortcut needs a second accumulator set: 32-channel wave tiles");
    const long long wplane_d = (long long)CG * P.Cout * 16;
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(SC ? P.sc_w : P.wq), 0, (int)(4 * wplane_d), 0x00020000);

    // B fragments: refilled in place behind the MFMAs that read them (two fragment rows per wave: six MFMAs of the other
    // fragments cover the LDS round trip), or -- one fragment row per wave, three MFMAs per fragment and tap -- two sets
    // alternating by tap, the next tap's reads issued a whole tap ahead
#ifndef V2CE_UP_NB_SC
#define V2CE_UP_NB_SC 1
#endif
    constexpr int NB = CO_FR == 1 ? (FUSE == 2 ? V2CE_UP_NB_SC : 2) : 1;
    // two-fragment lists (CO_FR = 2: the folded lists of a wave with two phases of two fragments, or with one phase of two): a fragment's
    // in-place refill has only the other fragment's six MFMAs (192 cycles) to land -- the B sets alternate by tap there too.  Only
    // the list's own fragments are live, so two sets of two cost what one set of four does (the skip chunks' lists keep one set).
#ifndef V2CE_UP_NB2F
#define V2CE_UP_NB2F 2
#endif
    constexpr int NB0 = (NB == 1 && NFG == 2) ? V2CE_UP_NB2F : NB;            // B sets of the folded lists
    constexpr int NBA = NB0 > NB ? NB0 : NB;
    f16x8 ah[3][CO_FR], al[3][CO_FR], bh[NBA][PO_FR], bl[NBA][PO_FR];
    int wlane[CO_FR];
#define V2CE_LOAD_A(slot_, soff_, lod_)                                                        \
    {                                                                                          \
        const int so_ = (soff_), sl_ = so_ + (lod_);                                           \
        _Pragma("unroll") for (int q = 0; q < CO_FR; ++q) {                                    \
            ah[slot_][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, wlane[q], so_, 0)); \
            al[slot_][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, wlane[q], sl_, 0)); \
        }                                                                                      \
    }
    // a tap list = consecutive tap slots of one weight region: tap t at abase + t * tstride (hi), + lod (lo)
    struct ListRef { int abase, tstride, lod; };
    auto phase_of = [&](int g) -> int { return wpo * NG + g; };              // phase of group g of this wave
    auto list0 = [&](int li, int cg) -> ListRef { return ListRef{U.fold_off + li * kUpListTaps * tap_stride0 + cg * cg_stride, tap_stride0, U.fold_plane}; };
    auto list1 = [&](int cg) -> ListRef { return ListRef{cg * cg_stride, tap_stride, lo_off}; };

    const int lpos = up_lane_position(U, l32);           // this lane's position inside each of its fragments
    int ring_key = -1;
    bool more = true;
    while (more) {
        [[maybe_unused]] const unsigned long long ts = TICK();
        const int co0 = T.co_t * CO_TILE + wco * CO_FR * 32;
        const float x_scale = scale_of(T.b);
        const float inv_scale = 1.0f / (x_scale * w_scale);
        const bool tile_h = U.odd_h && T.h0 + P.TH >= P.Hout;       // the tile holds the last (even) row
        const bool tile_w = U.odd_w && T.w0 + P.TW >= P.Wout;
        // piece index of this lane's position f at tap (0,0,0): in the low-resolution box while the upsampled chunks run,
        // in the skip box behind them (recomputed at the switch: one array live instead of two).  kind > 0: the bases of a
        // correction list -- lanes outside the output's last row (1) / last column (2) / corner (3) point into the zeroed region
        int bb[PO_FR];
        auto set_bases = [&](bool skip, int kind) {
#pragma unroll
            for (int f = 0; f < PO_FR; ++f) {
                const int g = f / NFG, p = phase_of(g), ph = p >> 1, pw = p & 1;
                const int s = (f - g * NFG) * 32 + lpos;
                bb[f] = half * chs;
                bool on = kind == 0;
                if (s < U.n_sub) {
                    const int tt = s / (U.SH * U.SW);
                    const int rem = s - tt * (U.SH * U.SW);
                    const int i = rem / U.SW;
                    const int j = rem - i * U.SW;
                    if (skip) bb[f] += tt * U.Q1 + (2 * i + ph) * U.P1 + j;
                    else {
                        const bool row = T.h0 + 2 * i + ph == P.Hout - 1, col = T.w0 + 2 * j + pw == P.Wout - 1;
                        on = on || (kind == 1 && row) || (kind == 2 && col) || (kind == 3 && row && col);
                        bb[f] += on ? tt * U.Q0 + (i + ph) * U.P0 + j + pw : U.zero0;
                    }
                } else if (!on) bb[f] += U.zero0;
            }
        };
        set_bases(false, 0);
        if (T.co_t != ring_key) {                             // uniform: (re)load the ring for this channel tile
#pragma unroll
            for (int q = 0; q < CO_FR; ++q) {
                int co = co0 + q * 32 + l32;
                co = co < P.Cout ? co : P.Cout - 1;
                wlane[q] = (co * 16 + 8 * half) * 2;
            }
            const ListRef L = list0(phase_of(0), 0);
            V2CE_LOAD_A(0, L.abase, L.lod)
            V2CE_LOAD_A(1, L.abase + L.tstride, L.lod)
        }

        ACC_T(t_set, ts);
        f32x16 acc[CO_FR][PO_FR];
#pragma unroll
        for (int q = 0; q < CO_FR; ++q)
#pragma unroll
            for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][f][r] = 0.0f;
        f32x16 accd[SC ? CO_FR : 1][SC ? PO_FR : 1];
        if constexpr (SC) {
#pragma unroll
            for (int q = 0; q < CO_FR; ++q)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accd[q][f][r] = 0.0f;
        }
        f16x8 ahd[CO_FR], ald[CO_FR];                        // this chunk's shortcut weights

        // ---- one tap list over fragments [F0, F0 + NF), fully unrolled.  Slots 0 / 1 of the ring hold its taps 0 / 1 on
        // entry and the next list's on exit.
        //   SKIP:   the skip chunks' 27 taps; the piece offset of (dh, dw) depends on the fragment's column parity pw
        //           (columns are de-interleaved in LDS)
        //   !SKIP:  12 folded taps (dt, a, b) of the low-resolution box (a correction list runs on bases that send the lanes
        //           it does not apply to into the zeroed region)
        //   rc:     (fused shortcut; -1 = none) the tap in [4, 8) that reads x[i][j] itself
        //   have:   the first tap's B fragments are already in flight (the previous group's list issued them: `pre`)
        //   pre:    also issue the first-tap reads of the NEXT group's fragments (same chunk, bases valid): its list then starts
        //           without the LDS round trip in front of its first MFMA
        auto run_list = [&](auto F0c, auto NFc, auto SKIPc, int rc, const ListRef cur, const ListRef nxt, const f16x8 *qb, bool have, bool pre) {
            constexpr int F0 = decltype(F0c)::value, NF = decltype(NFc)::value;
            constexpr bool SKIP = decltype(SKIPc)::value;
            constexpr int NT = SKIP ? K3 : kUpListTaps;
            auto addr = [&](int f, int tap) -> int {
                if constexpr (SKIP) {
                    const int pw = phase_of(f / NFG) & 1;                      // uniform
                    const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
                    return bb[f] + dt * U.Q1 + dh * U.P1 + ((pw + dw) & 1) * U.HWh + ((pw + dw) >> 1);
                } else {
                    const int dt = tap >> 2, a = (tap >> 1) & 1, b = tap & 1;
                    return bb[f] + dt * U.Q0 + a * U.P0 + b;
                }
            };
            if (!have) {
#pragma unroll
                for (int f = F0; f < F0 + NF; ++f) {
                    const int a = addr(f, 0);
                    bh[0][f] = qb[a];
                    bl[0][f] = qb[a + 2 * chs];
                }
            }
            if constexpr (!SKIP && F0 + NF < PO_FR) {
                if (pre) {
#pragma unroll
                    for (int f = F0 + NF; f < F0 + 2 * NF; ++f) {
                        const int a = addr(f, 0);
                        bh[0][f] = qb[a];
                        bl[0][f] = qb[a + 2 * chs];
                    }
                }
            }
            step_loop<0, NT>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                constexpr int pt = tap + 2;                   // the tap whose A fragments are fetched now
                constexpr int NBL = SKIP ? NB : NB0;                                             // B sets of this list
                constexpr int cs = NBL == 2 ? tap % 2 : 0, ns = NBL == 2 ? (tap + 1) % 2 : 0;    // B sets of this / the next tap
                if constexpr (pt < NT) {
                    V2CE_LOAD_A(pt % 3, cur.abase + pt * cur.tstride, cur.lod)
                } else {
                    V2CE_LOAD_A(pt % 3, nxt.abase + (pt - NT) * nxt.tstride, nxt.lod)
                }
                if constexpr (NBL == 2 && tap + 1 < NT) {      // the next tap's B fragments, a whole tap ahead
#pragma unroll
                    for (int f = F0; f < F0 + NF; ++f) {
                        const int a = addr(f, tap + 1);
                        bh[ns][f] = qb[a];
                        bl[ns][f] = qb[a + 2 * chs];
                    }
                }
                bool centre = false;
                if constexpr (SC && SKIP) centre = tap == 13;
                if constexpr (SC && !SKIP && tap / 4 == 1) centre = tap == rc;          // uniform
#pragma unroll
                for (int f = F0; f < F0 + NF; ++f) {
#pragma unroll
                    for (int q = 0; q < CO_FR; ++q) {
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % 3][q], bh[cs][f], acc[q][f], 0, 0, 0);
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % 3][q], bl[cs][f], acc[q][f], 0, 0, 0);
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tap % 3][q], bh[cs][f], acc[q][f], 0, 0, 0);
                    }
                    if constexpr (SC) {
                        if (centre) {
#pragma unroll
                            for (int q = 0; q < CO_FR; ++q) {
                                accd[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahd[q], bh[cs][f], accd[q][f], 0, 0, 0);
                                accd[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahd[q], bl[cs][f], accd[q][f], 0, 0, 0);
                                accd[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ald[q], bh[cs][f], accd[q][f], 0, 0, 0);
                            }
                        }
                    }
                    if constexpr (NBL == 1 && tap + 1 < NT) {  // refill in place for the next tap
                        const int a = addr(f, tap + 1);
                        bh[0][f] = qb[a];
                        bl[0][f] = qb[a + 2 * chs];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        using std::integral_constant;
        typedef integral_constant<bool, true> Yes;
        typedef integral_constant<bool, false> No;

        auto chunk_head = [&](int cg) -> const f16x8 * {
            if constexpr (SC) {
                const int wc = cg * cg_stride;
#pragma unroll
                for (int q = 0; q < CO_FR; ++q) {
                    ahd[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_d, wlane[q], wc, 0));
                    ald[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_d, wlane[q], wc + (int)(2 * wplane_d), 0));
                }
            }
            { [[maybe_unused]] const unsigned long long tb = TICK();
            __syncthreads();                                   // barrier gc: pieces[gc & 1] ready
            ACC_T(t_bar, tb); }
            // (opaque per chunk: otherwise the per-tap piece addresses -- bb[f] + tap offset, invariant over the chunks -- are
            // all hoisted out of the chunk loop and spilled, a scratch reload per tap and fragment)
#pragma unroll
            for (int f = 0; f < PO_FR; ++f) asm volatile("" : "+v"(bb[f]));
            return pieces + (gc & 1) * 4 * chs;
        };
        // ---- upsampled chunks: per phase group its folded list, then (edge tiles, even phases) the correction lists
        for (int cg = 0; cg < CG0; ++cg, ++gc) {
            const f16x8 *qb = chunk_head(cg);
            // after this chunk's last list: the next upsampled chunk's first list, or the first skip chunk's
            const ListRef after = cg + 1 < CG0 ? list0(phase_of(0), cg + 1) : (CGE > CG0 ? list1(CG0) : list0(phase_of(0), 0));
            step_loop<0, NG>([&](auto gcst) {
                constexpr int g = decltype(gcst)::value;
                const int p = phase_of(g), ph = p >> 1, pw = p & 1;
                const bool ch = tile_h && ph == 0, cw = tile_w && pw == 0;         // uniform
                const int npass = 1 + (ch ? 1 : 0) + (cw ? 1 : 0) + (ch && cw ? 1 : 0);
                ListRef grp_after = after;
                if constexpr (g + 1 < NG) grp_after = list0(phase_of(g + 1), cg);
                auto pass_list = [&](int k, int &kind) -> ListRef {      // k-th pass of this group
                    if (k == 0) { kind = 0; return list0(p, cg); }
                    if (ch && k == 1) { kind = 1; return list0(4 + pw, cg); }
                    if (cw && k == (ch ? 2 : 1)) { kind = 2; return list0(6 + ph, cg); }
                    kind = 3;
                    return list0(8, cg);
                };
                const int rc = SC ? 4 + (1 - ph) * 2 + (1 - pw) : -1;
                // (group 0 without correction passes hands group 1 its first B fragments: see run_list)
                const bool chain0 = NG == 2 && !(tile_h && (phase_of(0) >> 1) == 0) && !tile_w;
                for (int k = 0; k < npass; ++k) {
                    int kind, nkind;
                    const ListRef cur = pass_list(k, kind);
                    const ListRef nxt = k + 1 < npass ? pass_list(k + 1, nkind) : grp_after;
                    if (k > 0) set_bases(false, kind);                   // (edge tiles only)
                    run_list(integral_constant<int, g * NFG>{}, integral_constant<int, NFG>{}, No{}, k == 0 ? rc : -1, cur, nxt, qb,
                             g == 1 && k == 0 && chain0, g == 0 && chain0);
                }
                if (npass > 1) set_bases(false, 0);
            });
        }
        // ---- skip chunks
        set_bases(true, 0);
        for (int cg = CG0; cg < CGE; ++cg, ++gc) {
            const f16x8 *qb = chunk_head(cg);
            // the last chunk prefetches the first list of the next tile with the same channel tile
            const ListRef nxt = cg + 1 < CG ? list1(cg + 1) : list0(phase_of(0), 0);
            run_list(integral_constant<int, 0>{}, integral_constant<int, PO_FR>{}, Yes{}, -1, list1(cg), nxt, qb, false, false);
        }
        ring_key = T.co_t;

        int poff[PO_FR];
#pragma unroll
        for (int f = 0; f < PO_FR; ++f) {
            const int g = f / NFG, p = phase_of(g), ph = p >> 1, pw = p & 1;
            const int s = (f - g * NFG) * 32 + lpos;
            poff[f] = -1;
            if (s < U.n_sub) {
                const int tt = s / (U.SH * U.SW);
                const int rem = s - tt * (U.SH * U.SW);
                const int i = rem / U.SW;
                const int j = rem - i * U.SW;
                const int t = T.t0 + tt, h = T.h0 + 2 * i + ph, w = T.w0 + 2 * j + pw;
                if (t < P.T && h < P.Hout && w < P.Wout)
                    poff[f] = 4 * ((t * P.Cout) * (P.Hout * P.Woutp)) + 64 * (h * P.Woutp + w);
            }
        }
        [[maybe_unused]] const unsigned long long te = TICK();
        float *stage = reinterpret_cast<float *>(conv_smem + U.stage_off) + wave * 1024;       // this wave's 4 KB
        if (U.dbg == 1) {
#pragma unroll
            for (int f = 0; f < PO_FR; ++f) poff[f] = -1;
        }
        if (U.dbg != 2) {
        up_epilogue<CO_FR, PO_FR>(P, acc, poff, co0, lane, T.b, inv_scale, stage, P.y, P.scale, P.shift, P.act, P.y_absmax);
        if constexpr (SC) {                                     // shortcut: bn_d(conv_d x), no activation, no range slot
            const float wd_scale = reinterpret_cast<const float *>(P.sc_w + 2 * wplane_d)[1];
            up_epilogue<CO_FR, PO_FR>(P, accd, poff, co0, lane, T.b, 1.0f / (x_scale * wd_scale), stage, P.sc_y, P.sc_scale, P.sc_shift,
                                      V2CE_ACT_NONE, nullptr);
        }
        } else {
            float sink = 0.0f;
#pragma unroll
            for (int q = 0; q < CO_FR; ++q)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    sink += acc[q][f][0];
                    if constexpr (SC) sink += accd[q][f][0];
                }
            if (sink == 12345.678f) P.y[0] = sink;
        }
        ACC_T(t_epi, te);
        vb += (int)gridDim.x;
        more = next_tile(vb, T);
    }
#ifdef V2CE_STAMP
    if (lane == 0 && wave == 0) {
        P.stamps[((long long)blockIdx.x * 2 + 0) * 8 + 0] = TICK() - t_all;
        P.stamps[((long long)blockIdx.x * 2 + 0) * 8 + 1] = t_bar;
        P.stamps[((long long)blockIdx.x * 2 + 0) * 8 + 2] = t_epi;
        P.stamps[((long long)blockIdx.x * 2 + 0) * 8 + 3] = t_set;
    }
#endif
#undef V2CE_LOAD_A
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------
// weights: the folded region behind a v2ce_pack_weights_f16x2 buffer
// ---------------------------------------------------------------------------------------------
// up to eight layers per launch (the four decoder conv1 weights of a forward pass: one launch per pass instead of four)
constexpr int kUpFoldBatch = 8;
struct UpFoldLayer {
    const float *w, *sigma;
    float *tail;
    _Float16 *fold;
    int Cout, Cin, C0;
};
struct UpFoldBatch {
    UpFoldLayer L[kUpFoldBatch];
    int n;
    int blk[kUpFoldBatch + 1];    // prefix of the layers' workgroups
};

// max over every tap of every list of the folded region of |sum of (w / sigma)| for the C0 channels -> atomic max into tail[0]
// (the correction lists hold single weights or two-term sums: covered by the plain maximum and the folded lists')
__global__ __launch_bounds__(256) void up_fold_absmax_kernel(UpFoldBatch B) {
    int l = 0;
    while (l + 1 < B.n && (int)blockIdx.x >= B.blk[l + 1]) ++l;
    const float *__restrict__ w = B.L[l].w;
    const float *sigma = B.L[l].sigma;
    float *tail = B.L[l].tail;
    const int Cout = B.L[l].Cout, Cin = B.L[l].Cin, C0 = B.L[l].C0;
    const int bid = blockIdx.x - B.blk[l], nblk = B.blk[l + 1] - B.blk[l];
    // one thread per (co, ci < C0, dt): its nine (dh, dw) weights
    const long long n = (long long)Cout * C0 * 3;
    float m = 0.0f;
    for (long long e = (long long)bid * 256 + threadIdx.x; e < n; e += (long long)nblk * 256) {
        const int dt = (int)(e % 3);
        const long long r = e / 3;
        const int ci = (int)(r % C0), co = (int)(r / C0);
        const float *p = w + ((long long)co * Cin + ci) * 27 + dt * 9;
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            v[k] = p[k];
            if (sigma) v[k] = v[k] / sigma[0];
        }
#pragma unroll
        for (int li = 0; li < 9; ++li)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int mh, mw, sign;
                    up_list_masks(li, a, b, mh, mw, sign);
                    float s = 0.0f;
#pragma unroll
                    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                        for (int dw = 0; dw < 3; ++dw)
                            if ((mh >> dh & 1) && (mw >> dw & 1)) s += v[dh * 3 + dw];
                    m = fmaxf(m, fabsf(s));
                }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // (test before the atomic: 8 000 same-address atomics of a launch serialise in L2 -- 78 us for 19 MB of weights -- and the maximum
    // settles after the first few workgroups)
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __atomic_load_n(reinterpret_cast<unsigned *>(tail), __ATOMIC_RELAXED))
        atomicMax(reinterpret_cast<unsigned *>(tail), __float_as_uint(m));
}

// fold[plane][slot][cg][co][16] (108 slots) = fp16 hi / lo of  +-s * sum over the slot's source taps of w[co][cg*16+j][.] / sigma,
// s = pow2_prescale(tail[0]).  A workgroup takes 32 output channels x one 16-channel group: stages their 27 taps (scaled
// quotients, f32) in LDS -- 32 contiguous runs of W in -- and writes, per slot and plane, 1 KiB contiguous.
__global__ __launch_bounds__(256) void up_fold_pack_kernel(UpFoldBatch B) {
    __shared__ float st[32 * 16 * 27];
    int l = 0;
    while (l + 1 < B.n && (int)blockIdx.x >= B.blk[l + 1]) ++l;
    const float *__restrict__ w = B.L[l].w;
    const float *sigma = B.L[l].sigma;
    const float *tail = B.L[l].tail;
    _Float16 *__restrict__ fold = B.L[l].fold;
    const int Cout = B.L[l].Cout, Cin = B.L[l].Cin, C0 = B.L[l].C0;
    const int bid = blockIdx.x - B.blk[l];
    const int CG0 = C0 / 16, run = 16 * 27;
    const int cg = bid % CG0, co0 = (bid / CG0) * 32;
    const float w_scale = pow2_prescale(tail[0]);
    const float sg = sigma ? sigma[0] : 1.0f;
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int col = e / run, rem = e - col * run;                 // rem = j * 27 + tap
        float v = w[((long long)(co0 + col) * Cin + cg * 16) * 27 + rem];
        if (sigma) v = v / sg;
        st[e] = v * w_scale;
    }
    __syncthreads();
    const long long plane = (long long)kUpSlots * CG0 * Cout * 16;     // halves
    const int col = threadIdx.x >> 3, j2 = (threadIdx.x & 7) * 2;      // this thread: channel col, inputs j2, j2 + 1
    const float *s0 = st + (col * 16 + j2) * 27, *s1 = s0 + 27;
    for (int li = 0; li < 9; ++li) {
        for (int tau = 0; tau < kUpListTaps; ++tau) {
            const int dt = tau >> 2, a = (tau >> 1) & 1, b = tau & 1;
            int mh, mw, sign;
            up_list_masks(li, a, b, mh, mw, sign);
            float x0 = 0.0f, x1 = 0.0f;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw)
                    if ((mh >> dh & 1) && (mw >> dw & 1)) {
                        x0 += s0[dt * 9 + dh * 3 + dw];
                        x1 += s1[dt * 9 + dh * 3 + dw];
                    }
            if (sign < 0) { x0 = -x0; x1 = -x1; }
            typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
            const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
            const f16x2v hi{h0, h1}, lo{(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1)};
            const long long o = (((long long)(li * kUpListTaps + tau) * CG0 + cg) * Cout + co0 + col) * 16 + j2;
            *reinterpret_cast<f16x2v *>(fold + o) = hi;
            *reinterpret_cast<f16x2v *>(fold + plane + o) = lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: box, LDS pitches and lane order
// ---------------------------------------------------------------------------------------------
struct UpCfg {
    int tt, sh, sw;               // phase sub-box
    int P0, Q0, P1, Q1;           // LDS pitches (rows / time steps) of the low-resolution and of the skip box
    int lbit[5];                  // lane order (up_lane_position)
    double conf0, conf1;          // LDS cycles per 16-lane group of a B-fragment read (1 = conflict-free), upsampled / skip chunks
};

// LDS cycles per 16-lane group of the consumers' ds_read_b128 (each group is served in max-multiplicity-per-slot cycles: a 16-byte
// piece covers one of the sixteen slots of the 256-byte bank row), averaged over the groups of the nfg fragments of a phase
double up_conflicts(const int (&lbit)[5], int sh, int sw, int n_sub, int nfg, int pr, int qr, int rowmul) {
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    double sum = 0.0;
    for (int k = 0; k < nfg; ++k)
        for (int g = 0; g < 2; ++g) {
            int cnt[16] = {0}, worst = 1;
            int seen[16][16];
            for (int idx = 0; idx < 16; ++idx) {
                const int rank = g * 16 + idx;
                int pos = 0;
                for (int b = 0; b < 5; ++b) pos |= ((rank >> b) & 1) << lbit[b];
                const int s = k * 32 + pos;
                int a = 0;                                     // lanes past the sub-box read piece 0
                if (s < n_sub) {
                    const int tt = s / (sh * sw), rem = s - tt * (sh * sw), i = rem / sw, j = rem - i * sw;
                    a = tt * qr + rowmul * i * pr + j;
                }
                const int slot = a & 15;
                bool dup = false;                              // identical addresses broadcast
                for (int c = 0; c < cnt[slot]; ++c) dup = dup || seen[slot][c] == a;
                if (!dup) seen[slot][cnt[slot]++] = a;
                worst = cnt[slot] > worst ? cnt[slot] : worst;
            }
            (void)grp;
            sum += worst;
        }
    return sum / (2.0 * nfg);
}

// For a sub-box: the pitches and lane order with the fewest LDS cycles per read, the two chunk kinds searched independently per
// lane order (w0 / w1 = reads of a tile on upsampled / skip chunks); false when nothing fits the LDS budget (chs <= 1152 pieces)
bool up_pitches(UpCfg &c, int nfg, bool edges, double w0, double w1) {
    const int HT = c.tt + 2, HH0 = c.sh + 2, HW0 = c.sw + 2, HH = 2 * c.sh + 2, HWd = 2 * c.sw + 2, n_sub = c.tt * c.sh * c.sw;
    int perm[5] = {0, 1, 2, 3, 4};
    double best = -1.0;
    do {
        double b0 = -1.0, b1 = -1.0;
        int p0 = 0, q0 = 0, p1 = 0, q1 = 0;
        for (int P = HW0; P <= HW0 + 8; ++P)
            for (int pad = 0; pad < 16; ++pad) {
                const int Q = HH0 * P + pad, plane0 = HT * Q, z0 = (plane0 + 63) & ~63;
                if ((edges ? z0 + plane0 : plane0) > 1152) continue;
                const double cf = up_conflicts(perm, c.sh, c.sw, n_sub, nfg, P, Q, 1) + 1e-4 * plane0;
                if (b0 < 0 || cf < b0) { b0 = cf; p0 = P; q0 = Q; }
            }
        for (int P = HWd; P <= HWd + 8; ++P)
            for (int pad = 0; pad < 16; ++pad) {
                const int Q = HH * P + pad;
                if (HT * Q > 1152) continue;
                const double cf = up_conflicts(perm, c.sh, c.sw, n_sub, nfg, P, Q, 2) + 1e-4 * HT * Q;
                if (b1 < 0 || cf < b1) { b1 = cf; p1 = P; q1 = Q; }
            }
        if (b0 < 0 || b1 < 0) return false;
        const double tot = w0 * b0 + w1 * b1;
        if (best < 0 || tot < best) {
            best = tot;
            c.P0 = p0; c.Q0 = q0; c.P1 = p1; c.Q1 = q1;
            for (int b = 0; b < 5; ++b) c.lbit[b] = perm[b];
            c.conf0 = b0; c.conf1 = b1;
        }
    } while (std::next_permutation(perm, perm + 5));
    return true;
}

// phase sub-box (tt, sh, sw): tt * sh * sw <= n_sub_max positions per phase.  Fewest workgroup rounds over the CUs; among the
// boxes within 3 % of the fewest tiles the one whose B-fragment reads cost the fewest LDS cycles, then the smaller halo, then
// wide rows.  fixed: a caller-given box (only the pitches are searched)
bool choose_up_cfg(UpCfg &out, int B, int T, int Ho, int Wo, int n_sub_max, int nfg, int n_co_tiles, int n_cu, bool edges, double w0,
                   double w1, const UpCfg *fixed) {
    struct Cand { UpCfg c; long long rounds, blocks, halo; };
    std::vector<Cand> cands;
    for (int tt = 1; tt <= 16 && tt <= (T > 1 ? 2 * T - 1 : 1); tt *= 2)
        for (int sh = 1; sh <= 64 && 2 * (sh - 1) < Ho; ++sh)
            for (int sw = 1; sw <= 64 && 2 * (sw - 1) < Wo; ++sw) {
                if (fixed && (tt != fixed->tt || sh != fixed->sh || sw != fixed->sw)) continue;
                if (tt * sh * sw > n_sub_max) break;
                const long long halo1 = (long long)(tt + 2) * (2 * sh + 2) * (2 * sw + 2);
                if (halo1 > 1152) break;                     // (128 B of pieces per element + 16 KB of epilogue staging in 160 KB)
                const long long nsp = (long long)B * ((T + tt - 1) / tt) * ((Ho + 2 * sh - 1) / (2 * sh)) * ((Wo + 2 * sw - 1) / (2 * sw));
                Cand k{};
                k.c.tt = tt; k.c.sh = sh; k.c.sw = sw;
                k.blocks = 8 * ((nsp + 7) / 8) * n_co_tiles;
                k.rounds = (k.blocks + n_cu - 1) / n_cu;
                k.halo = halo1 + 2 * (long long)(tt + 2) * (sh + 2) * (sw + 2);
                cands.push_back(k);
            }
    if (cands.empty()) return false;
    long long min_rounds = cands[0].rounds, min_blocks = -1;
    for (const Cand &k : cands) min_rounds = k.rounds < min_rounds ? k.rounds : min_rounds;
    for (const Cand &k : cands)
        if (k.rounds == min_rounds && (min_blocks < 0 || k.blocks < min_blocks)) min_blocks = k.blocks;
    bool have = false;
    double best_cost = 0.0;
    long long best_halo = 0;
    for (Cand &k : cands) {
        if (k.rounds != min_rounds || k.blocks * 100 > min_blocks * 103) continue;
        if (!up_pitches(k.c, nfg, edges, w0, w1)) continue;
        const double cost = (double)k.blocks * (w0 * k.c.conf0 + w1 * k.c.conf1);
        if (!have || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && (k.halo < best_halo || (k.halo == best_halo && k.c.sw > out.sw)))) {
            have = true; best_cost = cost; best_halo = k.halo; out = k.c;
        }
    }
    return have;
}

thread_local char *g_up_name_out = nullptr;
thread_local size_t g_up_name_cap = 0;

template <int WCO, int CO_FR, int PO_FR, int FUSE>
int launch_up(UpParams U, const v2ce_conv3d_desc &d, hipStream_t stream) {
    ConvParams &P = U.C;
    constexpr int CO_TILE = WCO * CO_FR * 32, NFG = PO_FR / WCO;
    if (g_up_name_out) {
        snprintf(g_up_name_out, g_up_name_cap, "conv3d_up_kernel<%d,%d,%d,%d>", WCO, CO_FR, PO_FR, FUSE);
        return V2CE_OK;
    }
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n < 8 ? 8 : (n / 8) * 8;
    }();
    P.n_co_tiles = (d.Cout + CO_TILE - 1) / CO_TILE;
    // box, pitches and lane order: searched once per launch geometry
    UpCfg cfg{};
    {
        static std::mutex mu;
        static std::map<std::tuple<int, int, int, int, int, int, int, int, int, int>, UpCfg> cache;
        std::lock_guard<std::mutex> g(mu);
        const auto key = std::make_tuple(d.B, d.T, d.Hout, d.Wout, d.C0, d.C1, d.Cout, NFG + 16 * WCO, d.tile_t, d.tile_h * 4096 + d.tile_w);
        auto it = cache.find(key);
        if (it == cache.end()) {
            UpCfg fixed{};
            bool have_fixed = false;
            if (d.tile_t > 0 && d.tile_h > 0 && d.tile_w > 0) { fixed.tt = d.tile_t; fixed.sh = d.tile_h / 2; fixed.sw = d.tile_w / 2; have_fixed = true; }
            else {
                char name[64];
                snprintf(name, sizeof name, "V2CE_UPBOX_%dx%d_%d", d.Hout, d.Wout, NFG * 32);
                if (const char *e = getenv(name)) {
                    int a, b, c;
                    if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) { fixed.tt = a; fixed.sh = b / 2; fixed.sw = c / 2; have_fixed = true; }
                }
            }
            const bool ok = choose_up_cfg(cfg, d.B, d.T, d.Hout, d.Wout, NFG * 32, NFG, P.n_co_tiles, n_cu, U.odd_h || U.odd_w,
                                          12.0 * (d.C0 / 16), 27.0 * (d.C1 / 16), have_fixed ? &fixed : nullptr);
            V2CE_REQUIRE(ok, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_up2: no box fits (%d positions per phase; TH, TW must be even; halo <= 1152 elements)", NFG * 32);
            it = cache.emplace(key, cfg).first;
            if (getenv("V2CE_UP_VERBOSE"))
                fprintf(stderr, "[up cfg %dx%d C0=%d C1=%d Cout=%d] box %dx%dx%d pitches low %d/%d skip %d/%d lane bits %d%d%d%d%d LDS cycles per read %.2f / %.2f\n", d.Hout,
                        d.Wout, d.C0, d.C1, d.Cout, cfg.tt, 2 * cfg.sh, 2 * cfg.sw, cfg.P0, cfg.Q0, cfg.P1, cfg.Q1, cfg.lbit[0], cfg.lbit[1], cfg.lbit[2],
                        cfg.lbit[3], cfg.lbit[4], cfg.conf0, cfg.conf1);
        }
        cfg = it->second;
    }
    P.TT = cfg.tt; P.TH = 2 * cfg.sh; P.TW = 2 * cfg.sw;
    U.SH = cfg.sh; U.SW = cfg.sw; U.n_sub = cfg.tt * cfg.sh * cfg.sw;
    P.n_pos = 4 * U.n_sub;
    P.HT = P.TT + 2; P.HH = P.TH + 2; P.HWd = P.TW + 2;
    U.HH0 = U.SH + 2; U.HW0 = U.SW + 2;
    U.HWh = P.HWd / 2;
    U.P0 = cfg.P0; U.Q0 = cfg.Q0; U.P1 = cfg.P1; U.Q1 = cfg.Q1;
    for (int b = 0; b < 5; ++b) U.lbit[b] = cfg.lbit[b];
    P.plane = P.HT * U.Q1;                                 // pitched slots of a skip chunk
    U.plane0 = P.HT * U.Q0;
    V2CE_REQUIRE(P.plane <= 1152, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_up2: the box's halo does not fit (%d > 1152 elements)", P.plane);
    P.nT = (d.T + P.TT - 1) / P.TT; P.nH = (d.Hout + P.TH - 1) / P.TH; P.nW = (d.Wout + P.TW - 1) / P.TW;
    P.n_spatial = d.B * P.nT * P.nH * P.nW;
    P.xcd_remap = 1;
    P.per_xcd = (P.n_spatial + 7) / 8;
    const long long blocks = (long long)8 * P.per_xcd * P.n_co_tiles;
    V2CE_REQUIRE(blocks < (1ll << 31), V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_up2: grid too large");
    U.zero0 = (U.plane0 + 63) & ~63;
    int chs = (P.plane + 63) & ~63;
    if (chs < U.zero0 + U.plane0) chs = (U.zero0 + U.plane0 + 63) & ~63;      // (tiny boxes only)
    U.chs = chs;
    U.stage_off = chs * 128;
    const size_t lds = (size_t)chs * 128 + 4 * 4096;
    V2CE_REQUIRE(lds <= 160 * 1024, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_up2: %zu B of LDS", lds);
    auto kern = conv3d_up_kernel<WCO, CO_FR, PO_FR, FUSE>;
    V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    P.total_blocks = (int)blocks;
    const unsigned grid = (unsigned)(blocks > n_cu ? n_cu : blocks);
#ifdef V2CE_STAMP
    V2CE_HIP_CHECK(hipMalloc(&P.stamps, (size_t)grid * 16 * sizeof(unsigned long long)));
    V2CE_HIP_CHECK(hipMemset(P.stamps, 0, (size_t)grid * 16 * sizeof(unsigned long long)));
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, U);
    V2CE_HIP_CHECK(hipGetLastError());
#ifdef V2CE_STAMP
    {
        std::vector<unsigned long long> h((size_t)grid * 16);
        V2CE_HIP_CHECK(hipDeviceSynchronize());
        V2CE_HIP_CHECK(hipMemcpy(h.data(), P.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        V2CE_HIP_CHECK(hipFree(P.stamps));
        auto med = [&](int role, int a) {
            std::vector<unsigned long long> v;
            for (unsigned k = 0; k < grid; ++k) v.push_back(h[(k * 2 + role) * 8 + a]);
            std::sort(v.begin(), v.end());
            return (long long)v[v.size() / 2];
        };
        const double tiles = (double)blocks / grid;
        fprintf(stderr, "[stamp up<%d,%d,%d,%d> CG0=%d CG=%d box=%dx%dx%d plane0=%d plane1=%d blocks=%lld = %.2f tiles per workgroup] "
                "consumer wave 0 (cycles, median over workgroups): kernel %lld, in chunk barriers %lld (%.1f %%), epilogues %lld (%.1f %%), tile setup %lld (%.1f %%) | "
                "producer wave 4: kernel %lld, in chunk barriers %lld (%.1f %%)\n",
                WCO, CO_FR, PO_FR, FUSE, U.CG0, P.Cin / 16, P.TT, P.TH, P.TW, U.plane0, P.plane, blocks, tiles,
                med(0, 0), med(0, 1), 100.0 * med(0, 1) / med(0, 0), med(0, 2), 100.0 * med(0, 2) / med(0, 0), med(0, 3), 100.0 * med(0, 3) / med(0, 0),
                med(1, 0), med(1, 1), 100.0 * med(1, 1) / med(1, 0));
    }
#endif
    return V2CE_OK;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

static size_t up_fold_off(int Cout, int Cin) { return (size_t)Cout * Cin * 27 * 4 + 16; }   // plain planes + {max, scale, 0, 0}

extern "C" size_t v2ce_pack_weights_f16x2_up_bytes(int Cout, int C0, int C1) {
    if (Cout <= 0 || C0 <= 0 || C1 < 0) return 0;
    return up_fold_off(Cout, C0 + C1) + (size_t)kUpSlots * C0 * Cout * 4;
}

// enqueue the two fold passes behind a tail[0] that already holds max |w / sigma| (see v2ce_pack_weights_f16x2_up and
// v2ce_sn_update_batch, which call them around their own plain packs)
int v2ce::v2ce_up_fold_batch(const float *const *w, const int *Cout, const int *Cin, const int *C0, const float *const *sigma, void *const *w_up,
                             int n, int pass, hipStream_t st) {
    V2CE_REQUIRE(n >= 0 && n <= kUpFoldBatch, V2CE_ERR_BAD_ARG, "v2ce_up_fold_batch: 0..%d layers", kUpFoldBatch);
    if (n == 0) return V2CE_OK;
    UpFoldBatch B{};
    B.n = n;
    for (int l = 0; l < n; ++l) {
        char *base = static_cast<char *>(w_up[l]);
        B.L[l] = UpFoldLayer{w[l], sigma[l], reinterpret_cast<float *>(base + (size_t)Cout[l] * Cin[l] * 27 * 4),
                             reinterpret_cast<_Float16 *>(base + up_fold_off(Cout[l], Cin[l])), Cout[l], Cin[l], C0[l]};
        const long long items = (long long)Cout[l] * C0[l] * 3;
        const long long nb = (items + 255) / 256;
        B.blk[l + 1] = B.blk[l] + (pass == 0 ? (int)(nb < 2048 ? nb : 2048) : (Cout[l] / 32) * (C0[l] / 16));
    }
    if (pass == 0) hipLaunchKernelGGL(up_fold_absmax_kernel, dim3((unsigned)B.blk[n]), dim3(256), 0, st, B);
    else hipLaunchKernelGGL(up_fold_pack_kernel, dim3((unsigned)B.blk[n]), dim3(256), 0, st, B);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
int v2ce::v2ce_up_fold_absmax(const float *w, int Cout, int Cin, int C0, const float *sigma, void *w_up, hipStream_t st) {
    return v2ce_up_fold_batch(&w, &Cout, &Cin, &C0, &sigma, &w_up, 1, 0, st);
}
int v2ce::v2ce_up_fold_pack(const float *w, int Cout, int Cin, int C0, const float *sigma, void *w_up, hipStream_t st) {
    return v2ce_up_fold_batch(&w, &Cout, &Cin, &C0, &sigma, &w_up, 1, 1, st);
}

extern "C" int v2ce_pack_weights_f16x2_up(const float *w, int Cout, int C0, int C1, const float *sigma, void *w_up,
                                          v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(w && w_up && Cout > 0 && Cout % 32 == 0 && C0 > 0 && C0 % 16 == 0 && C1 > 0 && C1 % 16 == 0, V2CE_ERR_BAD_ARG,
                 "v2ce_pack_weights_f16x2_up: needs Cout %% 32 == 0 and C0, C1 positive multiples of 16");
    const int Cin = C0 + C1;
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(w_up);
    float *tail = reinterpret_cast<float *>(base + (size_t)Cout * Cin * 27 * 4);
    V2CE_HIP_CHECK(hipMemsetAsync(tail, 0, 4 * sizeof(float), st));
    // max |w / sigma| over the plain weights, then over the folded sums; then the two packs with the common scale
    int rc = v2ce_pack_weights_f16x2_absmax_only(w, Cout, Cin, 27, sigma, w_up, stream);
    if (rc != V2CE_OK) return rc;
    v2ce_up_fold_absmax(w, Cout, Cin, C0, sigma, w_up, st);
    rc = v2ce_pack_weights_f16x2_pack_only(w, Cout, Cin, 27, sigma, w_up, stream);
    if (rc != V2CE_OK) return rc;
    v2ce_up_fold_pack(w, Cout, Cin, C0, sigma, w_up, st);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

static int up_dispatch(const v2ce_conv3d_desc *desc, const float *x0, const float *x1, const void *w_up,
                       const float *scale, const float *shift, float *y, const float *x0_absmax,
                       const float *x1_absmax, float *y_absmax, const void *sc_w, const float *sc_scale,
                       const float *sc_shift, float *sc_y, v2ce_stream_t stream, bool part = false) {
    clear_error();
    V2CE_REQUIRE(desc && (g_up_name_out || (x0 && (x1 || part) && w_up && scale && shift && y)), V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2: null pointer");
    V2CE_REQUIRE(!part || !sc_w, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2_part: no fused shortcut on a partial launch");
    const v2ce_conv3d_desc &d = *desc;
    V2CE_REQUIRE(d.B > 0 && d.T > 0 && d.C0 > 0 && d.C1 > 0 && d.Hin > 0 && d.Win > 0 && d.Cout > 0, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2: bad shape");
    V2CE_REQUIRE(d.ksize == 3 && d.stride_hw == 1 && d.precision == V2CE_PRECISION_F16X2 && d.layout == V2CE_LAYOUT_C16, V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd_up2: a split-half 3x3x3 stride-1 conv on channels-last-16 activations");
    V2CE_REQUIRE(d.H0 == (d.Hin + 1) / 2 && d.W0 == (d.Win + 1) / 2 && d.Hout == d.Hin && d.Wout == d.Win, V2CE_ERR_BAD_ARG,
                 "v2ce_conv3d_fwd_up2: x0 must be the 2x nearest-upsample source (H0 = ceil(Hin / 2), W0 = ceil(Win / 2)), got %dx%d for %dx%d",
                 d.H0, d.W0, d.Hin, d.Win);
    V2CE_REQUIRE(d.C0 % 16 == 0 && d.C1 % 16 == 0 && d.Cout % 32 == 0, V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd_up2: channel counts must be multiples of 16 (inputs) / 32 (outputs)");
    V2CE_REQUIRE(d.act >= 0 && d.act <= 2, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2: act %d", d.act);
    const int W0p = d.W0_pitch > 0 ? d.W0_pitch : d.W0, Winp = d.Win_pitch > 0 ? d.Win_pitch : d.Win;
    const int Woutp = d.Wout_pitch > 0 ? d.Wout_pitch : d.Wout;
    V2CE_REQUIRE(W0p >= d.W0 && Winp >= d.Win && Woutp >= d.Wout, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2: a row pitch is smaller than its width");
    V2CE_REQUIRE((long long)d.T * d.C0 * d.H0 * W0p < (1ll << 29) && (long long)d.T * d.C1 * d.Hin * Winp < (1ll << 29) &&
                 (long long)d.T * d.Cout * d.Hout * Woutp < (1ll << 29), V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd_up2: a single sequence exceeds the 2 GiB buffer-descriptor range");
    V2CE_REQUIRE(x0_absmax || !x1_absmax, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2: x1_absmax without x0_absmax");
    V2CE_REQUIRE(!x0_absmax || x1_absmax, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2: x0_absmax without x1_absmax");
    V2CE_REQUIRE(!sc_w || (d.Cout <= 32 && sc_scale && sc_shift && (sc_y || g_up_name_out)), V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd_up2: the fused shortcut needs <= 32 output channels and its scale / shift / output");
    const size_t total = v2ce_pack_weights_f16x2_up_bytes(d.Cout, d.C0, d.C1);
    V2CE_REQUIRE(total < (1ull << 31), V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_up2: the weight buffer exceeds the 2 GiB buffer-descriptor range");

    UpParams U{};
    ConvParams &P = U.C;
    P.x0 = x0; P.x1 = x1; P.scale = scale; P.shift = shift; P.y = y;
    P.B = d.B; P.T = d.T; P.C0 = d.C0; P.H0 = d.H0; P.W0 = d.W0; P.C1 = d.C1; P.Hin = d.Hin; P.Win = d.Win;
    P.Cin = d.C0 + d.C1; P.Cout = d.Cout; P.Hout = d.Hout; P.Wout = d.Wout;
    P.W0p = W0p; P.Winp = Winp; P.Woutp = Woutp;
    P.c16 = 1;
    P.act = d.act;
    P.wq = static_cast<const _Float16 *>(w_up);
    P.x0_absmax = x0_absmax; P.x1_absmax = x1_absmax; P.y_absmax = y_absmax;
    P.guard = y_absmax ? y_absmax + 1 : nullptr;
    P.amax_bs = d.absmax_batch_stride;
    V2CE_REQUIRE(P.amax_bs == 0 || P.amax_bs >= 2, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_up2: absmax_batch_stride must be 0 or >= 2");
    P.sc_w = static_cast<const _Float16 *>(sc_w); P.sc_scale = sc_scale; P.sc_shift = sc_shift; P.sc_y = sc_y;
    U.CG0 = d.C0 / 16;
    U.part = part ? 1 : 0;
    U.odd_h = d.Hout & 1; U.odd_w = d.Wout & 1;
    U.fold_off = (int)up_fold_off(d.Cout, P.Cin);
    U.fold_plane = (int)((size_t)kUpSlots * d.C0 * d.Cout * 2);
    { const char *e = getenv("V2CE_UP_DBG"); U.dbg = e ? atoi(e) : 0; }
    hipStream_t st = as_stream(stream);
    if (sc_w) return launch_up<1, 1, 4, 2>(U, d, st);
    if (d.Cout <= 32) return launch_up<1, 1, 4, 0>(U, d, st);
    // >= 64 output channels: 128 channels x 256 positions (two phases per wave), 64 x 512 (one phase per wave) or 64 x 256 (the same at
    // half the positions) -- whichever walks the CUs in the fewest tile-equivalents (whole rounds of the persistent grid x tile size).
    // dec0 (33 x 44 planes, 256 channels): 800 tiles of 128 x 256 are FOUR rounds on 256 CUs where 3.1 would do; 1 600 half tiles
    // are seven half rounds.  V2CE_UP_TILE = 0 (choose) | 1 | 2 | 3 forces one of the three.
    static const int force = [] { const char *e = getenv("V2CE_UP_TILE"); return e ? atoi(e) : 0; }();
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n < 8 ? 8 : (n / 8) * 8;
    }();
    auto cost = [&](int co_tile, int n_sub_max, int nfg, double size) -> double {      // rounds x tile size, for the best box of that shape
        UpCfg c{};
        const int n_co = (d.Cout + co_tile - 1) / co_tile;
        if (!choose_up_cfg(c, d.B, d.T, d.Hout, d.Wout, n_sub_max, nfg, n_co, n_cu, U.odd_h || U.odd_w, 12.0 * (d.C0 / 16), 27.0 * (d.C1 / 16), nullptr))
            return 1e30;
        const long long nsp = (long long)d.B * ((d.T + c.tt - 1) / c.tt) * ((d.Hout + 2 * c.sh - 1) / (2 * c.sh)) * ((d.Wout + 2 * c.sw - 1) / (2 * c.sw));
        const long long blocks = 8 * ((nsp + 7) / 8) * n_co;
        return (double)((blocks + n_cu - 1) / n_cu) * size;
    };
    int pick = force;
    if (!pick) {
        {
            static std::mutex mu;
            static std::map<std::tuple<int, int, int, int, int, int, int>, int> cache;
            std::lock_guard<std::mutex> g(mu);
            const auto key = std::make_tuple(d.B, d.T, d.Hout, d.Wout, d.C0, d.C1, d.Cout);
            auto it = cache.find(key);
            if (it == cache.end()) {
                const double c1 = d.Cout >= 128 ? cost(128, 64, 2, 1.0) : 1e30, c2 = cost(64, 128, 4, 1.0), c3 = cost(64, 64, 2, 0.5);
                // (the half tile pays twice the weight stream per MFMA: it must save at least 8 % to be taken)
                int p = d.Cout >= 128 ? 1 : 2;
                double best = p == 1 ? c1 : c2;
                if (c2 < best - 1e-9) { p = 2; best = c2; }
                if (c3 < 0.92 * best) p = 3;
                it = cache.emplace(key, p).first;
            }
            pick = it->second;
        }
    }
    if (pick == 1 && d.Cout >= 128) return launch_up<2, 2, 4, 0>(U, d, st);
    if (pick == 3) return launch_up<1, 2, 2, 0>(U, d, st);
    return launch_up<1, 2, 4, 0>(U, d, st);
}

extern "C" int v2ce_conv3d_fwd_up2(const v2ce_conv3d_desc *desc, const float *x0, const float *x1, const void *w_up,
                                   const float *scale, const float *shift, float *y, const float *x0_absmax,
                                   const float *x1_absmax, float *y_absmax, const void *sc_w, const float *sc_scale,
                                   const float *sc_shift, float *sc_y, v2ce_stream_t stream) {
    g_up_name_out = nullptr;
    return up_dispatch(desc, x0, x1, w_up, scale, shift, y, x0_absmax, x1_absmax, y_absmax, sc_w, sc_scale, sc_shift, sc_y, stream);
}

extern "C" int v2ce_conv3d_fwd_up2_part(const v2ce_conv3d_desc *desc, const float *x0, const void *w_up, const float *scale, const float *shift,
                                        float *y, const float *x0_absmax, const float *x1_absmax, float *y_absmax, v2ce_stream_t stream) {
    g_up_name_out = nullptr;
    return up_dispatch(desc, x0, nullptr, w_up, scale, shift, y, x0_absmax, x1_absmax, y_absmax, nullptr, nullptr, nullptr, nullptr, stream, true);
}

extern "C" int v2ce_conv3d_up2_variant(const v2ce_conv3d_desc *desc, int with_shortcut, char *name, size_t cap) {
    V2CE_REQUIRE(name && cap > 0, V2CE_ERR_BAD_ARG, "v2ce_conv3d_up2_variant: no buffer");
    name[0] = '\0';
    g_up_name_out = name;
    g_up_name_cap = cap;
    static const float dummy = 0.0f;
    const int rc = up_dispatch(desc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               with_shortcut ? &dummy : nullptr, with_shortcut ? &dummy : nullptr, with_shortcut ? &dummy : nullptr,
                               nullptr, nullptr);
    g_up_name_out = nullptr;
    return rc;
}
