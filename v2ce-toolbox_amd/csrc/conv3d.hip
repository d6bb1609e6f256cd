// conv3d.hip -- fused 3-D convolution for the V2ce3d UNet on gfx950.
//
// Two kernels behind v2ce_conv3d_fwd: conv3d_kernel (exact f32 MFMA; described first) and
// conv3d_f16x2_ws_kernel (operands split into fp16 halves, wave-specialised, persistent; its own
// header further down), selected by desc.precision.
//
// Replaces nn.Conv3d + BatchNorm3d(eval) + ReLU/LeakyReLU + residual add + nearest-upsample +
// channel concat as used by /root/reference/scripts/submodules.py:96,115-124,226-264 and
// /root/reference/scripts/unet_2layer.py:341-374.
//
// Formulation: direct convolution as a GEMM per workgroup,
//     D[co][pos] += W[co][(ci,tap)] * X[(ci,tap)][pos]
// with v_mfma_f32_32x32x2_f32 (exact f32 FMA chain; 64 FLOP/clk/SIMD = the chip's f32 peak).
// Output POSITIONS sit on the MFMA N / lane dimension and output CHANNELS on the M / register
// dimension, so each accumulator register is one channel of 32 consecutive positions: the epilogue
// stores (and the residual loads) are 128-byte runs of the planar [B][T][C][H][W] layout.
//
// Per workgroup (256 threads = 4 waves): an output box (TT x TH x TW positions, <= POS_TILE) times
// CO_TILE output channels.  The K loop walks the input channels in chunks of CK; per chunk the
// input HALO box ((TT+2) x ((TH-1)s+3) x ((TW-1)s+3)) and the [tap][ci][co] weight slab are staged
// once in LDS and all 27 taps read their shifted B fragments straight out of the halo (no im2col
// duplication; LDS read bandwidth is far below the f32 MFMA's appetite: 6 ds_read_b32 per 8 MFMA =
// 512 matrix cycles).  The halo loader is a gather, which is what makes the decoder's virtual
// "nearest-upsample + concat" input free: channels < C0 are fetched from the low-resolution tensor
// through row/column index maps, the rest from the skip tensor -- neither the upsampled tensor nor
// the concatenation is ever written to HBM.
#include "conv3d_dev.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#ifdef V2CE_STAMP
#include <algorithm>
#include <vector>
#endif
#include <type_traits>

// producer-side epilogue (conv3d_f16x2_ws_kernel, OPT bit 0): position fragments per wave whose epilogue stays with the consumers
// (measured on dec3.conv2 + head: 2 of 4 -> 0.94 ms, 1 of 4 -> 0.98, 0 of 4 -> 0.99, all four = no hand-over -> 1.01)
#ifndef V2CE_PEPI_KEEP
#define V2CE_PEPI_KEEP 2
#endif

namespace v2ce {
namespace {

#if defined(__HIP_DEVICE_COMPILE__)

// byte offsets (relative to the sequence base of the source tensor) of this thread's halo elements;
// kOOB = zero padding (hardware range check of the buffer load supplies the zero).
// Element r of the halo plane <-> LDS offset r inside the channel.
// C16: the same for the channels-last-16 layout -- byte offset of the element's 64-byte channel group inside
// channel group 0 of its time step (the chunk's group adds (group) * Hs * Ws * 64).
template <int EPT, bool C16 = false>
__device__ __forceinline__ void halo_offsets(const ConvParams &P, int tin0, int hin0, int win0,
                                             bool src1, int tid, unsigned (&goff)[EPT], int gs = 1) {
    const int Cs = src1 ? P.C1 : P.C0;
    const int Hs = src1 ? P.Hin : P.H0, Ws = src1 ? P.Winp : P.W0p;      // row pitch of the source
    const bool mapped = !src1 && P.hmap != nullptr;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int r = tid + 256 * i;
        unsigned off = kOOB;
        if (r < P.plane) {
            const int ht = r / (P.HH * P.HWd);
            const int rem = r - ht * (P.HH * P.HWd);
            const int hh = rem / P.HWd;
            const int hw = rem - hh * P.HWd;
            const int t = tin0 + ht, h = hin0 + hh * gs, w = win0 + hw * gs;   // gs > 1: strided 1x1x1 gather
            if (t >= 0 && t < P.T && h >= 0 && h < P.Hin && w >= 0 && w < P.Win) {
                const int hs = mapped ? P.hmap[h] : h;
                const int ws = mapped ? P.wmap[w] : w;
                off = C16 ? 4u * (unsigned)((t * Cs) * (Hs * Ws)) + 64u * (unsigned)(hs * Ws + ws)
                          : 4u * (unsigned)((t * Cs) * (Hs * Ws) + hs * Ws + ws);
            }
        }
        goff[i] = off;
    }
}

// the same for the folded 1x1x1 tail: element r = output position r of the box (TT x TH x TW, row-major), read at
// (t, h * tS, w * tS) of the tail's virtual input (tx0 through thmap / twmap, or tx1); channels-last-16 sources
template <int EPT>
__device__ __forceinline__ void tail_offsets(const ConvParams &P, int t0, int h0, int w0, bool src1, int tid,
                                             unsigned (&goff)[EPT], unsigned &gsel) {
    const int Cs = src1 ? P.tC1 : P.tC0;
    const int Hs = src1 ? P.tHin : P.tH0, Ws = src1 ? P.tWinp : P.tW0p;
    const bool mapped = !src1 && P.thmap != nullptr;
    gsel = 0;                                              // 4 bits per element slot: its channel group inside the super-chunk
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + 256 * i;
        const int g = e / P.tNPP, r = e - g * P.tNPP;
        unsigned off = kOOB;
        if (g < P.tTCH && r < P.n_pos) {
            const int tt = r / (P.TH * P.TW);
            const int rem = r - tt * (P.TH * P.TW);
            const int th = rem / P.TW;
            const int tw = rem - th * P.TW;
            const int t = t0 + tt, h = (h0 + th) * P.tS, w = (w0 + tw) * P.tS;
            if (t < P.T && h < P.tHin && w < P.tWin) {
                const int hs = mapped ? P.thmap[h] : h;
                const int ws = mapped ? P.twmap[w] : w;
                off = 4u * (unsigned)((t * Cs) * (Hs * Ws)) + 64u * (unsigned)(hs * Ws + ws) + (unsigned)g * (unsigned)(Hs * Ws * 64);
            }
        }
        goff[i] = off;
        gsel |= (unsigned)(g < 15 ? g : 15) << (4 * i);
    }
}

template <int EPT, int WPT>
struct DmaState {
    unsigned goff[EPT];
    unsigned woff[WPT];
    int cur_src;
    __amdgpu_buffer_rsrc_t rs_in, rs_w;
    int src_cstride4, src_cbase;
};

// issue the LDS-DMA of input-channel chunk ci0 into the buffer starting at `hb`.
// part/nparts: the chunk's DMA instructions are dealt round-robin over `nparts` call sites so that
// they issue in the shadow of the MFMAs of the (dt,dh) tap groups instead of in one burst in front
// of them (an LDS-DMA costs ~60-180 issue cycles, MI355X_MICROARCH.md "LDS-DMA piece").
template <int KS, int CK, int EPT, int WPT, int W_ROWS>
__device__ __forceinline__ void issue_chunk(const ConvParams &P, DmaState<EPT, WPT> &D, int ci0,
                                            float *hb, int chs, int b, int wave, int tin0, int hin0,
                                            int win0, int part, int nparts) {
    constexpr int K3 = KS * KS * KS;
    if (part == 0) {
    const int want_src = ci0 < P.C0 ? 0 : 1;
    if (want_src != D.cur_src) {   // uniform; at most twice per kernel
        D.cur_src = want_src;
        halo_offsets<EPT>(P, tin0, hin0, win0, want_src == 1, wave * 64 + (int)(threadIdx.x & 63), D.goff);
        if (want_src == 0) {
            const long long seq = (long long)P.T * P.C0 * (P.H0 * P.W0p);
            D.rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0 + b * seq), 0,
                                                        (int)(seq * 4), 0x00020000);
            D.src_cstride4 = P.H0 * P.W0p * 4;
            D.src_cbase = 0;
        } else {
            const long long seq = (long long)P.T * P.C1 * (P.Hin * P.Winp);
            D.rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x1 + b * seq), 0,
                                                        (int)(seq * 4), 0x00020000);
            D.src_cstride4 = P.Hin * P.Winp * 4;
            D.src_cbase = P.C0;
        }
    }
    }
    float *wb = hb + CK * chs;
#pragma unroll
    for (int ci = 0; ci < CK; ++ci) {
        const int cg = ci0 + ci;
        const bool cok = cg < P.Cin;
        const int soff = (cg - D.src_cbase) * D.src_cstride4;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int r0 = wave * 64 + 256 * i;              // wave-uniform first element
            if (r0 < P.plane && (ci * EPT + i) % nparts == part) {
                const unsigned vo = cok ? D.goff[i] : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(D.rs_in, (lds_ptr_t)(hb + ci * chs + r0), 4, vo,
                                                         cok ? soff : 0, 0, 0);
            }
        }
    }
    const int wsoff = ci0 * K3 * P.Cout * 4;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int row = wave + 4 * j;
        if (row < W_ROWS && (CK * EPT + j) % nparts == part)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(D.rs_w, (lds_ptr_t)(wb + row * 256), 16, D.woff[j],
                                                     wsoff, 0, 0);
    }
}

#endif  // __HIP_DEVICE_COMPILE__


// LDS-DMA double-buffered direct convolution (see the file header).
//   per chunk of CK input channels:  halo  [CK][chs]            (chs = plane rounded up to 64)
//                                    weights [K3][CK][CO_TILE]  (+ pad to a whole wave row)
//   both staged by buffer_load ... lds (no VGPR round trip) into buffer (chunk & 1) while the
//   MFMAs consume the other buffer; one workgroup barrier per chunk.
// MW: workgroups the register budget must leave room for per CU (2 independent workgroups per CU
// hide each other's barrier / DMA-issue bubbles); IL: deal the DMA issue over the (dt,dh) tap groups
// (measured on MI355X: IL pays only together with MW = 2)
template <int KS, int S, int CO_FR, int PO_FR, int CK, int EPT, int MW, int IL>
__global__ __launch_bounds__(256, MW) void conv3d_kernel(ConvParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = ConvCfg<KS, S, CO_FR, PO_FR, CK, EPT>;
    constexpr int K3 = Cfg::K3, CO_TILE = Cfg::CO_TILE;
    constexpr int PAD = KS / 2;
    constexpr int V4 = CO_TILE / 4;                       // 16-byte units per weight row
    constexpr int W_UNITS = K3 * CK * V4;                 // 16-byte units per weight slab
    constexpr int W_ROWS = (W_UNITS + 63) / 64;           // wave instructions per slab
    constexpr int WPT = (W_ROWS + 3) / 4;                 // per wave
    constexpr int W_FLOATS = W_ROWS * 64 * 4;

    const int chs = (P.plane + 63) & ~63;
    const int buf_floats = CK * chs + W_FLOATS;
    float *smem = reinterpret_cast<float *>(conv_smem);

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (co tile, spatial box)
    // Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one): with xcd_remap the co tiles
    // of one input box run back to back on ONE XCD, so its halo is fetched into one L2 only.
    int bid = blockIdx.x, co_t;
    if (P.xcd_remap) {
        const int xcd = bid & 7, q = bid >> 3;
        co_t = q % P.n_co_tiles;
        bid = (q / P.n_co_tiles) * 8 + xcd;
        if (bid >= P.n_spatial) return;      // padding block (uniform)
    } else {
        co_t = bid % P.n_co_tiles;
        bid /= P.n_co_tiles;
    }
    const int iw = bid % P.nW;            bid /= P.nW;
    const int ih = bid % P.nH;            bid /= P.nH;
    const int it = bid % P.nT;            bid /= P.nT;
    const int b = bid;
    const int co0 = co_t * CO_TILE;
    const int t0 = it * P.TT, h0 = ih * P.TH, w0 = iw * P.TW;
    const int tin0 = t0 - PAD, hin0 = h0 * S - PAD, win0 = w0 * S - PAD;

    // per-lane position fragments
    int hoff[PO_FR];      // halo offset of the position (tap 0,0,0)
    int poff[PO_FR];      // output offset inside the sequence, channel 0; -1 = no such position
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) {
        const int m = (wave * PO_FR + f) * 32 + l32;
        hoff[f] = 0;
        poff[f] = -1;
        if (m < P.n_pos) {
            const int tt = m / (P.TH * P.TW);
            const int rem = m - tt * (P.TH * P.TW);
            const int th = rem / P.TW;
            const int tw = rem - th * P.TW;
            hoff[f] = (tt * P.HH + th * S) * P.HWd + tw * S;
            const int t = t0 + tt, h = h0 + th, w = w0 + tw;
            if (t < P.T && h < P.Hout && w < P.Wout)
                poff[f] = (t * P.Cout) * (P.Hout * P.Woutp) + h * P.Woutp + w;
        }
    }

    // per-lane weight-slab source offsets (bytes, relative to chunk base ci0*K3*Cout*4)
    unsigned woff[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int u = (wave + 4 * j) * 64 + lane;         // 16-byte unit inside the LDS slab
        const int row = u / V4, v = u - row * V4;         // row = tap*CK + ci
        const int tap = row / CK, ci = row - tap * CK;
        const int co = co0 + 4 * v;
        woff[j] = (u < W_UNITS && co < P.Cout) ? 4u * (unsigned)((ci * K3 + tap) * P.Cout + co) : kOOB;
    }
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.wp), 0, P.Cin * K3 * P.Cout * 4, 0x00020000);

    f32x16 acc[CO_FR][PO_FR];
#pragma unroll
    for (int q = 0; q < CO_FR; ++q)
#pragma unroll
        for (int f = 0; f < PO_FR; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][f][r] = 0.0f;

    DmaState<EPT, WPT> D;
    D.cur_src = -1;
    D.rs_in = rs_w;
    D.rs_w = rs_w;
    D.src_cstride4 = 0;
    D.src_cbase = 0;
#pragma unroll
    for (int j = 0; j < WPT; ++j) D.woff[j] = woff[j];
#define ISSUE(ci0_, buf_, part_, nparts_) issue_chunk<KS, CK, EPT, WPT, W_ROWS>(P, D, (ci0_), smem + (buf_) * buf_floats, chs, b, wave, tin0, hin0, win0, (part_), (nparts_))

    ISSUE(0, 0, 0, 1);
    int buf = 0;
    for (int ci0 = 0; ci0 < P.Cin; ci0 += CK, buf ^= 1) {
        // chunk ci0 has landed for every wave, and every wave is done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = ci0 + CK < P.Cin;
        constexpr int NPARTS = IL ? (KS == 1 ? CK / 2 : KS * KS) : 1;
        if (NPARTS == 1 && more) ISSUE(ci0 + CK, buf ^ 1, 0, 1);
        const float *hl = smem + buf * buf_floats;
        const float *wl = hl + CK * chs;
        // K3*CK/2 MFMA k-steps, fully unrolled, with the A/B fragments of step s+1 loaded into a
        // second register set BEFORE the MFMAs of step s: a fragment read issued right behind the
        // last MFMA that uses its register comes back ~50 cycles after the matrix pipe has drained
        // (ISA of the naive loop), costing ~12 % of every k-step.
        constexpr int KH = CK / 2, NSTEP = K3 * KH;
        float a[2][CO_FR], bq[2][PO_FR];
#define V2CE_LOAD_STEP(slot_, s_)                                                            \
        {                                                                                    \
            constexpr int tap_ = (s_) / KH, kk_ = (s_) % KH;                                 \
            constexpr int dt_ = tap_ / (KS * KS), dh_ = (tap_ / KS) % KS, dw_ = tap_ % KS;   \
            const int toff_ = (dt_ * P.HH + dh_) * P.HWd + dw_;                              \
            const float *wrow_ = wl + (tap_ * CK + 2 * kk_ + half) * CO_TILE + l32;          \
            const float *hrow_ = hl + (2 * kk_ + half) * chs + toff_;                        \
            _Pragma("unroll") for (int q = 0; q < CO_FR; ++q) a[slot_][q] = wrow_[q * 32];   \
            _Pragma("unroll") for (int f = 0; f < PO_FR; ++f) bq[slot_][f] = hrow_[hoff[f]]; \
        }
        step_loop<0, NSTEP>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s == 0) V2CE_LOAD_STEP(0, 0)
            if constexpr (KS > 1) {
                if constexpr (s % (KS * KH) == 0) {
                    if (NPARTS > 1 && more) ISSUE(ci0 + CK, buf ^ 1, s / (KS * KH), NPARTS);
                }
            } else {
                if (NPARTS > 1 && more) ISSUE(ci0 + CK, buf ^ 1, s, NPARTS);
            }
            if constexpr (s + 1 < NSTEP) V2CE_LOAD_STEP((s + 1) & 1, s + 1)
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE this step's MFMAs
#pragma unroll
            for (int q = 0; q < CO_FR; ++q)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f)
                    acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][q], bq[s & 1][f],
                                                                    acc[q][f], 0, 0, 0);
        });
#undef V2CE_LOAD_STEP
    }

#undef ISSUE
    // measured: the batched straight-line epilogue wins on the store-heavy 1x1x1 kernels (enc0.down 0.48 ->
    // 0.37 ms), the streaming one on the register-tight 3x3x3 kernels (127 vs 117 TF)
    if constexpr (KS == 1) conv_epilogue<CO_FR, PO_FR, true>(P, acc, poff, co0, half, b, 1.0f);
    else conv_epilogue_stream<CO_FR, PO_FR>(P, acc, poff, co0, half, b);
#endif  // __HIP_DEVICE_COMPILE__ (the host pass only needs the launch stub)
}

// ---------------------------------------------------------------------------------------------
// Split-half, wave-specialised conv.  Every operand is the sum of two fp16 numbers (x * s_x = xh + xl,
// w * s_w = wh + wl, power-of-two pre-scales s_x, s_w) and each 16-channel k-step is three
// v_mfma_f32_32x32x16_f16 (wh.xh + wh.xl + wl.xh; fp16 products are exact in the f32 accumulator):
// 22-bit operands at 5.3x the f32 MFMA rate per k-step; measured against f64 the result is as
// accurate as f32 arithmetic (profiles/r01_d_precision_report.json).  Same tensors, tiling, halo
// gather (zero padding, virtual upsample + concat) and fused epilogue as conv3d_kernel.
// The workgroup is 8 waves = 2 per SIMD with fixed roles.
//   waves 4-7 (producers): gather chunk c+1 of the f32 halo into registers (buffer loads; the
//       hardware range check supplies the zero padding), convert it to the fp16 hi/lo pieces of
//       buffer (c+1)&1, issue the loads of chunk c+2 (they land during the barrier wait).
//   waves 0-3 (consumers): per tap 8 ds_read_b128 + 4 global A-fragment loads + 24 MFMAs out of
//       pieces buffer c&1; B fragments are refilled in place for the next tap as soon as the MFMAs
//       that read them have issued, A fragments are double-buffered one tap ahead.
// One workgroup barrier per 16-channel chunk hands buffer (c+1)&1 over.  The conversion, the DMA
// issue and its branches never sit in the MFMA waves' instruction stream.
// LDS: 2 x 64 B of pieces per halo element (up to 1280 elements: 512-position boxes).
// ---------------------------------------------------------------------------------------------
// FUSE: 0 = plain, 1 = fused 1x1x1 head (pred_epilogue), 2 = fused 1x1x1 shortcut (second accumulator set),
//       3 = folded 1x1x1 tail: a residual block's shortcut as P.tCG more K chunks of the SAME accumulators (conv2 of the
//           block: relu(s2 (W2 * t + Wd' * x) + shift), Wd' = Wd sd / s2 folded on the host; no shortcut tensor at all)
//       4 = 3 + 1 (round 6): the folded tail AND the fused head -- conv2 of the last decoder block with its shortcut split by source:
//           the skip channels ride as the tail, the upsampled channels' share arrives as a low-resolution residual (P.res_up)
// RES: residual known at compile time (0 = none, 1 = present) or checked at run time (2), see conv_epilogue
// OPT bit 0, "PEPI" (round 6): the tile's epilogue is SHARED between the roles.  The 32-channel tile with the fused head has two chunks
//       of 324 MFMAs per wave and an epilogue (residual, scale / shift, ReLU, the head's MFMAs, 20 planar channels of stores) that
//       took 17-19 k of its 46 k cycles with the producers idle at barriers for half of the launch.  Now the consumers leave the
//       accumulators of V2CE_PEPI_KEEP .. PO_FR - 1 of a wave's position fragments in the pieces buffer the tile's last chunk has just
//       freed (a barrier in front of the dump: every consumer wave must be done with those pieces; one behind it: the producers may
//       start) and run the epilogue of the others themselves, while producer wave w + 4 -- the lane layout of consumer wave w --
//       takes the dumped ones.  Both use ONE lean form (epi_fragment): tables in LDS instead of global loads, one LDS round trip
//       per fragment, a power-of-two pre-scale per position instead of a wave maximum, the residual requested ahead of the
//       hand-over; nothing in it waits for a store.  dec3.conv2 + head: 1.00-1.04 -> 0.93-0.97 ms (with what did not
//       work: all four fragments on the producers, B fragments reused across the time taps).
template <int KS, int S, int WCO, int CO_FR, int PO_FR, int NA, int FUSE = 0, int RES = 2, int OPT = 0>
__global__ __launch_bounds__(512, 1) void conv3d_f16x2_ws_kernel(ConvParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PEPI = OPT & 1;                              // (OPT: a bit set; bit 0 = the producer-side epilogue)
    // bit 1, "G4": the producers gather with FOUR LANES PER ELEMENT -- load instruction k of a quad of lanes fetches the four 16-byte
    // quarters of the element lane k of the quad owns (its offset arrives by a quad broadcast), so an instruction touches 16 whole
    // 64-byte lines instead of a quarter of each of 64: the texture path the consumers' weight loads share is what the gather
    // occupies.  A lane then holds four channels of four elements instead of sixteen of one: 8-byte pieces.
    constexpr bool G4 = (OPT & 2) != 0;
    static_assert(!G4 || (KS == 3 && FUSE != 3 && FUSE != 4), "four-lane gather: plain 3x3x3 chunks (no folded tail)");
    // KS = 1: the "halo box" is the output box itself (positions gathered with stride S), one tap
    constexpr int K3 = KS * KS * KS, CK = 16, EPT = 5, PAD = KS / 2, GS = KS == 1 ? S : 1;
    constexpr int CO_TILE = WCO * CO_FR * 32;
    static_assert(!PEPI || ((FUSE == 1 || FUSE == 2) && CO_FR == 1),
                  "shared epilogue: one 32-channel fragment row per wave, with the fused head or the fused shortcut");
    // PEPI: the consumers keep the first kKeepFr position fragments of a wave for their own epilogue and hand the others over -- the
    // two epilogues run side by side between the tile's last chunk and the next tile's first
    constexpr int kKeepFr = PEPI ? (FUSE == 1 ? V2CE_PEPI_KEEP : PO_FR / 2) : PO_FR, kDumpFr = PO_FR - kKeepFr;
    constexpr int kDumpSets = FUSE == 2 ? 2 : 1;             // (the fused shortcut's second accumulator set travels too)
    constexpr int kDumpWave = kDumpFr * kDumpSets * 4 * 64;  // f32x4 per consumer wave in the accumulator dump (PEPI)
    int chs_ = (FUSE == 3 || FUSE == 4) ? P.tCHS : (P.plane + 63) & ~63;
    if (PEPI && chs_ < kDumpWave) chs_ = kDumpWave;          // a pieces buffer (4 chs x 16 B) holds the four waves' dumps
    const int chs = chs_;
    f16x8 *pieces = reinterpret_cast<f16x8 *>(conv_smem);                      // [2][4][chs] x 16 B
    [[maybe_unused]] unsigned *pepi_flag = reinterpret_cast<unsigned *>(conv_smem + (size_t)chs * 128);   // PEPI: producer waves done reading a dump
    if constexpr (PEPI != 0) {                               // (all of it visible behind the first chunk barrier, long before its first use)
        // LDS behind the pieces, 16 B in: scale[32] shift[32] of the conv, bias[32] of the head | the head's A fragments (4 KB)
        // -- or (fused shortcut) scale[Cout] | shift[Cout] | sc_scale[Cout] | sc_shift[Cout]
        if constexpr (FUSE == 2) {
            float *tab = reinterpret_cast<float *>(pepi_flag + 4);
            for (int i = threadIdx.x; i < P.Cout; i += 512) {
                tab[i] = P.scale[i];
                tab[P.Cout + i] = P.shift[i];
                tab[2 * P.Cout + i] = P.sc_scale[i];
                tab[3 * P.Cout + i] = P.sc_shift[i];
            }
        } else {
        if (threadIdx.x < 32) {
            float *tab = reinterpret_cast<float *>(pepi_flag + 4);
            tab[threadIdx.x] = P.scale[threadIdx.x];
            tab[32 + threadIdx.x] = P.shift[threadIdx.x];
            tab[64 + threadIdx.x] = P.pred_b[threadIdx.x];
        }
        if (threadIdx.x < 256) reinterpret_cast<f16x8 *>(pepi_flag + 100)[threadIdx.x] = reinterpret_cast<const f16x8 *>(P.pred_w)[threadIdx.x];
        }
    }

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CG = P.Cin / CK;
    const long long wplane = (long long)K3 * CG * P.Cout * 16;               // halves per plane

    // Persistent workgroups: virtual block vb = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple
    // of 8, so a workgroup's tiles keep their XCD / L2).  Both roles walk the same tile sequence and
    // meet at one barrier per 16-channel chunk; the producers run one chunk ahead ACROSS tile
    // boundaries, so a tile's first chunk is gathered and converted while the consumers are still in
    // the previous tile's last chunk and epilogue.
    struct TileId { int b, co_t, t0, h0, w0; };
    // XCD x = vb & 7 owns the contiguous range [x * per_xcd, (x + 1) * per_xcd) of spatial boxes and walks
    // it in order, all channel tiles of a box back to back: the 32 workgroups of an XCD work on
    // neighbouring boxes at the same time, so the halo overlap (2.4x the tensor for 512-position
    // boxes) is served by that XCD's L2 instead of being fetched once per XCD.
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
    auto next_tile = [&](int &vb, TileId &T) -> bool {     // first valid virtual block at or after vb
        for (; vb < P.total_blocks; vb += (int)gridDim.x)
            if (decode(vb, T)) return true;
        return false;
    };

    const float w_scale = reinterpret_cast<const float *>(P.wq + 2 * wplane)[1];
    // max |x| of batch element b as its producers recorded it (one slot per element with amax_bs > 0, else one per
    // tensor); without tracking the fixed pre-scale kActScale applies (|x| < 4094 required)
    auto amax_of = [&](int b) -> float {
        if (!P.x0_absmax) return 4094.0f;
        float am = P.x0_absmax[b * P.amax_bs];
        if (P.x1_absmax) am = fmaxf(am, P.x1_absmax[b * P.amax_bs]);
        return am;
    };
    auto scale_of = [&](int b) -> float { return P.x0_absmax ? pow2_prescale(amax_of(b)) : kActScale; };
    constexpr bool TAIL = FUSE == 3 || FUSE == 4;
    static_assert(!TAIL || KS == 3, "the folded tail rides behind a 3x3x3 conv");
    const int CGT = TAIL ? CG + P.tSC : CG;                  // barriers (chunks / tail super-chunks) per tile
    auto tamax_of = [&](int b) -> float {                    // the same for the tail's input
        if (!P.tx0_absmax) return 4094.0f;
        float am = P.tx0_absmax[b * P.amax_bs];
        if (P.tx1_absmax) am = fmaxf(am, P.tx1_absmax[b * P.amax_bs]);
        return am;
    };
    auto tscale_of = [&](int b) -> float { return P.tx0_absmax ? pow2_prescale(tamax_of(b)) : kActScale; };
    // Range guard (y_absmax[1]): a bound on what the one-scale-per-tensor split can cost this launch's
    // outputs.  An operand whose scaled magnitude is below 2^-3 has an fp16-SUBNORMAL lo half: hi + lo
    // then misses it by up to 2^-25 (scaled), instead of by 2^-22 relative.  Worst case over a K-term
    // dot product, times the folded BatchNorm scale:
    //   E = max|scale| * K * 2^-25 * (max|w| / x_scale + max|x| / w_scale)
    // (first term: activations flushed, second: weights flushed).  The host compares E with its limit
    // after the run (v2ce_3d.V2ce3d.range_guard_value) and repeats the clip on the exact-f32 kernels.
    // With per-element slots every batch element reports its own bound (slot b, second float).
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
            if (TAIL) {                                     // both parts land in the same output: the bounds add
                const float *tl = reinterpret_cast<const float *>(P.sc_w + 2 * (long long)P.tCG * P.Cout * 16);
                const float tam = tamax_of(b), txs = tscale_of(b);
                E += sm * (float)(P.tCG * 16) * 0x1p-25f * (tl[0] / txs + tam / tl[1]);
                // the accumulators are rescaled by (txs tl[1]) / (xs tail[1]) between the two parts: a power of two, exact
                // unless it is so extreme that they leave the f32 range -- then report "unbounded"
                const float rho = (txs * tl[1]) / (xs * tail[1]);
                if (!(rho > 0x1p-40f && rho < 0x1p40f)) E = __builtin_inff();
            }
            P.guard[b * P.amax_bs] = E;
        }
    }

    int vb = blockIdx.x;
    TileId T;
    if (!next_tile(vb, T)) return;
    int gc = 0;                                             // chunks handled so far: pieces buffer gc & 1

    // ---- PEPI: the lean epilogue of ONE position fragment, used by both roles (consumer wave w and producer wave w + 4 share the
    // lane layout of wave w's accumulator tiles).  conv_epilogue's arithmetic (scale, shift, residual, activation, range tracking;
    // conv3d_dev.h) on the sixteen registers of the fragment, then pred_epilogue's: split into fp16 halves at a power of two per
    // POSITION, six MFMAs, bias, ReLU, planar stores.  Scale / shift / bias and the head's A fragments come from LDS -- a global load
    // here would sit behind earlier stores in the in-order vector-memory counter -- in one round trip; nothing in it waits for a store.
    typedef float f32x4d __attribute__((ext_vector_type(4)));
    [[maybe_unused]] int epk[PO_FR];                        // this lane's position in each fragment, packed tt << 20 | th << 10 | tw (-1: none)
    if constexpr (PEPI != 0) {
#pragma unroll
        for (int f = 0; f < PO_FR; ++f) {
            const int m = (((wave & 3) / WCO) * PO_FR + f) * 32 + l32;
            epk[f] = -1;
            if (m < P.n_pos) {
                const int tt = m / (P.TH * P.TW);
                const int rem = m - tt * (P.TH * P.TW);
                const int th = rem / P.TW;
                epk[f] = (tt << 20) | (th << 10) | (rem - th * P.TW);
            }
        }
    }
    // this lane's offsets in fragment f of tile E: vo = bytes of its 16-byte channel quad inside channel group 0 of the conv's output /
    // residual, vp = bytes of its head output in channel 4 half (kOOB: outside the tensor)
    [[maybe_unused]] auto epi_offsets = [&](const TileId &E, int f, unsigned &vo, unsigned &vp) __attribute__((always_inline)) {
        int pk = epk[f];
        asm volatile("" : "+v"(pk));                        // (unpacked at the use, not kept unpacked for the life of the kernel)
        vo = vp = kOOB;
        if (pk >= 0) {
            const int tt = pk >> 20, th = (pk >> 10) & 1023, tw = pk & 1023;
            const int t = E.t0 + tt, h = E.h0 + th, w = E.w0 + tw;
            if (t < P.T && h < P.Hout && w < P.Wout) {
                vo = (unsigned)(4 * ((t * P.Cout) * (P.Hout * P.Woutp)) + 64 * (h * P.Woutp + w) + 16 * half);
                vp = (unsigned)((((t * P.pred_cout) * (P.Hout * P.Wout)) + h * P.Wout + w) * 4 + 4 * half * (P.Hout * P.Wout * 4));
            }
        }
    };
    // the residual of fragment f of tile E: channels 8 r4 + 4 half + {0..3} of register quad r4 = group r4 >> 1, bytes 32 (r4 & 1) + 16 half
    [[maybe_unused]] auto epi_load_res = [&](const TileId &E, int f, f32x4d (&rv)[4]) __attribute__((always_inline)) {
        if constexpr (PEPI != 0 && RES == 1) {
            const long long seq = (long long)P.T * P.Cout * (P.Hout * P.Woutp);
            const int gstride = P.Hout * P.Woutp * 64;
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.res + E.b * seq), 0, (int)(seq * 4), 0x00020000);
            unsigned vo, vp;
            epi_offsets(E, f, vo, vp);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                rv[r4] = __builtin_bit_cast(f32x4d, __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo, (r4 >> 1) * gstride + 32 * (r4 & 1), 0));
        }
    };
    // a = the fragment's sixteen accumulators as four register quads, rv = its residual; `between` runs after the last use of rv and
    // before the fragment's stores (the place to request the next residual)
    // inv = 1 / (the tile's activation pre-scale x the weights' pre-scale), pw_scale = the head weights' pre-scale: fetched once per
    // tile by the caller (scalar loads whose round trip would otherwise open every fragment)
    [[maybe_unused]] auto epi_fragment = [&](const TileId &E, int f, const f32x4d (&a)[4], const f32x4d (&rv)[4], unsigned &ymax,
                                             float inv, float pw_scale, auto &&between) __attribute__((always_inline)) {
        if constexpr (PEPI != 0 && FUSE == 1) {
            typedef unsigned u32x4d __attribute__((ext_vector_type(4)));
            const float *tab = reinterpret_cast<const float *>(pepi_flag + 4);                      // scale[32] | shift[32] | bias[32]
            const f16x8 *atab = reinterpret_cast<const f16x8 *>(pepi_flag + 100);                   // the head's A fragments (4 KB)
            const long long seq = (long long)P.T * P.Cout * (P.Hout * P.Woutp);
            const int gstride = P.Hout * P.Woutp * 64;            // bytes between 16-channel groups of a time step
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y ? P.y + E.b * seq : const_cast<float *>(P.scale), 0,
                                                                                  P.y ? (int)(seq * 4) : 0, 0x00020000);
            const long long pseq = (long long)P.T * P.pred_cout * (P.Hout * P.Wout);
            const int pstride4 = P.Hout * P.Wout * 4;
            const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(P.pred_y + E.b * pseq, 0, (int)(pseq * 4), 0x00020000);
            const float slope = act_slope(P.act);
            unsigned vo, vp;
            epi_offsets(E, f, vo, vp);
            const unsigned vmask = vo != kOOB ? 0x7fffffffu : 0u;
            // every LDS read of the fragment in front of ONE wait (with a wait per register quad the epilogue was 100 LDS round trips
            // in a row behind the consumers' B-fragment reads: ~1 k cycles each)
            f32x4d scq[4], shq[4], bq[4];
            f16x8 ahp[2], alp[2];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                scq[r4] = *reinterpret_cast<const f32x4d *>(tab + 8 * r4 + 4 * half);
                shq[r4] = *reinterpret_cast<const f32x4d *>(tab + 32 + 8 * r4 + 4 * half);
                bq[r4] = *reinterpret_cast<const f32x4d *>(tab + 64 + 8 * r4 + 4 * half);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                ahp[k] = atab[((k * 2 + 0) * 32 + l32) * 2 + half];
                alp[k] = atab[((k * 2 + 1) * 32 + l32) * 2 + half];
            }
            float v[16];
            float mx = 0.0f;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4d y4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = a[r4][k] * (scq[r4][k] * inv) + shq[r4][k];
                    if constexpr (RES == 1) t += rv[r4][k];
                    t = apply_act(t, slope);
                    y4[k] = t;
                    const unsigned av = __builtin_bit_cast(unsigned, t) & vmask;
                    ymax = av > ymax ? av : ymax;
                    const float kept = __builtin_bit_cast(float, av == 0u ? 0u : __builtin_bit_cast(unsigned, t));
                    v[4 * r4 + k] = kept;
                    mx = fmaxf(mx, fabsf(kept));
                }
                if (P.y) {                                      // uniform: the decoder output itself is rarely wanted
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4d, y4), rs_y, vo, (r4 >> 1) * gstride + 32 * (r4 & 1), 0);
                    asm volatile("s_nop 1" : "+v"(y4));        // (16-byte store data hazard: conv_epilogue)
                }
            }
            between();
            // pre-scale of the split: a power of two per POSITION (a column of the B fragment, whose sixteen channels of a k-step sit in
            // this lane and in lane ^ 32; the same column of the output lands in this lane), so one exchange between the wave's halves
            // replaces the six-step wave maximum of pred_epilogue
            {
                // (inline asm: through __builtin_amdgcn_permlane32_swap this build's compiler takes BOTH results from the first register
                // and the lower half of the wave never sees the upper half's maximum -- found by the golden test)
                float ma = mx, mb = mx;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ma), "+v"(mb));
                mx = fmaxf(mx, fmaxf(ma, mb));
            }
            const float v_scale = pow2_prescale(mx);
            f32x16 out;
#pragma unroll
            for (int r = 0; r < 16; ++r) out[r] = 0.0f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
                u32x4c ph, pl;                                  // hi = f16(x s), lo = f16(x s - hi): the producers' four mixed-precision FMAs per pair
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    unsigned h, l;
                    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
                        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(h), "=&v"(l) : "v"(v[8 * k + 2 * c2]), "v"(v[8 * k + 2 * c2 + 1]), "v"(v_scale));
                    ph[c2] = h;
                    pl[c2] = l;
                }
                const f16x8 bhp = __builtin_bit_cast(f16x8, ph), blp = __builtin_bit_cast(f16x8, pl);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahp[k], bhp, out, 0, 0, 0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahp[k], blp, out, 0, 0, 0);
                out = __builtin_amdgcn_mfma_f32_32x32x16_f16(alp[k], bhp, out, 0, 0, 0);
            }
            const float invp = 1.0f / (v_scale * pw_scale);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (8 * r4 >= P.pred_cout) continue;            // uniform: rows of the 32 that the head does not have (20 channels: a quarter of the stores)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int oq = k + 8 * r4;                  // output channel minus 4 * half
                    const bool ook = oq + 4 * half < P.pred_cout;
                    float y = out[4 * r4 + k] * invp + bq[r4][k];
                    y = fmaxf(y, 0.0f) + 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rs_p, ook ? vp : kOOB, oq * pstride4, 0);
                }
            }
        }
    };

    // the same for the conv with the fused shortcut (FUSE 2): y = act(scale a + shift), sc_y = sc_scale ad + sc_shift, both in the
    // channels-last-16 layout (one 16-byte store per register quad and output)
    [[maybe_unused]] auto epi_fragment_sc = [&](const TileId &E, int f, const f32x4d (&a)[4], const f32x4d (&ad)[4], unsigned &ymax) __attribute__((always_inline)) {
        if constexpr (PEPI != 0 && FUSE == 2) {
            typedef unsigned u32x4d __attribute__((ext_vector_type(4)));
            const int co0 = E.co_t * CO_TILE + ((wave & 3) % WCO) * 32;
            const long long wpl_d = (long long)CG * P.Cout * 16;
            const float xs = scale_of(E.b);
            const float inv = 1.0f / (xs * w_scale), invd = 1.0f / (xs * reinterpret_cast<const float *>(P.sc_w + 2 * wpl_d)[1]);
            const float *tab = reinterpret_cast<const float *>(pepi_flag + 4);
            const long long seq = (long long)P.T * P.Cout * (P.Hout * P.Woutp);
            const int gstride = P.Hout * P.Woutp * 64;            // bytes between 16-channel groups of a time step
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y + E.b * seq, 0, (int)(seq * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(P.sc_y + E.b * seq, 0, (int)(seq * 4), 0x00020000);
            const float slope = act_slope(P.act);
            unsigned vo, vp;
            epi_offsets(E, f, vo, vp);
            const unsigned vmask = vo != kOOB ? 0x7fffffffu : 0u;
            f32x4d scq[4], shq[4], dq[4], dsh[4];                 // one LDS round trip
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int cb = co0 + 8 * r4 + 4 * half;
                scq[r4] = *reinterpret_cast<const f32x4d *>(tab + cb);
                shq[r4] = *reinterpret_cast<const f32x4d *>(tab + P.Cout + cb);
                dq[r4] = *reinterpret_cast<const f32x4d *>(tab + 2 * P.Cout + cb);
                dsh[r4] = *reinterpret_cast<const f32x4d *>(tab + 3 * P.Cout + cb);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const bool cok = co0 + 8 * r4 < P.Cout;            // uniform (Cout need not fill the last channel tile)
                const int so = (co0 / 16 + (r4 >> 1)) * gstride + 32 * (r4 & 1);
                f32x4d y4, d4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = a[r4][k] * (scq[r4][k] * inv) + shq[r4][k];
                    t = apply_act(t, slope);
                    y4[k] = t;
                    const unsigned av = __builtin_bit_cast(unsigned, t) & (cok ? vmask : 0u);
                    ymax = av > ymax ? av : ymax;
                    d4[k] = ad[r4][k] * (dq[r4][k] * invd) + dsh[r4][k];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4d, y4), rs_y, cok ? vo : kOOB, so, 0);
                asm volatile("s_nop 1" : "+v"(y4));            // (16-byte store data hazard: conv_epilogue)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4d, d4), rs_s, cok ? vo : kOOB, so, 0);
                asm volatile("s_nop 1" : "+v"(d4));
            }
        }
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        // a chunk travels global -> registers (gather with the halo offsets; out-of-range offsets
        // read as the zero padding) -> fp16 hi/lo pieces in LDS.  Two register buffers: the loads of
        // chunk c+2 are issued right after chunk c has been converted, so every gather has a whole
        // consumer chunk to land in (measured: with one buffer the producers spent 18-28 k cycles per
        // chunk, most of it waiting for their own loads, and set the pace of every layer).  The load
        // cursor (tile, chunk) therefore runs two chunks ahead of the conversion, across tiles.
        const int ptid = tid - 256;
        float x_scale = scale_of(T.b);                      // pre-scale of the tile being converted
        float t_scale = TAIL ? tscale_of(T.b) : 1.0f;       // ... and of its tail chunks
        unsigned goff[EPT];
        unsigned gsel = 0;                                   // tail: channel group of every element slot (tail_offsets)
        float R0[CK][EPT], R1[CK][EPT];
        __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0), 0, 0, 0x00020000);
        int cur_src = -1, src_cstride4 = 0, src_cbase = 0;
        auto load_chunk = [&](const TileId &L, int cidx, float (&R)[CK][EPT]) {
            const bool tail = TAIL && cidx >= CG;            // uniform
            // tail super-chunk k = cidx - CG: tx0's groups first (tSC0 super-chunks of up to tTCH groups), then tx1's
            const int tk = cidx - CG;
            const int tg0 = tail ? (tk < P.tSC0 ? tk : tk - P.tSC0) * P.tTCH : 0;           // first group inside its source
            const int tng = tail ? min(P.tTCH, (tk < P.tSC0 ? P.tC0 : P.tC1) / 16 - tg0) : 0;   // groups of this super-chunk
            const int ci0 = tail ? tg0 * CK + (tk < P.tSC0 ? 0 : P.tC0) : cidx * CK;
            const int want_src = tail ? (tk < P.tSC0 ? 2 : 3) : (ci0 < P.C0 ? 0 : 1);
            if (want_src != cur_src) {   // uniform; at most four times per tile
                cur_src = want_src;
                if (tail) {
                    tail_offsets<EPT>(P, L.t0, L.h0, L.w0, want_src == 3, ptid, goff, gsel);
                    if (want_src == 2) {
                        const long long seq = (long long)P.T * P.tC0 * (P.tH0 * P.tW0p);
                        rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.tx0 + L.b * seq), 0, (int)(seq * 4), 0x00020000);
                        src_cstride4 = P.tH0 * P.tW0p * 4;
                        src_cbase = 0;
                    } else {
                        const long long seq = (long long)P.T * P.tC1 * (P.tHin * P.tWinp);
                        rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.tx1 + L.b * seq), 0, (int)(seq * 4), 0x00020000);
                        src_cstride4 = P.tHin * P.tWinp * 4;
                        src_cbase = P.tC0;
                    }
                } else {
                // (PEPI: the decomposition of the element index -- the same for every tile -- is NOT to be hoisted out of the tile loop:
                // fifteen registers for the life of the kernel that the producer-side epilogue cannot afford; ~400 VALU per tile instead)
                int ptid_l = ptid;
                if constexpr (PEPI != 0) asm volatile("" : "+v"(ptid_l));
                halo_offsets<EPT, true>(P, L.t0 - PAD, L.h0 * S - PAD, L.w0 * S - PAD, want_src == 1, ptid_l, goff, GS);
                if (want_src == 0) {
                    const long long seq = (long long)P.T * P.C0 * (P.H0 * P.W0p);
                    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0 + L.b * seq), 0, (int)(seq * 4), 0x00020000);
                    src_cstride4 = P.H0 * P.W0p * 4;
                    src_cbase = 0;
                } else {
                    const long long seq = (long long)P.T * P.C1 * (P.Hin * P.Winp);
                    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x1 + L.b * seq), 0, (int)(seq * 4), 0x00020000);
                    src_cstride4 = P.Hin * P.Winp * 4;
                    src_cbase = P.C0;
                }
                }
            }
            const int lim = tail ? tng * P.tNPP : P.plane;            // element slots of this chunk
#ifdef V2CE_ABLATE_UP
            const int ne_abl = (!tail && want_src == 0 && P.hmap) ? V2CE_ABLATE_UP : EPT;     // timing ablation: an upsampled source gathered once per SOURCE element
#endif
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
#ifdef V2CE_ABLATE_UP
                if (i >= ne_abl) continue;
#endif
                if ((wave - 4) * 64 + 256 * i < lim) {                // wave-uniform
                    // (tail: the slot's group must exist in this super-chunk -- the last one of a source may be short)
                    const unsigned vo = (tail && (int)((gsel >> (4 * i)) & 15u) >= tng) ? kOOB : goff[i];
                    // the chunk is one 16-channel group: the element's 64 bytes in four 16-byte loads (one cache
                    // line per element; the planar layout needs 16 loads from 16 lines)
                    typedef float f32x4g __attribute__((ext_vector_type(4)));
                    if constexpr (G4) {
                        // R[4 k + c][i] = channel 4 (lane & 3) + c of the element that lane k of this lane's quad owns
                        const int so = ((ci0 - src_cbase) / 16) * (src_cstride4 * 16);
                        step_loop<0, 4>([&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            const unsigned vk = (unsigned)__builtin_amdgcn_mov_dpp((int)vo, k | (k << 2) | (k << 4) | (k << 6), 0xf, 0xf, true);
                            const f32x4g v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vk + 16u * (unsigned)(ptid & 3), so, 0));
                            R[4 * k][i] = v[0]; R[4 * k + 1][i] = v[1]; R[4 * k + 2][i] = v[2]; R[4 * k + 3][i] = v[3];
                        });
                    } else
#pragma unroll
                    for (int k4 = 0; k4 < CK / 4; ++k4) {
                        const f32x4g v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(
                            rs_in, vo, ((ci0 - src_cbase) / 16) * (src_cstride4 * 16) + 16 * k4, 0));
                        R[4 * k4][i] = v[0]; R[4 * k4 + 1][i] = v[1]; R[4 * k4 + 2][i] = v[2]; R[4 * k4 + 3][i] = v[3];
                    }
                }
            }
        };
        // load cursor
        int vbL = vb, cgL = 0;
        TileId TL = T;
        bool moreL = true;
        auto load_next = [&](float (&R)[CK][EPT]) {
            if (!moreL) return;
            load_chunk(TL, cgL, R);
            if (++cgL == CGT) {
                cgL = 0;
                vbL += (int)gridDim.x;
                moreL = next_tile(vbL, TL);
                cur_src = -1;
            }
        };
        int cgC = 0;                                         // conversion cursor: chunk inside the tile
        auto convert = [&](const float (&R)[CK][EPT]) {
            f16x8 *qb = pieces + (gc & 1) * 4 * chs;
            const bool tail = TAIL && cgC >= CG;             // uniform
            const float c_scale = tail ? t_scale : x_scale;
            const int lim = tail ? P.tTCH * P.tNPP : P.plane;       // (slots of absent groups hold zeros: harmless)
#ifdef V2CE_ABLATE_UP
            const int ne_abl = (!tail && cgC * CK < P.C0 && P.hmap) ? V2CE_ABLATE_UP : EPT;
#endif
            int rb = ptid;
            if constexpr (PEPI != 0) asm volatile("" : "+v"(rb));   // (the piece addresses are formed here, not kept for the life of the kernel)
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
#ifdef V2CE_ABLATE_UP
                if (i >= ne_abl) continue;
#endif
                const int r = rb + 256 * i;
                if constexpr (G4) {
                    if ((wave - 4) * 64 + 256 * i < lim) {            // wave-uniform
                        // this lane's four channels 4 j .. 4 j + 3 (j = lane & 3) of the quad's four elements: 8 bytes of the hi piece and
                        // 8 of the lo piece of 8-channel group j >> 1, second half of the piece for odd j
                        typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
                        const int j = rb & 3;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            u32x2c ph, pl;
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2) {
                                const float xa = R[4 * k + 2 * c2][i], xb = R[4 * k + 2 * c2 + 1][i];
                                unsigned h, l;
                                asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                                    "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                                    "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
                                    "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                    : "=&v"(h), "=&v"(l) : "v"(xa), "v"(xb), "v"(c_scale));
                                ph[c2] = h;
                                pl[c2] = l;
                            }
                            const int rk = (r & ~3) + k;
                            reinterpret_cast<u32x2c *>(qb + (j >> 1) * chs + rk)[j & 1] = ph;
                            reinterpret_cast<u32x2c *>(qb + (2 + (j >> 1)) * chs + rk)[j & 1] = pl;
                        }
                    }
                } else
                if ((wave - 4) * 64 + 256 * i < lim) {                // wave-uniform (lanes past the box write padding)
#pragma unroll
                    for (int hg = 0; hg < 2; ++hg) {
                        // hi = f16(x s), lo = f16(x s - hi): four mixed-precision FMAs per pair of values (x s is exact,
                        // s a power of two, and x s - hi is exact in f32, so this is bit-identical to multiply, convert,
                        // convert back, subtract, convert -- which the compiler emitted for most pairs: 6 instructions)
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
                                : "=&v"(h), "=&v"(l) : "v"(xa), "v"(xb), "v"(c_scale));
                            ph[c2] = h;
                            pl[c2] = l;
                        }
                        qb[hg * chs + r] = __builtin_bit_cast(f16x8, ph);
                        qb[(2 + hg) * chs + r] = __builtin_bit_cast(f16x8, pl);
                    }
                }
            }
        };
        // the conversion cursor counts the chunks of this workgroup's tiles
        bool moreC = true;
        [[maybe_unused]] TileId Tprev = T;                  // PEPI: the tile whose chunks have all been staged
        auto advance = [&]() {
            ++gc;
            if (++cgC == CGT) {
                cgC = 0;
                if constexpr (PEPI != 0) Tprev = T;
                vb += (int)gridDim.x;
                moreC = next_tile(vb, T);
                if (moreC) {
                    x_scale = scale_of(T.b);
                    if (TAIL) t_scale = tscale_of(T.b);
                }
            }
        };
        [[maybe_unused]] unsigned long long t_all = TICK(), t_bar = 0, t_epi = 0, t_cvt = 0, t_ld = 0;   // (-DV2CE_STAMP: where the producers' time goes)
        [[maybe_unused]] f32x4d epi_rv[4];                             // residual of ONE fragment, rolling (epi_load_res)
        [[maybe_unused]] auto epi_prefetch = [&](const TileId &E) __attribute__((always_inline)) { epi_load_res(E, kKeepFr, epi_rv); };
        // PEPI: the epilogue of the fragments kKeepFr .. PO_FR - 1 of tile E, whose consumers left those accumulators in the pieces buffer
        // of its last chunk (g_last): this wave takes the dump of consumer wave (wave - 4) -- the same lane layout.  The residual of
        // fragment f + 1 is requested as soon as fragment f's values have been consumed and BEFORE fragment f's stores.
        [[maybe_unused]] auto tile_epilogue = [&](const TileId &E, int g_last) __attribute__((always_inline)) {
            if constexpr (PEPI != 0) {
                const f32x4d *dump = reinterpret_cast<const f32x4d *>(pieces + (g_last & 1) * 4 * chs) + (wave - 4) * kDumpWave;
                unsigned ymax = 0u;
                [[maybe_unused]] float inv = 1.0f, pw_scale = 1.0f;
                if constexpr (FUSE == 1) {
                    inv = 1.0f / (scale_of(E.b) * w_scale);
                    pw_scale = reinterpret_cast<const float *>(P.pred_w + 2048)[0];
                }
                step_loop<kKeepFr, PO_FR>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;   // the wave's fragment; its accumulators are fragment f - kKeepFr of the dump
                    f32x4d a[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) a[r4] = dump[(((f - kKeepFr) * kDumpSets) * 4 + r4) * 64 + lane];
                    if constexpr (FUSE == 2) {
                        f32x4d ad[4];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) ad[r4] = dump[(((f - kKeepFr) * kDumpSets + 1) * 4 + r4) * 64 + lane];
                        epi_fragment_sc(E, f, a, ad, ymax);
                    } else {
                        epi_fragment(E, f, a, epi_rv, ymax, inv, pw_scale, [&]() __attribute__((always_inline)) {
                            if constexpr (f + 1 < PO_FR) epi_load_res(E, f + 1, epi_rv);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (P.y_absmax) absmax_commit(__builtin_bit_cast(float, ymax), P.y_absmax + E.b * P.amax_bs);
            }
        };
        // one chunk: convert it, issue the loads of the chunk two ahead, meet the consumers.  PEPI: a chunk that opens a tile (not the
        // first) is preceded by the barrier behind which the consumers dump the previous tile's accumulators, and followed by that
        // tile's epilogue.
        auto chunk_step = [&](float (&R)[CK][EPT], int stamp_a, int stamp_b) __attribute__((always_inline)) {
            { [[maybe_unused]] const unsigned long long tc = TICK();
            convert(R);
            ACC_T(t_cvt, tc); }
            if (gc == 0 && stamp_a >= 0) STAMP(1, 2);
            if constexpr (PEPI != 0) {
                // R is reloaded BEHIND the rendezvous (its loads still have a whole consumer chunk to land in) and declared dead in
                // front of it: the epilogue has its eighty registers.  (With both chunks in flight next to the epilogue 236 registers
                // went to scratch and the launch took 3.7 ms instead of 1.0; with the reload on two paths the register allocator
                // spilled the chunks themselves.)
                const bool opens = cgC == 0 && gc != 0;               // uniform
                if (opens) {
                    epi_prefetch(Tprev);
                    { [[maybe_unused]] const unsigned long long tb = TICK();
                    lds_barrier();                                    // (the consumers' pieces of the previous tile are dead)
                    lds_barrier();                                    // their accumulators are in that buffer
                    ACC_T(t_bar, tb); }
                    { [[maybe_unused]] const unsigned long long te = TICK();
                    tile_epilogue(Tprev, gc - 1);                      // (beside the consumers' own half of it)
                    ACC_T(t_epi, te); }
                }
                { [[maybe_unused]] const unsigned long long tb = TICK();
                __syncthreads();                                      // barrier gc: pieces[gc & 1] ready; every producer wave is done with the dump
                ACC_T(t_bar, tb); }
#pragma unroll
                for (int c = 0; c < CK; ++c)
#pragma unroll
                    for (int i = 0; i < EPT; ++i) R[c][i] = 0.0f;
                { [[maybe_unused]] const unsigned long long tl = TICK();
                load_next(R);
                ACC_T(t_ld, tl); }
            } else {
                load_next(R);
                if (gc == 0 && stamp_a >= 0) STAMP(1, 3);
                __syncthreads();                                      // barrier gc: pieces[gc & 1] ready
            }
            if (gc == 0 && stamp_a >= 0) STAMP(1, 4);
            if (gc == 1 && stamp_b >= 0) STAMP(1, 5);
            advance();
        };
        STAMP(1, 0);
        load_next(R0);
        load_next(R1);
        STAMP(1, 1);
        while (moreC) {
            chunk_step(R0, 0, -1);
            if (!moreC) break;
            chunk_step(R1, -1, 0);
        }
        if constexpr (PEPI != 0) {                                    // the last tile's accumulators
            epi_prefetch(Tprev);
            lds_barrier();
            lds_barrier();
            tile_epilogue(Tprev, gc - 1);
#ifdef V2CE_STAMP
            if (lane == 0 && wave == 4) {
                unsigned long long *o = P.stamps + (long long)gridDim.x * 16 + ((long long)blockIdx.x * 2 + 1) * 8;
                o[0] = TICK() - t_all; o[1] = t_bar; o[2] = t_epi; o[3] = t_cvt; o[4] = t_ld;
            }
#endif
        }
        STAMP(1, 6);
        return;
    }

    // ---------------------------------------------------------------------- consumers
    STAMP(0, 0);
    __builtin_amdgcn_s_setprio(2);
    const int wco = wave % WCO, wpo = wave / WCO;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(P.wq), 0, (int)(4 * wplane), 0x00020000);
    const int lo_off = (int)(2 * wplane);                  // bytes from the hi plane to the lo plane
    const int tap_stride = CG * P.Cout * 32;               // bytes between taps
    const int cg_stride = P.Cout * 32;                     // bytes between 16-channel groups
    constexpr bool SC = FUSE == 2;
    static_assert(!SC || (KS == 3 && CO_FR == 1), "the fused shortcut needs a second accumulator set: 32-channel wave tiles");
    const long long wplane_d = (long long)CG * P.Cout * 16;                  // halves per plane of the shortcut weights
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(SC ? P.sc_w : P.wq), 0, (int)(4 * wplane_d), 0x00020000);

    // A fragments come from L2 (a chunk's weights exceed the 32 KiB L1): ring of NA slots, loaded
    // NA-1 taps ahead; 27 % NA == 0, so slot = tap % NA stays static across chunk boundaries.  The
    // last chunk of a tile prefetches chunk 0 again: exactly what the workgroup's next tile starts
    // with when it has the same channel tile (always, for power-of-two tile counts).
    f16x8 ah[KS == 1 ? 2 : NA][CO_FR], al[KS == 1 ? 2 : NA][CO_FR], bh[PO_FR], bl[PO_FR];
    int wlane[CO_FR];
#define V2CE_LOAD_A(slot_, soff_)                                                              \
    {                                                                                          \
        const int so_ = (soff_);                                                               \
        _Pragma("unroll") for (int q = 0; q < CO_FR; ++q) {                                    \
            ah[slot_][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, wlane[q], so_, 0));          \
            al[slot_][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, wlane[q], so_ + lo_off, 0)); \
        }                                                                                      \
    }
    int ring_co_t = -1;
    bool more = true;
    [[maybe_unused]] unsigned long long tc_all = TICK(), tc_bar = 0, tc_dump = 0, tc_own = 0, tc_mma = 0;
    while (more) {
        const int co0 = T.co_t * CO_TILE + wco * CO_FR * 32;     // this wave's first channel
        const float x_scale = scale_of(T.b);
        const float inv_scale = 1.0f / (x_scale * w_scale);      // a power of two: exact
        int bhb[PO_FR];
#pragma unroll
        for (int f = 0; f < PO_FR; ++f) {
            const int m = (wpo * PO_FR + f) * 32 + l32;
            bhb[f] = half * chs;
            if (m < P.n_pos) {
                const int tt = m / (P.TH * P.TW);
                const int rem = m - tt * (P.TH * P.TW);
                const int th = rem / P.TW;
                const int tw = rem - th * P.TW;
                bhb[f] += KS == 1 ? m : (tt * P.HH + th * S) * P.HWd + tw * S;
            }
        }
        if (T.co_t != ring_co_t) {                          // uniform: (re)load the ring for this channel tile
            ring_co_t = T.co_t;
#pragma unroll
            for (int q = 0; q < CO_FR; ++q) {
                int co = co0 + q * 32 + l32;
                co = co < P.Cout ? co : P.Cout - 1;
                wlane[q] = (co * 16 + 8 * half) * 2;
            }
            if constexpr (KS == 1) {
                V2CE_LOAD_A(1, 0)                           // chunk 0 (moved to slot 0 at the top of the chunk)
            } else {
                step_loop<0, NA - 1>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    V2CE_LOAD_A(t, t * tap_stride)          // chunk 0, taps 0 .. NA-2
                });
            }
        }

        f32x16 acc[CO_FR][PO_FR];
#pragma unroll
        for (int q = 0; q < CO_FR; ++q)
#pragma unroll
            for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][f][r] = 0.0f;

        f32x16 accd[SC ? CO_FR : 1][SC ? PO_FR : 1];         // shortcut accumulators
        if constexpr (SC) {
#pragma unroll
            for (int q = 0; q < CO_FR; ++q)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accd[q][f][r] = 0.0f;
        }
        for (int cg = 0; cg < CG; ++cg, ++gc) {
            const f16x8 *qb = pieces + (gc & 1) * 4 * chs;
            const int wc = cg * cg_stride;
            f16x8 ahd[CO_FR], ald[CO_FR];                    // this chunk's shortcut weights (used at the centre tap)
            if constexpr (SC) {
#pragma unroll
                for (int q = 0; q < CO_FR; ++q) {
                    ahd[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_d, wlane[q], wc, 0));
                    ald[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_d, wlane[q], wc + (int)(2 * wplane_d), 0));
                }
            }
            const int wn = cg + 1 < CG ? wc + cg_stride : 0;    // last chunk: chunk 0 again (the next tile's start)
            if constexpr (KS == 1) {                            // one tap per chunk: A double-buffered over chunks
#pragma unroll
                for (int q = 0; q < CO_FR; ++q) { ah[0][q] = ah[1][q]; al[0][q] = al[1][q]; }
                V2CE_LOAD_A(1, wn)
            }
            { [[maybe_unused]] const unsigned long long tb = TICK();
            __syncthreads();                                   // barrier gc: pieces[gc & 1] ready
            ACC_T(tc_bar, tb); }
            [[maybe_unused]] const unsigned long long tm = TICK();
            if (gc == 0) STAMP(0, 1);
            if (gc == 1) STAMP(0, 2);
#pragma unroll
            for (int f = 0; f < PO_FR; ++f) {
                bh[f] = qb[bhb[f]];
                bl[f] = qb[bhb[f] + 2 * chs];
            }
            if constexpr (KS == 1) {
#pragma unroll
                for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                    for (int q = 0; q < CO_FR; ++q) {
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][q], bh[f], acc[q][f], 0, 0, 0);
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][q], bl[f], acc[q][f], 0, 0, 0);
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0][q], bh[f], acc[q][f], 0, 0, 0);
                    }
            } else
            step_loop<0, K3>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                constexpr int nt = tap + 1;
                constexpr int dt = nt / 9, dh = (nt / 3) % 3, dw = nt % 3;
                constexpr int pt = tap + NA - 1;              // the tap whose A fragments are fetched now
#if !(defined(V2CE_ABLATE_TAPS) && (V2CE_ABLATE_TAPS & 1))   // diagnostic build (tools/tap_ablate.sh): the ring is never refilled -- WRONG results
                if constexpr (pt < K3) {
                    V2CE_LOAD_A(pt % NA, wc + pt * tap_stride)
                } else {
                    V2CE_LOAD_A(pt % NA, wn + (pt - K3) * tap_stride)
                }
#endif
                const int toff = (dt * P.HH + dh) * P.HWd + dw;          // next tap's offset in the halo box
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
#pragma unroll
                    for (int q = 0; q < CO_FR; ++q) {
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % NA][q], bh[f], acc[q][f], 0, 0, 0);
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % NA][q], bl[f], acc[q][f], 0, 0, 0);
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tap % NA][q], bh[f], acc[q][f], 0, 0, 0);
                    }
                    if constexpr (SC && tap == 13) {         // centre tap = the positions the 1x1x1 shortcut reads
#pragma unroll
                        for (int q = 0; q < CO_FR; ++q) {
                            accd[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahd[q], bh[f], accd[q][f], 0, 0, 0);
                            accd[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahd[q], bl[f], accd[q][f], 0, 0, 0);
                            accd[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ald[q], bh[f], accd[q][f], 0, 0, 0);
                        }
                    }
#if !(defined(V2CE_ABLATE_TAPS) && (V2CE_ABLATE_TAPS & 2))   // diagnostic build: tap 0's B fragments serve every tap -- WRONG results
                    if constexpr (nt < K3) {                 // refill in place for the next tap
                        bh[f] = qb[bhb[f] + toff];
                        bl[f] = qb[bhb[f] + toff + 2 * chs];
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            ACC_T(tc_mma, tm);
        }
        float out_inv_scale = inv_scale;
        if constexpr (TAIL) {
            // ---- folded 1x1x1 tail: P.tCG more chunks into the same accumulators.  They arrive at another power-of-two
            // scale (the tail input's and the tail weights' pre-scales): the accumulators are rescaled first (exact).
            const long long tplane = (long long)P.tCG * P.Cout * 16;              // halves per plane of the tail weights
            const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<_Float16 *>(P.sc_w), 0, (int)(4 * tplane), 0x00020000);
            const float t_scale = tscale_of(T.b);
            const float wt_scale = reinterpret_cast<const float *>(P.sc_w + 2 * tplane)[1];
            const float rho = (t_scale * wt_scale) / (x_scale * w_scale);
#pragma unroll
            for (int q = 0; q < CO_FR; ++q)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][f][r] *= rho;
            out_inv_scale = 1.0f / (t_scale * wt_scale);
            int bpt[PO_FR];                                  // B fragments of a tail chunk: indexed by output position
#pragma unroll
            for (int f = 0; f < PO_FR; ++f) {
                const int m = (wpo * PO_FR + f) * 32 + l32;
                bpt[f] = half * chs + (m < P.n_pos ? m : 0);
            }
            f16x8 ath[2][CO_FR], atl[2][CO_FR];              // A fragments, double-buffered over the channel groups
#pragma unroll
            for (int q = 0; q < CO_FR; ++q) {
                ath[0][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], 0, 0));
                atl[0][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], (int)(2 * tplane), 0));
            }
            int cgt = 0;                                     // channel group of the tail (weights order: tx0's, then tx1's)
            for (int k = 0; k < P.tSC; ++k, ++gc) {          // super-chunks: up to tTCH groups of one source per barrier
                const f16x8 *qb = pieces + (gc & 1) * 4 * chs;
                const int g0 = (k < P.tSC0 ? k : k - P.tSC0) * P.tTCH;
                const int ng = min(P.tTCH, (k < P.tSC0 ? P.tC0 : P.tC1) / 16 - g0);
                __syncthreads();                               // barrier gc: pieces[gc & 1] ready
                for (int g = 0; g < ng; ++g, ++cgt) {
                    if (cgt) {
#pragma unroll
                        for (int q = 0; q < CO_FR; ++q) { ath[0][q] = ath[1][q]; atl[0][q] = atl[1][q]; }
                    }
                    const int wn = (cgt + 1 < P.tCG ? cgt + 1 : cgt) * cg_stride;
#pragma unroll
                    for (int q = 0; q < CO_FR; ++q) {
                        ath[1][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], wn, 0));
                        atl[1][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], wn + (int)(2 * tplane), 0));
                    }
#pragma unroll
                    for (int f = 0; f < PO_FR; ++f) {
                        bh[f] = qb[bpt[f] + g * P.tNPP];
                        bl[f] = qb[bpt[f] + g * P.tNPP + 2 * chs];
                    }
#pragma unroll
                    for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                        for (int q = 0; q < CO_FR; ++q) {
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ath[0][q], bh[f], acc[q][f], 0, 0, 0);
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ath[0][q], bl[f], acc[q][f], 0, 0, 0);
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(atl[0][q], bh[f], acc[q][f], 0, 0, 0);
                        }
                }
            }
        }
        if (gc == CG) STAMP(0, 3);
        int poff[PO_FR];                                    // output offsets (not kept live across the main loop)
#pragma unroll
        for (int f = 0; f < PO_FR; ++f) {
            const int m = (wpo * PO_FR + f) * 32 + l32;
            poff[f] = -1;
            if (m < P.n_pos) {
                const int tt = m / (P.TH * P.TW);
                const int rem = m - tt * (P.TH * P.TW);
                const int th = rem / P.TW;
                const int tw = rem - th * P.TW;
                const int t = T.t0 + tt, h = T.h0 + th, w = T.w0 + tw;
                if (t < P.T && h < P.Hout && w < P.Wout)      // channels-last-16: byte offset of the position's group
                    poff[f] = 4 * ((t * P.Cout) * (P.Hout * P.Woutp)) + 64 * (h * P.Woutp + w);
            }
        }
#ifdef V2CE_ABLATE_EPI   // diagnostic build only: what the step would cost if the epilogue were free (nothing is written)
        if (P.ablate == 1) {
            float sink = 0.0f;
#pragma unroll
            for (int q = 0; q < CO_FR; ++q)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    sink += acc[q][f][0];
                    if constexpr (SC) sink += accd[q][f][0];
                }
            if (sink == 12345.678f && poff[0] >= 0) P.y[0] = sink;
        } else
#endif
        if constexpr (PEPI != 0) {
            // the producers run the epilogue of this wave's fragments kKeepFr .. PO_FR - 1 (tile_epilogue) out of the buffer of the tile's
            // last chunk, this wave the others', side by side
            [[maybe_unused]] const unsigned long long td = TICK();
            [[maybe_unused]] float pw_scale_c = 1.0f;
            if constexpr (FUSE == 1) pw_scale_c = reinterpret_cast<const float *>(P.pred_w + 2048)[0];
            f32x4d rvK[kKeepFr][4];                             // the residual of this wave's own fragments, requested in front of the hand-over
#pragma unroll
            for (int f = 0; f < kKeepFr; ++f) epi_load_res(T, f, rvK[f]);
            lds_barrier();                                      // every consumer wave is done with those pieces
            f32x4d *dump = reinterpret_cast<f32x4d *>(pieces + ((gc - 1) & 1) * 4 * chs) + wave * kDumpWave;
#pragma unroll
            for (int f = kKeepFr; f < PO_FR; ++f)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    dump[(((f - kKeepFr) * kDumpSets) * 4 + r4) * 64 + lane] = f32x4d{acc[0][f][4 * r4], acc[0][f][4 * r4 + 1], acc[0][f][4 * r4 + 2], acc[0][f][4 * r4 + 3]};
                    if constexpr (SC)
                        dump[(((f - kKeepFr) * kDumpSets + 1) * 4 + r4) * 64 + lane] = f32x4d{accd[0][f][4 * r4], accd[0][f][4 * r4 + 1], accd[0][f][4 * r4 + 2], accd[0][f][4 * r4 + 3]};
                }
            lds_barrier();                                      // the producers may start
            ACC_T(tc_dump, td);
            {
                [[maybe_unused]] const unsigned long long to = TICK();
                unsigned ymax = 0u;
                step_loop<0, kKeepFr>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    f32x4d a[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) a[r4] = f32x4d{acc[0][f][4 * r4], acc[0][f][4 * r4 + 1], acc[0][f][4 * r4 + 2], acc[0][f][4 * r4 + 3]};
                    if constexpr (SC) {
                        f32x4d ad[4];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) ad[r4] = f32x4d{accd[0][f][4 * r4], accd[0][f][4 * r4 + 1], accd[0][f][4 * r4 + 2], accd[0][f][4 * r4 + 3]};
                        epi_fragment_sc(T, f, a, ad, ymax);
                    } else
                    epi_fragment(T, f, a, rvK[f], ymax, out_inv_scale, pw_scale_c, []() {});
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (P.y_absmax) absmax_commit(__builtin_bit_cast(float, ymax), P.y_absmax + T.b * P.amax_bs);
                ACC_T(tc_own, to);
            }
        } else
        if constexpr (FUSE == 1 || FUSE == 4) {                 // 32-channel conv with the fused 1x1x1 head
            static_assert(KS == 3 && S == 1 && WCO == 1 && CO_FR == 1, "the fused head rides on a 32-channel tile");
            if constexpr (FUSE == 4 && RES == 1) {
                int rp[PO_FR];                                  // the low-resolution residual's offsets (conv_epilogue, rpoff)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    const int m = (wpo * PO_FR + f) * 32 + l32;
                    rp[f] = -1;
                    if (m < P.n_pos) {
                        const int tt = m / (P.TH * P.TW);
                        const int rem = m - tt * (P.TH * P.TW);
                        const int th = rem / P.TW;
                        const int tw = rem - th * P.TW;
                        const int t = T.t0 + tt, h = T.h0 + th, w = T.w0 + tw;
                        if (t < P.T && h < P.Hout && w < P.Wout)
                            rp[f] = 4 * ((t * P.Cout) * (P.rH * P.rWp)) + 64 * ((h >> 1) * P.rWp + (w >> 1));
                    }
                }
                conv_epilogue<CO_FR, PO_FR, true, true, RES, true>(P, acc, poff, co0, half, T.b, out_inv_scale, &rp);
            } else {
                conv_epilogue<CO_FR, PO_FR, true, true, RES, true>(P, acc, poff, co0, half, T.b, out_inv_scale);
            }
            pred_epilogue<PO_FR>(P, acc, wpo * PO_FR * 32, lane, T.b, T.t0, T.h0, T.w0);
        } else {
            conv_epilogue<CO_FR, PO_FR, true, false, RES, true>(P, acc, poff, co0, half, T.b, out_inv_scale);   // Cout need not fill the last channel tile
        }
#ifdef V2CE_ABLATE_EPI
        if (P.ablate != 1)
#endif
        if constexpr (SC && PEPI == 0) {                        // shortcut: bn_d(conv_d x), no activation, no residual
            ConvParams Q = P;
            Q.scale = P.sc_scale; Q.shift = P.sc_shift; Q.res = nullptr; Q.y = P.sc_y; Q.act = V2CE_ACT_NONE;
            Q.y_absmax = nullptr;
            const float wd_scale = reinterpret_cast<const float *>(P.sc_w + 2 * wplane_d)[1];
            conv_epilogue<CO_FR, PO_FR, true, false, 0, true>(Q, accd, poff, co0, half, T.b, 1.0f / (x_scale * wd_scale));
        }
        if (gc == CG) STAMP(0, 4);
        vb += (int)gridDim.x;
        more = next_tile(vb, T);
    }
    if constexpr (PEPI != 0) {
#ifdef V2CE_STAMP
        if (lane == 0 && wave == 0) {
            unsigned long long *o = P.stamps + (long long)gridDim.x * 16 + (long long)blockIdx.x * 2 * 8;
            o[0] = TICK() - tc_all; o[1] = tc_bar; o[2] = tc_dump; o[3] = tc_own; o[4] = tc_mma;
        }
#endif
    }
#undef V2CE_LOAD_A
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------
// Head convolution (unet_2layer.py:341: Conv3d(2, 32, 3, padding 1) + LeakyReLU): Cin = 2 makes it a
// K = 54 problem that writes 16x what it reads -- bound by the 32-channel output stream (HBM), not
// by arithmetic.  The generic exact-f32 kernel stages 2-channel chunks through its channel-chunked
// LDS pipeline and reaches 1 TB/s of output (0.73 ms per 64 frame-pairs).  Here: lane = one output
// position holding all 32 channels in 16 packed-f32 accumulators; per tap one ds_read_b32 of the
// 2-channel halo box (8 KB of LDS) feeds 16 v_pk_fma_f32 whose weights are scalar operands (the
// 54 x 32 table is read with s_load: it is the same for every lane); a channel's 64 positions leave
// as one 256-byte store.  ~50 VGPRs, so eight waves per SIMD cover the LDS / store latencies.
// (A one-pass v_mfma_f32_32x32x2_f32 version -- weights in 27 registers, one ds_read_b32 per MFMA --
// measured 0.53 ms: 188 registers, two waves per SIMD, every MFMA waits for its LDS operand.)
// Measured 0.42 ms per 64 frame-pairs; without the stores 0.26, with 2 of the 54 taps 0.37: the 737 MB
// output stream (32 planes, 256-byte pieces) is the bound at ~2 TB/s.
// f32 FMA chain over the 54 terms; the order differs from the generic kernel's: 1e-7 relative.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kHeadTT = 2, kHeadTH = 2, kHeadTW = 64;       // wave = (time step, row), lane = column
__global__ __launch_bounds__(256) void conv3d_head_kernel(ConvParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TT = kHeadTT, TH = kHeadTH, TW = kHeadTW;
    constexpr int HT = TT + 2, HH = TH + 2, HWd = TW + 2, PLANE = HT * HH * HWd;
    static_assert(TT * TH == 4, "one wave per (time step, row)");
    __shared__ float halo[2 * PLANE > 4096 ? 2 * PLANE : 4096];   // halo box, then 4 KB per wave for the output transpose
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups are dealt round-robin over the 8 XCDs: XCD x walks the contiguous range [x n/8, (x+1) n/8)
    // of boxes, so the boxes that share output cache lines (rows are not 128-byte aligned) and halo
    // rows meet in one L2
    int bid = (int)(blockIdx.x & 7) * P.per_xcd + (int)(blockIdx.x >> 3);
    if (bid >= P.n_spatial) return;
    const int iw = bid % P.nW; bid /= P.nW;
    const int ih = bid % P.nH; bid /= P.nH;
    const int it = bid % P.nT;
    const int b = bid / P.nT;
    const int t0 = it * TT, h0 = ih * TH, w0 = iw * TW;

    // halo box: rows of HWd consecutive floats (one row per (channel, ht, hh)), coalesced along W
    const float *xb = P.x0 + (long long)b * P.T * 2 * (P.H0 * P.W0p);
    for (int row = wave; row < 2 * HT * HH; row += 4) {
        const int ci = row / (HT * HH), r = row - ci * (HT * HH);
        const int ht = r / HH, hh = r - ht * HH;
        const int t = t0 + ht - 1, h = h0 + hh - 1;
        const bool rok = t >= 0 && t < P.T && h >= 0 && h < P.H0;
        const float *src = xb + ((long long)(rok ? t : 0) * 2 + ci) * (P.H0 * P.W0p) + (rok ? h : 0) * P.W0p;
        for (int ww = lane; ww < HWd; ww += 64) {
            const int w = w0 + ww - 1;
            halo[row * HWd + ww] = (rok && w >= 0 && w < P.W0) ? src[w] : 0.0f;
        }
    }
    __syncthreads();

    const int tt = wave / TH, th = wave - tt * TH;
    const float *hp = halo + (tt * HH + th) * HWd + lane;   // (tt, th, tw = lane) in the halo box
    f32x2 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x2{0.0f, 0.0f};
    // weights [kk = ci * 27 + tap][32]: the same for every lane => SGPR operands of the packed FMAs.  Written
    // as explicit s_load_dwordx16 pairs, one tap ahead (ping-pong A / B): left to the compiler, the 1728
    // uniform loads are hoisted to the top and spilled into VGPR lanes (2 v_readlane per FMA).  The waits are
    // asm statements that "modify" the registers, so no FMA can be scheduled above its wait.
    typedef float f32x16s __attribute__((ext_vector_type(16)));
    f32x16s wA0, wA1, wB0, wB1;
    const float *wptr = P.wp;
    // (x rides through the load statement so that the FMAs reading it are issued after the loads, not before)
#define V2CE_SLOAD(lo_, hi_, kk_)                                                                                      \
    asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5" : "=&s"(lo_), "=&s"(hi_), "+v"(x)          \
                 : "s"(wptr), "n"((kk_) * 128), "n"((kk_) * 128 + 64))
#define V2CE_SWAIT(lo_, hi_) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lo_), "+s"(hi_))
    {
        float x = 0.0f;
        V2CE_SLOAD(wA0, wA1, 0);
    }
    step_loop<0, 54>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        constexpr int off = (kk / 27) * PLANE + (((kk % 27) / 9) * HH + ((kk % 27) / 3) % 3) * HWd + (kk % 3);
        float x = hp[off];
        if constexpr (kk % 2 == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wA0), "+s"(wA1), "+v"(x));
            if constexpr (kk + 1 < 54) V2CE_SLOAD(wB0, wB1, kk + 1);
            const f32x2 xx{x, x};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_elementwise_fma(f32x2{wA0[2 * j], wA0[2 * j + 1]}, xx, acc[j]);
                acc[8 + j] = __builtin_elementwise_fma(f32x2{wA1[2 * j], wA1[2 * j + 1]}, xx, acc[8 + j]);
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wB0), "+s"(wB1), "+v"(x));
            if constexpr (kk + 1 < 54) V2CE_SLOAD(wA0, wA1, kk + 1);
            const f32x2 xx{x, x};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_elementwise_fma(f32x2{wB0[2 * j], wB0[2 * j + 1]}, xx, acc[j]);
                acc[8 + j] = __builtin_elementwise_fma(f32x2{wB1[2 * j], wB1[2 * j + 1]}, xx, acc[8 + j]);
            }
        }
        // pin the tap's FMAs between its wait and the next tap's loads: instruction selection is free to sink
        // them below every (volatile) asm otherwise -- and did, spilling all 54 taps' weights
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                     "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]),
                     "+v"(acc[14]), "+v"(acc[15]));
    });
#undef V2CE_SLOAD
#undef V2CE_SWAIT

    const int t = t0 + tt, h = h0 + th, w = w0 + lane;
    const bool ok = t < P.T && h < P.Hout && w < P.Wout;
    const int hw = P.Hout * P.Woutp;
    const long long seq = (long long)P.T * P.Cout * hw;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y + b * seq, 0, (int)(seq * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.res ? P.res + b * seq : P.scale), 0, P.res ? (int)(seq * 4) : 0, 0x00020000);
    const float slope = act_slope(P.act);
    const float *__restrict__ scale = P.scale;
    const float *__restrict__ shift = P.shift;
    float ymax = 0.0f;
    if (P.c16) {
        // channels-last-16 output (what the residual blocks' kernels gather): the lane's 32 channels are two
        // 64-byte groups, eight 16-byte stores
        typedef float f32x4h __attribute__((ext_vector_type(4)));
        typedef unsigned u32x4h __attribute__((ext_vector_type(4)));
        const unsigned vo = ok ? (unsigned)(4 * ((t * P.Cout) * hw) + 64 * (h * P.Woutp + w)) : kOOB;
        if (!P.res) {
            // The wave's 64 positions x 16 channels of a group are 4 KB of contiguous memory.  Stored from the
            // MFMA-free layout (lane = position) every instruction would touch 64 lines with 16 bytes each; through
            // LDS (the halo box is dead by now) lane L of store j writes bytes [1024 j + 16 L, + 16): whole lines.
            __syncthreads();                                 // every wave is done reading the halo box
            f32x4h *tr = reinterpret_cast<f32x4h *>(halo) + wave * 256;          // 4 KB per wave
            const int wlast = P.Wout - w0;                   // positions of this row inside the tensor (wave-uniform)
            const bool row_ok = t < P.T && h < P.Hout;
            const unsigned vrow = (unsigned)(4 * ((t * P.Cout) * hw) + 64 * (h * P.Woutp + w0));
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    f32x4h o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int co = 16 * g + 4 * c4 + k;
                        float v = acc[co >> 1][co & 1] * scale[co] + shift[co];
                        v = apply_act(v, slope);
                        o[k] = v;
                        ymax = fmaxf(ymax, ok ? fabsf(v) : 0.0f);
                    }
                    tr[lane * 4 + c4] = o;                   // position-major: element `lane`, quarter c4
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 64 * j + lane;             // 16-byte piece q of the 4 KB: position q >> 2, quarter q & 3
                    f32x4h o = tr[q];
                    const bool pok = row_ok && (q >> 2) < wlast;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, o), rs_y, pok ? vrow + 16u * (unsigned)q : kOOB,
                                                           g * (hw * 64), 0);
                    asm volatile("s_nop 1" : "+v"(o));       // store-data hazard of 16-byte stores, see conv_epilogue
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const int so = (c4 >> 2) * (hw * 64) + 16 * (c4 & 3);
            f32x4h r{0.0f, 0.0f, 0.0f, 0.0f}, o;
            r = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo, so, 0));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = 4 * c4 + k;
                float v = acc[co >> 1][co & 1] * scale[co] + shift[co];
                v += r[k];
                v = apply_act(v, slope);
                o[k] = v;
                ymax = fmaxf(ymax, ok ? fabsf(v) : 0.0f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, o), rs_y, vo, so, 0);
            asm volatile("s_nop 1" : "+v"(o));           // store-data hazard of 16-byte stores, see conv_epilogue
        }
        }
    } else {
        const unsigned vo = ok ? (unsigned)(((t * P.Cout) * hw + h * P.Woutp + w) * 4) : kOOB;
#pragma unroll
        for (int co = 0; co < 32; ++co) {
            float v = acc[co >> 1][co & 1] * scale[co] + shift[co];
            if (P.res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, vo, co * hw * 4, 0));
            v = apply_act(v, slope);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, vo, co * hw * 4, 0);
            ymax = fmaxf(ymax, ok ? fabsf(v) : 0.0f);
        }
    }
    if (P.y_absmax) absmax_commit(ymax, P.y_absmax + b * P.amax_bs);
#endif
}

// ---------------------------------------------------------------------------------------------
// host side: tile choice + dispatch
// ---------------------------------------------------------------------------------------------
struct Tile { int tt, th, tw; };

// best (tt,th,tw) with tt*th*tw <= pos_tile and halo plane <= max_plane: maximise the fraction of
// MFMA lanes that compute real outputs, then prefer wide rows (coalescing) and small halos.
Tile choose_tile(int T, int Ho, int Wo, int ks, int s, int pos_tile, int max_plane) {
    Tile best{1, 1, 1};
    double best_eff = -1.0;
    long long best_halo = 0;
    for (int tt = 1; tt <= T && tt <= 16; tt *= 2) {
        for (int th = 1; th <= Ho && th <= 64; ++th) {
            const int max_tw = pos_tile / (tt * th);
            if (max_tw < 1) break;
            for (int tw = 1; tw <= Wo && tw <= max_tw; ++tw) {
                const long long plane = (long long)(tt + ks - 1) * ((th - 1) * s + ks) * ((tw - 1) * s + ks);
                if (plane > max_plane) break;
                const long long ntiles = (long long)((T + tt - 1) / tt) * ((Ho + th - 1) / th) *
                                         ((Wo + tw - 1) / tw);
                const double eff = (double)T * Ho * Wo / ((double)ntiles * pos_tile);
                const bool better = eff > best_eff + 1e-9 ||
                                    (eff > best_eff - 1e-9 && (tw > best.tw || (tw == best.tw && plane < best_halo)));
                if (better) { best = {tt, th, tw}; best_eff = eff; best_halo = plane; }
            }
        }
    }
    return best;
}

// Box override for in-network sweeps (tools/box_sweep.py): V2CE_BOX_<Ho>x<Wo>_<stride>_<positions>=tt,th,tw.
// Sweeps of 5-6 candidate boxes per layer class INSIDE the network (bench.py, +-0.05 ms of 23.7 ms per step)
// found the lane-efficiency search's choices within 1 % of the best everywhere -- while the same candidates
// timed on isolated layers (tools/tile_probe.py: one kernel in a loop, inputs resident in L2, another power
// state) had differed by 8-16 % and ranked the other way round on the 130x173 layers.  Only in-network winners
// enter the small table below.
Tile measured_box(int T, int Ho, int Wo, int s, int pos_tile) {
    char key[64];
    snprintf(key, sizeof key, "V2CE_BOX_%dx%d_%d_%d", Ho, Wo, s, pos_tile);
    if (const char *e = getenv(key)) {
        Tile t{0, 0, 0};
        if (sscanf(e, "%d,%d,%d", &t.tt, &t.th, &t.tw) == 3 && t.tt > 0 && t.th > 0 && t.tw > 0 && t.tt * t.th * t.tw <= pos_tile) return t;
    }
    // Boxes that the in-network sweep (channels-last-16 layout) put 3-4 sigma ahead of the search's choice: smaller halo
    // volume at the same lane use.  346x260: (8,4,16) 21.47 vs (4,4,32) 21.66 ms per step; 87x65: (8,4,8) 21.58 vs
    // (16,2,8) 21.73; 44x33: (2,11,11) 21.70 vs (1,11,23) 21.76.  V2CE_MEASURED_BOXES=0 disables the table.
    static const bool on = [] { const char *e = getenv("V2CE_MEASURED_BOXES"); return !(e && e[0] == '0'); }();
    struct Row { int Ho, Wo, s, pos; Tile t; };
    static const Row rows[] = {{260, 346, 1, 512, {8, 4, 16}}, {65, 87, 1, 256, {8, 4, 8}}, {33, 44, 1, 256, {2, 11, 11}}};
    if (on)
        for (const Row &r : rows)
            if (r.Ho == Ho && r.Wo == Wo && r.s == s && r.pos == pos_tile && T % r.t.tt == 0) return r.t;
    return Tile{0, 0, 0};
}

thread_local char *g_name_out = nullptr;   // non-null: report the variant instead of launching
thread_local size_t g_name_cap = 0;

double tile_efficiency(int T, int Ho, int Wo, int ks, int s, int pos_tile, int max_plane) {
    const Tile t = choose_tile(T, Ho, Wo, ks, s, pos_tile, max_plane);
    const long long ntiles = (long long)((T + t.tt - 1) / t.tt) * ((Ho + t.th - 1) / t.th) * ((Wo + t.tw - 1) / t.tw);
    return (double)T * Ho * Wo / ((double)ntiles * pos_tile);
}

template <int KS, int S, int CO_FR, int PO_FR, int CK, int EPT, int MW, int IL>
int launch(ConvParams P, const v2ce_conv3d_desc &d, hipStream_t stream) {
    using Cfg = ConvCfg<KS, S, CO_FR, PO_FR, CK, EPT>;
    if (g_name_out) {
        snprintf(g_name_out, g_name_cap, "conv3d_kernel<%d,%d,%d,%d,%d,%d,%d,%d>", KS, S, CO_FR, PO_FR, CK, EPT, MW, IL);
        return V2CE_OK;
    }
    Tile t{d.tile_t, d.tile_h, d.tile_w};
    if (t.tt <= 0 || t.th <= 0 || t.tw <= 0) {
        static std::mutex mu;
        static std::map<std::tuple<int, int, int, int, int, int, int>, Tile> cache;
        std::lock_guard<std::mutex> g(mu);
        auto key = std::make_tuple(d.T, d.Hout, d.Wout, KS, S, Cfg::POS_TILE, Cfg::MAX_PLANE);
        auto it = cache.find(key);
        if (it == cache.end())
            it = cache.emplace(key, choose_tile(d.T, d.Hout, d.Wout, KS, S, Cfg::POS_TILE, Cfg::MAX_PLANE)).first;
        t = it->second;
    }
    P.TT = t.tt; P.TH = t.th; P.TW = t.tw;
    P.n_pos = t.tt * t.th * t.tw;
    P.HT = t.tt + KS - 1; P.HH = (t.th - 1) * S + KS; P.HWd = (t.tw - 1) * S + KS;
    P.plane = P.HT * P.HH * P.HWd;
    V2CE_REQUIRE(P.n_pos <= Cfg::POS_TILE && P.plane <= Cfg::MAX_PLANE, V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd: tile %dx%dx%d does not fit (positions %d/%d, halo %d/%d)", t.tt,
                 t.th, t.tw, P.n_pos, Cfg::POS_TILE, P.plane, Cfg::MAX_PLANE);
    P.nT = (d.T + t.tt - 1) / t.tt; P.nH = (d.Hout + t.th - 1) / t.th; P.nW = (d.Wout + t.tw - 1) / t.tw;
    P.n_co_tiles = (d.Cout + Cfg::CO_TILE - 1) / Cfg::CO_TILE;
    P.n_spatial = d.B * P.nT * P.nH * P.nW;
    static const int remap_env = [] { const char *e = getenv("V2CE_XCD_REMAP"); return e ? atoi(e) : 1; }();
    P.xcd_remap = (remap_env && P.n_co_tiles > 1) ? 1 : 0;
    const long long blocks = P.xcd_remap ? (long long)((P.n_spatial + 7) / 8) * 8 * P.n_co_tiles
                                         : (long long)P.n_spatial * P.n_co_tiles;
    V2CE_REQUIRE(blocks > 0 && blocks < (1ll << 31), V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd: grid too large");
    // two buffers of { halo [CK][plane rounded to 64], weight slab rounded to whole wave rows }
    const int chs = (P.plane + 63) & ~63;
    const int w_units = Cfg::K3 * CK * (Cfg::CO_TILE / 4);
    const int w_floats = ((w_units + 63) / 64) * 256;
    const size_t lds = (size_t)2 * (CK * chs + w_floats) * sizeof(float);
    V2CE_REQUIRE(lds <= 160 * 1024, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd: %zu B of LDS", lds);
    auto kern = conv3d_kernel<KS, S, CO_FR, PO_FR, CK, EPT, MW, IL>;
    V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, P);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

// tail[0] = max |w / sigma| (tail[0] zeroed by the caller)
__global__ __launch_bounds__(256) void weights_absmax_kernel(const float *__restrict__ w, long long n,
                                                             const float *sigma, float *tail) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = w[i];
        if (sigma) v = v / sigma[0];
        m = fmaxf(m, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(tail), __float_as_uint(m));
}

template <int KS, int S, int WCO, int CO_FR, int PO_FR, int NA, int FUSE = 0, int RES = 2, int OPT = 0>
int launch_f16x2_ws(ConvParams P, const v2ce_conv3d_desc &d, hipStream_t stream) {
    constexpr int PEPI = OPT & 1;
    static_assert(27 % NA == 0 && NA >= 2, "the A-fragment ring must divide the 27 taps");
    constexpr int CO_TILE = WCO * CO_FR * 32, POS_TILE = (4 / WCO) * PO_FR * 32;
    constexpr int MAX_PLANE = PEPI ? 1152 : 1280;      // 128 B of LDS per halo element; 5 elements per producer lane (PEPI: + up to 8.2 KB of tables)
    if (g_name_out) {
        snprintf(g_name_out, g_name_cap, "conv3d_f16x2_ws_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d>", KS, S, WCO, CO_FR, PO_FR, NA, FUSE, RES, OPT);
        return V2CE_OK;
    }
    Tile t{d.tile_t, d.tile_h, d.tile_w};
    if (t.tt <= 0 || t.th <= 0 || t.tw <= 0) {
        t = choose_tile(d.T, d.Hout, d.Wout, KS, KS == 1 ? 1 : S, POS_TILE, MAX_PLANE);
        if (KS == 3) {
            const Tile m = measured_box(d.T, d.Hout, d.Wout, S, POS_TILE);
            if (m.tt > 0) t = m;
        }
    }
    P.TT = t.tt; P.TH = t.th; P.TW = t.tw;
    P.n_pos = t.tt * t.th * t.tw;
    if (KS == 1) { P.HT = t.tt; P.HH = t.th; P.HWd = t.tw; }          // the gathered box is the output box
    else { P.HT = t.tt + KS - 1; P.HH = (t.th - 1) * S + KS; P.HWd = (t.tw - 1) * S + KS; }
    P.plane = P.HT * P.HH * P.HWd;
    V2CE_REQUIRE(P.n_pos <= POS_TILE && P.plane <= MAX_PLANE, V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd(f16x2 ws): tile does not fit");
    V2CE_REQUIRE((long long)d.T * d.Cout * d.Hout * d.Wout < (1ll << 29), V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd(f16x2): an output sequence exceeds the 2 GiB buffer-descriptor range");
    P.nT = (d.T + t.tt - 1) / t.tt; P.nH = (d.Hout + t.th - 1) / t.th; P.nW = (d.Wout + t.tw - 1) / t.tw;
    P.n_co_tiles = (d.Cout + CO_TILE - 1) / CO_TILE;
    P.n_spatial = d.B * P.nT * P.nH * P.nW;
    P.xcd_remap = 1;
    P.per_xcd = (P.n_spatial + 7) / 8;
    const long long blocks = (long long)8 * P.per_xcd * P.n_co_tiles;
    int chs = (P.plane + 63) & ~63;
    if (FUSE == 3 || FUSE == 4) {
        // tail super-chunks: as many channel groups per barrier as the producers' 5 x 256 element slots (= 160 KB of pieces) hold
        P.tNPP = (P.n_pos + 63) & ~63;
        const int tch = 1280 / P.tNPP;
        P.tTCH = tch < 1 ? 1 : (tch > 15 ? 15 : tch);
        P.tSC0 = (P.tC0 / 16 + P.tTCH - 1) / P.tTCH;
        P.tSC = P.tSC0 + (P.tC1 / 16 + P.tTCH - 1) / P.tTCH;
        chs = chs > P.tTCH * P.tNPP ? chs : P.tTCH * P.tNPP;
        P.tCHS = chs;
    }
    if (PEPI) {                             // a pieces buffer doubles as the accumulator dump of the four consumer waves (16 B x 64 lanes per register quad)
        constexpr int keep = FUSE == 1 ? V2CE_PEPI_KEEP : PO_FR / 2;
        constexpr int dump = (PO_FR - keep) * (FUSE == 2 ? 2 : 1) * 4 * 64;     // (kDumpWave of the kernel)
        chs = chs > dump ? chs : dump;
        V2CE_REQUIRE(FUSE == 1 || d.Cout <= 512, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd(f16x2 ws, shared epilogue): more than 512 output channels");
    }
    const size_t lds = (size_t)chs * (2 * 4 * 16) + (PEPI ? (FUSE == 1 ? 400 + 4096 : 16 + 16 * (size_t)d.Cout) : 0);   // (PEPI: + the epilogue's tables)
    V2CE_REQUIRE(lds <= 160 * 1024, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd(f16x2 ws): %zu B of LDS", lds);
    auto kern = conv3d_f16x2_ws_kernel<KS, S, WCO, CO_FR, PO_FR, NA, FUSE, RES, OPT>;
    V2CE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef V2CE_STAMP
    V2CE_HIP_CHECK(hipMalloc(&P.stamps, (size_t)blocks * 32 * sizeof(unsigned long long)));    // (PEPI: accumulators behind the first grid * 16)
    V2CE_HIP_CHECK(hipMemset(P.stamps, 0, (size_t)blocks * 32 * sizeof(unsigned long long)));
#endif
    // persistent: one workgroup per CU walks the virtual blocks (a multiple of 8 keeps tiles on their XCD)
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n < 8 ? 8 : (n / 8) * 8;
    }();
    P.total_blocks = (int)blocks;
    // (measured in the network, same box: 2200 vs 2143 frame-pairs/s against one tile per workgroup)
    const unsigned grid = (unsigned)(blocks > n_cu ? n_cu : blocks);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, P);
    V2CE_HIP_CHECK(hipGetLastError());
#ifdef V2CE_STAMP
    {
        std::vector<unsigned long long> h((size_t)blocks * 32);
        V2CE_HIP_CHECK(hipDeviceSynchronize());
        V2CE_HIP_CHECK(hipMemcpy(h.data(), P.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        V2CE_HIP_CHECK(hipFree(P.stamps));
        auto med = [&](int role, int a, int b) {
            std::vector<long long> v;
            for (long long k = 0; k < blocks; ++k) {
                const unsigned long long x = h[(k * 2 + role) * 8 + a], y = h[(k * 2 + role) * 8 + b];
                if (x && y) v.push_back((long long)(y - x));
            }
            if (v.empty()) return -1ll;
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
        };
        fprintf(stderr, "[stamp ws<%d,%d,%d,%d,%d,%d> CG=%d plane=%d blocks=%lld] consumer: setup->bar0 %lld, chunk0 %lld, bar1->end-of-loop %lld, epilogue %lld, total %lld | "
                "producer: issue0 %lld, wait+convert0 %lld, issue1 %lld, bar0 wait %lld, chunk1 iter %lld, total %lld\n",
                KS, S, WCO, CO_FR, PO_FR, NA, P.Cin / 16, P.plane, blocks, med(0, 0, 1), med(0, 1, 2), med(0, 2, 3), med(0, 3, 4), med(0, 0, 4),
                med(1, 0, 1), med(1, 1, 2), med(1, 2, 3), med(1, 3, 4), med(1, 4, 5), med(1, 0, 6));
        if (PEPI) {
            const unsigned grid_ = (unsigned)(blocks > n_cu ? n_cu : blocks);
            auto acc = [&](int role, int k) {
                std::vector<long long> v;
                for (unsigned b = 0; b < grid_; ++b) {
                    const unsigned long long x = h[(size_t)grid_ * 16 + ((size_t)b * 2 + role) * 8 + k];
                    if (h[(size_t)grid_ * 16 + ((size_t)b * 2 + role) * 8]) v.push_back((long long)x);
                }
                if (v.empty()) return -1ll;
                std::sort(v.begin(), v.end());
                return v[v.size() / 2];
            };
            const double tiles = (double)blocks / grid_;
            fprintf(stderr, "    [pepi, s_memtime ticks per workgroup, %.1f tiles each] consumer: total %lld, chunk barriers %lld, hand-over barriers + dump %lld | "
                    "producer: total %lld, barriers %lld, epilogues %lld, convert %lld, load issue %lld\n", tiles,
                    acc(0, 0), acc(0, 1), acc(0, 2), acc(1, 0), acc(1, 1), acc(1, 2), acc(1, 3), acc(1, 4));
            fprintf(stderr, "    [pepi] consumer: tap loops %lld, own half of the epilogues %lld\n", acc(0, 4), acc(0, 3));

        }
    }
#endif
    return V2CE_OK;
}

// wq[plane][tap][cg][co][16] = fp16 hi / lo of s * w[co][cg*16 + j][tap] / sigma, s = the power of two
// that puts max |w/sigma| in [2^14, 2^15); tail = { max |w/sigma|, s } behind the two planes
__global__ __launch_bounds__(256) void pack_weights_f16x2_kernel(const float *__restrict__ w, int Cout,
                                                                 int Cin, int k3, const float *sigma,
                                                                 _Float16 *__restrict__ wq) {
    const long long n = (long long)Cout * Cin * k3;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // ((tap*CG + cg)*Cout + co)*16 + j
    float *tail = reinterpret_cast<float *>(wq + 2 * n);
    const float w_scale = pow2_prescale(tail[0]);
    if (i == 0) tail[1] = w_scale;
    if (i >= n) return;
    const int j = (int)(i & 15);
    long long r = i >> 4;
    const int co = (int)(r % Cout); r /= Cout;
    const int CG = Cin / 16;
    const int cg = (int)(r % CG);
    const int tap = (int)(r / CG);
    float v = w[((long long)co * Cin + cg * 16 + j) * k3 + tap];
    if (sigma) v = v / sigma[0];
    v *= w_scale;
    const _Float16 h = (_Float16)v;
    wq[i] = h;
    wq[n + i] = (_Float16)(v - (float)h);
}

// A fragments of the fused 1x1x1 head: table[k][plane][o][16] fp16 (o < 32, rows >= cout zero) with
// table[k][.][o][8 * half + j] = hi / lo of s * w[o][(j & 3) + 8 (j >> 2) + 16 k + 4 half]  (the channel
// order in which a lane of the 32-channel conv holds its outputs, see pred_epilogue), then { s } as a
// float; s = the power of two that puts max |w| in [2^14, 2^15).  One workgroup.
__global__ __launch_bounds__(256) void pack_pred_weights_kernel(const float *__restrict__ w, int cout,
                                                                _Float16 *__restrict__ table) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int i = threadIdx.x; i < cout * 32; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float s = pow2_prescale(red[0]);
    for (int e = threadIdx.x; e < 2 * 32 * 16; e += 256) {
        const int k = e >> 9, o = (e >> 4) & 31, kk = e & 15, half = kk >> 3, j = kk & 7;
        const int c = (j & 3) + 8 * (j >> 2) + 16 * k + 4 * half;
        const float v = o < cout ? w[o * 32 + c] * s : 0.0f;
        const _Float16 h = (_Float16)v;
        table[((k * 2 + 0) * 32 + o) * 16 + kk] = h;
        table[((k * 2 + 1) * 32 + o) * 16 + kk] = (_Float16)(v - (float)h);
    }
    if (threadIdx.x == 0) reinterpret_cast<float *>(table + 2048)[0] = s;
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ w, int Cout,
                                                           int Cin, int k3, const float *sigma,
                                                           float *__restrict__ wp) {
    // wp[ci][tap][co] = w[co][ci][tap] / sigma     (spectral_norm.py:31: elementwise true division)
    const long long n = (long long)Cout * Cin * k3;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % Cout);
    const long long r = i / Cout;          // ci*k3 + tap
    float v = w[(long long)co * Cin * k3 + r];
    if (sigma) v = v / sigma[0];
    wp[i] = v;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

static int conv3d_dispatch(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                           const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                           const float *scale, const float *shift, const float *residual,
                           float *y, const float *x0_absmax, const float *x1_absmax, float *y_absmax,
                           v2ce_stream_t stream, const void *pred_w = nullptr, const float *pred_b = nullptr,
                           int pred_cout = 0, float *pred_y = nullptr, const void *sc_w = nullptr,
                           const float *sc_scale = nullptr, const float *sc_shift = nullptr, float *sc_y = nullptr,
                           const v2ce_conv3d_desc *tail = nullptr, const float *tx0 = nullptr, const float *tx1 = nullptr,
                           const int32_t *thmap = nullptr, const int32_t *twmap = nullptr,
                           const float *tx0_absmax = nullptr, const float *tx1_absmax = nullptr, int res_h = 0, int res_wp = 0) {
    clear_error();
    V2CE_REQUIRE(desc && (g_name_out || (x0 && w_packed && scale && shift && (y || pred_w))), V2CE_ERR_BAD_ARG,
                 "v2ce_conv3d_fwd: null pointer");
    const v2ce_conv3d_desc &d = *desc;
    V2CE_REQUIRE(d.B > 0 && d.T > 0 && d.C0 > 0 && d.C1 >= 0 && d.Hin > 0 && d.Win > 0 && d.Cout > 0,
                 V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: bad shape");
    V2CE_REQUIRE(d.ksize == 1 || d.ksize == 3, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd: ksize %d", d.ksize);
    V2CE_REQUIRE(d.stride_hw == 1 || d.stride_hw == 2, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd: stride %d", d.stride_hw);
    V2CE_REQUIRE(d.act >= 0 && d.act <= 2, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: act %d", d.act);
    V2CE_REQUIRE(d.C1 == 0 || x1 || g_name_out, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: C1>0 needs x1");
    V2CE_REQUIRE((hmap == nullptr) == (wmap == nullptr), V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: hmap/wmap");
    V2CE_REQUIRE(hmap || (d.H0 == d.Hin && d.W0 == d.Win), V2CE_ERR_BAD_ARG,
                 "v2ce_conv3d_fwd: source 0 is %dx%d, logical input %dx%d: index maps required", d.H0,
                 d.W0, d.Hin, d.Win);
    const int pad = d.ksize / 2, s = d.stride_hw;
    V2CE_REQUIRE(d.Hout == (d.Hin + 2 * pad - d.ksize) / s + 1 && d.Wout == (d.Win + 2 * pad - d.ksize) / s + 1,
                 V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: output size %dx%d inconsistent", d.Hout, d.Wout);
    V2CE_REQUIRE(d.Cout % 4 == 0, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd: Cout %% 4 != 0");
    const int W0p = d.W0_pitch > 0 ? d.W0_pitch : d.W0, Winp = d.Win_pitch > 0 ? d.Win_pitch : d.Win;
    const int Woutp = d.Wout_pitch > 0 ? d.Wout_pitch : d.Wout;
    V2CE_REQUIRE(W0p >= d.W0 && Winp >= d.Win && Woutp >= d.Wout, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: a row pitch is smaller than its width");
    const long long seq_in0 = (long long)d.T * d.C0 * d.H0 * W0p, seq_in1 = (long long)d.T * d.C1 * d.Hin * Winp;
    const long long seq_out = (long long)d.T * d.Cout * d.Hout * Woutp;
    V2CE_REQUIRE(seq_in0 < (1ll << 29) && seq_in1 < (1ll << 29) && seq_out < (1ll << 29),
                 V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd: a single sequence exceeds the 2 GiB buffer-descriptor range");

    ConvParams P{};
    P.x0 = x0; P.x1 = x1; P.hmap = hmap; P.wmap = wmap; P.wp = w_packed; P.scale = scale;
    P.shift = shift; P.res = residual; P.y = y;
    P.B = d.B; P.T = d.T; P.C0 = d.C0; P.H0 = d.H0; P.W0 = d.W0; P.C1 = d.C1; P.Hin = d.Hin;
    P.Win = d.Win; P.Cin = d.C0 + d.C1; P.Cout = d.Cout; P.Hout = d.Hout; P.Wout = d.Wout;
    P.W0p = W0p; P.Winp = Winp; P.Woutp = Woutp;
    P.c16 = d.layout == V2CE_LAYOUT_C16 ? 1 : 0;
    V2CE_REQUIRE(d.layout == V2CE_LAYOUT_PLANAR || d.layout == V2CE_LAYOUT_C16, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: layout %d", d.layout);
    P.act = d.act;
    P.x0_absmax = x0_absmax; P.x1_absmax = d.C1 > 0 ? x1_absmax : nullptr; P.y_absmax = y_absmax;
    P.guard = y_absmax ? y_absmax + 1 : nullptr;
    P.amax_bs = d.absmax_batch_stride;
    V2CE_REQUIRE(P.amax_bs == 0 || P.amax_bs >= 2, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: absmax_batch_stride must be 0 or >= 2");
#ifdef V2CE_ABLATE_EPI
    { const char *e = getenv("V2CE_ABLATE_EPI"); P.ablate = e ? atoi(e) : 0; }
#endif
    if (pred_w && !y) P.y_absmax = nullptr;      // no y is materialised: only the guard value is reported
    P.pred_w = static_cast<const _Float16 *>(pred_w); P.pred_b = pred_b; P.pred_cout = pred_cout; P.pred_y = pred_y;
    P.sc_w = static_cast<const _Float16 *>(sc_w); P.sc_scale = sc_scale; P.sc_shift = sc_shift; P.sc_y = sc_y;
    if (tail) {
        const v2ce_conv3d_desc &t = *tail;
        // (round 6, v2ce_conv3d_fwd_tail_pred: the 32-channel conv with the fused head takes a tail too, and with it a low-resolution
        // residual -- the last decoder block's shortcut split by source)
        const bool tail_pred = pred_w && d.Cout == 32 && (!residual || res_h > 0);
        V2CE_REQUIRE(d.precision == V2CE_PRECISION_F16X2 && d.ksize == 3 && d.stride_hw == 1 && sc_w && !sc_y &&
                     (tail_pred || (d.Cout >= 64 && !pred_w && !residual)), V2CE_ERR_UNSUPPORTED,
                     "v2ce_conv3d_fwd_tail: the folded tail rides behind a split-half 3x3x3 stride-1 conv with >= 64 output channels, "
                     "no residual and no fused head -- or behind the 32-channel conv with the fused head (v2ce_conv3d_fwd_tail_pred)");
        if (residual && res_h > 0) {
            V2CE_REQUIRE(res_h == (d.Hout + 1) / 2 && res_wp >= (d.Wout + 1) / 2, V2CE_ERR_BAD_ARG,
                         "v2ce_conv3d_fwd_tail_pred: an upsampled residual has ceil(Hout / 2) rows of at least ceil(Wout / 2) columns");
            V2CE_REQUIRE((long long)d.T * d.Cout * res_h * res_wp < (1ll << 29), V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_tail_pred: residual too large");
            P.res_up = 1; P.rH = res_h; P.rWp = res_wp;
        }
        V2CE_REQUIRE(t.ksize == 1 && (t.stride_hw == 1 || t.stride_hw == 2) && t.layout == V2CE_LAYOUT_C16 && t.B == d.B && t.T == d.T &&
                     t.Cout == d.Cout && t.Hout == d.Hout && t.Wout == d.Wout && t.C0 > 0 && t.C0 % 16 == 0 && t.C1 >= 0 &&
                     t.C1 % 16 == 0 && t.Hout == (t.Hin - 1) / t.stride_hw + 1 && t.Wout == (t.Win - 1) / t.stride_hw + 1,
                     V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_tail: the tail must be a 1x1x1 conv (channels-last-16, channel counts multiples "
                     "of 16) producing exactly the main conv's output shape");
        V2CE_REQUIRE((g_name_out || tx0) && (t.C1 == 0 || tx1 || g_name_out) && (thmap == nullptr) == (twmap == nullptr) &&
                     (thmap || (t.H0 == t.Hin && t.W0 == t.Win)), V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_tail: tail inputs / index maps");
        V2CE_REQUIRE((tx0_absmax != nullptr) == (x0_absmax != nullptr) && (t.C1 == 0 || !tx0_absmax || tx1_absmax), V2CE_ERR_BAD_ARG,
                     "v2ce_conv3d_fwd_tail: range slots of the tail inputs");
        P.tx0 = tx0; P.tx1 = tx1; P.thmap = thmap; P.twmap = twmap;
        P.tC0 = t.C0; P.tH0 = t.H0; P.tW0p = t.W0_pitch > 0 ? t.W0_pitch : t.W0; P.tC1 = t.C1; P.tHin = t.Hin; P.tWin = t.Win;
        P.tWinp = t.Win_pitch > 0 ? t.Win_pitch : t.Win; P.tS = t.stride_hw; P.tCG = (t.C0 + t.C1) / 16;
        P.tx0_absmax = tx0_absmax; P.tx1_absmax = t.C1 > 0 ? tx1_absmax : nullptr;
        V2CE_REQUIRE((long long)t.T * t.C0 * t.H0 * P.tW0p < (1ll << 29) && (long long)t.T * t.C1 * t.Hin * P.tWinp < (1ll << 29),
                     V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_tail: a single sequence exceeds the 2 GiB buffer-descriptor range");
    } else if (sc_w) {
        V2CE_REQUIRE(d.precision == V2CE_PRECISION_F16X2 && d.ksize == 3 && (d.stride_hw == 2 || d.Cout <= 32) &&
                     !pred_w && sc_scale && sc_shift && sc_y, V2CE_ERR_UNSUPPORTED,
                     "v2ce_conv3d_fwd_sc: the fused shortcut needs a split-half 3x3x3 conv that is strided or has <= 32 "
                     "output channels (one 32-channel fragment row per wave)");
    }
    if (pred_w) {
        V2CE_REQUIRE(d.precision == V2CE_PRECISION_F16X2 && d.ksize == 3 && d.stride_hw == 1 && d.Cout == 32 &&
                     d.act == V2CE_ACT_RELU && pred_b && pred_y && pred_cout > 0 && pred_cout <= 32,
                     V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_pred: the fused head needs a split-half 3x3x3 stride-1 conv with "
                     "32 output channels and ReLU, and 1..32 head channels");
        V2CE_REQUIRE((long long)d.T * pred_cout * d.Hout * d.Wout < (1ll << 29), V2CE_ERR_UNSUPPORTED,
                     "v2ce_conv3d_fwd_pred: an output sequence exceeds the 2 GiB buffer-descriptor range");
    }
    hipStream_t st = as_stream(stream);

    const bool small_co = d.Cout <= 32;
    if (d.precision == V2CE_PRECISION_F16X2) {
        // split-half path: w_packed is the fp16 hi/lo buffer of v2ce_pack_weights_f16x2
        V2CE_REQUIRE(P.Cin % 16 == 0 && (d.C1 == 0 || d.C0 % 16 == 0) && d.Cout % 32 == 0,
                     V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd(f16x2): needs input channel counts that are multiples "
                     "of 16 and an output channel count that is a multiple of 32");
        V2CE_REQUIRE(P.c16, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd(f16x2): the split-half kernels take and produce activations "
                     "in the channels-last-16 layout (desc.layout = V2CE_LAYOUT_C16)");
        P.wq = reinterpret_cast<const _Float16 *>(w_packed);
        V2CE_REQUIRE(x0_absmax || !x1_absmax, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd(f16x2): x1_absmax without x0_absmax");
        V2CE_REQUIRE(d.C1 == 0 || !x0_absmax || x1_absmax, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd(f16x2): x0_absmax without x1_absmax");
        if (d.ksize == 1) {
            // 1x1x1 (shortcuts): one tap per chunk, so the producers set the pace; still ~2x the exact-f32
            // kernel on >= 128 output channels and on the strided ones
            if (s == 1) {
                if (small_co) return launch_f16x2_ws<1, 1, 1, 1, 4, 3>(P, d, st);
                if (d.Cout >= 128) return launch_f16x2_ws<1, 1, 2, 2, 4, 3>(P, d, st);
                return launch_f16x2_ws<1, 1, 1, 2, 4, 3>(P, d, st);
            }
            if (small_co) return launch_f16x2_ws<1, 2, 1, 1, 4, 3>(P, d, st);
            if (d.Cout >= 128) return launch_f16x2_ws<1, 2, 2, 2, 4, 3>(P, d, st);
            return launch_f16x2_ws<1, 2, 1, 2, 4, 3>(P, d, st);
        }
        if (s == 1) {
            // measured (tools/conv_bench.py, TF-equivalent): 128 channels x 256 positions per workgroup
            // 390-450; 64 x 512: 370-430 (64 x 256: 248-358); 32 x 512: 300-350 (32 x 256: 114-205)
            // no residual: a compile-time property (conv_epilogue RES 0: +2-6 %); with a residual the pipelined
            // form (RES 1) pays on the 32-channel tiles only (+3 %; the two-fragment-row tiles lose 6-12 % to its
            // registers), the others keep the run-time form
#define V2CE_WS_RES(R_, ...) (P.res ? launch_f16x2_ws<__VA_ARGS__, R_>(P, d, st) : launch_f16x2_ws<__VA_ARGS__, 0>(P, d, st))
            // (one 32-channel fragment row per wave: 12 MFMAs per tap -- a ring of nine taps covers the weight loads' L2 latency
            // where three do not; V2CE_NA9=0: the three-slot ring)
            static const bool na9 = [] { const char *e = getenv("V2CE_NA9"); return !(e && e[0] == '0'); }();
            if (small_co && P.pred_w && tail)              // the fused head behind a conv with a folded tail (round 6)
                return P.res ? launch_f16x2_ws<3, 1, 1, 1, 4, 9, 4, 1>(P, d, st) : launch_f16x2_ws<3, 1, 1, 1, 4, 9, 4, 0>(P, d, st);
            // (round 6: its epilogue on the producer waves; V2CE_PEPI=0: the consumers' own)
            static const bool pepi = [] { const char *e = getenv("V2CE_PEPI"); return !(e && e[0] == '0'); }();
            // (nine ring slots: the consumers' path has the registers -- the 229 of the kernel are the producers' -- 0.94 -> 0.91 ms)
            static const bool g4s = [] { const char *e = getenv("V2CE_G4"); return e && e[0] == '1'; }();
            if (small_co && P.pred_w && pepi && na9 && g4s)
                return P.res ? launch_f16x2_ws<3, 1, 1, 1, 4, 9, 1, 1, 3>(P, d, st) : launch_f16x2_ws<3, 1, 1, 1, 4, 9, 1, 0, 3>(P, d, st);
            if (small_co && P.pred_w && pepi && na9)
                return P.res ? launch_f16x2_ws<3, 1, 1, 1, 4, 9, 1, 1, 1>(P, d, st) : launch_f16x2_ws<3, 1, 1, 1, 4, 9, 1, 0, 1>(P, d, st);
            if (small_co && P.pred_w && pepi)
                return P.res ? launch_f16x2_ws<3, 1, 1, 1, 4, 3, 1, 1, 1>(P, d, st) : launch_f16x2_ws<3, 1, 1, 1, 4, 3, 1, 0, 1>(P, d, st);
            if (small_co && P.pred_w && na9) return V2CE_WS_RES(1, 3, 1, 1, 1, 4, 9, 1);
            if (small_co && P.pred_w) return V2CE_WS_RES(1, 3, 1, 1, 1, 4, 3, 1);
            if (small_co && P.sc_w) return launch_f16x2_ws<3, 1, 1, 1, 4, 3, 2, 0>(P, d, st);
            if (small_co) return V2CE_WS_RES(1, 3, 1, 1, 1, 4, 3, 0);
            if (tail) {                                   // folded shortcut: no residual by construction
                if (d.Cout >= 128) {
                    auto cost = [&](int pos_tile) {
                        const Tile t = choose_tile(d.T, d.Hout, d.Wout, 3, 1, pos_tile, 1280);
                        const long long nsp = (long long)d.B * ((d.T + t.tt - 1) / t.tt) * ((d.Hout + t.th - 1) / t.th) *
                                              ((d.Wout + t.tw - 1) / t.tw);
                        const long long blocks = 8 * ((nsp + 7) / 8) * ((d.Cout + 127) / 128);
                        return ((blocks + 255) / 256) * pos_tile;
                    };
                    const bool po3 = d.tile_t > 0 ? d.tile_t * d.tile_h * d.tile_w <= 192 : cost(192) < cost(256);
                    if (po3) return launch_f16x2_ws<3, 1, 2, 2, 3, 3, 3, 0>(P, d, st);
                    return launch_f16x2_ws<3, 1, 2, 2, 4, 3, 3, 0>(P, d, st);
                }
                return launch_f16x2_ws<3, 1, 1, 2, 4, 3, 3, 0>(P, d, st);
            }
            if (d.Cout >= 128) {
                // 256- or 192-position boxes (4 or 3 position fragments per wave): whichever needs fewer
                // (workgroup rounds x box size).  17x22 planes: (16,2,8) boxes use 87 % of the MFMA lanes and
                // 448 virtual blocks are 1.75 rounds over 256 CUs; (1,17,11) boxes use 97 % and make exactly 2
                // rounds of 3/4 the size: 25 % less MFMA time on the five 512-channel launches
                auto cost = [&](int pos_tile) {
                    const Tile t = choose_tile(d.T, d.Hout, d.Wout, 3, 1, pos_tile, 1280);
                    const long long nsp = (long long)d.B * ((d.T + t.tt - 1) / t.tt) * ((d.Hout + t.th - 1) / t.th) *
                                          ((d.Wout + t.tw - 1) / t.tw);
                    const long long blocks = 8 * ((nsp + 7) / 8) * ((d.Cout + 127) / 128);
                    return ((blocks + 255) / 256) * pos_tile;
                };
                const bool po3 = d.tile_t > 0 ? d.tile_t * d.tile_h * d.tile_w <= 192 : cost(192) < cost(256);
                if (po3) return V2CE_WS_RES(2, 3, 1, 2, 2, 3, 3, 0);
                return V2CE_WS_RES(2, 3, 1, 2, 2, 4, 3, 0);
            }
            return V2CE_WS_RES(2, 3, 1, 1, 2, 4, 3, 0);
#undef V2CE_WS_RES
        }
        // stride 2: the halo box is ~4x the output box, so 128-position boxes; one 32-channel fragment
        // row per wave measured best (Cout >= 128: 300-320; Cout = 64: 245)
        if (P.sc_w) {
            static const bool na9 = [] { const char *e = getenv("V2CE_NA9"); return !(e && e[0] == '0'); }();
            // (128-channel tiles: both accumulator sets of four fragments and a nine-slot ring do not fit -- 173 spilled registers)
            // (round 6: the epilogue shared with the producer waves as on the conv with the fused head -- measured SLOWER here, opt-in
            // with V2CE_PEPI_SC=1: the three 128-channel launches 0.495 -> 0.503 ms, enc0.conv1 0.670 -> 0.713)
            static const bool pepi_sc = [] { const char *e = getenv("V2CE_PEPI_SC"); return e && e[0] == '1'; }();
            // (round 6: four lanes per element in the producers' gather, VERDICT r5 #3 (a) -- measured no faster, opt-in with V2CE_G4=1:
            // the three 128-channel launches 0.482 -> 0.481 ms, enc0.conv1 0.717 -> 0.745, dec3.conv2 + head 0.88 -> 0.89)
            static const bool g4 = [] { const char *e = getenv("V2CE_G4"); return e && e[0] == '1'; }();
            if (g4 && d.Cout >= 128) return launch_f16x2_ws<3, 2, 4, 1, 4, 3, 2, 0, 2>(P, d, st);
            if (g4 && !small_co && na9) return launch_f16x2_ws<3, 2, 2, 1, 2, 9, 2, 0, 2>(P, d, st);
            if (pepi_sc && d.Cout >= 128 && d.Cout <= 512 && !P.res) return launch_f16x2_ws<3, 2, 4, 1, 4, 3, 2, 0, 1>(P, d, st);
            if (pepi_sc && !small_co && d.Cout < 128 && na9 && !P.res) return launch_f16x2_ws<3, 2, 2, 1, 2, 9, 2, 0, 1>(P, d, st);
            if (d.Cout >= 128) return launch_f16x2_ws<3, 2, 4, 1, 4, 3, 2, 0>(P, d, st);
            if (!small_co && na9) return launch_f16x2_ws<3, 2, 2, 1, 2, 9, 2, 0>(P, d, st);
            if (!small_co) return launch_f16x2_ws<3, 2, 2, 1, 2, 3, 2, 0>(P, d, st);
            return launch_f16x2_ws<3, 2, 1, 1, 1, 3, 2, 0>(P, d, st);
        }
        if (d.Cout >= 128) return launch_f16x2_ws<3, 2, 4, 1, 4, 3>(P, d, st);
        if (!small_co) return launch_f16x2_ws<3, 2, 2, 1, 2, 3>(P, d, st);
        return launch_f16x2_ws<3, 2, 1, 1, 1, 3>(P, d, st);
    }
    V2CE_REQUIRE(d.precision == V2CE_PRECISION_F32, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd: precision %d", d.precision);
    if (d.ksize == 3 && s == 1 && d.C0 == 2 && d.C1 == 0 && d.Cout == 32 && !hmap && !wmap && d.tile_t <= 0 &&
        (long long)d.T * d.Cout * d.Hout * d.Wout < (1ll << 29)) {
        // the UNet's head: dedicated one-pass kernel (conv3d_head_kernel)
        if (g_name_out) {
            snprintf(g_name_out, g_name_cap, "conv3d_head_kernel");
            return V2CE_OK;
        }
        P.nT = (d.T + kHeadTT - 1) / kHeadTT; P.nH = (d.Hout + kHeadTH - 1) / kHeadTH; P.nW = (d.Wout + kHeadTW - 1) / kHeadTW;
        P.n_spatial = d.B * P.nT * P.nH * P.nW;
        P.per_xcd = (P.n_spatial + 7) / 8;
        const long long blocks = 8ll * P.per_xcd;
        V2CE_REQUIRE((long long)d.B * P.nT * P.nH * P.nW < (1ll << 30), V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd(head): too many tiles");
        hipLaunchKernelGGL(conv3d_head_kernel, dim3((unsigned)blocks), dim3(256), 0, st, P);
        V2CE_HIP_CHECK(hipGetLastError());
        return V2CE_OK;
    }
    V2CE_REQUIRE(!P.c16, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd(f32): the exact-f32 kernels take planar activations "
                 "(desc.layout = V2CE_LAYOUT_PLANAR); only the 2-channel head convolution can write the channels-last-16 layout");
    // CK per (ksize, stride): sized so 2 workgroups share a CU's 160 KiB of LDS
    // fewer positions per workgroup when the launch would otherwise leave CUs idle
    // (decided per SEQUENCE, as for a batch of four: the choice sets CK and with it the summation order, which
    // must not depend on how many sequences share the launch -- a sequence's result is the same in any batch)
    const long long pos_total = 4ll * d.T * d.Hout * d.Wout;
    const int co_tiles = small_co ? 1 : (d.Cout + 63) / 64;
    const bool small_pos = (pos_total / 512) * co_tiles < 512;
#define V2CE_CK_OK(CK) V2CE_REQUIRE(d.C1 == 0 || d.C0 % (CK) == 0, V2CE_ERR_UNSUPPORTED, \
                                    "v2ce_conv3d_fwd: C0=%d not a multiple of the channel chunk %d", d.C0, (CK))
    // <KS, S, CO_FR, PO_FR, CK, EPT, MW, IL>; the (CK, MW, IL) choices are A/B-measured
    if (d.ksize == 3 && s == 1) {
        if (small_pos) {   // 256-position boxes: 80 KiB of LDS, two workgroups per CU as is
            V2CE_CK_OK(4);
            if (small_co) return launch<3, 1, 1, 2, 4, 8, 1, 0>(P, d, st);
            // 384-position boxes waste fewer MFMA lanes on 17x22 planes (16x1x22 = 352 of 384)
            if (tile_efficiency(d.T, d.Hout, d.Wout, 3, 1, 384, 2048) >
                tile_efficiency(d.T, d.Hout, d.Wout, 3, 1, 256, 2048)) {
                V2CE_CK_OK(2);
                return launch<3, 1, 2, 3, 2, 8, 1, 0>(P, d, st);
            }
            return launch<3, 1, 2, 2, 4, 8, 1, 0>(P, d, st);
        }
        // large launches: CK = 2 keeps two workgroups per CU.  384-position boxes (6 MFMAs per
        // k-step, accumulators in AGPRs, no DMA interleave) measured 0-6 % faster than 512-position
        // ones at equal tile efficiency, so they win unless they waste clearly more MFMA lanes.
        V2CE_CK_OK(2);
        const bool po3 = tile_efficiency(d.T, d.Hout, d.Wout, 3, 1, 384, 2048) + 0.01 >=
                         tile_efficiency(d.T, d.Hout, d.Wout, 3, 1, 512, 2048);
        if (small_co) {
            if (po3) return launch<3, 1, 1, 3, 2, 8, 1, 0>(P, d, st);
            return launch<3, 1, 1, 4, 2, 8, 2, 1>(P, d, st);
        }
        if (po3) return launch<3, 1, 2, 3, 2, 8, 1, 0>(P, d, st);
        return launch<3, 1, 2, 4, 2, 8, 2, 1>(P, d, st);
    }
#define V2CE_DISPATCH(KS, S, CK, EPT)                                                    \
    do {                                                                                 \
        V2CE_CK_OK(CK);                                                                  \
        if (small_co) {                                                                  \
            if (small_pos) return launch<KS, S, 1, 2, CK, EPT, 1, 0>(P, d, st);          \
            return launch<KS, S, 1, 4, CK, EPT, 1, 0>(P, d, st);                         \
        }                                                                                \
        if (small_pos) return launch<KS, S, 2, 2, CK, EPT, 1, 0>(P, d, st);              \
        return launch<KS, S, 2, 4, CK, EPT, 1, 0>(P, d, st);                             \
    } while (0)
    if (d.ksize == 3 && s == 2 && !small_co) {
        // stride 2: the halo box is ~4x the output box, so small boxes keep two workgroups per CU;
        // measured: 256-position boxes 8-10 % faster than 512, 384 better only on 17x22 planes
        V2CE_CK_OK(2);
        if (tile_efficiency(d.T, d.Hout, d.Wout, 3, 2, 384, 3584) >
            tile_efficiency(d.T, d.Hout, d.Wout, 3, 2, 256, 3584) + 0.02)
            return launch<3, 2, 2, 3, 2, 14, 1, 0>(P, d, st);
        return launch<3, 2, 2, 2, 2, 14, 1, 0>(P, d, st);
    }
    if (d.ksize == 3 && s == 2) V2CE_DISPATCH(3, 2, 2, 14);
    if (d.ksize == 1 && s == 1) {
        // 1x1x1: no tap reuse, so the LDS-DMA issue rate (one 256-byte piece per 4 MFMAs) and HBM
        // bound these; measured best: DMA issue dealt over the k-steps, small boxes for Cout >= 64
        if (small_co) { V2CE_CK_OK(8); return launch<1, 1, 1, 4, 8, 2, 2, 1>(P, d, st); }
        // Cout >= 64: chunks of 8 channels and two workgroups per CU measured best (0.51 vs 0.60 ms on the
        // 192 -> 64 shortcut; 16 / 32 channels per chunk with one workgroup 0.59 / 0.52)
        V2CE_CK_OK(8);
        return launch<1, 1, 2, 2, 8, 2, 2, 1>(P, d, st);
    }
    V2CE_DISPATCH(1, 2, 8, 8);
    return V2CE_ERR_UNSUPPORTED;
#undef V2CE_DISPATCH
#undef V2CE_CK_OK
}

extern "C" int v2ce_conv3d_fwd(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                               const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                               const float *scale, const float *shift, const float *residual,
                               float *y, const float *x0_absmax, const float *x1_absmax,
                               float *y_absmax, v2ce_stream_t stream) {
    g_name_out = nullptr;
    return conv3d_dispatch(desc, x0, x1, hmap, wmap, w_packed, scale, shift, residual, y, x0_absmax,
                           x1_absmax, y_absmax, stream);
}

extern "C" int v2ce_conv3d_fwd_pred(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                                    const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                                    const float *scale, const float *shift, const float *residual,
                                    float *y, const float *x0_absmax, const float *x1_absmax,
                                    float *y_absmax, const void *pred_w, const float *pred_b, int pred_cout,
                                    float *pred_y, v2ce_stream_t stream) {
    g_name_out = nullptr;
    clear_error();
    V2CE_REQUIRE(pred_w, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_pred: null head weights");
    return conv3d_dispatch(desc, x0, x1, hmap, wmap, w_packed, scale, shift, residual, y, x0_absmax,
                           x1_absmax, y_absmax, stream, pred_w, pred_b, pred_cout, pred_y);
}

extern "C" int v2ce_conv3d_fwd_sc(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                                  const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                                  const float *scale, const float *shift, float *y,
                                  const float *x0_absmax, const float *x1_absmax, float *y_absmax,
                                  const void *sc_w, const float *sc_scale, const float *sc_shift, float *sc_y,
                                  v2ce_stream_t stream) {
    g_name_out = nullptr;
    clear_error();
    V2CE_REQUIRE(sc_w, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_sc: null shortcut weights");
    return conv3d_dispatch(desc, x0, x1, hmap, wmap, w_packed, scale, shift, nullptr, y, x0_absmax,
                           x1_absmax, y_absmax, stream, nullptr, nullptr, 0, nullptr, sc_w, sc_scale, sc_shift, sc_y);
}

extern "C" int v2ce_conv3d_fwd_tail(const v2ce_conv3d_desc *desc, const float *x0, const float *x1,
                                    const int32_t *hmap, const int32_t *wmap, const float *w_packed,
                                    const float *scale, const float *shift, float *y,
                                    const float *x0_absmax, const float *x1_absmax, float *y_absmax,
                                    const v2ce_conv3d_desc *tail_desc, const float *tx0, const float *tx1,
                                    const int32_t *thmap, const int32_t *twmap, const void *tail_w,
                                    const float *tx0_absmax, const float *tx1_absmax, v2ce_stream_t stream) {
    g_name_out = nullptr;
    clear_error();
    V2CE_REQUIRE(tail_desc && tail_w, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_tail: null tail description / weights");
    return conv3d_dispatch(desc, x0, x1, hmap, wmap, w_packed, scale, shift, nullptr, y, x0_absmax,
                           x1_absmax, y_absmax, stream, nullptr, nullptr, 0, nullptr, tail_w, nullptr, nullptr, nullptr,
                           tail_desc, tx0, tx1, thmap, twmap, tx0_absmax, tx1_absmax);
}

extern "C" int v2ce_conv3d_fwd_tail_pred(const v2ce_conv3d_desc *desc, const float *x0, const float *w_packed, const float *scale,
                                         const float *shift, float *y, const float *x0_absmax, float *y_absmax, const void *pred_w,
                                         const float *pred_b, int pred_cout, float *pred_y, const v2ce_conv3d_desc *tail_desc,
                                         const float *tx0, const float *tx1, const int32_t *thmap, const int32_t *twmap,
                                         const void *tail_w, const float *tx0_absmax, const float *tx1_absmax, const float *residual,
                                         int res_h, int res_w_pitch, v2ce_stream_t stream) {
    g_name_out = nullptr;
    clear_error();
    V2CE_REQUIRE(pred_w && tail_desc && tail_w, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_tail_pred: null head weights / tail description / tail weights");
    V2CE_REQUIRE(!residual || res_h > 0, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_tail_pred: the residual is the low-resolution one (res_h > 0)");
    return conv3d_dispatch(desc, x0, nullptr, nullptr, nullptr, w_packed, scale, shift, residual, y, x0_absmax, nullptr, y_absmax, stream,
                           pred_w, pred_b, pred_cout, pred_y, tail_w, nullptr, nullptr, nullptr, tail_desc, tx0, tx1, thmap, twmap,
                           tx0_absmax, tx1_absmax, res_h, res_w_pitch);
}

extern "C" size_t v2ce_pack_pred_weights_f16x2_bytes(void) { return 2048 * 2 + 16; }

extern "C" int v2ce_pack_pred_weights_f16x2(const float *w, int cout, int cin, void *table, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(w && table && cout > 0 && cout <= 32 && cin == 32, V2CE_ERR_BAD_ARG,
                 "v2ce_pack_pred_weights_f16x2: needs [cout <= 32][32] weights");
    hipLaunchKernelGGL(pack_pred_weights_kernel, dim3(1), dim3(256), 0, as_stream(stream), w, cout,
                       static_cast<_Float16 *>(table));
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" int v2ce_conv3d_variant_fused(const v2ce_conv3d_desc *desc, int mapped, int fuse_in, char *name, size_t cap) {
    const bool tail_pred = (fuse_in & 8) != 0;              // the tail behind the conv with the fused head (v2ce_conv3d_fwd_tail_pred)
    const int fuse = tail_pred ? 3 : fuse_in & 3;
    const bool with_res = (fuse_in & 4) != 0;
    V2CE_REQUIRE(name && cap > 0, V2CE_ERR_BAD_ARG, "v2ce_conv3d_variant: no buffer");
    name[0] = '\0';
    g_name_out = name;
    g_name_cap = cap;
    static const int32_t dummy_map = 0;
    static const float dummy = 0.0f;
    const int32_t *m = mapped ? &dummy_map : nullptr;
    v2ce_conv3d_desc td{};
    if (fuse == 3 && desc) {                                 // a stand-in tail of the right output shape
        td = *desc;
        td.ksize = 1; td.stride_hw = 1; td.C0 = 16; td.C1 = 0; td.H0 = td.Hin = desc->Hout; td.W0 = td.Win = desc->Wout;
        td.W0_pitch = td.Win_pitch = 0;
    }
    const int rc = conv3d_dispatch(desc, nullptr, nullptr, m, m, nullptr, nullptr, nullptr, with_res ? &dummy : nullptr,
                                   nullptr, nullptr, nullptr, nullptr, nullptr,
                                   (fuse == 1 || tail_pred) ? &dummy : nullptr, (fuse == 1 || tail_pred) ? &dummy : nullptr,
                                   (fuse == 1 || tail_pred) ? 1 : 0, (fuse == 1 || tail_pred) ? const_cast<float *>(&dummy) : nullptr,
                                   fuse >= 2 ? &dummy : nullptr, fuse == 2 ? &dummy : nullptr,
                                   fuse == 2 ? &dummy : nullptr, fuse == 2 ? const_cast<float *>(&dummy) : nullptr,
                                   fuse == 3 ? &td : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   (tail_pred && with_res && desc) ? (desc->Hout + 1) / 2 : 0, (tail_pred && with_res && desc) ? (desc->Wout + 1) / 2 : 0);
    g_name_out = nullptr;
    return rc;
}

extern "C" int v2ce_conv3d_variant(const v2ce_conv3d_desc *desc, int mapped, char *name, size_t cap) {
    return v2ce_conv3d_variant_fused(desc, mapped, 0, name, cap);
}

extern "C" size_t v2ce_pack_weights_f16x2_bytes(int Cout, int Cin, int k3) {
    return (size_t)Cout * Cin * k3 * 4 + 2 * sizeof(float);
}

// the two passes of v2ce_pack_weights_f16x2, for callers that put their own maxima into tail[0] between them
// (conv3d_up.hip: the folded sums of the decoder weights); the caller zeroes the tail first
int v2ce::v2ce_pack_weights_f16x2_absmax_only(const float *w, int Cout, int Cin, int k3, const float *sigma, void *w_f16x2,
                                              v2ce_stream_t stream) {
    const long long n = (long long)Cout * Cin * k3;
    float *tail = reinterpret_cast<float *>(static_cast<_Float16 *>(w_f16x2) + 2 * n);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(weights_absmax_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, as_stream(stream), w, n,
                       sigma, tail);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
int v2ce::v2ce_pack_weights_f16x2_pack_only(const float *w, int Cout, int Cin, int k3, const float *sigma, void *w_f16x2,
                                            v2ce_stream_t stream) {
    const long long n = (long long)Cout * Cin * k3;
    hipLaunchKernelGGL(pack_weights_f16x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w, Cout, Cin, k3, sigma, static_cast<_Float16 *>(w_f16x2));
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" int v2ce_pack_weights_f16x2(const float *w, int Cout, int Cin, int k3, const float *sigma,
                                       void *w_f16x2, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(w && w_f16x2 && Cout > 0 && Cin > 0 && Cin % 16 == 0 && (k3 == 27 || k3 == 1), V2CE_ERR_BAD_ARG,
                 "v2ce_pack_weights_f16x2: needs a 3x3x3 or 1x1x1 kernel and Cin %% 16 == 0");
    const long long n = (long long)Cout * Cin * k3;
    float *tail = reinterpret_cast<float *>(static_cast<_Float16 *>(w_f16x2) + 2 * n);
    V2CE_HIP_CHECK(hipMemsetAsync(tail, 0, 2 * sizeof(float), as_stream(stream)));
    int rc = v2ce_pack_weights_f16x2_absmax_only(w, Cout, Cin, k3, sigma, w_f16x2, stream);
    if (rc != V2CE_OK) return rc;
    return v2ce_pack_weights_f16x2_pack_only(w, Cout, Cin, k3, sigma, w_f16x2, stream);
}

extern "C" int v2ce_pack_weights(const float *w, int Cout, int Cin, int k3, const float *sigma,
                                 float *w_packed, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(w && w_packed && Cout > 0 && Cin > 0 && (k3 == 1 || k3 == 27), V2CE_ERR_BAD_ARG,
                 "v2ce_pack_weights: bad argument");
    const long long n = (long long)Cout * Cin * k3;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w, Cout, Cin, k3, sigma, w_packed);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
