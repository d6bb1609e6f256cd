// conv3d_dev.h -- device-side pieces shared by the conv translation units (conv3d.hip: generic kernels;
// conv3d_up.hip: the phase-folded decoder kernel): launch parameters, the fused epilogues, range tracking.
#pragma once
#include "common.h"

#include <type_traits>

namespace v2ce {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));




struct ConvParams {
    const float *x0, *x1;
    const int *hmap, *wmap;
    const float *wp, *scale, *shift, *res;
    float *y;
    int B, T, C0, H0, W0, C1, Hin, Win, Cin, Cout, Hout, Wout;
    int W0p, Winp, Woutp;         // row pitches (floats) of x0, x1 and of y / residual / sc_y (>= the logical widths)
    int c16;                      // activation layout of x0, x1, y, residual, sc_y: 0 = planar [B][T][C][H][Wp],
                                  // 1 = channel groups of 16 innermost [B][T][C/16][H][Wp][16] (V2CE_LAYOUT_C16)
    int act;
    int TT, TH, TW;       // output box
    int nT, nH, nW;       // boxes per dimension
    int HT, HH, HWd;      // halo box
    int plane;            // HT*HH*HWd  (LDS stride between channels)
    int res_up, rH, rWp;  // conv3d_wt.hip: the residual is a 2x nearest-upsampled LOW-resolution tensor [B][T][Cout/16][rH][rWp][16], read at
                          // (h >> 1, w >> 1) (the upsampled source's share of a decoder's folded shortcut, computed where it is small)
    int flat;             // > 0 (conv3d_wt.hip): a tile's positions are a RANGE of `flat` consecutive positions of the row-major
                          // Hout x Wout plane (per time block) instead of a TH x TW rectangle -- planes like 17 x 22 split into three
                          // ranges of 125 where rectangles of <= 128 positions need four; nH = 1, nW = ranges per plane, TW = Wout,
                          // the halo box = the range's rows (+-1) x (Wout + 2)
    int n_co_tiles;
    int n_pos;            // TT*TH*TW
    int n_spatial;        // B*nT*nH*nW
    int xcd_remap;        // 1: blocks that share an input box (different co tiles) share an XCD/L2
    int total_blocks;     // persistent kernels: number of virtual blocks to walk
    int per_xcd;          // persistent kernels: spatial boxes per XCD
    // split-half path (conv3d_f16x2_ws_kernel): weights as fp16 hi/lo planes [2][K3][Cin/16][Cout][16]
    // followed by { max |w|, power-of-two pre-scale } as two floats (pack_weights_f16x2_kernel)
    const _Float16 *wq;
    // dynamic range tracking: y_absmax[0] (may be null) receives max |y| of this launch by atomic max,
    // y_absmax[1] the split-half kernel's range-guard bound (see conv3d_f16x2_ws_kernel);
    // x0_absmax / x1_absmax (may be null) are the slots the producers of x0 / x1 wrote -- the
    // split-half kernel derives its power-of-two activation pre-scale from them
    const float *x0_absmax, *x1_absmax;
    float *y_absmax;
    float *guard;                 // = y_absmax + 1
    int amax_bs;                  // floats between the slots of consecutive batch elements (desc.absmax_batch_stride);
                                  // 0 = one slot per tensor.  With a stride every sequence of the batch carries its own
                                  // range and pre-scale, so its result does not depend on what else is in the batch
#ifdef V2CE_ABLATE_EPI
    int ablate;                   // diagnostic build: skip the epilogue of this launch (tools/epi_ablate.sh)
#endif
    // fused 1x1x1 head behind a 32-channel conv (v2ce_conv3d_fwd_pred): pred_y[o] = relu(pred_w[o][:] . y + pred_b[o])
    const _Float16 *pred_w;       // table of v2ce_pack_pred_weights_f16x2 (A fragments hi/lo + pre-scale)
    const float *pred_b;          // [32], zero padded
    float *pred_y;                // [B][T][pred_cout][Hout][Wout]
    int pred_cout;
    // fused 1x1x1 shortcut of a residual block (v2ce_conv3d_fwd_sc): the centre tap of the 3x3x3 conv reads exactly
    // the (strided) positions the shortcut conv reads, so it rides along in a second accumulator set
    const _Float16 *sc_w;         // v2ce_pack_weights_f16x2 buffer of the [Cout][Cin][1] shortcut weights
    const float *sc_scale, *sc_shift;
    float *sc_y;                  // [B][T][Cout][Hout][Wout]
    // folded 1x1x1 tail (v2ce_conv3d_fwd_tail, FUSE 3): tCG more 16-channel K chunks behind the 3x3x3 conv's own, one
    // tap each, gathered at the OUTPUT positions (stride tS) from a second virtual input tx0 (++ tx1); weights = sc_w
    const float *tx0, *tx1;
    const int *thmap, *twmap;
    int tC0, tH0, tW0p, tC1, tHin, tWin, tWinp, tS, tCG;
    int tCHS;                     // pieces per plane of the LDS buffers of a tail launch (>= the halo plane, >= tTCH * tNPP)
    int tNPP, tTCH, tSC0, tSC;    // a tail SUPER-chunk = up to tTCH channel groups of ONE source staged per barrier, group g at
                                  // pieces [g * tNPP, + n_pos) (tNPP = n_pos rounded up to 64); tSC0 / tSC super-chunks of tx0 / in all
    const float *tx0_absmax, *tx1_absmax;
#ifdef V2CE_STAMP
    unsigned long long *stamps;   // diagnostic build only: [block][role][8] s_memtime stamps
#endif
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
[[maybe_unused]] constexpr float kActScale = 16.0f;      // pre-scale when the caller tracks no range: |x| < 4094 required

template <int KS, int S, int CO_FR, int PO_FR, int CK, int EPT>
struct ConvCfg {
    static constexpr int K3 = KS * KS * KS;
    static constexpr int CO_TILE = CO_FR * 32;
    static constexpr int POS_TILE = 4 * PO_FR * 32;
    static constexpr int MAX_PLANE = 256 * EPT;
};

#if defined(__HIP_DEVICE_COMPILE__)   // buffer-descriptor types exist only in the device pass
constexpr unsigned kOOB = 0x80000000u;   // buffer voffset that is always out of range => load returns 0

typedef __attribute__((address_space(3))) void *lds_ptr_t;
#ifdef V2CE_STAMP
#define TICK() __builtin_amdgcn_s_memtime()
#define ACC_T(var_, t0_) var_ += TICK() - (t0_)
#else
#define TICK() 0ull
#define ACC_T(var_, t0_) do {} while (0)
#endif
#ifdef V2CE_STAMP
#define STAMP(role_, k_) do { if (lane == 0 && (wave == 0 || wave == 4)) P.stamps[((long long)blockIdx.x * 2 + (role_)) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(role_, k_) do {} while (0)
#endif

// workgroup barrier that waits only for this wave's LDS traffic (__syncthreads also drains the vector-memory counter: an
// epilogue's stores would be waited for at every rendezvous)
[[maybe_unused]] __device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave-level max of m (>= 0), then one atomic max per wave on the float's bit pattern
__device__ __forceinline__ void absmax_commit(float m, float *slot) {
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // the running maximum settles after the first few workgroups: test before paying for the atomic
    if ((threadIdx.x & 63) == 0 &&
        __float_as_uint(m) > __atomic_load_n(reinterpret_cast<unsigned *>(slot), __ATOMIC_RELAXED))
        atomicMax(reinterpret_cast<unsigned *>(slot), __float_as_uint(m));
}

// Activation without control flow: act(t) = max(t, slope * t) + 0 with slope 1 (none), 0 (ReLU) or
// 0.01 (LeakyReLU); the "+ 0" turns the -0 that 0 * t leaves for negative t into the +0 of max(t, 0).
__device__ __forceinline__ float act_slope(int act) {
    return act == V2CE_ACT_RELU ? 0.0f : (act == V2CE_ACT_LEAKY ? 0.01f : 1.0f);
}
__device__ __forceinline__ float apply_act(float t, float slope) { return fmaxf(t, slope * t) + 0.0f; }

// y = act(acc * scale + shift (+ residual)), max |y| tracking.  Accumulator register r of fragment
// row q is channel co0 + 32 q + (r & 3) + 8 (r >> 2) + 4 half, lane l32 of fragment column f is the
// position poff[f] (< 0: outside the tensor).  Straight-line code (a per-element branch costs more
// than the store it guards: the first version of this epilogue spent 200 cycles per output in
// scalar branches): loads and stores are buffer operations whose per-lane offset is pushed out of
// range for masked lanes (the hardware range check returns 0 / drops the store), the channel offset
// rides in the scalar offset.  Requires Cout % 32 == 0, sequences < 2 GiB, y not aliasing the inputs.
// RES: 0 = no residual, 1 = residual (both decided at compile time), 2 = P.res checked at run time.
// The vector-memory counter of gfx9 is shared by loads and stores and drains in issue order, so a wait for a
// load also waits for every store issued before it.  The run-time form (a uniform branch around each batch of
// loads, then s_waitcnt vmcnt(0)) therefore pays the write-acknowledge latency of the previous batch's stores
// 4 * CO_FR times per tile even when there is no residual (in-kernel stamps: 25 k -> 16 k cycles on
// dec3.conv1).  RES 0 has no wait inside the loop; RES 1 issues the loads of batch i+1 in front of the stores
// of batch i (and all scale / shift loads in front of everything), so a wait covers only stores that are two
// batches old.
// C16: y / residual in the channels-last-16 layout; poff[f] is then the BYTE offset of the position's 64-byte
// group inside channel group 0 of its time step (or < 0), and the four channels (r & 3) a lane holds per r >> 2
// are 16 contiguous bytes: one 16-byte store / residual load per (r >> 2, fragment) instead of four dwords.
template <int CO_FR, int PO_FR, bool CHECK_CO = false, bool KEEP = false, int RES = 2, bool C16 = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams &P, f32x16 (&acc)[CO_FR][PO_FR],
                                              const int (&poff)[PO_FR], int co0, int half, int b,
                                              float inv_scale, const int (*rpoff)[PO_FR] = nullptr) {
    // rpoff (C16, a residual): the residual is a 2x nearest-upsampled LOW-resolution tensor [B][T][Cout/16][rH][rWp][16]
    // (ConvParams::res_up): (*rpoff)[f] = byte offset of the position's group at (h >> 1, w >> 1) inside channel group 0 of its time
    // step there, or < 0
    typedef float f32x4q __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
    const float *__restrict__ scale = P.scale;
    const float *__restrict__ shift = P.shift;
    const long long seq = (long long)P.T * P.Cout * (P.Hout * P.Woutp);
    const int cstride4 = P.Hout * P.Woutp * 4;             // planar: bytes between channels; C16: x 16 = bytes between groups
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y ? P.y + b * seq : const_cast<float *>(P.scale), 0,
                                                                          P.y ? (int)(seq * 4) : 0, 0x00020000);
    const bool has_res = RES == 1 || (RES == 2 && P.res != nullptr);
    const bool rup = C16 && rpoff != nullptr;
    const long long seq_r = rup ? (long long)P.T * P.Cout * (P.rH * P.rWp) : seq;
    const int rstride4 = rup ? P.rH * P.rWp * 4 : cstride4;
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(has_res ? P.res + b * seq_r : P.scale), 0, has_res ? (int)(seq_r * 4) : 0, 0x00020000);
    const int cbase = co0 + 4 * half;                      // this lane's channel for (q, r) = (0, 0)
    const float slope = act_slope(P.act);
    unsigned vo[PO_FR], vmask[PO_FR];
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) {
        if (C16) vo[f] = poff[f] >= 0 ? (unsigned)(poff[f] + 16 * half) : kOOB;
        else vo[f] = poff[f] >= 0 ? (unsigned)(poff[f] * 4 + cbase * cstride4) : kOOB;
        vmask[f] = poff[f] >= 0 ? 0x7fffffffu : 0u;        // |v| of a masked lane counts as 0
    }
    unsigned vor[PO_FR];                                   // the residual's offsets: the output's, or the low-resolution tensor's
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) vor[f] = rup ? ((*rpoff)[f] >= 0 ? (unsigned)((*rpoff)[f] + 16 * half) : kOOB) : vo[f];
#ifdef V2CE_ABLATE_EPI   // diagnostic build: 2 = every store is issued but dropped by the range check, 3 = every residual load
    unsigned vo_st[PO_FR], vo_ld[PO_FR];
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) {
        vo_st[f] = P.ablate == 2 ? kOOB : vo[f];
        vo_ld[f] = P.ablate == 3 ? kOOB : vor[f];
    }
#define V2CE_VO_ST vo_st
#define V2CE_VO_LD vo_ld
#else
#define V2CE_VO_ST vo
#define V2CE_VO_LD vor
#endif
    // scalar (wave-uniform) byte offset of batch (q, r4) = channels cbase + 32 q + 8 r4 + {0..3}
    auto soff = [&](int q, int r4, int k) -> int {
        if (C16) return (co0 / 16 + 2 * q + (r4 >> 1)) * (cstride4 * 16) + 32 * (r4 & 1);
        return (q * 32 + k + 8 * r4) * cstride4;
    };
    auto load_res = [&](int step, float (&rv)[4][PO_FR]) {  // batch `step` = (q, r4): 4 channels x PO_FR positions
        const int q = step >> 2, r4 = step & 3;
        if constexpr (C16) {
            const bool cok = !CHECK_CO || co0 + q * 32 + 8 * r4 < P.Cout;
#pragma unroll
            for (int f = 0; f < PO_FR; ++f) {
                const int so = rup ? (co0 / 16 + 2 * q + (r4 >> 1)) * (rstride4 * 16) + 32 * (r4 & 1) : soff(q, r4, 0);
                const f32x4q v = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(rs_r, cok ? V2CE_VO_LD[f] : kOOB, so, 0));
                rv[0][f] = v[0]; rv[1][f] = v[1]; rv[2][f] = v[2]; rv[3][f] = v[3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    const bool cok = !CHECK_CO || cbase + q * 32 + k + 8 * r4 < P.Cout;
                    rv[k][f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rs_r, cok ? V2CE_VO_LD[f] : kOOB, soff(q, r4, k), 0));
                }
        }
    };
    float rva[4][PO_FR], rvb[4][PO_FR];
    if constexpr (RES == 1) load_res(0, rva);
    constexpr bool HOIST = RES == 1 && CO_FR == 1;
    float sc[HOIST ? CO_FR : 1][16], sh[HOIST ? CO_FR : 1][16];
    auto load_affine = [&](int q, float (&scq)[16], float (&shq)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int co = cbase + q * 32 + (r & 3) + 8 * (r >> 2);
            if (CHECK_CO) co = co < P.Cout ? co : P.Cout - 1;
            scq[r] = scale[co] * inv_scale;
            shq[r] = shift[co];
        }
    };
    if constexpr (HOIST) {
#pragma unroll
        for (int q = 0; q < CO_FR; ++q) load_affine(q, sc[q], sh[q]);
    }
    unsigned ymax = 0u;                                    // max |y| as a bit pattern (non-negative floats order as integers)
#pragma unroll
    for (int q = 0; q < CO_FR; ++q) {
        float (&scq)[16] = sc[HOIST ? q : 0];
        float (&shq)[16] = sh[HOIST ? q : 0];
        if constexpr (!HOIST) load_affine(q, scq, shq);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int step = q * 4 + r4;
            float (&rv)[4][PO_FR] = (step & 1) ? rvb : rva;
            if constexpr (RES == 1) {
                if (step + 1 < 4 * CO_FR) load_res(step + 1, (step & 1) ? rva : rvb);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int f = 0; f < PO_FR; ++f) rv[k][f] = 0.0f;
                if (RES == 2 && has_res) load_res(step, rv);            // uniform
            }
            if constexpr (C16) {
                const bool cok = !CHECK_CO || co0 + q * 32 + 8 * r4 < P.Cout;
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    f32x4q out;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = 4 * r4 + k;
                        float v = acc[q][f][r] * scq[r] + shq[r];
                        v += rv[k][f];
                        v = apply_act(v, slope);
                        out[k] = v;
                        const unsigned av = __builtin_bit_cast(unsigned, v) & (cok ? vmask[f] : 0u);
                        if (KEEP) acc[q][f][r] = __builtin_bit_cast(float, av == 0u ? 0u : __builtin_bit_cast(unsigned, v));
                        ymax = av > ymax ? av : ymax;
                    }
                    if (!KEEP || P.y) {                         // uniform: a fused head may not want y itself
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, out),
                                                               rs_y, cok ? V2CE_VO_ST[f] : kOOB, soff(q, r4, 0), 0);
                        // A 16-byte store reads its data registers for several cycles after issue; a VALU write to
                        // them in the next slot corrupts dword 1 of lanes 12-15 / 28-31 (seen on gfx950: the
                        // compiler's hazard table exempts stores with an SGPR soffset).  The data registers stay
                        // live through this statement, so nothing can overwrite them before the wait states.
                        asm volatile("s_nop 1" : "+v"(out));
                    }
                }
            } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * r4 + k;
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    float v = acc[q][f][r] * scq[r] + shq[r];
                    v += rv[k][f];
                    v = apply_act(v, slope);
                    const bool cok = !CHECK_CO || cbase + q * 32 + k + 8 * r4 < P.Cout;
                    if (!KEEP || P.y)                           // uniform: a fused head may not want y itself
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, cok ? V2CE_VO_ST[f] : kOOB,
                                                              soff(q, r4, k), 0);
                    const unsigned av = __builtin_bit_cast(unsigned, v) & (cok ? vmask[f] : 0u);
                    if (KEEP) acc[q][f][r] = __builtin_bit_cast(float, av == 0u ? 0u : __builtin_bit_cast(unsigned, v));
                    ymax = av > ymax ? av : ymax;
                }
            }
            }
        }
    }
    if (P.y_absmax) absmax_commit(__builtin_bit_cast(float, ymax), P.y_absmax + b * P.amax_bs);
#undef V2CE_VO_ST
#undef V2CE_VO_LD
}

// Fused 1x1x1 head (the UNet's `pred` layer, unet_2layer.py:374) behind a 32-channel conv: the
// wave's 32 x (PO_FR*32) block of post-activation outputs v is still in its accumulator registers
// (conv_epilogue<KEEP>), so out[o][pos] = relu(sum_c Wp[o][c] v[c][pos] + b[o]) is two more k-steps
// of the same split-half MFMA with the roles kept: lane (pos, half) already holds exactly the 8
// channels c(j) = (j & 3) + 8 (j >> 2) + 16 k + 4 half of k-step k that a B fragment needs, so no
// value moves between lanes; the host packs Wp's columns in that order (v2ce_pack_pred_weights_f16x2).
// v is pre-scaled by a power of two from the wave's own maximum.
template <int PO_FR>
__device__ __forceinline__ void pred_epilogue(const ConvParams &P, const f32x16 (&v)[1][PO_FR], int pos0,
                                              int lane, int b, int t0, int h0, int w0) {
    const int l32 = lane & 31, half = lane >> 5;
    float m = 0.0f;
#pragma unroll
    for (int f = 0; f < PO_FR; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[0][f][r]));
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float v_scale = pow2_prescale(m);
    const float w_scale = reinterpret_cast<const float *>(P.pred_w + 2048)[0];
    f16x8 ahp[2], alp[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ahp[k] = *reinterpret_cast<const f16x8 *>(P.pred_w + ((k * 2 + 0) * 32 + l32) * 16 + 8 * half);
        alp[k] = *reinterpret_cast<const f16x8 *>(P.pred_w + ((k * 2 + 1) * 32 + l32) * 16 + 8 * half);
    }
    f32x16 out[PO_FR];
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[f][r] = 0.0f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f16x8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = v[0][f][8 * k + j] * v_scale;
                const _Float16 hh = (_Float16)x;
                bh[j] = hh;
                bl[j] = (_Float16)(x - (float)hh);
            }
            out[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahp[k], bh, out[f], 0, 0, 0);
            out[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahp[k], bl, out[f], 0, 0, 0);
            out[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alp[k], bh, out[f], 0, 0, 0);
        }
    }
    const float inv = 1.0f / (v_scale * w_scale);
    const long long seq = (long long)P.T * P.pred_cout * (P.Hout * P.Wout);
    const int cstride4 = P.Hout * P.Wout * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(P.pred_y + b * seq, 0, (int)(seq * 4), 0x00020000);
    const float *__restrict__ pb = P.pred_b;
    float bias[16];                                          // batch of loads ahead of the stores
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = pb[(r & 3) + 8 * (r >> 2) + 4 * half];
    unsigned vo[PO_FR];
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) {
        const int mm = pos0 + f * 32 + l32;
        vo[f] = kOOB;
        if (mm < P.n_pos) {
            const int tt = mm / (P.TH * P.TW);
            const int rem = mm - tt * (P.TH * P.TW);
            const int th = rem / P.TW;
            const int tw = rem - th * P.TW;
            const int t = t0 + tt, h = h0 + th, w = w0 + tw;
            if (t < P.T && h < P.Hout && w < P.Wout)
                vo[f] = (unsigned)((((t * P.pred_cout) * (P.Hout * P.Wout)) + h * P.Wout + w) * 4 + 4 * half * cstride4);
        }
    }
    // straight-line stores: rows >= pred_cout are pushed out of range instead of branched around
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oq = (r & 3) + 8 * (r >> 2);               // output channel minus 4 * half
        const bool ook = oq + 4 * half < P.pred_cout;
#pragma unroll
        for (int f = 0; f < PO_FR; ++f) {
            float y = out[f][r] * inv + bias[r];
            y = fmaxf(y, 0.0f) + 0.0f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rs, ook ? vo[f] : kOOB, oq * cstride4, 0);
        }
    }
}

// The same epilogue in streaming order (constants, residual and store per element): used by the
// exact-f32 kernels, whose two-workgroups-per-CU register budgets leave no room for the batches.
template <int CO_FR, int PO_FR>
__device__ __forceinline__ void conv_epilogue_stream(const ConvParams &P, const f32x16 (&acc)[CO_FR][PO_FR],
                                                     const int (&poff)[PO_FR], int co0, int half, int b) {
    const long long ybase = (long long)b * P.T * P.Cout * (P.Hout * P.Woutp);
    const int cstride = P.Hout * P.Woutp;
    const float slope = act_slope(P.act);
    float ymax = 0.0f;
#pragma unroll
    for (int q = 0; q < CO_FR; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < P.Cout) {
                const float sc = P.scale[co], sh = P.shift[co];
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    if (poff[f] >= 0) {
                        const long long idx = ybase + poff[f] + (long long)co * cstride;
                        float v = acc[q][f][r] * sc + sh;
                        if (P.res) v += P.res[idx];
                        v = apply_act(v, slope);
                        P.y[idx] = v;
                        ymax = fmaxf(ymax, fabsf(v));
                    }
                }
            }
        }
    }
    if (P.y_absmax) absmax_commit(ymax, P.y_absmax + b * P.amax_bs);
}

#endif  // __HIP_DEVICE_COMPILE__

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void step_loop(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        step_loop<I + 1, N>(f);
    }
}

[[maybe_unused]] extern __shared__ __attribute__((aligned(16))) unsigned char conv_smem[];

}  // namespace
}  // namespace v2ce
