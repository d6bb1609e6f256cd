// conv3d_wt.hip -- 3x3x3, stride-1 convolution of a residual block (submodules.py:249-264: conv2 of every block, conv1 of the
// two middle blocks) with the Winograd transform F(2,3) along T, on the split-half fp16 MFMA arithmetic of conv3d.hip.  gfx950.
//
// Two outputs y(2p), y(2p+1) of the same pixel need, over the three time taps g0 g1 g2 and the four inputs d0..d3 = x(2p-1 .. 2p+2),
// six products per (dh, dw, ci) in the direct form and FOUR in the transformed one:
//     m0 = (d0 - d2) g0          m1 = (d1 + d2) (g0 + g1 + g2)/2       m2 = (d2 - d1) (g0 - g1 + g2)/2       m3 = (d1 - d3) g2
//     y(2p) = m0 + m1 + m2       y(2p+1) = m1 - m2 - m3
// i.e. four independent 3x3 (H, W) convolutions -- "transform slots" j = 0..3 -- over transformed inputs D_j with transformed
// weights G_j, summed over (dh, dw, ci) BEFORE the output transform.  The executed MFMA work of the layer drops by 1.5x (36 instead
// of 54 k-steps per output pair and channel chunk); the chip is power-limited on fp16 MFMA, so the executed flop is
// the lever that is left.  Numerics: transforms, products (22-bit split operands) and sums in f32 -- the network-level deviation from
// the direct f32 form is 1-2e-6 like the direct form's own distance from f64 (tools/winograd_t_sim.py), the parity bar is 1e-5.
//
// Mapping on the wave-specialised kernel (waves 0-3 consume, 4-7 produce, one barrier per 16-channel chunk):
//   * consumer wave j owns transform slot j for the WHOLE tile: CO_FR x PO_FR accumulator tiles (32 channels x 32 pair-positions
//     (p, h, w)), A fragments of (slot, tap, chunk) reused across the PO_FR position fragments exactly like the direct kernel's;
//   * a producer lane owns one (p, hh, hw) element of the halo box: four 64-byte loads (the four time steps), the input transform
//     on sixteen channels, the hi/lo split, and sixteen 16-byte pieces (4 slots x hi/lo x channel halves) into LDS -- at most 256
//     elements per box, two chunks in flight in registers;
//   * output transform across the four waves at the end of a tile, through the LDS buffer the tile's last chunk has just freed:
//     per channel fragment every wave leaves its 16 KB of accumulators in ITS OWN slot region (the only region it was still reading),
//     barrier, wave f sums the four slots of position fragment f into y(2p), y(2p+1) and runs conv_epilogue on them (scale, shift,
//     residual, activation, range tracking -- the same code as the direct kernel).  The producers take part in the 2 * CO_FR extra
//     barriers before they refill that buffer.
#include "conv3d_dev.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace v2ce {
namespace {

constexpr int kWtSlot = 256;             // element slots per transform slot in a quarter plane (>= elements of a halo box)
constexpr int kWtChs = 4 * kWtSlot;      // pieces per quarter plane (hi ch 0-7 | hi ch 8-15 | lo 0-7 | lo 8-15)
constexpr int kWtTaps = 36;              // 4 transform slots x 9 (dh, dw) taps

// TAIL: the block's folded 1x1x1 shortcut rides behind the 3x3x3 conv's chunks (v2ce_conv3d_fwd_tail's contract: P.tx0 (++ tx1) read at
// the output positions, P.sc_w = the [Cout][tC0 + tC1][1] weights, P.tCG channel groups).  In the transform domain a term z(t) = Wd x(t)
// of the OUTPUT splits over the slots as  m0 += WdA x(2p),  m3 -= WdA x(2p+1),  m1 += WdB (x(2p) + x(2p+1)) / 2,
// m2 += WdB (x(2p) - x(2p+1)) / 2  (A / B = first / second half of the tail's channel groups): y(2p) = m0 + m1 + m2 and
// y(2p+1) = m1 - m2 - m3 then gain Wd x(2p) and Wd x(2p+1), and every wave does a quarter of the tail's MFMAs.  A producer lane owns an
// output pair-position and one of the two k-steps of a barrier: four 64-byte loads (two time steps x group of half A, group of half B),
// the same register shape as a chunk of the main loop.
// A tail barrier carries only 48 MFMAs per wave (a chunk of the main loop: 216), less than the latency of the gathers behind it, so the
// tail barriers are INTERLEAVED with the main chunks (M0 T0 M1 T1 ...: a tail barrier's loads are issued two barriers = more than one
// main chunk ahead; all tail barriers behind the main loop measured 3x their MFMA time, the producers waiting for loads).  The two parts
// therefore accumulate at ONE power-of-two scale S = min(x_scale w_scale, t_scale wt_scale): the part with the larger natural scale
// gives up log2 of the ratio in headroom (0-3 bits on this network), which the range-guard bound -- computed from the scales really
// used -- accounts for.
template <int CO_FR, int PO_FR, int RES, bool TAIL = false>
__global__ __launch_bounds__(512, 1) void conv3d_wt_kernel(ConvParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
    [[maybe_unused]] unsigned long long t_all = TICK(), t_bar = 0, t_epi = 0, t_xbar = 0, t_cvt = 0, t_e0 = 0, t_e1 = 0, t_e2 = 0;
    static_assert(PO_FR == 4, "the output transform hands position fragment f to consumer wave f");
    // who runs the epilogue behind the output transform: the producers (their vector-memory counter takes the tile's 64 KB of stores, the
    // consumers go on with the next tile), or -- the tail variants, whose producers have no registers left for it (63-186 spilled
    // registers, dec2's launch 0.83 -> 0.88 ms) -- the consumers themselves
    constexpr bool PEPI = !TAIL;
    // (TAIL with RES: a decoder's folded shortcut split by source -- the skip channels ride as the tail, the upsampled channels'
    // share arrives as a low-resolution residual, ConvParams::res_up)
    constexpr int CK = 16, NA = 3, CO_TILE = CO_FR * 32, chs = kWtChs;
    f16x8 *pieces = reinterpret_cast<f16x8 *>(conv_smem);            // [2][4][chs] x 16 B
    // scale | shift of the tile's CO_TILE channels, read by the epilogue with LDS loads: a global load there would sit behind the
    // previous round's stores in the in-order vector-memory counter (stamps: 2 k cycles per round waiting for write acknowledges)
    float *aff = reinterpret_cast<float *>(conv_smem + (size_t)2 * 4 * chs * 16);     // [tile parity][scale | shift][CO_TILE]

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CG = P.Cin / CK;
    const long long wplane = (long long)kWtTaps * CG * P.Cout * 16;  // halves per plane

    struct TileId { int b, co_t, t0, h0, w0, s0; };
    auto decode = [&](int vb, TileId &T) -> bool {                   // (the direct kernel's walk: conv3d.hip)
        const int xcd = vb & 7, q = vb >> 3;
        // xcd_remap 1: all channel tiles of a box back to back (the XCD's L2 keeps the box's input); 2: the XCD's boxes back to
        // back per channel tile (its L2 keeps that tile's weights: deep layers, whose weights are larger than their activations)
        T.co_t = P.xcd_remap == 2 ? q / P.per_xcd : q % P.n_co_tiles;
        const int sp = P.xcd_remap == 2 ? q - T.co_t * P.per_xcd : q / P.n_co_tiles;
        int bid = xcd * P.per_xcd + sp;
        if (sp >= P.per_xcd || bid >= P.n_spatial) return false;
        const int iw = bid % P.nW;            bid /= P.nW;
        const int ih = bid % P.nH;            bid /= P.nH;
        const int it = bid % P.nT;            bid /= P.nT;
        T.b = bid;
        T.t0 = it * P.TT; T.h0 = ih * P.TH; T.w0 = iw * P.TW; T.s0 = 0;
        if (P.flat) { T.s0 = iw * P.flat; T.h0 = T.s0 / P.Wout; T.w0 = 0; }     // a range of the plane: its first row, all columns
        return true;
    };
    // pair-position m of a tile -> (pair, row, column) relative to the tile's origin (t0, h0, w0); false: not an output
    auto pos_of = [&](int m, const TileId &T, int &pp, int &th, int &tw) -> bool {
        if (P.flat) {
            pp = m / P.flat;
            const int s = T.s0 + (m - pp * P.flat);
            const int h = s / P.Wout;
            th = h - T.h0;
            tw = s - h * P.Wout;
            return m < P.n_pos && s < P.Hout * P.Wout;
        }
        pp = m / (P.TH * P.TW);
        const int rem = m - pp * (P.TH * P.TW);
        th = rem / P.TW;
        tw = rem - th * P.TW;
        return m < P.n_pos;
    };
    auto next_tile = [&](int &vb, TileId &T) -> bool {
        for (; vb < P.total_blocks; vb += (int)gridDim.x)
            if (decode(vb, T)) return true;
        return false;
    };

    const float *tail = reinterpret_cast<const float *>(P.wq + 2 * wplane);
    const float w_scale = tail[1];
    auto amax_of = [&](int b) -> float { return P.x0_absmax ? P.x0_absmax[b * P.amax_bs] : 4094.0f; };
    // the transformed inputs reach 2 max |x|: one binade less than the direct kernel's pre-scale
    auto scale_of = [&](int b) -> float { return P.x0_absmax ? pow2_prescale(2.0f * amax_of(b)) : 0.5f * kActScale; };
    const int NTB = TAIL ? P.tCG / 4 : 0;                             // tail barriers per tile: two k-steps per wave each
    const int CGT = CG + NTB;
    auto tamax_of = [&](int b) -> float {
        if (!P.tx0_absmax) return 4094.0f;
        float am = P.tx0_absmax[b * P.amax_bs];
        if (P.tx1_absmax) am = fmaxf(am, P.tx1_absmax[b * P.amax_bs]);
        return am;
    };
    auto tscale_of = [&](int b) -> float { return P.tx0_absmax ? pow2_prescale(tamax_of(b)) : kActScale; };
    const float wt_scale = TAIL ? reinterpret_cast<const float *>(P.sc_w + 2 * (long long)P.tCG * P.Cout * 16)[1] : 1.0f;
    // TAIL: the common accumulator scale of batch element b, and the input pre-scales that produce it (all powers of two)
    auto acc_scale_of = [&](int b) -> float { return TAIL ? fminf(scale_of(b) * w_scale, tscale_of(b) * wt_scale) : scale_of(b) * w_scale; };
    auto xs_of = [&](int b) -> float { return TAIL ? acc_scale_of(b) / w_scale : scale_of(b); };
    auto ts_of = [&](int b) -> float { return acc_scale_of(b) / wt_scale; };
    // the order of a tile's CGT barriers: main chunk and tail barrier alternate while both remain
    const int nI = CG < NTB ? CG : NTB;
    auto is_tail = [&](int k) -> bool { return TAIL && (k < 2 * nI ? (k & 1) != 0 : NTB > CG); };
    auto idx_of = [&](int k) -> int { return !TAIL ? k : (k < 2 * nI ? k >> 1 : k - nI); };
    // range guard (conv3d_f16x2_ws_kernel): K = 9 Cin terms per slot, three slots per output, operands 2 max |x| and max |G|
    if (P.guard && blockIdx.x == 0 && wave == 0) {
        float sm = 0.0f;
        for (int co = lane; co < P.Cout; co += 64) sm = fmaxf(sm, fabsf(P.scale[co]));
#pragma unroll
        for (int o = 32; o; o >>= 1) sm = fmaxf(sm, __shfl_xor(sm, o));
        const int nb = P.amax_bs ? P.B : 1;
        for (int b = lane; b < nb; b += 64) {
            const float am = 2.0f * amax_of(b), xs = xs_of(b);
            float E = sm * (float)(P.Cin * 27) * 0x1p-25f * (tail[0] / xs + am / tail[1]);
            if (TAIL) {                                               // (conv3d_f16x2_ws_kernel, FUSE 3: the bounds add)
                const float *tl = reinterpret_cast<const float *>(P.sc_w + 2 * (long long)P.tCG * P.Cout * 16);
                const float tam = tamax_of(b), txs = ts_of(b);
                E += sm * (float)(P.tCG * 16) * 0x1p-25f * (tl[0] / txs + tam / tl[1]);
                if (!(xs > 0x1p-100f && txs > 0x1p-100f)) E = __builtin_inff();     // (a scale gap beyond the f32 range)
            }
            P.guard[b * P.amax_bs] = E;
        }
    }

    int vb = blockIdx.x;
    TileId T;
    if (!next_tile(vb, T)) return;
    int gc = 0;                                                       // chunks handled so far: pieces buffer gc & 1
    // In rectangle mode a lane's (pair, row, column) inside the box does not depend on the tile: decoded once (two integer divisions
    // by run-time values per position -- ~100 VALU instructions; per tile they were 2 k cycles of a 64-channel layer's 47 k).
    // Range tiles (P.flat) decode per tile: their rows depend on where the range starts.
    struct Pos { int pp, th, tw; bool ok; };
    auto pos_cached = [&](int m) -> Pos {
        Pos q{0, 0, 0, false};
        if (!P.flat) q.ok = pos_of(m, T, q.pp, q.th, q.tw);
        return q;
    };
    auto pos_get = [&](const Pos &c, int m, const TileId &L) -> Pos {
        if (!P.flat) return c;
        Pos q;
        q.ok = pos_of(m, L, q.pp, q.th, q.tw);
        return q;
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int ptid = tid - 256;                                   // = the element (p, hh, hw) of the halo box this lane owns
        const bool wave_on = (wave - 4) * 64 < P.plane;               // wave-uniform
        const int e_pp = ptid / (P.HH * P.HWd), e_hh = (ptid - e_pp * (P.HH * P.HWd)) / P.HWd, e_hw = ptid - e_pp * (P.HH * P.HWd) - e_hh * P.HWd;
        const Pos tpos = pos_cached(ptid & 127);                      // (tail: the output pair-position this lane gathers for)
        float x_scale = xs_of(T.b);
        float t_scale = TAIL ? ts_of(T.b) : 1.0f;
        TileId Tprev = T;                                             // the tile whose chunks are all staged: its epilogue is the producers' (tile_rendezvous)
        unsigned toff[2][2];                                          // tail: [source][time step 2p, 2p+1] of this lane's output position
        [[maybe_unused]] int tm_pp = 0, tm_hi = 0, tm_wi = 0, tm_hs = 0, tm_ws = 0;
        [[maybe_unused]] bool tm_ok = false;
        __amdgpu_buffer_rsrc_t rs_t0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0), 0, 0, 0x00020000), rs_t1 = rs_t0;
        unsigned goff[4];                                             // the element's four time steps 2p-1 .. 2p+2 (kOOB: zero padding)
        float R0[4][CK], R1[4][CK];
        __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0), 0, 0, 0x00020000);
        const long long seq = (long long)P.T * P.Cin * (P.Hin * P.Winp);
        const int cg_bytes = P.Hin * P.Winp * 64;                     // bytes between 16-channel groups of a time step
        auto load_chunk = [&](const TileId &L, int kseq, float (&R)[4][CK]) {
            const int cidx = idx_of(kseq);                            // main chunk / tail barrier number
            if (is_tail(kseq)) {                                      // uniform: a tail barrier's two k-steps
                typedef float f32x4g __attribute__((ext_vector_type(4)));
                if (cidx == 0) {
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) toff[sidx][0] = toff[sidx][1] = kOOB;
                    if (tm_ok) {                                      // (position and map entries: fetched with the tile's first chunk)
                        {
                            const int hi = tm_hi, wi = tm_wi, hs = tm_hs, ws = tm_ws, pp = tm_pp;
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int t = L.t0 + 2 * pp + i;
                                if (t < P.T) {
                                    toff[0][i] = 4u * (unsigned)((t * P.tC0) * (P.tH0 * P.tW0p)) + 64u * (unsigned)(hs * P.tW0p + ws);
                                    toff[1][i] = 4u * (unsigned)((t * P.tC1) * (P.tHin * P.tWinp)) + 64u * (unsigned)(hi * P.tWinp + wi);
                                }
                            }
                        }
                    }
                    const long long seq0 = (long long)P.T * P.tC0 * (P.tH0 * P.tW0p), seq1 = (long long)P.T * P.tC1 * (P.tHin * P.tWinp);
                    rs_t0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.tx0 + L.b * seq0), 0, (int)(seq0 * 4), 0x00020000);
                    rs_t1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.tx1 ? P.tx1 + L.b * seq1 : P.tx0), 0, P.tx1 ? (int)(seq1 * 4) : 0, 0x00020000);
                }
                const int gs = (wave - 4) >> 1;                       // lanes 0-127: the barrier's first k-step, 128-255: its second
                const int CG0 = P.tC0 / 16;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {                      // half A's group, half B's group
                    const int g = hb * (P.tCG / 2) + 2 * cidx + gs;
                    const bool s1 = g >= CG0;                         // uniform: which source holds group g
                    const int gbytes = s1 ? (g - CG0) * (P.tHin * P.tWinp * 64) : g * (P.tH0 * P.tW0p * 64);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int k4 = 0; k4 < CK / 4; ++k4) {
                            const f32x4g v = s1 ? __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rs_t1, toff[1][i], gbytes + 16 * k4, 0))
                                                : __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rs_t0, toff[0][i], gbytes + 16 * k4, 0));
                            R[2 * hb + i][4 * k4] = v[0]; R[2 * hb + i][4 * k4 + 1] = v[1]; R[2 * hb + i][4 * k4 + 2] = v[2]; R[2 * hb + i][4 * k4 + 3] = v[3];
                        }
                }
                return;
            }
            if (cidx == 0) {                                          // uniform: a new tile
                if constexpr (TAIL) {
                    // the tail's gather position of this lane, its index-map entries requested now and used one barrier later
                    const Pos q = pos_get(tpos, ptid & 127, L);
                    tm_ok = q.ok && L.h0 + q.th < P.Hout && L.w0 + q.tw < P.Wout;
                    tm_pp = q.pp;
                    tm_hi = tm_ok ? (L.h0 + q.th) * P.tS : 0;
                    tm_wi = tm_ok ? (L.w0 + q.tw) * P.tS : 0;
                    tm_hs = P.thmap ? P.thmap[tm_hi] : tm_hi;
                    tm_ws = P.twmap ? P.twmap[tm_wi] : tm_wi;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) goff[i] = kOOB;
                if (ptid < P.plane) {
                    const int pp = e_pp, hh = e_hh, hw = e_hw;
                    const int h = L.h0 - 1 + hh, w = L.w0 - 1 + hw;
                    if (h >= 0 && h < P.Hin && w >= 0 && w < P.Win) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int t = L.t0 + 2 * pp - 1 + i;
                            if (t >= 0 && t < P.T) goff[i] = 4u * (unsigned)((t * P.Cin) * (P.Hin * P.Winp)) + 64u * (unsigned)(h * P.Winp + w);
                        }
                    }
                }
                rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x0 + L.b * seq), 0, (int)(seq * 4), 0x00020000);
            }
            if (wave_on) {
                typedef float f32x4g __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k4 = 0; k4 < CK / 4; ++k4) {
#if defined(V2CE_ABLATE_TAPS) && (V2CE_ABLATE_TAPS & 4)   // diagnostic build: a quarter of the gather's load instructions -- WRONG results
                        if (k4) { R[i][4 * k4] = R[i][0]; R[i][4 * k4 + 1] = R[i][1]; R[i][4 * k4 + 2] = R[i][2]; R[i][4 * k4 + 3] = R[i][3]; continue; }
#endif
                        const f32x4g v = __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(rs_in, goff[i], cidx * cg_bytes + 16 * k4, 0));
                        R[i][4 * k4] = v[0]; R[i][4 * k4 + 1] = v[1]; R[i][4 * k4 + 2] = v[2]; R[i][4 * k4 + 3] = v[3];
                    }
            }
        };
        int vbL = vb, cgL = 0;                                        // load cursor: two chunks ahead of the conversion, across tiles
        TileId TL = T;
        bool moreL = true;
        auto load_next = [&](float (&R)[4][CK]) {
            if (!moreL) return;
            load_chunk(TL, cgL, R);
            if (++cgL == CGT) {
                cgL = 0;
                vbL += (int)gridDim.x;
                moreL = next_tile(vbL, TL);
            }
        };
        int cgC = 0;                                                  // conversion cursor: chunk inside the tile
        // one transformed, split element (slot j, channels 8 hg .. 8 hg + 7) into LDS; `tf(j, c)`: the transformed value of channel c
        auto convert_with = [&](float c_scale0, float c_scale12, auto tf) __attribute__((always_inline)) {
            f16x8 *qb = pieces + (gc & 1) * 4 * chs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c_scale = (j == 1 || j == 2) ? c_scale12 : c_scale0;
#pragma unroll
                for (int hg = 0; hg < 2; ++hg) {
                    typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
                    u32x4c ph, pl;
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) {
                        const float xa = tf(j, 8 * hg + 2 * c2), xb = tf(j, 8 * hg + 2 * c2 + 1);
                        unsigned h, l;                                // hi = f16(x s), lo = f16(x s - hi)  (conv3d.hip, producers)
                        asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                            "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
                            "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "=&v"(h), "=&v"(l) : "v"(xa), "v"(xb), "v"(c_scale));
                        ph[c2] = h;
                        pl[c2] = l;
                    }
                    qb[hg * chs + j * kWtSlot + ptid] = __builtin_bit_cast(f16x8, ph);
                    qb[(2 + hg) * chs + j * kWtSlot + ptid] = __builtin_bit_cast(f16x8, pl);
                }
            }
        };
        auto convert = [&](const float (&R)[4][CK]) {
            const bool tail = is_tail(cgC);                           // uniform
            if (!wave_on && !tail) return;
            [[maybe_unused]] const unsigned long long tcv = TICK();
            if (tail) {
                // R = { xA(2p), xA(2p+1), xB(2p), xB(2p+1) }:  xA(2p) | (xB(2p) + xB(2p+1)) / 2 | (xB(2p) - xB(2p+1)) / 2 | -xA(2p+1)
                // (the halving rides in the pre-scale of slots 1 and 2: exact)
                convert_with(t_scale, 0.5f * t_scale, [&](int j, int c) -> float {
                    return j == 0 ? R[0][c] : j == 1 ? R[2][c] + R[3][c] : j == 2 ? R[2][c] - R[3][c] : -R[1][c];
                });
            } else {
                // input transform B^T d (f32): d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3
                convert_with(x_scale, x_scale, [&](int j, int c) -> float {
                    return j == 0 ? R[0][c] - R[2][c] : j == 1 ? R[1][c] + R[2][c] : j == 2 ? R[2][c] - R[1][c] : R[1][c] - R[3][c];
                });
            }
            ACC_T(t_cvt, tcv);
        };
        bool moreC = true;
        auto advance = [&]() {
            ++gc;
            if (++cgC == CGT) {
                cgC = 0;
                Tprev = T;
                vb += (int)gridDim.x;
                moreC = next_tile(vb, T);
                if (moreC) {
                    x_scale = xs_of(T.b);
                    if (TAIL) t_scale = ts_of(T.b);
                }
            }
        };
        // The epilogue of the tile BEFORE the one being staged (Tprev), by the producers: consumer wave j has left the accumulators of
        // slot j in LDS; producer wave f sums the four slots of position fragment f into y(2p), y(2p+1), applies scale, shift, residual
        // and activation (conv_epilogue's arithmetic, conv3d_dev.h) and stores.  The stores -- 64 KB per tile through a path that takes
        // ~8 k cycles for them, in a vector-memory counter that is in order with the loads behind them -- drain in the producers'
        // time: the consumers go on with the next tile's MFMAs after the two barriers of the last round.
        const int fr = wave - 4;                                      // the position fragment this wave stores
        int spk;                                                      // (pair << 20 | row << 10 | column) of this lane's position in it, -1: none
        {
            const Pos q = pos_cached(fr * 32 + l32);
            spk = q.ok ? (q.pp << 20) | (q.th << 10) | q.tw : -1;
        }
        int seen_b = -1;                                              // range tracking: the batch element and the maximum already committed
        unsigned seen_max = 0u;
        int ntile_p = 0;
        auto tile_rendezvous = [&]() {
            [[maybe_unused]] const unsigned long long tb = TICK();
            if constexpr (!PEPI) {                                    // the consumers' own epilogue: only its barriers
#pragma unroll
                for (int k = 0; k < 2 * CO_FR; ++k) lds_barrier();
                ACC_T(t_xbar, tb);
                return;
            }
            typedef float f32x4t __attribute__((ext_vector_type(4)));
            typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
            const TileId &E = Tprev;
            const int co0 = E.co_t * CO_TILE;
            const float out_inv_scale = 1.0f / acc_scale_of(E.b);
            const float *a2 = aff + (ntile_p & 1) * 2 * CO_TILE;
            ++ntile_p;
            int poff[2], roff[2];                                     // byte offsets of the two outputs, and of their residuals
            {
                const int m = fr * 32 + l32;
                poff[0] = poff[1] = -1;
                roff[0] = roff[1] = -1;
                Pos q;
                if (!P.flat) { q.ok = spk >= 0; q.pp = spk >> 20; q.th = (spk >> 10) & 1023; q.tw = spk & 1023; }
                else q.ok = pos_of(m, E, q.pp, q.th, q.tw);
                if (q.ok) {
                    const int t = E.t0 + 2 * q.pp, h = E.h0 + q.th, w = E.w0 + q.tw;
                    if (h < P.Hout && w < P.Wout) {
                        if (t < P.T) poff[0] = 4 * ((t * P.Cout) * (P.Hout * P.Woutp)) + 64 * (h * P.Woutp + w);
                        if (t + 1 < P.T) poff[1] = 4 * (((t + 1) * P.Cout) * (P.Hout * P.Woutp)) + 64 * (h * P.Woutp + w);
                        if (RES && P.res_up) {
                            if (t < P.T) roff[0] = 4 * ((t * P.Cout) * (P.rH * P.rWp)) + 64 * ((h >> 1) * P.rWp + (w >> 1));
                            if (t + 1 < P.T) roff[1] = 4 * (((t + 1) * P.Cout) * (P.rH * P.rWp)) + 64 * ((h >> 1) * P.rWp + (w >> 1));
                        } else {
                            roff[0] = poff[0]; roff[1] = poff[1];
                        }
                    }
                }
            }
            const f32x4t *xch = reinterpret_cast<const f32x4t *>(pieces + ((gc - 1) & 1) * 4 * chs);   // the buffer of that tile's last chunk
            const long long yseq = (long long)P.T * P.Cout * (P.Hout * P.Woutp);
            const int gstride = P.Hout * P.Woutp * 64;                // bytes between 16-channel groups of a time step
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y + E.b * yseq, 0, (int)(yseq * 4), 0x00020000);
            const bool r_up = RES && P.res_up;                        // uniform
            const long long rseq = r_up ? (long long)P.T * P.Cout * (P.rH * P.rWp) : yseq;
            const int gstride_r = r_up ? P.rH * P.rWp * 64 : gstride;
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(RES ? P.res + E.b * rseq : P.scale), 0,
                                                                                  RES ? (int)(rseq * 4) : 0, 0x00020000);
            unsigned vr[2], vo[2], vmask[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                vr[o] = roff[o] >= 0 ? (unsigned)(roff[o] + 16 * half) : kOOB;
                vo[o] = poff[o] >= 0 ? (unsigned)(poff[o] + 16 * half) : kOOB;
                vmask[o] = poff[o] >= 0 ? 0x7fffffffu : 0u;
            }
            const float slope = act_slope(P.act);
            unsigned ymax = 0u;
            // channels co0 + 32 q + 8 r4 + 4 half + {0..3}: group (co0 / 16 + 2 q + (r4 >> 1)), bytes 32 (r4 & 1) + 16 half inside it
#pragma unroll
            for (int q = 0; q < CO_FR; ++q) {
                f32x4t rv[2][4];                                      // the round's residual, requested before the round is waited for
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        if constexpr (RES) rv[o][r4] = __builtin_bit_cast(f32x4t, __builtin_amdgcn_raw_buffer_load_b128(
                                                           rs_r, vr[o], (co0 / 16 + 2 * q + (r4 >> 1)) * gstride_r + 32 * (r4 & 1), 0));
                        else rv[o][r4] = f32x4t{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                lds_barrier();                                        // the consumers' round q is in LDS
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    // (one r4 at a time: sixteen registers of transformed sums next to the two chunks in flight -- reading the whole round
                    // first, or half of it, spills 50-500 registers and costs more than the consumers' shorter wait gains)
                    f32x4t mm[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) mm[jj] = xch[fr * chs + jj * kWtSlot + r4 * 64 + lane];
                    if (r4 == 3) lds_barrier();                       // every read of the round is done: the consumers may overwrite it
                    const f32x4t scq = *reinterpret_cast<const f32x4t *>(a2 + 32 * q + 8 * r4 + 4 * half);
                    const f32x4t shq = *reinterpret_cast<const f32x4t *>(a2 + CO_TILE + 32 * q + 8 * r4 + 4 * half);
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        f32x4t out;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float y = o == 0 ? (mm[0][k] + mm[1][k]) + mm[2][k] : (mm[1][k] - mm[2][k]) - mm[3][k];
                            float v = y * (scq[k] * out_inv_scale) + shq[k];
                            v += rv[o][r4][k];
                            v = apply_act(v, slope);
                            out[k] = v;
                            const unsigned av = __builtin_bit_cast(unsigned, v) & vmask[o];
                            ymax = av > ymax ? av : ymax;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4t, out), rs_y, vo[o],
                                                               (co0 / 16 + 2 * q + (r4 >> 1)) * gstride + 32 * (r4 & 1), 0);
                        asm volatile("s_nop 1" : "+v"(out));          // (16-byte store data hazard: conv_epilogue)
                    }
                }
            }
            if (P.y_absmax) {
                // max |y| of the tile -> the launch's range slot; the wave remembers the largest value it has already committed for this
                // batch element and sends an atomic (no return value, nothing to wait for) only beyond that
#pragma unroll
                for (int o = 32; o; o >>= 1) {
                    const unsigned other = (unsigned)__shfl_xor((int)ymax, o);
                    ymax = other > ymax ? other : ymax;
                }
                if (E.b != seen_b) { seen_b = E.b; seen_max = 0u; }
                if (ymax > seen_max) {                                // uniform
                    seen_max = ymax;
                    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(P.y_absmax + E.b * P.amax_bs), ymax);
                }
            }
            ACC_T(t_xbar, tb);
        };
        load_next(R0);
        load_next(R1);
        while (moreC) {
            convert(R0);
            load_next(R0);
            if (cgC == 0 && gc != 0) tile_rendezvous();
            { [[maybe_unused]] const unsigned long long tb = TICK();
            __syncthreads();                                          // barrier gc: pieces[gc & 1] ready
            ACC_T(t_bar, tb); }
            advance();
            if (!moreC) break;
            convert(R1);
            load_next(R1);
            if (cgC == 0 && gc != 0) tile_rendezvous();
            { [[maybe_unused]] const unsigned long long tb = TICK();
            __syncthreads();
            ACC_T(t_bar, tb); }
            advance();
        }
        tile_rendezvous();                                            // the last tile's
#ifdef V2CE_STAMP
        if (lane == 0 && wave == 4) {
            unsigned long long *o = P.stamps + ((long long)blockIdx.x * 2 + 1) * 8;
            o[0] = TICK() - t_all; o[1] = t_bar; o[2] = t_xbar; o[3] = t_cvt;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers: wave = transform slot
    __builtin_amdgcn_s_setprio(2);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(P.wq), 0, (int)(4 * wplane), 0x00020000);
    const int lo_off = (int)(2 * wplane);                             // bytes from the hi plane to the lo plane
    const int tap_stride = CG * P.Cout * 32;                          // bytes between taps
    const int cg_stride = P.Cout * 32;                                // bytes between 16-channel groups
    const int slot_off = wave * 9 * tap_stride;                       // this slot's nine taps
    f16x8 ah[NA][CO_FR], al[NA][CO_FR], bh[PO_FR], bl[PO_FR];
    int wlane[CO_FR];
#define V2CE_LOAD_A(slot_, soff_)                                                              \
    {                                                                                          \
        const int so_ = (soff_);                                                               \
        _Pragma("unroll") for (int q = 0; q < CO_FR; ++q) {                                    \
            ah[slot_][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, wlane[q], so_, 0));          \
            al[slot_][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, wlane[q], so_ + lo_off, 0)); \
        }                                                                                      \
    }
    // this lane's offset in the slot's plane for each of its fragments (-1: no position) -- one register each across the main loop
    int crel[PO_FR];
#pragma unroll
    for (int f = 0; f < PO_FR; ++f) {
        const Pos q = pos_cached(f * 32 + l32);
        crel[f] = q.ok ? (q.pp * P.HH + q.th) * P.HWd + q.tw : -1;
    }

    int ring_co_t = -1;
    int ntile = 0;                                                    // tiles of this workgroup so far: parity of the scale / shift table
    [[maybe_unused]] int spk = -1;                                    // (!PEPI: the position this lane stores, packed; range tracking state)
    if constexpr (!PEPI) {
        const Pos q = pos_cached(wave * 32 + l32);
        spk = q.ok ? (q.pp << 20) | (q.th << 10) | q.tw : -1;
    }
    [[maybe_unused]] int seen_b = -1;
    [[maybe_unused]] unsigned seen_max = 0u;
    bool more = true;
    while (more) {
        const int co0 = T.co_t * CO_TILE;
        const float inv_scale = 1.0f / acc_scale_of(T.b);             // a power of two: exact
        int bhb[PO_FR];
#pragma unroll
        for (int f = 0; f < PO_FR; ++f) {
            const int m = f * 32 + l32;                               // pair-position (p, h, w) of the box
            bhb[f] = half * chs + wave * kWtSlot;
            if (!P.flat) {
                bhb[f] += crel[f] >= 0 ? crel[f] : 0;
            } else {
                Pos q;
                q.ok = pos_of(m, T, q.pp, q.th, q.tw);
                if (q.ok) bhb[f] += (q.pp * P.HH + q.th) * P.HWd + q.tw;
            }
        }
        if (T.co_t != ring_co_t) {                                    // uniform: (re)load the ring for this channel tile
            ring_co_t = T.co_t;
        }
        {
            // scale | shift of this tile's channels for the producers' epilogue, in the table of the tile's parity (they read it while
            // the consumers are already in the next tile, which writes the other one); every consumer wave writes the same values
            float *a2 = aff + (ntile & 1) * 2 * CO_TILE;
            if (lane < CO_TILE) {
                a2[lane] = P.scale[co0 + lane];
                a2[CO_TILE + lane] = P.shift[co0 + lane];
            }
            ++ntile;
#pragma unroll
            for (int q = 0; q < CO_FR; ++q) {
                int co = co0 + q * 32 + l32;
                co = co < P.Cout ? co : P.Cout - 1;
                wlane[q] = (co * 16 + 8 * half) * 2;
            }
            step_loop<0, NA - 1>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                V2CE_LOAD_A(t, slot_off + t * tap_stride)             // chunk 0, taps 0 .. NA-2
            });
        }
        f32x16 acc[CO_FR][PO_FR];
#pragma unroll
        for (int q = 0; q < CO_FR; ++q)
#pragma unroll
            for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][f][r] = 0.0f;

        // tail state (TAIL): this slot's half of the channel groups, A fragments double-buffered over the k-steps
        const long long tplane = TAIL ? (long long)P.tCG * P.Cout * 16 : 0;      // halves per plane of the tail weights
        const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(TAIL ? P.sc_w : P.wq), 0, (int)(4 * tplane), 0x00020000);
        const int gbase = (wave == 0 || wave == 3) ? 0 : P.tCG / 2;
        int bpt[TAIL ? PO_FR : 1];
        f16x8 ath[2][TAIL ? CO_FR : 1], atl[2][TAIL ? CO_FR : 1];
        if constexpr (TAIL) {
#pragma unroll
            for (int f = 0; f < PO_FR; ++f) bpt[f] = half * chs + wave * kWtSlot + f * 32 + l32;
        }
        auto chunk_barrier = [&]() -> const f16x8 * {
            const f16x8 *qb = pieces + (gc & 1) * 4 * chs;
            [[maybe_unused]] const unsigned long long tb = TICK();
            __syncthreads();                                          // barrier gc: pieces[gc & 1] ready
            ACC_T(t_bar, tb);
            ++gc;
            return qb;
        };
        auto main_chunk = [&](int cg) __attribute__((always_inline)) {          // a 16-channel chunk of the 3x3x3 conv
            const f16x8 *qb = chunk_barrier();
                const int wc = slot_off + cg * cg_stride;
                const int wn = cg + 1 < CG ? wc + cg_stride : slot_off;   // last chunk: chunk 0 again (the next tile's start)
#pragma unroll
                for (int f = 0; f < PO_FR; ++f) {
                    bh[f] = qb[bhb[f]];
                    bl[f] = qb[bhb[f] + 2 * chs];
                }
                step_loop<0, 9>([&](auto tc) {
                    constexpr int tap = decltype(tc)::value;
                    constexpr int nt = tap + 1;
                    constexpr int dh = nt / 3, dw = nt % 3;
                    constexpr int pt = tap + NA - 1;                      // the tap whose A fragments are fetched now
#if !(defined(V2CE_ABLATE_TAPS) && (V2CE_ABLATE_TAPS & 1))   // diagnostic build (tools/tap_ablate.sh): the ring is never refilled -- WRONG results
                    if constexpr (pt < 9) {
                        V2CE_LOAD_A(pt % NA, wc + pt * tap_stride)
                    } else {
                        V2CE_LOAD_A(pt % NA, wn + (pt - 9) * tap_stride)
                    }
#endif
                    const int toff = dh * P.HWd + dw;                     // next tap's offset in the slot's plane
#pragma unroll
                    for (int f = 0; f < PO_FR; ++f) {
#pragma unroll
                        for (int q = 0; q < CO_FR; ++q) {
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % NA][q], bh[f], acc[q][f], 0, 0, 0);
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % NA][q], bl[f], acc[q][f], 0, 0, 0);
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tap % NA][q], bh[f], acc[q][f], 0, 0, 0);
                        }
#if !(defined(V2CE_ABLATE_TAPS) && (V2CE_ABLATE_TAPS & 2))   // diagnostic build: tap 0's B fragments serve every tap -- WRONG results
                        if constexpr (nt < 9) {                           // refill in place for the next tap
                            bh[f] = qb[bhb[f] + toff];
                            bl[f] = qb[bhb[f] + toff + 2 * chs];
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
        };
        auto tail_prefetch = [&](int kb) __attribute__((always_inline)) {       // the first step's A fragments of tail barrier kb
            const int w0 = (gbase + 2 * kb) * cg_stride;
#pragma unroll
            for (int q = 0; q < (TAIL ? CO_FR : 0); ++q) {
                ath[0][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], w0, 0));
                atl[0][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], w0 + (int)(2 * tplane), 0));
            }
        };
        auto tail_barrier = [&](int kb) __attribute__((always_inline)) {        // two k-steps of the folded 1x1x1 tail
            if constexpr (TAIL) {
                const f16x8 *qb = chunk_barrier();
                const int w1 = (gbase + 2 * kb + 1) * cg_stride;      // the second step's A fragments (the first step's are in flight since
#pragma unroll                                                        // the end of the previous barrier's work)
                for (int q = 0; q < CO_FR; ++q) {
                    ath[1][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], w1, 0));
                    atl[1][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_t, wlane[q], w1 + (int)(2 * tplane), 0));
                }
#pragma unroll
                for (int gs = 0; gs < 2; ++gs) {
#pragma unroll
                    for (int f = 0; f < PO_FR; ++f) {
                        bh[f] = qb[bpt[f] + gs * 128];
                        bl[f] = qb[bpt[f] + gs * 128 + 2 * chs];
                    }
#pragma unroll
                    for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                        for (int q = 0; q < CO_FR; ++q) {
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ath[gs][q], bh[f], acc[q][f], 0, 0, 0);
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ath[gs][q], bl[f], acc[q][f], 0, 0, 0);
                            acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(atl[gs][q], bh[f], acc[q][f], 0, 0, 0);
                        }
                }
            }
        };
        if constexpr (!TAIL) {
            for (int cg = 0; cg < CG; ++cg) main_chunk(cg);
        } else {
            // M0 T0 M1 T1 ... while both remain (is_tail / idx_of: the producers walk the same order), then the rest of the longer part
            for (int i = 0; i < nI; ++i) {
                main_chunk(i);
                tail_prefetch(i);
                tail_barrier(i);
            }
            for (int i = nI; i < CG; ++i) main_chunk(i);
            for (int i = nI; i < NTB; ++i) {
                tail_prefetch(i);
                tail_barrier(i);
            }
        }

        if constexpr (PEPI) {
        // ---- output transform: the accumulators go to the producers through LDS, one channel fragment row per round
        [[maybe_unused]] const unsigned long long te = TICK();
        typedef float f32x4t __attribute__((ext_vector_type(4)));
        f32x4t *xch = reinterpret_cast<f32x4t *>(pieces + ((gc - 1) & 1) * 4 * chs);   // the buffer of the tile's last chunk
#pragma unroll
        for (int q = 0; q < CO_FR; ++q) {
            // (round q > 0: the producers' reads of the previous round are behind its second barrier)
#pragma unroll
            for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4t v;
                    v[0] = acc[q][f][4 * r4]; v[1] = acc[q][f][4 * r4 + 1]; v[2] = acc[q][f][4 * r4 + 2]; v[3] = acc[q][f][4 * r4 + 3];
                    xch[f * chs + wave * kWtSlot + r4 * 64 + lane] = v;
                }
            [[maybe_unused]] const unsigned long long tb = TICK();
            lds_barrier();                                            // the round is in LDS
            lds_barrier();                                            // the producers have read it
            ACC_T(t_xbar, tb);
        }
        ACC_T(t_epi, te);
        } else {
        const float *aff_c = aff + ((ntile - 1) & 1) * 2 * CO_TILE;   // (this tile's table)
        // ---- output transform + epilogue.  This wave stores position fragment f = wave: outputs t = 2p, 2p + 1.
        [[maybe_unused]] const unsigned long long te = TICK();
        int poff[2], roff[2];                                         // byte offsets of the two outputs, and of their residuals
        {
            const int m = wave * 32 + l32;
            poff[0] = poff[1] = -1;
            roff[0] = roff[1] = -1;
            Pos q;
            if (!P.flat) { q.ok = spk >= 0; q.pp = spk >> 20; q.th = (spk >> 10) & 1023; q.tw = spk & 1023; }
            else q.ok = pos_of(m, T, q.pp, q.th, q.tw);
            if (q.ok) {
                const int t = T.t0 + 2 * q.pp, h = T.h0 + q.th, w = T.w0 + q.tw;
                if (h < P.Hout && w < P.Wout) {
                    if (t < P.T) poff[0] = 4 * ((t * P.Cout) * (P.Hout * P.Woutp)) + 64 * (h * P.Woutp + w);
                    if (t + 1 < P.T) poff[1] = 4 * (((t + 1) * P.Cout) * (P.Hout * P.Woutp)) + 64 * (h * P.Woutp + w);
                    if (RES && P.res_up) {
                        if (t < P.T) roff[0] = 4 * ((t * P.Cout) * (P.rH * P.rWp)) + 64 * ((h >> 1) * P.rWp + (w >> 1));
                        if (t + 1 < P.T) roff[1] = 4 * (((t + 1) * P.Cout) * (P.rH * P.rWp)) + 64 * ((h >> 1) * P.rWp + (w >> 1));
                    } else {
                        roff[0] = poff[0]; roff[1] = poff[1];
                    }
                }
            }
        }
        typedef float f32x4t __attribute__((ext_vector_type(4)));
        typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
        f32x4t *xch = reinterpret_cast<f32x4t *>(pieces + ((gc - 1) & 1) * 4 * chs);   // the buffer of the tile's last chunk
        // The epilogue proper is conv_epilogue's arithmetic (conv3d_dev.h: y = act(acc * scale * inv_scale + shift + residual), max |y|,
        // 16-byte stores of the four channels a lane holds per r >> 2) with its loads -- residual, scale, shift -- issued BEFORE the
        // accumulators go through LDS, so that their latency is covered by the exchange instead of following it.
        const long long yseq = (long long)P.T * P.Cout * (P.Hout * P.Woutp);
        const int gstride = P.Hout * P.Woutp * 64;                    // bytes between 16-channel groups of a time step
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y + T.b * yseq, 0, (int)(yseq * 4), 0x00020000);
        const bool r_up = RES && P.res_up;                            // uniform
        const long long rseq = r_up ? (long long)P.T * P.Cout * (P.rH * P.rWp) : yseq;
        const int gstride_r = r_up ? P.rH * P.rWp * 64 : gstride;
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(RES ? P.res + T.b * rseq : P.scale), 0,
                                                                              RES ? (int)(rseq * 4) : 0, 0x00020000);
        unsigned vr[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) vr[o] = roff[o] >= 0 ? (unsigned)(roff[o] + 16 * half) : kOOB;
        unsigned vo[2], vmask[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            vo[o] = poff[o] >= 0 ? (unsigned)(poff[o] + 16 * half) : kOOB;
            vmask[o] = poff[o] >= 0 ? 0x7fffffffu : 0u;
        }
        const float slope = act_slope(P.act);
        unsigned ymax = 0u;
        [[maybe_unused]] unsigned long long te2 = 0;
        f32x4t rvq[2][2][4];                                          // residual of a round: [buffer][output 2p | 2p+1][r4]
        // channels co0 + 32 q + 8 r4 + 4 half + {0..3}: group (co0 / 16 + 2 q + (r4 >> 1)), bytes 32 (r4 & 1) + 16 half inside it
        auto soff_of = [&](int q, int r4) -> int { return (co0 / 16 + 2 * q + (r4 >> 1)) * gstride + 32 * (r4 & 1); };
        auto load_res = [&](int q, f32x4t (&rv)[2][4]) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    if constexpr (RES) rv[o][r4] = __builtin_bit_cast(f32x4t, __builtin_amdgcn_raw_buffer_load_b128(
                                                       rs_r, vr[o], (co0 / 16 + 2 * q + (r4 >> 1)) * gstride_r + 32 * (r4 & 1), 0));
                    else rv[o][r4] = f32x4t{0.0f, 0.0f, 0.0f, 0.0f};
                }
        };
#pragma unroll
        for (int q = 0; q < CO_FR; ++q) {
            auto soff = [&](int r4) -> int { return soff_of(q, r4); };
            f32x4t scq[4], shq[4];
            f32x4t (&rv)[2][4] = rvq[q & 1];
            if (q == 0) load_res(0, rvq[0]);
            // (round q > 0: the previous round's reads are behind its second barrier)
#pragma unroll
            for (int f = 0; f < PO_FR; ++f)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4t v;
                    v[0] = acc[q][f][4 * r4]; v[1] = acc[q][f][4 * r4 + 1]; v[2] = acc[q][f][4 * r4 + 2]; v[3] = acc[q][f][4 * r4 + 3];
                    xch[f * chs + wave * kWtSlot + r4 * 64 + lane] = v;
                }
            // the next round's residual: issued now (its accumulators' registers are free) and IN FRONT of this round's stores
            if (q + 1 < CO_FR) load_res(q + 1, rvq[(q + 1) & 1]);
            ACC_T(t_e0, q == 0 ? te : te2);
            { [[maybe_unused]] const unsigned long long tb = TICK();
            lds_barrier();
            ACC_T(t_xbar, tb); }
            [[maybe_unused]] const unsigned long long te1 = TICK();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                scq[r4] = *reinterpret_cast<const f32x4t *>(aff_c + 32 * q + 8 * r4 + 4 * half);
                shq[r4] = *reinterpret_cast<const f32x4t *>(aff_c + CO_TILE + 32 * q + 8 * r4 + 4 * half);
            }
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {                          // two r4 per pass: 32 registers of transformed sums at a time
                f32x4t mm[2][4];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) mm[rr][jj] = xch[wave * chs + jj * kWtSlot + (2 * hp + rr) * 64 + lane];
                if (hp == 1) {
                    ACC_T(t_e1, te1);
                    [[maybe_unused]] const unsigned long long tb = TICK();
                    lds_barrier();                                    // reads done: the next round / the producers may write
                    ACC_T(t_xbar, tb);
                    te2 = TICK();
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int r4 = 2 * hp + rr;
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        f32x4t out;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float y = o == 0 ? (mm[rr][0][k] + mm[rr][1][k]) + mm[rr][2][k] : (mm[rr][1][k] - mm[rr][2][k]) - mm[rr][3][k];
                            float v = y * (scq[r4][k] * inv_scale) + shq[r4][k];
                            v += rv[o][r4][k];
                            v = apply_act(v, slope);
                            out[k] = v;
                            const unsigned av = __builtin_bit_cast(unsigned, v) & vmask[o];
                            ymax = av > ymax ? av : ymax;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4t, out), rs_y, vo[o], soff(r4), 0);
                        asm volatile("s_nop 1" : "+v"(out));          // (16-byte store data hazard: conv_epilogue)
                    }
                }
            }
        }
        if (P.y_absmax) {
            // max |y| of the tile -> the launch's range slot.  absmax_commit reads the slot first (a global load the wave then waits
            // for: ~1.5 k cycles per tile); here the wave remembers the largest value it has already committed for this batch element
            // and sends an atomic (no return value, nothing to wait for) only beyond that
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                const unsigned other = (unsigned)__shfl_xor((int)ymax, o);
                ymax = other > ymax ? other : ymax;
            }
            if (T.b != seen_b) { seen_b = T.b; seen_max = 0u; }
            if (ymax > seen_max) {                                    // uniform
                seen_max = ymax;
                if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(P.y_absmax + T.b * P.amax_bs), ymax);
            }
        }
        ACC_T(t_epi, te);
        ACC_T(t_e2, te2);
        }
        vb += (int)gridDim.x;
        more = next_tile(vb, T);
    }
#ifdef V2CE_STAMP
    if (lane == 0 && wave == 0) {
        unsigned long long *o = P.stamps + ((long long)blockIdx.x * 2 + 0) * 8;
        o[0] = TICK() - t_all; o[1] = t_bar; o[2] = t_xbar; o[3] = t_epi; o[4] = (unsigned long long)gc; o[5] = t_e0; o[6] = t_e1; o[7] = t_e2;
    }
#endif
#undef V2CE_LOAD_A
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------------
// weights: G_j of W / sigma as fp16 hi / lo planes [2][slot * 9 + dh * 3 + dw][Cin / 16][Cout][16], tail { bound on |G|, pre-scale, 0, 0 }.
// A workgroup takes 32 output channels x one 16-channel group (32 runs of 432 contiguous floats of W), like sn_batch_pack_kernel.
// PASS 0: the bound 1.5 max |W / sigma| >= max |G| (|(g0 +- g1 + g2) / 2| <= 1.5 max |g|) into tail[0] (atomic max on the bit pattern;
// the caller zeroes it) -- the spectral-norm batch knows max |W| from its own passes and writes the same value without this pass
// (sn.hip, 0.1 ms per call for the ten layers); PASS 1: the planes, pre-scaled by the power of two derived from tail[0].
// ---------------------------------------------------------------------------------------------
constexpr int kWtMaxBatch = 16;
struct WtLayer {
    const float *w, *sigma;       // [Cout][cin_total][27]; sigma: device scalar or null (1)
    _Float16 *packed;
    int rows, cin;                // output channels; input channels PACKED: [ci0, ci0 + cin) of the tensor's cin_total
    int cin_total, ci0;
    long long tail_halves;        // where the tail { bound, pre-scale, 0, 0 } sits: 2 * rows * (channels of the buffer) * 36 halves into `packed`
};
struct WtBatch {
    WtLayer L[kWtMaxBatch];
    int n;
    int blk[kWtMaxBatch + 1];     // prefix of (rows / 32) * (cin / 16)
};
constexpr size_t kWtPackLds = (size_t)32 * 432 * 4 + (size_t)2 * kWtTaps * 514 * 2;

template <int PASS>
__global__ __launch_bounds__(512) void wt_pack_kernel(WtBatch B) {   // (129 KB of LDS: one workgroup per CU -- eight waves of it)
    extern __shared__ __attribute__((aligned(16))) unsigned char wt_smem[];
    float *wf = reinterpret_cast<float *>(wt_smem);                   // [32][16][27] = W / sigma
    _Float16 *hi = reinterpret_cast<_Float16 *>(wt_smem + 32 * 432 * 4), *lo = hi + kWtTaps * 514;
    int l = 0;
    while (l + 1 < B.n && (int)blockIdx.x >= B.blk[l + 1]) ++l;
    const WtLayer &P = B.L[l];
    const int CG = P.cin / 16;
    const int blk = blockIdx.x - B.blk[l];
    const int cg = blk % CG, co0 = (blk / CG) * 32;
    const long long n = (long long)P.rows * P.cin * kWtTaps;
    float *tail = reinterpret_cast<float *>(P.packed + P.tail_halves);
    const float sigma = P.sigma ? P.sigma[0] : 1.0f;
    for (int e = threadIdx.x; e < 32 * 432; e += 512) {
        const int col = e / 432, rem = e - col * 432;
        wf[e] = P.w[((long long)(co0 + col) * P.cin_total + P.ci0 + cg * 16) * 27 + rem] / sigma;
    }
    __syncthreads();
    const float w_scale = PASS ? pow2_prescale(tail[0]) : 1.0f;
    if (PASS && blk == 0 && threadIdx.x == 0) tail[1] = w_scale;
    float m = 0.0f;
    for (int it = threadIdx.x; it < 32 * 16 * 9; it += 512) {
        const int col = it / 144, rem = it - col * 144, j = rem / 9, t9 = rem - j * 9;
        const float *g = wf + col * 432 + j * 27 + t9;
        const float g0 = g[0], g1 = g[9], g2 = g[18];
        float G[4];
        G[0] = g0;
        G[1] = ((g0 + g1) + g2) * 0.5f;
        G[2] = ((g0 - g1) + g2) * 0.5f;
        G[3] = g2;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (PASS) {
                const float v = G[s] * w_scale;
                const _Float16 h = (_Float16)v;
                hi[(s * 9 + t9) * 514 + col * 16 + j] = h;
                lo[(s * 9 + t9) * 514 + col * 16 + j] = (_Float16)(v - (float)h);
            } else {
                m = fmaxf(m, fabsf(s == 1 ? g1 : G[s]));              // max |g| over the three time taps (G[0] = g0, G[3] = g2)
            }
        }
    }
    if (!PASS) {
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        m *= 1.5f;
        if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __atomic_load_n(reinterpret_cast<unsigned *>(tail), __ATOMIC_RELAXED))
            atomicMax(reinterpret_cast<unsigned *>(tail), __float_as_uint(m));
        return;
    }
    __syncthreads();
    const unsigned *hi32 = reinterpret_cast<const unsigned *>(hi), *lo32 = reinterpret_cast<const unsigned *>(lo);
    for (int tap = 0; tap < kWtTaps; ++tap) {
        const long long o = (((long long)tap * CG + cg) * P.rows + co0) * 16;       // halves
        const int t = threadIdx.x & 255;                              // threads 0-255: the hi plane's 256 dwords of this tap, 256-511: the lo plane's
        if (threadIdx.x < 256) reinterpret_cast<unsigned *>(P.packed + o)[t] = hi32[tap * 257 + t];
        else reinterpret_cast<unsigned *>(P.packed + n + o)[t] = lo32[tap * 257 + t];
    }
}

int wt_pack_launch(const WtBatch &B, int pass, hipStream_t st) {
    static const int once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wt_pack_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWtPackLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wt_pack_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWtPackLds);
        return 0;
    }();
    (void)once;
    if (B.blk[B.n] == 0) return V2CE_OK;
    if (pass == 0) hipLaunchKernelGGL(wt_pack_kernel<0>, dim3(B.blk[B.n]), dim3(512), kWtPackLds, st, B);
    else hipLaunchKernelGGL(wt_pack_kernel<1>, dim3(B.blk[B.n]), dim3(512), kWtPackLds, st, B);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

thread_local char *g_wt_name_out = nullptr;
thread_local size_t g_wt_name_cap = 0;

struct WtBox { int pp, th, tw, flat; };     // flat > 0: ranges of `flat` positions of the plane instead of th x tw rectangles

// The box (pairs x rows x columns) of a tile: at most PO_FR * 32 pair-positions, at most 256 halo elements (one per producer
// lane); fewest rounds of the persistent grid first, then the least halo per output.  V2CE_WT_BOX=pp,th,tw forces one.
WtBox choose_wt_box(int B, int T, int H, int W, int n_co, int n_cu, int pos_tile) {
    if (const char *e = getenv("V2CE_WT_BOX")) {
        WtBox b{0, 0, 0, 0};
        if (sscanf(e, "%d,%d,%d", &b.pp, &b.th, &b.tw) == 3 && b.pp > 0 && b.th > 0 && b.tw > 0 && b.pp * b.th * b.tw <= pos_tile &&
            b.pp * (b.th + 2) * (b.tw + 2) <= kWtSlot)
            return b;
    }
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int>, WtBox> cache;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_tuple(B, T, H, W, n_co, pos_tile);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int pairs = (T + 1) / 2;
    WtBox best{0, 0, 0, 0};
    double best_cost = 1e30;
    static const int flat_mode = [] { const char *e = getenv("V2CE_WT_FLAT"); return e ? atoi(e) : 1; }();
    // ranges of the row-major plane (narrow planes): n positions span at most (W - 1 + n - 1) / W + 1 rows of W + 2 halo columns
    for (int pp = 1; flat_mode && pp <= pairs && pp <= 16; ++pp) {
        const int n_max = pos_tile / pp;
        if (n_max < 1) break;
        const int nR = (H * W + n_max - 1) / n_max, n = (H * W + nR - 1) / nR;
        const int rows = std::min(H, (W - 1 + n - 1) / W + 1);
        if (pp * (rows + 2) * (W + 2) > kWtSlot) continue;
        const long long nsp = (long long)B * ((pairs + pp - 1) / pp) * nR;
        const long long blocks = 8 * ((nsp + 7) / 8) * n_co;
        const double rounds = (double)((blocks + n_cu - 1) / n_cu);
        const double halo = (double)pp * (rows + 2) * (W + 2) / kWtSlot;
        const double cost = rounds * (1.0 + 0.15 * halo) + 1e-3 * (double)blocks / n_cu;
        if (cost < best_cost) { best_cost = cost; best = WtBox{pp, rows, W, n}; }
    }
    for (int pp = 1; pp <= pairs && pp <= 16; ++pp)
        for (int th = 1; th <= H && th <= 64; ++th)
            for (int tw = 1; tw <= W && tw <= 128; ++tw) {
                if (pp * th * tw > pos_tile || pp * (th + 2) * (tw + 2) > kWtSlot) continue;
                const long long nsp = (long long)B * ((pairs + pp - 1) / pp) * ((H + th - 1) / th) * ((W + tw - 1) / tw);
                const long long blocks = 8 * ((nsp + 7) / 8) * n_co;
                const double rounds = (double)((blocks + n_cu - 1) / n_cu);
                // a tile costs its MFMA work (fixed) plus what its producers gather: 4 loads per halo element
                const double halo = (double)pp * (th + 2) * (tw + 2) / kWtSlot;
                const double cost = rounds * (1.0 + 0.15 * halo) + 1e-3 * (double)blocks / n_cu;
                if (cost < best_cost - 1e-9) { best_cost = cost; best = WtBox{pp, th, tw, 0}; }
            }
    cache.emplace(key, best);
    return best;
}

template <int CO_FR, int PO_FR, int RES, bool TAIL = false>
int launch_wt(ConvParams P, const v2ce_conv3d_desc &d, hipStream_t stream) {
    constexpr int CO_TILE = CO_FR * 32, POS_TILE = PO_FR * 32;
    if (g_wt_name_out) {
        snprintf(g_wt_name_out, g_wt_name_cap, "conv3d_wt_kernel<%d,%d,%d,%d>", CO_FR, PO_FR, RES, (int)TAIL);
        return V2CE_OK;
    }
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n < 8 ? 8 : (n / 8) * 8;
    }();
    P.n_co_tiles = (d.Cout + CO_TILE - 1) / CO_TILE;
    WtBox bx{d.tile_t / 2, d.tile_h, d.tile_w, 0};
    if (bx.pp <= 0 || bx.th <= 0 || bx.tw <= 0) bx = choose_wt_box(d.B, d.T, d.Hout, d.Wout, P.n_co_tiles, n_cu, POS_TILE);
    V2CE_REQUIRE(bx.pp > 0 && bx.th > 0 && bx.tw > 0, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_wt: no box fits");
    P.TT = 2 * bx.pp; P.TH = bx.th; P.TW = bx.tw;
    P.flat = bx.flat;
    P.n_pos = bx.flat ? bx.pp * bx.flat : bx.pp * bx.th * bx.tw;      // pair-positions
    P.HT = bx.pp; P.HH = bx.th + 2; P.HWd = bx.tw + 2;
    P.plane = P.HT * P.HH * P.HWd;                         // halo elements (p, hh, hw)
    P.nT = (d.T + P.TT - 1) / P.TT; P.nH = (d.Hout + bx.th - 1) / bx.th; P.nW = (d.Wout + bx.tw - 1) / bx.tw;
    if (bx.flat) { P.nH = 1; P.nW = (d.Hout * d.Wout + bx.flat - 1) / bx.flat; }
    V2CE_REQUIRE(P.n_pos <= POS_TILE && P.plane <= kWtSlot, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_wt: tile does not fit");
    P.n_spatial = d.B * P.nT * P.nH * P.nW;
    P.per_xcd = (P.n_spatial + 7) / 8;
    {
        // weights of a channel tile vs the input of the XCD's boxes: walk the larger one once
        static const int force = [] { const char *e = getenv("V2CE_WT_ORDER"); return e ? atoi(e) : 0; }();
        const double w_bytes = (double)CO_TILE * P.Cin * kWtTaps * 4, x_bytes = (double)P.per_xcd * P.n_pos * 2 * P.Cin * 4;
        P.xcd_remap = force ? force : (P.n_co_tiles > 1 && w_bytes * P.n_co_tiles > x_bytes ? 2 : 1);
    }
    const long long blocks = (long long)8 * P.per_xcd * P.n_co_tiles;
    P.total_blocks = (int)blocks;
    const size_t lds = (size_t)kWtChs * (2 * 4 * 16) + 2 * 2 * CO_TILE * sizeof(float);  // 128 KB of pieces + two scale | shift tables
    auto kern = conv3d_wt_kernel<CO_FR, PO_FR, RES, TAIL>;
    static const int once = [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return 0;
    }();
    (void)once;
    if (getenv("V2CE_WT_VERBOSE"))
        fprintf(stderr, "[wt<%d,%d,%d%s> %dx%dx%dx%d C %d -> %d] box %d pairs x %d x %d%s (%d of %d positions, %d halo elements), %lld tiles\n", CO_FR, PO_FR, RES, TAIL ? ",tail" : "", d.B, d.T,
                d.Hout, d.Wout, P.Cin, P.Cout, bx.pp, bx.th, bx.tw, bx.flat ? " rows: flat ranges" : "", P.n_pos, POS_TILE, P.plane, blocks);
    const unsigned grid = (unsigned)(blocks > n_cu ? n_cu : blocks);
#ifdef V2CE_STAMP
    V2CE_HIP_CHECK(hipMalloc(&P.stamps, (size_t)grid * 16 * sizeof(unsigned long long)));
    V2CE_HIP_CHECK(hipMemset(P.stamps, 0, (size_t)grid * 16 * sizeof(unsigned long long)));
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, P);
    V2CE_HIP_CHECK(hipGetLastError());
#ifdef V2CE_STAMP
    {
        std::vector<unsigned long long> h((size_t)grid * 16);
        V2CE_HIP_CHECK(hipDeviceSynchronize());
        V2CE_HIP_CHECK(hipMemcpy(h.data(), P.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        V2CE_HIP_CHECK(hipFree(P.stamps));
        auto mean = [&](int role, int a) {
            double sm = 0;
            for (unsigned k = 0; k < grid; ++k) sm += (double)h[((size_t)k * 2 + role) * 8 + a];
            return sm / grid;
        };
        const double tot = mean(0, 0), chunks = mean(0, 4), tiles = chunks / (P.Cin / 16);
        fprintf(stderr, "[stamp wt<%d,%d,%d> C %d -> %d %dx%d box %dx%dx%d] per workgroup: %.1f tiles, %.0f chunks, %.0f cycles (s_memtime, 100 MHz) | consumer: chunk barriers %.1f %%, "
                "epilogue %.1f %% (its barriers %.1f %%), main loop %.1f %% = %.1f ticks per chunk | producer: chunk barriers %.1f %%, rendezvous %.1f %%, convert %.1f %%\n",
                CO_FR, PO_FR, RES, P.Cin, P.Cout, d.Hout, d.Wout, bx.pp, bx.th, bx.tw, tiles, chunks, tot, 100 * mean(0, 1) / tot, 100 * mean(0, 3) / tot,
                100 * mean(0, 2) / tot, 100 * (tot - mean(0, 1) - mean(0, 3)) / tot, (tot - mean(0, 1) - mean(0, 3)) / chunks, 100 * mean(1, 1) / mean(1, 0),
                100 * mean(1, 2) / mean(1, 0), 100 * mean(1, 3) / mean(1, 0));
        fprintf(stderr, "    epilogue per tile (cycles): loads + dump (incl. the previous round's compute and stores) %.0f, first barrier .. reads %.0f, last compute + stores + range %.0f, barriers %.0f\n",
                mean(0, 5) / tiles, mean(0, 6) / tiles, mean(0, 7) / tiles, mean(0, 2) / tiles);
    }
#endif
    return V2CE_OK;
}

struct WtTail {
    const v2ce_conv3d_desc *desc;
    const float *tx0, *tx1;
    const int *thmap, *twmap;
    const void *w;
    const float *tx0_absmax, *tx1_absmax;
    int res_h, res_wp;            // > 0 with a residual: it is a 2x-upsampled low-resolution tensor of res_h rows at a row pitch of res_wp
};

int wt_dispatch(const v2ce_conv3d_desc *desc, const float *x, const void *w_wt, const float *scale, const float *shift,
                const float *residual, float *y, const float *x_absmax, float *y_absmax, v2ce_stream_t stream, const WtTail *tl = nullptr) {
    clear_error();
    V2CE_REQUIRE(desc && (g_wt_name_out || (x && w_wt && scale && shift && y)), V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt: null pointer");
    const v2ce_conv3d_desc &d = *desc;
    V2CE_REQUIRE(d.B > 0 && d.T > 0 && d.C0 > 0 && d.C1 == 0 && d.Hin > 0 && d.Win > 0 && d.Cout > 0, V2CE_ERR_BAD_ARG,
                 "v2ce_conv3d_fwd_wt: bad shape (one source: C1 must be 0)");
    V2CE_REQUIRE(d.ksize == 3 && d.stride_hw == 1 && d.precision == V2CE_PRECISION_F16X2 && d.layout == V2CE_LAYOUT_C16, V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd_wt: a split-half 3x3x3 stride-1 conv on channels-last-16 activations");
    V2CE_REQUIRE(d.H0 == d.Hin && d.W0 == d.Win && d.Hout == d.Hin && d.Wout == d.Win, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt: input and output planes must have one size");
    V2CE_REQUIRE(d.C0 % 16 == 0 && d.Cout % 64 == 0, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_wt: Cin must be a multiple of 16, Cout of 64");
    V2CE_REQUIRE(d.act >= 0 && d.act <= 2, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt: act %d", d.act);
    const int Winp = d.W0_pitch > 0 ? d.W0_pitch : d.W0, Woutp = d.Wout_pitch > 0 ? d.Wout_pitch : d.Wout;
    V2CE_REQUIRE(Winp >= d.Win && Woutp >= d.Wout, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt: a row pitch is smaller than its width");
    V2CE_REQUIRE((long long)d.T * d.C0 * d.Hin * Winp < (1ll << 29) && (long long)d.T * d.Cout * d.Hout * Woutp < (1ll << 29), V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd_wt: a single sequence exceeds the 2 GiB buffer-descriptor range");
    V2CE_REQUIRE(v2ce_pack_weights_f16x2_wt_bytes(d.Cout, d.C0) < (1ull << 31), V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_fwd_wt: the weight buffer exceeds the 2 GiB buffer-descriptor range");
    ConvParams P{};
    P.x0 = x; P.scale = scale; P.shift = shift; P.res = residual; P.y = y;
    P.B = d.B; P.T = d.T; P.C0 = d.C0; P.H0 = d.H0; P.W0 = d.W0; P.Hin = d.Hin; P.Win = d.Win;
    P.Cin = d.C0; P.Cout = d.Cout; P.Hout = d.Hout; P.Wout = d.Wout;
    P.W0p = Winp; P.Winp = Winp; P.Woutp = Woutp;
    P.c16 = 1;
    P.act = d.act;
    P.wq = static_cast<const _Float16 *>(w_wt);
    P.x0_absmax = x_absmax; P.y_absmax = y_absmax;
    P.guard = y_absmax ? y_absmax + 1 : nullptr;
    P.amax_bs = d.absmax_batch_stride;
    V2CE_REQUIRE(P.amax_bs == 0 || P.amax_bs >= 2, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt: absmax_batch_stride must be 0 or >= 2");
    hipStream_t st = as_stream(stream);
    if (tl) {
        // (the contract of v2ce_conv3d_fwd_tail, conv3d.hip)
        V2CE_REQUIRE(tl->desc && tl->w, V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt_tail: null tail description / weights");
        if (residual && tl->res_h > 0) {
            V2CE_REQUIRE(tl->res_h == (d.Hout + 1) / 2 && tl->res_wp >= (d.Wout + 1) / 2, V2CE_ERR_BAD_ARG,
                         "v2ce_conv3d_fwd_wt_tail: an upsampled residual has ceil(Hout / 2) rows of at least ceil(Wout / 2) columns");
            V2CE_REQUIRE((long long)d.T * d.Cout * tl->res_h * tl->res_wp < (1ll << 29), V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_wt_tail: residual too large");
            P.res_up = 1; P.rH = tl->res_h; P.rWp = tl->res_wp;
        }
        const v2ce_conv3d_desc &t = *tl->desc;
        V2CE_REQUIRE(t.ksize == 1 && (t.stride_hw == 1 || t.stride_hw == 2) && t.layout == V2CE_LAYOUT_C16 && t.B == d.B && t.T == d.T &&
                     t.Cout == d.Cout && t.Hout == d.Hout && t.Wout == d.Wout && t.C0 > 0 && t.C0 % 16 == 0 && t.C1 >= 0 &&
                     t.C1 % 16 == 0 && t.Hout == (t.Hin - 1) / t.stride_hw + 1 && t.Wout == (t.Win - 1) / t.stride_hw + 1,
                     V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt_tail: the tail must be a 1x1x1 conv (channels-last-16, channel counts multiples "
                     "of 16) producing exactly the main conv's output shape");
        V2CE_REQUIRE(((t.C0 + t.C1) / 16) % 4 == 0, V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_wt_tail: the tail's channel count must be a multiple of 64");
        V2CE_REQUIRE((g_wt_name_out || tl->tx0) && (t.C1 == 0 || tl->tx1 || g_wt_name_out) && (tl->thmap == nullptr) == (tl->twmap == nullptr) &&
                     (tl->thmap || (t.H0 == t.Hin && t.W0 == t.Win)), V2CE_ERR_BAD_ARG, "v2ce_conv3d_fwd_wt_tail: tail inputs / index maps");
        V2CE_REQUIRE((tl->tx0_absmax != nullptr) == (x_absmax != nullptr) && (t.C1 == 0 || !tl->tx0_absmax || tl->tx1_absmax), V2CE_ERR_BAD_ARG,
                     "v2ce_conv3d_fwd_wt_tail: range slots of the tail inputs");
        P.tx0 = tl->tx0; P.tx1 = t.C1 > 0 ? tl->tx1 : nullptr; P.thmap = tl->thmap; P.twmap = tl->twmap;
        P.tC0 = t.C0; P.tH0 = t.H0; P.tW0p = t.W0_pitch > 0 ? t.W0_pitch : t.W0; P.tC1 = t.C1; P.tHin = t.Hin; P.tWin = t.Win;
        P.tWinp = t.Win_pitch > 0 ? t.Win_pitch : t.Win; P.tS = t.stride_hw; P.tCG = (t.C0 + t.C1) / 16;
        P.tx0_absmax = tl->tx0_absmax; P.tx1_absmax = t.C1 > 0 ? tl->tx1_absmax : nullptr;
        P.sc_w = static_cast<const _Float16 *>(tl->w);
        V2CE_REQUIRE((long long)t.T * t.C0 * t.H0 * P.tW0p < (1ll << 29) && (long long)t.T * t.C1 * t.Hin * P.tWinp < (1ll << 29),
                     V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_fwd_wt_tail: a single sequence exceeds the 2 GiB buffer-descriptor range");
        if (residual) return launch_wt<2, 4, 1, true>(P, d, st);
        return launch_wt<2, 4, 0, true>(P, d, st);
    }
    if (residual || (g_wt_name_out && scale)) return launch_wt<2, 4, 1>(P, d, st);
    return launch_wt<2, 4, 0>(P, d, st);
}

}  // namespace

// internals for sn.hip: the Winograd planes of spectral-norm layers, all layers of a batch in one launch per pass
int v2ce_wt_pack_batch(const float *const *w, const float *const *sigma, void *const *packed, const int *rows, const int *cin, int n, int pass,
                       hipStream_t st, const int *cin_total, const int *ci0) {
    V2CE_REQUIRE(n >= 0 && n <= kWtMaxBatch, V2CE_ERR_BAD_ARG, "v2ce_wt_pack_batch: 0..%d layers", kWtMaxBatch);
    WtBatch B{};
    B.n = n;
    for (int l = 0; l < n; ++l) {
        V2CE_REQUIRE(rows[l] % 32 == 0 && cin[l] % 16 == 0, V2CE_ERR_UNSUPPORTED, "v2ce_wt_pack_batch: Cout %% 32, Cin %% 16");
        B.L[l] = WtLayer{w[l], sigma[l], static_cast<_Float16 *>(packed[l]), rows[l], cin[l], cin_total ? cin_total[l] : cin[l], ci0 ? ci0[l] : 0,
                         2ll * rows[l] * cin[l] * kWtTaps};
        if (pass == 0) {        // the bound is taken over the WHOLE tensor (what the spectral-norm batch knows): walk all channels, same tail
            B.L[l].cin = B.L[l].cin_total;
            B.L[l].ci0 = 0;
        }
        V2CE_REQUIRE(B.L[l].ci0 >= 0 && B.L[l].ci0 + cin[l] <= B.L[l].cin_total, V2CE_ERR_BAD_ARG, "v2ce_wt_pack_batch: channel slice out of range");
        B.blk[l + 1] = B.blk[l] + (rows[l] / 32) * (B.L[l].cin / 16);
    }
    if (n == 0) return V2CE_OK;
    return wt_pack_launch(B, pass, st);
}

}  // namespace v2ce

using namespace v2ce;

extern "C" size_t v2ce_pack_weights_f16x2_wt_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)2 * kWtTaps * Cout * Cin * sizeof(_Float16) + 16;
}

extern "C" int v2ce_pack_weights_f16x2_wt_slice(const float *w, int Cout, int Cin_total, int ci0, int Cin, const float *sigma, void *w_wt,
                                                v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(w && w_wt && Cout > 0 && Cin > 0 && ci0 >= 0 && ci0 + Cin <= Cin_total, V2CE_ERR_BAD_ARG, "v2ce_pack_weights_f16x2_wt: bad argument");
    V2CE_REQUIRE(Cout % 32 == 0 && Cin % 16 == 0 && ci0 % 16 == 0, V2CE_ERR_UNSUPPORTED,
                 "v2ce_pack_weights_f16x2_wt: Cout must be a multiple of 32, the channel slice of 16");
    hipStream_t st = as_stream(stream);
    unsigned char *tail = static_cast<unsigned char *>(w_wt) + (size_t)2 * kWtTaps * Cout * Cin * sizeof(_Float16);
    V2CE_HIP_CHECK(hipMemsetAsync(tail, 0, 16, st));
    void *pk = w_wt;
    int rc = v2ce_wt_pack_batch(&w, &sigma, &pk, &Cout, &Cin, 1, 0, st, &Cin_total, &ci0);
    if (rc != V2CE_OK) return rc;
    return v2ce_wt_pack_batch(&w, &sigma, &pk, &Cout, &Cin, 1, 1, st, &Cin_total, &ci0);
}

extern "C" int v2ce_pack_weights_f16x2_wt(const float *w, int Cout, int Cin, const float *sigma, void *w_wt, v2ce_stream_t stream) {
    return v2ce_pack_weights_f16x2_wt_slice(w, Cout, Cin, 0, Cin, sigma, w_wt, stream);
}

extern "C" int v2ce_conv3d_fwd_wt(const v2ce_conv3d_desc *desc, const float *x, const void *w_wt, const float *scale, const float *shift,
                                  const float *residual, float *y, const float *x_absmax, float *y_absmax, v2ce_stream_t stream) {
    g_wt_name_out = nullptr;
    return wt_dispatch(desc, x, w_wt, scale, shift, residual, y, x_absmax, y_absmax, stream);
}

extern "C" int v2ce_conv3d_fwd_wt_tail(const v2ce_conv3d_desc *desc, const float *x, const void *w_wt, const float *scale, const float *shift,
                                       float *y, const float *x_absmax, float *y_absmax, const v2ce_conv3d_desc *tail_desc, const float *tx0,
                                       const float *tx1, const int32_t *thmap, const int32_t *twmap, const void *tail_w,
                                       const float *tx0_absmax, const float *tx1_absmax, const float *residual, int res_h, int res_w_pitch,
                                       v2ce_stream_t stream) {
    g_wt_name_out = nullptr;
    const WtTail tl{tail_desc, tx0, tx1, thmap, twmap, tail_w, tx0_absmax, tx1_absmax, res_h, res_w_pitch};
    return wt_dispatch(desc, x, w_wt, scale, shift, residual, y, x_absmax, y_absmax, stream, &tl);
}

extern "C" int v2ce_conv3d_wt_variant(const v2ce_conv3d_desc *desc, int with_residual, char *name, size_t cap) {
    V2CE_REQUIRE(name && cap > 0, V2CE_ERR_BAD_ARG, "v2ce_conv3d_wt_variant: no buffer");
    name[0] = '\0';
    g_wt_name_out = name;
    g_wt_name_cap = cap;
    static const float dummy = 0.0f;
    int rc;
    if (with_residual == 2) {                              // the tail variant (any valid tail description names the same kernel)
        v2ce_conv3d_desc t = *desc;
        t.ksize = 1; t.stride_hw = 1; t.C0 = 64; t.C1 = 0; t.H0 = t.Hin = desc->Hout; t.W0 = t.Win = desc->Wout; t.layout = V2CE_LAYOUT_C16;
        const WtTail tl{&t, nullptr, nullptr, nullptr, nullptr, &dummy, nullptr, nullptr, 0, 0};
        rc = wt_dispatch(desc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &tl);
    } else if (with_residual == 3) {                       // the tail variant with a residual
        v2ce_conv3d_desc t = *desc;
        t.ksize = 1; t.stride_hw = 1; t.C0 = 64; t.C1 = 0; t.H0 = t.Hin = desc->Hout; t.W0 = t.Win = desc->Wout; t.layout = V2CE_LAYOUT_C16;
        const WtTail tl{&t, nullptr, nullptr, nullptr, nullptr, &dummy, nullptr, nullptr, 0, 0};
        rc = wt_dispatch(desc, nullptr, nullptr, nullptr, nullptr, &dummy, nullptr, nullptr, nullptr, nullptr, &tl);
    } else {
        rc = wt_dispatch(desc, nullptr, nullptr, with_residual ? &dummy : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    g_wt_name_out = nullptr;
    return rc;
}
