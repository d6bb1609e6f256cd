// conv3d_head.hip -- the UNet's first layer on gfx950 in split-half arithmetic:
// /root/reference/scripts/unet_2layer.py:341 (head = ConvLayer3D(2, 32, 3, padding 1), LeakyReLU) and
// /root/reference/scripts/submodules.py:96,115-124.
//
// Cin = 2 makes it a K = 54 problem that writes 16x what it reads (737 MB per 64 frame-pairs).  The exact-f32 head
// (conv3d.hip: conv3d_head_kernel, one lane = one position, 864 packed f32 FMAs with scalar weights) spends 0.26 of its
// 0.45 ms on arithmetic that does not overlap its own stores: 1.6 TB/s of output.  Here the arithmetic is 12 fp16 MFMAs
// per 32 positions -- every f32 operand split into two fp16 halves like all other layers of the default path (22-bit
// operands, f32 accumulation) -- and the kernel is what it should be, a stream of 1 KB stores.
//
// K is ORDERED for the MFMA's operand layout: a lane (position l32, half h) of a B fragment holds k = 8 h + j of a 16-wide
// k-step, so k-step s carries input channel ci = h, taps 8 s + j (j < 8; taps 27..31 are zero weights): a lane reads its
// eight values of a k-step from ONE channel plane of the halo box at eight compile-time tap offsets (ds_read_b32 with
// immediate offsets, no address arithmetic), converts them to the hi / lo halves and feeds three MFMAs.  Four k-steps.
// Workgroup = 4 waves on a (4, 4, 64) output box: wave = four (t, h) rows of two 32-position fragments; halo box
// 2 x 6 x 6 x 66 f32 in LDS.  Epilogue: bias, LeakyReLU, max |y| tracking, then the wave's fragment leaves through 4 KB of
// wave-private LDS as lane = (position, channel quarter): four 1 KB contiguous stores of the channels-last-16 layout.
#include "conv3d_dev.h"

namespace v2ce {
namespace {

constexpr int kHTT = 4, kHTH = 4, kHTW = 64;                 // output box
[[maybe_unused]] constexpr int kHHT = kHTT + 2, kHHH = kHTH + 2, kHHW = kHTW + 2, kHPitch = 68;      // halo box, row pitch in floats
[[maybe_unused]] constexpr int kHPlane = kHHT * kHHH * kHPitch;               // floats per input channel

struct HeadParams {
    const float *x;              // [B][T][2][H][W0p] planar
    const _Float16 *wt;          // table of v2ce_pack_head_weights_f16x2
    const float *bias;           // [32]
    float *y;                    // [B][T][2][H][Woutp][16]
    const float *x_absmax;       // per batch element (stride amax_bs), or NULL: |x| < 4094 required
    float *y_absmax;             // [b * amax_bs]: max |y|, + 1: range-guard value
    int B, T, H, W, W0p, Woutp, amax_bs;
    int nT, nH, nW, n_spatial, per_xcd;
    float slope;
};

#if defined(__HIP_DEVICE_COMPILE__)
__global__ __launch_bounds__(256) void conv3d_head_f16x2_kernel(HeadParams P) {
    __shared__ float halo[2 * kHPlane];                       // 19.6 KB
    __shared__ __attribute__((aligned(16))) float stage_all[4 * 1024];      // 4 KB per wave: the epilogue's transpose
    typedef float f32x4q __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = (int)(blockIdx.x & 7) * P.per_xcd + (int)(blockIdx.x >> 3);       // XCD x walks a contiguous range of boxes
    if (bid >= P.n_spatial) return;
    const int iw = bid % P.nW; bid /= P.nW;
    const int ih = bid % P.nH; bid /= P.nH;
    const int it = bid % P.nT;
    const int b = bid / P.nT;
    const int t0 = it * kHTT, h0 = ih * kHTH, w0 = iw * kHTW;

    // A fragments of the four k-steps (the whole layer's weights: 32 registers)
    f16x8 ah[4], al[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        ah[s] = *reinterpret_cast<const f16x8 *>(P.wt + ((s * 2 + 0) * 32 + l32) * 16 + 8 * half);
        al[s] = *reinterpret_cast<const f16x8 *>(P.wt + ((s * 2 + 1) * 32 + l32) * 16 + 8 * half);
    }
    const float *tail = reinterpret_cast<const float *>(P.wt + 4 * 2 * 32 * 16);        // { max |w|, pre-scale }
    const float w_scale = tail[1];
    const float amax = P.x_absmax ? P.x_absmax[b * P.amax_bs] : 4094.0f;
    const float x_scale = P.x_absmax ? pow2_prescale(amax) : kActScale;
    const float inv = 1.0f / (x_scale * w_scale);
    if (P.y_absmax && blockIdx.x == 0) {
        // range guard of this launch (include/v2ce_hip.h): K = 54 products per output, folded scale 1
        for (int e = tid; e < (P.amax_bs ? P.B : 1); e += 256) {
            const float am = P.x_absmax ? P.x_absmax[e * P.amax_bs] : 4094.0f;
            const float xs = P.x_absmax ? pow2_prescale(am) : kActScale;
            P.y_absmax[e * P.amax_bs + 1] = 54.0f * 0x1p-25f * (tail[0] / xs + am / w_scale);
        }
    }

    // halo box: 2 x 6 x 6 rows of 66 floats, zero padded
    const float *xb = P.x + (long long)b * P.T * 2 * (P.H * P.W0p);
    for (int row = wave; row < 2 * kHHT * kHHH; row += 4) {
        const int ci = row / (kHHT * kHHH), r = row - ci * (kHHT * kHHH);
        const int ht = r / kHHH, hh = r - ht * kHHH;
        const int t = t0 + ht - 1, h = h0 + hh - 1;
        const bool rok = t >= 0 && t < P.T && h >= 0 && h < P.H;
        const float *src = xb + ((long long)(rok ? t : 0) * 2 + ci) * (P.H * P.W0p) + (rok ? h : 0) * P.W0p;
        float *dst = halo + ci * kHPlane + (ht * kHHH + hh) * kHPitch;
        const int w = w0 + lane - 1;
        dst[lane] = (rok && w >= 0 && w < P.W) ? src[w] : 0.0f;
        if (lane < kHHW - 64) {
            const int w2 = w0 + 64 + lane - 1;
            dst[64 + lane] = (rok && w2 < P.W) ? src[w2] : 0.0f;
        }
    }
    __syncthreads();

    const long long seq = (long long)P.T * 32 * (P.H * P.Woutp);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y + b * seq, 0, (int)(seq * 4), 0x00020000);
    const int gstride = P.H * P.Woutp * 64;                   // bytes between the two 16-channel groups
    f32x4q *stage = reinterpret_cast<f32x4q *>(stage_all + wave * 1024);
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = P.bias[(r & 3) + 8 * (r >> 2) + 4 * half];
    unsigned ymax = 0u;
    const float *hp = halo + half * kHPlane + l32;             // this lane's channel plane, column l32

    // wave = rows 4 wave .. 4 wave + 3 of the box's 16 (t, h) rows, two fragments each
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr, tt = row / kHTH, th = row - tt * kHTH;
        const int t = t0 + tt, h = h0 + th;
        if (t >= P.T || h >= P.H) continue;                   // uniform
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            if (w0 + 32 * fr >= P.W) continue;                // uniform
            const float *p = hp + (tt * kHHH + th) * kHPitch + 32 * fr;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            step_loop<0, 4>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int tap = 8 * s + j;
                    if (tap < 27) {
                        const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
                        v[j] = p[(dt * kHHH + dh) * kHPitch + dw];
                    } else v[j] = 0.0f;
                }
                typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
                u32x4c ph, pl;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    unsigned hh_, ll_;
                    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
                        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
                        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
                        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(hh_), "=&v"(ll_) : "v"(v[2 * c2]), "v"(v[2 * c2 + 1]), "v"(x_scale));
                    ph[c2] = hh_;
                    pl[c2] = ll_;
                }
                const f16x8 bh = __builtin_bit_cast(f16x8, ph), bl = __builtin_bit_cast(f16x8, pl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, acc, 0, 0, 0);
            });
            // epilogue: bias + LeakyReLU, then lane = (position, quarter) through the wave's LDS: 1 KB contiguous per store
            const bool pok = w0 + 32 * fr + l32 < P.W;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4q out;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * r4 + k;
                    float y = acc[r] * inv + bias[r];
                    y = fmaxf(y, P.slope * y) + 0.0f;
                    out[k] = y;
                    const unsigned av = __builtin_bit_cast(unsigned, y) & (pok ? 0x7fffffffu : 0u);
                    ymax = av > ymax ? av : ymax;
                }
                stage[((r4 >> 1) * 32 + l32) * 4 + (r4 & 1) * 2 + half] = out;
            }
            const unsigned vrow = (unsigned)(4 * ((t * 32) * (P.H * P.Woutp)) + 64 * (h * P.Woutp + w0 + 32 * fr));
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x4q o = stage[(gq * 32 + 16 * k) * 4 + lane];
                    const bool ok = w0 + 32 * fr + 16 * k + (lane >> 2) < P.W;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, o), rs_y,
                                                           ok ? vrow + 1024u * (unsigned)k + 16u * (unsigned)lane : kOOB, gq * gstride, 0);
                    asm volatile("s_nop 1" : "+v"(o));        // store-data hazard of 16-byte stores, see conv_epilogue
                }
        }
    }
    if (P.y_absmax) absmax_commit(__builtin_bit_cast(float, ymax), P.y_absmax + b * P.amax_bs);
}
#else
__global__ void conv3d_head_f16x2_kernel(HeadParams) {}
#endif

// table[s][plane][co][8 h + j] = fp16 hi / lo of  scale * w[co][ci = h][tap = 8 s + j]  (taps >= 27: zero), then { max |w|, scale }
__global__ __launch_bounds__(256) void pack_head_weights_kernel(const float *__restrict__ w, _Float16 *__restrict__ table) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int i = threadIdx.x; i < 32 * 54; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float sc = pow2_prescale(red[0]);
    for (int e = threadIdx.x; e < 4 * 32 * 16; e += 256) {
        const int s = e >> 9, co = (e >> 4) & 31, kk = e & 15, h = kk >> 3, j = kk & 7, tap = 8 * s + j;
        const float v = tap < 27 ? w[(co * 2 + h) * 27 + tap] * sc : 0.0f;
        const _Float16 hi = (_Float16)v;
        table[((s * 2 + 0) * 32 + co) * 16 + kk] = hi;
        table[((s * 2 + 1) * 32 + co) * 16 + kk] = (_Float16)(v - (float)hi);
    }
    if (threadIdx.x == 0) {
        float *tail = reinterpret_cast<float *>(table + 4 * 2 * 32 * 16);
        tail[0] = red[0];
        tail[1] = sc;
    }
}

// slots[b * stride] = max |x[b][0 .. n)|   (slots zeroed by the caller)
__global__ __launch_bounds__(256) void absmax_batch_kernel(const float *__restrict__ x, long long n, float *slots, int stride) {
    const int b = blockIdx.y;
    const float *xb = x + (long long)b * n;
    float m = 0.0f;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {        // uniform: 16-byte loads
        // (four loads in flight per thread and enough workgroups to fill the chip: the kernel ran at 1.4 TB/s with one and 64 per
        // sequence -- 33 us in front of every forward)
        const long long step = (long long)gridDim.x * 1024;
        long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        for (; i + 3 * step < n; i += 4 * step) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4 *>(xb + i + k * step);
#pragma unroll
            for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
        }
        for (; i < n; i += step) {
            const float4 v = *reinterpret_cast<const float4 *>(xb + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(xb[i]));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(slots + (long long)b * stride), __float_as_uint(m));
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

extern "C" size_t v2ce_pack_head_weights_f16x2_bytes(void) { return 4 * 2 * 32 * 16 * 2 + 16; }

extern "C" int v2ce_pack_head_weights_f16x2(const float *w, void *table, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(w && table, V2CE_ERR_BAD_ARG, "v2ce_pack_head_weights_f16x2: null pointer");
    hipLaunchKernelGGL(pack_head_weights_kernel, dim3(1), dim3(256), 0, as_stream(stream), w, static_cast<_Float16 *>(table));
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" int v2ce_absmax_batch(const float *x, int B, long long n, float *slots, int stride, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(x && slots && B > 0 && n > 0 && stride >= 1, V2CE_ERR_BAD_ARG, "v2ce_absmax_batch: bad argument");
    const long long nb = (n / 4 + 255) / 256;
    const long long cap = B >= 8 ? 64 : 512 / B;            // ~512 workgroups in all
    hipLaunchKernelGGL(absmax_batch_kernel, dim3((unsigned)(nb < 1 ? 1 : (nb < cap ? nb : cap)), (unsigned)B), dim3(256), 0, as_stream(stream), x, n, slots, stride);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" int v2ce_conv3d_head_f16x2(const v2ce_conv3d_desc *desc, const float *x, const void *w_table, const float *bias,
                                      float *y, const float *x_absmax, float *y_absmax, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(desc && x && w_table && bias && y, V2CE_ERR_BAD_ARG, "v2ce_conv3d_head_f16x2: null pointer");
    const v2ce_conv3d_desc &d = *desc;
    V2CE_REQUIRE(d.B > 0 && d.T > 0 && d.C0 == 2 && d.C1 == 0 && d.Cout == 32 && d.ksize == 3 && d.stride_hw == 1 && d.H0 == d.Hin &&
                 d.W0 == d.Win && d.Hout == d.Hin && d.Wout == d.Win && d.layout == V2CE_LAYOUT_C16, V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_head_f16x2: the UNet's head is a 3x3x3 stride-1 conv of 2 planar input channels into 32 channels-last-16 ones");
    V2CE_REQUIRE(d.act == V2CE_ACT_LEAKY || d.act == V2CE_ACT_RELU || d.act == V2CE_ACT_NONE, V2CE_ERR_BAD_ARG, "v2ce_conv3d_head_f16x2: act %d", d.act);
    const int W0p = d.W0_pitch > 0 ? d.W0_pitch : d.W0, Woutp = d.Wout_pitch > 0 ? d.Wout_pitch : d.Wout;
    V2CE_REQUIRE(W0p >= d.W0 && Woutp >= d.Wout, V2CE_ERR_BAD_ARG, "v2ce_conv3d_head_f16x2: a row pitch is smaller than its width");
    V2CE_REQUIRE((long long)d.T * 32 * d.Hout * Woutp < (1ll << 29), V2CE_ERR_UNSUPPORTED,
                 "v2ce_conv3d_head_f16x2: an output sequence exceeds the 2 GiB buffer-descriptor range");
    V2CE_REQUIRE(d.absmax_batch_stride == 0 || d.absmax_batch_stride >= 2, V2CE_ERR_BAD_ARG, "v2ce_conv3d_head_f16x2: absmax_batch_stride must be 0 or >= 2");
    HeadParams P{};
    P.x = x; P.wt = static_cast<const _Float16 *>(w_table); P.bias = bias; P.y = y; P.x_absmax = x_absmax; P.y_absmax = y_absmax;
    P.B = d.B; P.T = d.T; P.H = d.Hin; P.W = d.Win; P.W0p = W0p; P.Woutp = Woutp; P.amax_bs = d.absmax_batch_stride;
    P.nT = (d.T + kHTT - 1) / kHTT; P.nH = (d.Hin + kHTH - 1) / kHTH; P.nW = (d.Win + kHTW - 1) / kHTW;
    P.n_spatial = d.B * P.nT * P.nH * P.nW;
    P.per_xcd = (P.n_spatial + 7) / 8;
    P.slope = d.act == V2CE_ACT_RELU ? 0.0f : (d.act == V2CE_ACT_LEAKY ? 0.01f : 1.0f);
    V2CE_REQUIRE((long long)P.n_spatial < (1ll << 28), V2CE_ERR_UNSUPPORTED, "v2ce_conv3d_head_f16x2: too many boxes");
    hipLaunchKernelGGL(conv3d_head_f16x2_kernel, dim3((unsigned)(8 * P.per_xcd)), dim3(256), 0, as_stream(stream), P);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
