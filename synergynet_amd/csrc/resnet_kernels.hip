// gfx950 kernels for the ResNet-50 backbone variant (BASELINE config 5; reference backbone_nets/resnet_backbone.py:90-136 Bottleneck, :139-254 ResNet).
// What is in this file, in the order of a forward (round 6):
//   resnet_stem_mfma_kernel   7x7/2 stem + BN + ReLU on v_mfma_f32_32x32x16_f16 from raw uint8 crops, the 3x3/2 max-pool in its epilogue
//                             (fp32 crops / small batches: resnet_stem_kernel + maxpool3x3s2_kernel)
//   conv_h2s_kernel           implicit GEMM, 128 pixels x 64 channels, weights through LDS chunks: layer1.0's conv1
//   conv_c3f_kernel           a bottleneck's conv2 (C2F) + conv3 + identity (or in-kernel downsample, DS) + ReLU + the NEXT bottleneck's conv1: layers 1 / 2
//   conv_lp_kernel            the LDS-tiled implicit GEMM as a pipeline (LDS-direct loads, three stages, two fragment sets): long-K convolutions of
//                             layers 2-4; DUAL: conv3 + stride-2 downsample branch as one GEMM (layer3.0 / 4.0)
//   conv_lt_kernel            the same tile with a register ring, two workgroups per CU: the short-K 1x1 convolutions (conv3, downsample)
//   conv_f16x2_kernel         64-bit addressing fallback (tensors of 2 GiB and more);  conv_kernel: exact fp32 MFMA (fp32 handles, weight-unsafe convolutions)
//   pool_fc_generic_kernel    average pool + the four heads + the verdict of the run-time range guard
// Every fp16 x2 kernel takes its operands as two fp16 pieces (fp32-class results: fused_block_f16.hip); activations travel between the kernels
// in the PAIR format below; one epilogue function (conv_epilogue) serves them all.
#include <cstdlib>
#include <type_traits>

#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// =====================================================================================
// Implicit-GEMM convolution.   out[m][n] = act( sum_{tap,k} in[pix(m,tap)][k] * W[n][tap*Cin + k] * scale[n] + shift[n] (+ res[m][n]) )
// Same operand convention as pointwise_kernel (backbone_kernels.hip): MFMA "A" rows = 16 output channels (weights,
// [Npad][KH*KW*Cin] row-major), MFMA "B" cols = 16 output pixels; a lane owns 4 consecutive channels of one pixel.
// (exact fp32-MFMA kernel, conv_kernel)  The K axis is walked tap by tap, 16 input channels at a time; a lane's operand fetch for tap (ky,kx) is one float4
// of the input pixel (oy*s-p+ky, ox*s-p+kx), or zeros outside the image (zero padding).  Operands go straight from
// L1/L2 to registers (no LDS): fp32 MFMA is slow enough that 8 x 1 KiB of operand traffic per 64 MFMAs is noise.
// Requires Cin % 16 == 0 and Cout % 4 == 0 (true for every ResNet-50 convolution after the stem).
// act: 0 none, 1 ReLU (applied AFTER the residual add: out = relu(bn3(conv3) + identity), resnet_backbone.py:130-134).
// =====================================================================================
// Run-time range guard of the fp16 x2 convolutions (ReLU networks have no static activation bound: synergy_abi.hip run_resnet50):
// every kernel whose output a later convolution takes as fp16 pieces folds max |output| into the status array of the forward --
// one v_max per output element, one wave reduction and ONE fire-and-forget atomic per wave.  Non-negative floats order like their
// bit patterns, so the atomic is an unsigned max.  Tens of thousands of waves report per launch, and device-scope atomics on ONE
// address serialise in that address's memory-side channel (measured: +100 us per convolution, ResNet-50 8.3 -> 13.7 ms; filtering
// them by a load of the current maximum costs an L2 round trip per wave instead: 9.2 - 11.6 ms), so a tensor has kRangeSub
// sub-slots kRangeStride floats apart and a wave reports into the one its workgroup index selects; pool_fc_generic_kernel takes the
// maximum over the sub-slots and poisons its results with NaN when a tensor left the fp16 window.
__device__ __forceinline__ void range_note(float *stat, float m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<unsigned *>(stat + (size_t)(blockIdx.x % kRangeSub) * kRangeStride), __builtin_bit_cast(unsigned, m));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// =====================================================================================
// The PAIR activation format (round 6).  An fp16 x2 convolution takes every operand as two fp16 pieces, x = hi + lo with
// hi = rtz_f16(x), lo = rtz_f16(x - hi) (22 significant bits, fused_block_f16.hip).  Rounds 2-5 kept the activations between the
// convolutions in fp32 and every CONSUMER split them -- a third of the vector instructions of the LDS-tiled GEMM's loop, repeated by every
// workgroup along the output-channel axis.  Now the PRODUCER's epilogue splits once and stores the pieces in the order the consumers'
// matrix instructions take them:
//   tensor [M pixels][C channels], C % 32 == 0: per pixel and 32-channel chunk kc 128 bytes = 32 high halves (64 B) | 32 low halves (64 B);
//   the B operand of v_mfma_f32_16x16x32_f16 for (pixel r, k-group g) of chunk kc is the 16 bytes at dword m C + 32 kc + 4 g, its low
//   piece the 16 bytes 16 dwords further.  The same 4 bytes per element as fp32, the same bits in the GEMM as the consumer-side split
//   (it is deterministic); what changes numerically is the identity branch of a bottleneck, which now adds hi + lo (|x - (hi + lo)| <=
//   2^-22 |x|) instead of x.
// So that a lane's accumulators ARE whole 16-byte pieces, the output channels of a 32-channel block sit on the MFMA rows in PAIR ORDER:
//   row rho of tile 2 b + ip  <->  channel 32 b + 8 (rho >> 2) + 4 ip + (rho & 3)   (weights packed so by synergy_abi.hip), i.e. lane group
//   g holds channels 32 b + 8 g + 0..3 in tile 2 b and + 4..7 in tile 2 b + 1: eight consecutive channels = k-group g of chunk b.
// Tensors no convolution consumes (the downsample branch, the last block's output) stay fp32; every epilogue takes the two formats
// (kOutPair / kResPair).  The exact fp32-MFMA convolution (conv_kernel: fp32 handles, weight-unsafe convolutions) reads and writes
// pairs too when it runs inside an fp16 x2 forward.
// =====================================================================================
constexpr int kOutPair = kFmtOutPair, kResPair = kFmtResPair;

__device__ __forceinline__ void split8(const f32x4 &x0, const f32x4 &x1, u32x4 (&pc)[2]) {
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
        const float x0 = x[2 * d], x1 = x[2 * d + 1];
        const unsigned a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
        pc[0][d] = a;
        pc[1][d] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
}
// four values -> their high and low fp16 pieces (two packed dwords each), the same arithmetic as split8
__device__ __forceinline__ void split4(const f32x4 &x, u32x2 &hi, u32x2 &lo) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float x0 = x[2 * d], x1 = x[2 * d + 1];
        const unsigned a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
        hi[d] = a;
        lo[d] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
}
// eight values back from their pieces: hi + lo is exact in fp32 (at most 22 significant bits)
__device__ __forceinline__ void unpair8(const u32x4 &hi, const u32x4 &lo, f32x4 &v0, f32x4 &v1) {
    const f16x8 h = __builtin_bit_cast(f16x8, hi), l = __builtin_bit_cast(f16x8, lo);
#pragma unroll
    for (int t = 0; t < 4; ++t) { v0[t] = (float)h[t] + (float)l[t]; v1[t] = (float)h[4 + t] + (float)l[4 + t]; }
}
__device__ __forceinline__ f32x4 unpair4(const u32x2 &hi, const u32x2 &lo) {
    // (scalars first: __builtin_bit_cast of a vector ELEMENT expression copies from the vector's address -- element 0 whatever the index)
    const unsigned hu0 = hi[0], hu1 = hi[1], lu0 = lo[0], lu1 = lo[1];
    const f16x2 h0 = __builtin_bit_cast(f16x2, hu0), h1 = __builtin_bit_cast(f16x2, hu1);
    const f16x2 l0 = __builtin_bit_cast(f16x2, lu0), l1 = __builtin_bit_cast(f16x2, lu1);
    return f32x4{(float)h0[0] + (float)l0[0], (float)h0[1] + (float)l0[1], (float)h1[0] + (float)l1[0], (float)h1[1] + (float)l1[1]};
}

// The epilogue every convolution kernel of this file ends with: acc[j][i] = tile i (pair order) of the lane's pixel mbase + 16 j + r16, the
// wave's channels nbase .. nbase + 16 NT - 1 (nbase % 32 == 0, NT even, N % 32 == 0).  out = act(acc x scale x inv_s + shift (+ residual)).
// In three passes: every load (BN scale / shift, residual) first and branch-free (rows clamped into the matrix), then the arithmetic, then
// the stores.  Interleaved, each residual load sat between two stores and its wait (vmcnt is in order, and conservative around an `if`)
// covered the previous store's acknowledgement: 8 serialised round trips per tile.
// A (pixel, tile) register quad is 16 bytes in either format: fp32 = channels 8 g + 4 ip + 0..3 of the block, pair = the lane's eight
// channels' high (ip = 0) or low (ip = 1) halves -- one 16-byte access per quad, at a format-dependent dword offset.
template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[MT][NT], const float *__restrict__ scale, const float *__restrict__ shift, float inv_s,
                                              const float *__restrict__ residual, float *__restrict__ out, int M, int N, int mbase, int nbase,
                                              int r16, int g, int act, int fmt, float *__restrict__ stat) {
    static_assert(NT % 2 == 0, "tile pairs");
    const bool out_pair = fmt & kOutPair, res_pair = fmt & kResPair;
    f32x4 scv[NT], shv[NT];
    u32x4 rraw[MT][NT];
    int qo_res[NT], qo_out[NT];                          // dword offset of quad i inside a pixel row
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int b32 = nbase + 32 * (i >> 1), ip = i & 1;
        qo_res[i] = b32 + (res_pair ? 4 * g + 16 * ip : 8 * g + 4 * ip);
        qo_out[i] = b32 + (out_pair ? 4 * g + 16 * ip : 8 * g + 4 * ip);
        const int n = b32 + 8 * g + 4 * ip;
        scv[i] = *(const f32x4 *)&scale[n] * inv_s;
        shv[i] = *(const f32x4 *)&shift[n];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            int m = mbase + j * 16 + r16;
            m = m < M ? m : 0;
            if (residual) rraw[j][i] = *(const u32x4 *)&residual[(size_t)m * N + qo_res[i]];     // (kernel-uniform condition)
        }
    }
    float vmax = 0.f;
#pragma unroll
    for (int p = 0; p < NT / 2; ++p)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            f32x4 v0 = acc[j][2 * p] * scv[2 * p] + shv[2 * p], v1 = acc[j][2 * p + 1] * scv[2 * p + 1] + shv[2 * p + 1];
            if (residual) {
                f32x4 r0, r1;
                if (res_pair) unpair8(rraw[j][2 * p], rraw[j][2 * p + 1], r0, r1);
                else { r0 = __builtin_bit_cast(f32x4, rraw[j][2 * p]); r1 = __builtin_bit_cast(f32x4, rraw[j][2 * p + 1]); }
                v0 += r0; v1 += r1;
            }
            if (act) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { v0[t] = fmaxf(v0[t], 0.0f); v1[t] = fmaxf(v1[t], 0.0f); }
            }
            if (stat && mbase + j * 16 + r16 < M)
                vmax = fmaxf(vmax, fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                                         fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3])))));
            if (out_pair) {
                u32x4 pc[2];
                split8(v0, v1, pc);
                acc[j][2 * p] = __builtin_bit_cast(f32x4, pc[0]);
                acc[j][2 * p + 1] = __builtin_bit_cast(f32x4, pc[1]);
            } else { acc[j][2 * p] = v0; acc[j][2 * p + 1] = v1; }
            asm volatile("" : "+v"(acc[j][2 * p]), "+v"(acc[j][2 * p + 1]));         // (keeps the arithmetic from being sunk into the store's branch)
        }
    if (stat) range_note(stat, vmax);                   // (kernel-uniform condition)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = mbase + j * 16 + r16;
            if (m >= M) continue;
            *(f32x4 *)&out[(size_t)m * N + qo_out[i]] = acc[j][i];
        }
}

// PAIR: the convolution runs inside an fp16 x2 forward -- its input is in the pair format (a lane rebuilds its four fp32 channels from two
// 8-byte pieces), its weight rows are taken in pair order and the epilogue writes what `fmt` says.  !PAIR: fp32 in and out (an fp32 handle).
template <int MT, int NT, bool PAIR>
__global__ __launch_bounds__(256) void conv_kernel(const float *__restrict__ in, const float *__restrict__ W,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   const float *__restrict__ residual, float *__restrict__ out, int M,
                                                   int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad,
                                                   int act, int n_tiles, int m_tiles, float *__restrict__ stat, int fmt) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt_idx = q % n_tiles;
    const int mt_idx = (q / n_tiles) * 8 + xcd;           // all channel tiles of one pixel tile share an XCD's L2
    if (mt_idx >= m_tiles) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = (mt_idx * 4 + wave) * (MT * 16);
    const int n0 = nt_idx * (NT * 16);
    if (m0 >= M) return;
    const int KCH = Cin >> 4, K = KH * KW * Cin, steps = KH * KW * KCH;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    int pb[MT], py[MT], px[MT];                           // batch index, top-left input coords of this lane's pixels
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        int m = m0 + j * 16 + r16;
        m = m < M ? m : M - 1;
        const int hw = Hout * Hout;
        pb[j] = m / hw;
        const int r = m - pb[j] * hw;
        py[j] = (r / Hout) * stride - pad;
        px[j] = (r % Hout) * stride - pad;
    }
    const float *wp[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        // MFMA row r16 of tile i: the channel in pair order (the epilogue's layout), W rows themselves are in natural order
        const int ch = n0 + 32 * (i >> 1) + 8 * (r16 >> 2) + 4 * (i & 1) + (r16 & 3);
        wp[i] = W + (size_t)ch * K + 4 * g;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = z4;

    auto fetch = [&](int s, f32x4(&wf)[NT], f32x4(&af)[MT]) {
        const int tap = s / KCH, kc = s - tap * KCH;
        const int ky = tap / KW, kx = tap - ky * KW;
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[i] = *(const f32x4 *)(wp[i] + tap * Cin + kc * 16);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int iy = py[j] + ky, ix = px[j] + kx;
            const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;
            const float *pp = in + ((size_t)(pb[j] * Hin + (ok ? iy : 0)) * Hin + (ok ? ix : 0)) * Cin;
            f32x4 v;
            if (PAIR) {   // channels 16 kc + 4 g + 0..3: chunk kc >> 1, halves 16 (kc & 1) + 4 g ... = dwords 8 (kc & 1) + 2 g, low piece 16 dwords on
                const float *q2 = pp + 32 * (kc >> 1) + 8 * (kc & 1) + 2 * g;
                v = unpair4(*(const u32x2 *)q2, *(const u32x2 *)(q2 + 16));
            } else v = *(const f32x4 *)(pp + kc * 16 + 4 * g);
            af[j] = ok ? v : z4;
        }
    };
    f32x4 wf[NT], af[MT], wn[NT], an[MT];
    fetch(0, wf, af);
    for (int s = 0; s < steps; ++s) {
        if (s + 1 < steps) fetch(s + 1, wn, an);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][t], af[j][t], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[i] = wn[i];
#pragma unroll
        for (int j = 0; j < MT; ++j) af[j] = an[j];
    }
    conv_epilogue<MT, NT>(acc, scale, shift, 1.0f, residual, out, M, N, m0, n0, r16, g, act, fmt, stat);
}

template <int MT, int NT>
static void launch_conv_t(const float *in, const float *W, const float *scale, const float *shift, const float *residual,
                          float *out, int M, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad,
                          int act, hipStream_t s, float *stat, int in_pair, int fmt) {
    const int n_tiles = N / (NT * 16);
    const int m_tiles = (M + 4 * MT * 16 - 1) / (4 * MT * 16);
    const int grid = ((m_tiles + 7) / 8) * n_tiles * 8;
    if (in_pair) conv_kernel<MT, NT, true><<<grid, 256, 0, s>>>(in, W, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, n_tiles, m_tiles, stat, fmt);
    else conv_kernel<MT, NT, false><<<grid, 256, 0, s>>>(in, W, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, n_tiles, m_tiles, stat, fmt);
}

// in_pair: the input is in the pair format (Cin % 32 == 0); fmt: kOutPair | kResPair.  Every ResNet-50 Cout is a multiple of 64 (required).
void launch_conv(const float *in, const float *W, const float *scale, const float *shift, const float *residual, float *out,
                 int B, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act, hipStream_t s, float *stat, int in_pair, int fmt) {
    const int M = B * Hout * Hout;
    const long tiles64 = ((long)M + 255) / 256 * (N / 64);
    if (tiles64 >= 2048) launch_conv_t<4, 4>(in, W, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, in_pair, fmt);
    else if (tiles64 >= 512) launch_conv_t<2, 4>(in, W, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, in_pair, fmt);
    else launch_conv_t<1, 4>(in, W, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, in_pair, fmt);
}

// =====================================================================================
// The same implicit GEMM on v_mfma_f32_16x16x32_f16 with fp32-equivalent accuracy: every operand as two fp16 pieces (x = a + b,
// 22 significant bits), three partial products per K=32 block -- see fused_block_f16.hip.  Weights are scaled by a power of two
// S (max |w| S in [2^13, 2^14)), split and lane-ordered offline, output channels in pair order (above):
//   W3[n_tile][tap*Cin/32 + kc][piece 2][lane][4 dwords], {S, 1/S};  lane (row l&15, k-group l>>4) holds k = 32*kc + 8*g + e.
// Activations arrive in the pair format: a lane's two operand pieces are two 16-byte loads, no arithmetic.  64-bit addressing (the one
// convolution kernel for tensors of 2 GiB and more).  Requires Cin % 32 == 0, N % 64 == 0.
// =====================================================================================
template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_f16x2_kernel(const float *__restrict__ in, const unsigned *__restrict__ W3,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       const float *__restrict__ residual, float *__restrict__ out, int M,
                                                       int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad,
                                                       int act, int n_tiles, int m_tiles, float *__restrict__ stat, int fmt) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt_idx = q % n_tiles;
    const int mt_idx = (q / n_tiles) * 8 + xcd;
    if (mt_idx >= m_tiles) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = (mt_idx * 4 + wave) * (MT * 16);
    const int n0 = nt_idx * (NT * 16);
    if (m0 >= M) return;
    const int KCH = Cin >> 5, steps = KH * KW * KCH;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 zu = {0u, 0u, 0u, 0u};
    int pb[MT], py[MT], px[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        int m = m0 + j * 16 + r16;
        m = m < M ? m : M - 1;
        const int hw = Hout * Hout;
        pb[j] = m / hw;
        const int r = m - pb[j] * hw;
        py[j] = (r / Hout) * stride - pad;
        px[j] = (r % Hout) * stride - pad;
    }
    const unsigned *wp[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wp[i] = W3 + (size_t)(n0 / 16 + i) * steps * 512 + lane * 4;
    const float inv_s = __builtin_bit_cast(float, W3[(size_t)(N / 16) * steps * 512 + 1]);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = z4;

    auto fetch = [&](int s, u32x4(&wf)[NT][2], u32x4(&bp)[MT][2]) {
        const int tap = s / KCH, kc = s - tap * KCH;
        const int ky = tap / KW, kx = tap - ky * KW;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) wf[i][p] = *(const u32x4 *)(wp[i] + ((size_t)s * 2 + p) * 256);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int iy = py[j] + ky, ix = px[j] + kx;
            const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;
            const float *p = in + ((size_t)(pb[j] * Hin + (ok ? iy : 0)) * Hin + (ok ? ix : 0)) * Cin + kc * 32 + 4 * g;
            const u32x4 v0 = *(const u32x4 *)p, v1 = *(const u32x4 *)(p + 16);
            bp[j][0] = ok ? v0 : zu;
            bp[j][1] = ok ? v1 : zu;
        }
    };
    auto mm = [](u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    };
    u32x4 wf[NT][2], wn[NT][2], bp[MT][2], bn[MT][2];
    fetch(0, wf, bp);
    for (int s = 0; s < steps; ++s) {
        if (s + 1 < steps) fetch(s + 1, wn, bn);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            constexpr int pa[3] = {1, 0, 0}, pbk[3] = {0, 0, 1};      // (w lo, a hi), (w hi, a hi), (w hi, a lo): conv_lt_kernel's order
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[j][i] = mm(wf[i][pa[t]], bp[j][pbk[t]], acc[j][i]);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) wf[i][p] = wn[i][p];
#pragma unroll
        for (int j = 0; j < MT; ++j) { bp[j][0] = bn[j][0]; bp[j][1] = bn[j][1]; }
    }
    conv_epilogue<MT, NT>(acc, scale, shift, inv_s, residual, out, M, N, m0, n0, r16, g, act, fmt, stat);
}

// Same convolution with the weight fragments shared through LDS: the four waves of a workgroup compute four M tiles of the SAME
// 64 output channels, so each of them loading all 8 KB of fragments per k32 step put 48 KB per step through the CU's vector
// cache (118 B/cycle wanted, 64 B/cycle there) for 96 MFMAs.  Here a chunk of KS k32 steps' fragments (NT x KS x 2 KB) is fetched
// once per workgroup -- a quarter per wave, into registers while the previous chunk computes, then into the other half of a
// double buffer -- and every wave reads its A operands from LDS; one barrier per chunk.
template <int MT, int NT, int KS>
__global__ __launch_bounds__(256) void conv_h2s_kernel(const float *__restrict__ in, const unsigned *__restrict__ W3,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       const float *__restrict__ residual, float *__restrict__ out, int M,
                                                       int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad,
                                                       int act, int n_tiles, int m_tiles, float *__restrict__ stat, unsigned in_bytes, int fmt) {
    constexpr int CH_DW = NT * KS * 2 * 256;                     // one chunk of fragments: [tile NT][step KS][piece 2][64][4]
    constexpr int NPW = NT * KS * 2 / 4;                         // fragments per wave and chunk
    static_assert(NT * KS * 2 % 4 == 0, "a quarter of a chunk per wave");
    __shared__ __attribute__((aligned(16))) unsigned wl[2 * CH_DW];
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt_idx = q % n_tiles;
    const int mt_idx = (q / n_tiles) * 8 + xcd;
    if (mt_idx >= m_tiles) return;                               // (workgroup-uniform)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = (mt_idx * 4 + wave) * (MT * 16);
    const int n0 = nt_idx * (NT * 16);
    const int KCH = Cin >> 5, steps = KH * KW * KCH, chunks = steps / KS;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // activations (pair format: the two pieces of a lane's k-group are 64 bytes apart) through buffer loads (byte offset = pixel base + a scalar per step; a tap outside the image gets an offset past the
    // end of the tensor and reads zeros = the padding): no select behind a load -- with one, the compiler waits for every load right
    // after issuing it and the whole memory latency stands in every step (conv_lt_kernel below)
    int py[MT], px[MT];
    unsigned pbase[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        int m = m0 + j * 16 + r16;
        m = m < M ? m : M - 1;                                   // (a wave past the end computes on clamped rows and stores nothing)
        const int hw = Hout * Hout;
        const int pbi = m / hw;
        const int r = m - pbi * hw;
        py[j] = (r / Hout) * stride - pad;
        px[j] = (r % Hout) * stride - pad;
        pbase[j] = (unsigned)(((pbi * Hin + py[j]) * Hin + px[j]) * Cin + 4 * g) * 4u;      // (wraps for padding rows: only used when the tap is inside)
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00027000);
    const float inv_s = __builtin_bit_cast(float, W3[(size_t)(N / 16) * steps * 512 + 1]);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = z4;

    // this wave's share of a chunk: fragments fi = wave + 4 k, fi = (tile i, step ks, piece p) in LDS order
    u32x4 wf[NPW];
    auto fetch_w = [&](int c) {
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int fi = wave + 4 * k, i = fi / (KS * 2), ks = (fi / 2) % KS, p = fi & 1;
            wf[k] = *(const u32x4 *)(W3 + ((size_t)(n0 / 16 + i) * steps + c * KS + ks) * 512 + p * 256 + lane * 4);
        }
    };
    auto park_w = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NPW; ++k) *(u32x4 *)&wl[buf * CH_DW + (wave + 4 * k) * 256 + lane * 4] = wf[k];
    };
    int f_s = 0, f_kc = 0, f_kx = 0, f_ky = 0;                   // the fetch pointer walks the steps in order (no division per step); it stops at the last one
    auto fetch_a = [&](u32x4(&bq)[MT][2]) {
        const unsigned dlt = (unsigned)((f_ky * Hin + f_kx) * Cin + f_kc * 32) * 4u;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int iy = py[j] + f_ky, ix = px[j] + f_kx;
            const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;
            const unsigned off = ok ? pbase[j] + dlt : 0x80000000u;
            bq[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
            bq[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 64, 0);
        }
        const int adv = f_s + 1 < steps ? 1 : 0;
        f_s += adv;
        f_kc += adv;
        const int c1 = f_kc == KCH ? 1 : 0;
        f_kc = c1 ? 0 : f_kc;
        f_kx += c1;
        const int c2 = f_kx == KW ? 1 : 0;
        f_kx = c2 ? 0 : f_kx;
        f_ky += c2;
    };
    auto mm = [](u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    };
    u32x4 bp[MT][2], bn[MT][2];
    fetch_w(0);
    fetch_a(bp);
    park_w(0);
    for (int c = 0; c < chunks; ++c) {
        // chunk c is in LDS, chunk c - 1 is read.  A raw barrier (__syncthreads() drains the activation loads in flight across it), and
        // no branches in the loop (past the end the last chunk is fetched / parked again, unread): the compiler's vmcnt bookkeeping
        // gives up at joins
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fetch_w(c + 1 < chunks ? c + 1 : c);
        const unsigned *wc = wl + (c & 1) * CH_DW + lane * 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            fetch_a(bn);                                         // step s + 1 (past the end: the last step again, unused)
            u32x4 wa[NT][2];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) wa[i][p] = *(const u32x4 *)(wc + ((i * KS + ks) * 2 + p) * 256);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                constexpr int pa[3] = {1, 0, 0}, pbk[3] = {0, 0, 1};      // (w lo, a hi), (w hi, a hi), (w hi, a lo): conv_lt_kernel's order
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) acc[j][i] = mm(wa[i][pa[t]], bp[j][pbk[t]], acc[j][i]);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) { bp[j][0] = bn[j][0]; bp[j][1] = bn[j][1]; }
        }
        park_w((c + 1) & 1);
    }
    if (m0 >= M) return;
    conv_epilogue<MT, NT>(acc, scale, shift, inv_s, residual, out, M, N, m0, n0, r16, g, act, fmt, stat);
}

template <int MT, int NT, int KS>
static void launch_conv_h2s_t(const float *in, const unsigned *W3, const float *scale, const float *shift, const float *residual,
                              float *out, int M, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act,
                              hipStream_t s, float *stat, int fmt) {
    const int n_tiles = (N + NT * 16 - 1) / (NT * 16);
    const int m_tiles = (M + 4 * MT * 16 - 1) / (4 * MT * 16);
    const int grid = ((m_tiles + 7) / 8) * n_tiles * 8;
    const unsigned in_bytes = (unsigned)((size_t)(M / (Hout * Hout)) * Hin * Hin * Cin * 4);
    conv_h2s_kernel<MT, NT, KS><<<grid, 256, 0, s>>>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad,
                                                     act, n_tiles, m_tiles, stat, in_bytes, fmt);
}

// =====================================================================================
// The deep layers (3 / 4: 55 % of the forward) on a workgroup tile of 64 MTW pixels x 128 output channels, BOTH operands through LDS.
// conv_h2s_kernel's tile is 64 (MT = 1) or 128 pixels x 64 channels: per k32 step a CU moves 128 B x (pixels + channels) through its
// vector cache (64 B/cycle) for 12 (pixels / 16)(channels / 16) matrix-pipe cycles -- 256 against 192 cycles for 64 x 64, more bytes than the
// cache delivers in the time the MFMAs take.  Here eight waves (4 along pixels x 2 along channels, 16 MTW pixels x 64 channels of accumulators each)
// share a step's operands: each wave fetches 1/8 of the step's activations (pair format, implicit-GEMM addressing as above) and 1/8 of its
// weight fragments into registers steps AHEAD and parks both as ready-made MFMA fragments in the
// other half of a double buffer ([pixel tile][piece][slot 64][4] / [channel tile][piece][lane][4]: a fragment read is a lane's own
// 16 bytes); one barrier per step.  Per step and CU (MTW = 4): 48 KB through the vector cache = 768 cycles against
// 1536 matrix-pipe cycles per SIMD.
// What made it fast (91 -> 55 us for layer 3's conv1; the tile alone: 91 vs conv_h2s_kernel's 92) is that NOTHING consumes a loaded
// register before the step that parks it: buffer loads whose out-of-range offsets return the padding zeros (a select behind the load
// made the compiler wait for it at once), a branch-free loop body (its vmcnt bookkeeping gives up at joins), a raw barrier, loads two
// steps ahead in a register ring.  DESIGN 7 has the ablations, the phase profile (tools/lt_prof.sh, -DLT_PROF=1) and what did not
// help (ping-pong wave halves, pinned MFMA / VALU interleave, a deeper ring).
// Same K order, same three products per step, same epilogue expression as conv_h2s_kernel: BIT-IDENTICAL results
// (tests/test_gpu_parity.py), so the two are interchangeable per convolution.  Requires Cin % 64 == 0 (an even number of k32 steps),
// N % 128 == 0, tensors below 2 GiB (32-bit buffer offsets).
// =====================================================================================
#ifndef LT_PROF
#define LT_PROF 0                      // 1: s_memtime sums per phase of waves 0 / 5 of one workgroup, printed per launch (tools/lt_prof.sh)
#endif
#define LT_LAP(i) do { if (LT_PROF) { __builtin_amdgcn_sched_barrier(0); tn = __builtin_amdgcn_s_memtime(); pt_[i] += tn - tk; tk = tn; __builtin_amdgcn_sched_barrier(0); } } while (0)
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F &&>(f));
    }
}
template <int MTW, bool FRAG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MTW == 2 ? 4 : 2, MTW == 2 ? 4 : 2)))
void conv_lt_kernel(const float *__restrict__ in, const unsigned *__restrict__ W3, const float *__restrict__ scale,
                    const float *__restrict__ shift, const float *__restrict__ residual, float *__restrict__ out, int M, int Hin,
                    int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act, int n_tiles, int m_tiles,
                    float *__restrict__ stat, unsigned in_bytes, int fmt) {
    constexpr int PT = 4 * MTW;                                  // 16-pixel tiles of the workgroup
    constexpr int U = MTW / 2;                                   // ... staged per wave
    static_assert(MTW == 2 || MTW == 4, "128 | 256 pixels per workgroup");
    constexpr int BF_DW = PT * 512, AF_DW = 8 * 512, ST_DW = BF_DW + AF_DW;      // one step: activation | weight fragments
    __shared__ __attribute__((aligned(16))) unsigned sm[2 * ST_DW];
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt_idx = q % n_tiles;
    const int mt_idx = (q / n_tiles) * 8 + xcd;                  // the channel tiles of one pixel tile share an XCD's L2
    if (mt_idx >= m_tiles) return;                               // (workgroup-uniform)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = mt_idx * (PT * 16);
    const int n0 = nt_idx * 128;
    const int KCH = Cin >> 5, steps = KH * KW * KCH;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // Staging (round 6).  In the pair format the 128 bytes of a (pixel, k32 chunk) are the high pieces of k-groups 0..3, then their low pieces:
    // every 16-byte chunk IS a fragment entry -- a staging lane copies it, no arithmetic.  A wave stages its U tiles with 2 U instructions per
    // step, in one of two shapes:
    //   FRAG  (3x3 convolutions: taps re-read from L2) a wave instruction fetches one 1 KB fragment PLANE -- lane (pixel r = l & 15, k-group
    //         gg = l >> 4) the 16 bytes at pixel row + 16 gg (+ 64: low piece) -- and parks it lane-linear: conflict-free writes and reads
    //         (linear 16-byte accesses fit the hardware's lane groups, MI355X_MICROARCH LDS table).  96 against 102 us for layer 3's conv2;
    //   !FRAG (1x1 convolutions: activations streamed from HBM) a wave instruction fetches 8 pixels x 128 B -- lane (pixel l >> 3 of half
    //         h, chunk sc = l & 7 = (piece sc >> 2, k-group sc & 3)): whole lines.  The plane-shaped loads (64-byte runs) cost these
    //         convolutions 7-10 % (layer 3 conv1 56 -> 61 us, layer 2.0 conv1 142 -> 158).  Its fragment entries go to slot
    //         gg 16 + ((r + 4 (gg >> 1) + 2 p) & 15) of the plane: the reads stay conflict-free in the lane groups ({0-3, 12-15, 20-27}, ...:
    //         k-groups 2 q and 2 q + 1 must share a rotation), the writes of an 8-lane group (one pixel: 4 k-groups x 2 pieces) are 2-way
    //         -- 16 LDS cycles under the 13 a ds_write_b128 takes to move its data anyway.  (Rounds 3-5: rotation 4 gg, reads 2-way.)
    // Buffer loads: a 32-bit byte offset per lane = pixel base + a scalar per (tap, k32 step); a tap outside the image takes an offset
    // past the end of the tensor, which the buffer hardware answers with zeros = the zero padding -- no select behind the load, so
    // nothing consumes a loaded register before the step that parks it (a v_cndmask there made the compiler wait for every load
    // right after issuing it).  in_bytes < 2^31 (launcher).
    const int sp = lane >> 3, sc = lane & 7;
    auto slot = [](int r, int gg, int p) { return gg * 16 + ((r + 4 * (gg >> 1) + 2 * p) & 15); };
    int py[U][2], px[U][2];                                       // [u][h]; FRAG: h = 0 only (one pixel per lane and tile)
    unsigned pbase[U][2];
    int woff[2];                                                  // dword offset inside a tile's two planes of what this lane parks ([h] | FRAG: [piece])
    int roff[2];                                                  // ... of this lane's fragment entry of piece p as a reader
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < (FRAG ? 1 : 2); ++h) {
            int m = m0 + (wave * U + u) * 16 + (FRAG ? r16 : 8 * h + sp);
            m = m < M ? m : M - 1;                               // (rows past the end: clamped loads, no stores)
            const int hw = Hout * Hout;
            const int pbi = m / hw;
            const int r = m - pbi * hw;
            py[u][h] = (r / Hout) * stride - pad;
            px[u][h] = (r % Hout) * stride - pad;
            pbase[u][h] = (unsigned)(((pbi * Hin + py[u][h]) * Hin + px[u][h]) * Cin + 4 * (FRAG ? g : sc)) * 4u;     // (wraps for padding rows: only used when the tap is inside)
        }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        woff[h] = FRAG ? h * 256 + lane * 4 : (sc >> 2) * 256 + slot(8 * h + sp, sc & 3, sc >> 2) * 4;
        roff[h] = FRAG ? lane * 4 : slot(r16, g, h) * 4;
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(W3), 0, 0x7fffffff, 0x00027000);
    const float inv_s = __builtin_bit_cast(float, W3[(size_t)(N / 16) * steps * 512 + 1]);
    f32x4 acc[MTW][4];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = z4;

    // Operands travel D steps ahead of their use in a register ring (slot = step % D): a step's loads are issued D barriers before the
    // step that parks them -- first-touch activations come from HBM (~2 us under load), one step of MFMAs is ~0.7 us.
    constexpr int D = 2;                                         // (4: the same time -- what mattered was that nothing waits for a load right behind it; PMC / variants in DESIGN 7)
    unsigned long long pt_[5] = {0, 0, 0, 0, 0}, tk = LT_PROF ? __builtin_amdgcn_s_memtime() : 0, tn;
    u32x4 sa[D][U][2];
    u32x4 sw[D][2];
    const unsigned wbase = (unsigned)((n0 / 16 + wave) * steps) * 2048u;        // this wave's channel tile: both pieces of a step (bytes)
    const unsigned l16 = lane * 16;
    // the fetch pointer walks the steps in order: (tap row, tap column, k32 chunk) as running scalars (a division per fetch cost ~400
    // cycles of a 3600-cycle step); past the last step it stays there (the last D fetches re-read it into slots nobody parks)
    int f_s = 0, f_kc = 0, f_kx = 0, f_ky = 0, w_s = 0;
    auto fetch = [&](auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        const unsigned dlt = (unsigned)((f_ky * Hin + f_kx) * Cin + f_kc * 32) * 4u;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < (FRAG ? 1 : 2); ++h) {
                const int iy = py[u][h] + f_ky, ix = px[u][h] + f_kx;
                const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;
                const unsigned off = ok ? pbase[u][h] + dlt : 0x80000000u;
                sa[SL][u][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
                if (FRAG) sa[SL][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 64, 0);
            }
        const int adv = f_s + 1 < steps ? 1 : 0;
        f_s += adv;
        f_kc += adv;
        const int c1 = f_kc == KCH ? 1 : 0;
        f_kc = c1 ? 0 : f_kc;
        f_kx += c1;
        const int c2 = f_kx == KW ? 1 : 0;
        f_kx = c2 ? 0 : f_kx;
        f_ky += c2;
    };
    auto fetch_w = [&](auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) sw[SL][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, l16, wbase + w_s * 2048 + p * 1024, 0);
        w_s += w_s + 1 < steps ? 1 : 0;
    };
    auto park = [&](int buf, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        unsigned *b = sm + buf * ST_DW;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) *(u32x4 *)(b + (wave * U + u) * 512 + woff[h]) = sa[SL][u][h];
#pragma unroll
        for (int p = 0; p < 2; ++p) *(u32x4 *)&b[BF_DW + (wave * 2 + p) * 256 + lane * 4] = sw[SL][p];
    };
    auto mm = [](u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    };
    u32x4 wa[4][2], bp[MTW][2];
    auto read_a = [&](int s) {                                   // operands of the first product: weights low, activations high
        const unsigned *b = sm + (s & 1) * ST_DW;
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[i][1] = *(const u32x4 *)(b + BF_DW + ((wn * 4 + i) * 2 + 1) * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < MTW; ++j) bp[j][0] = *(const u32x4 *)(b + ((wm * MTW + j) * 2 + 0) * 256 + roff[0]);
    };
    auto read_b = [&](int s) {                                   // ... the other two: weights high, activations low
        const unsigned *b = sm + (s & 1) * ST_DW;
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[i][0] = *(const u32x4 *)(b + BF_DW + ((wn * 4 + i) * 2 + 0) * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < MTW; ++j) bp[j][1] = *(const u32x4 *)(b + ((wm * MTW + j) * 2 + 1) * 256 + roff[1]);
    };
    // The three products of a step, and WHEN they run (round 6).  All eight waves pass the barrier together, so a step used to be two phases
    // in lockstep: every wave reading its 16 KB of fragments (LDS bandwidth: ~1200 cycles per CU with the parking writes) with the matrix
    // pipe idle, then every wave on the matrix pipe (1536 cycles per SIMD) with the LDS idle.  Now the third product of step s - 1
    // (weights high x activations low) is DEFERRED past the barrier of step s: it needs no register the first product of step s
    // (weights low x activations high) loads, so a wave issues the reads of those, runs the 16 deferred MFMAs under their latency, issues the
    // reads that overwrite the deferred product's operands (in flight under the deferred MFMAs' execution), parks, fetches, and goes on
    // with products one and two.  No extra registers; the accumulation order per step is (lo x hi), (hi x hi), (hi x lo) in every kernel.
    auto prod = [&](int pw, int pb_) {
#pragma unroll
        for (int j = 0; j < MTW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mm(wa[i][pw], bp[j][pb_], acc[j][i]);
    };
    // Raw barriers: __syncthreads() would drain the loads in flight.  No branches inside a step (the last steps park / fetch clamped
    // leftovers nobody reads): the compiler's vmcnt bookkeeping falls back to vmcnt(0) at every join.
#define LT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifndef SYN_LT_DEFER
#define SYN_LT_DEFER 1
#endif
#ifndef SYN_LT_PIN
#define SYN_LT_PIN 1
#endif
    // step s: slot_c = ring slot of step s + 1 (parked now), which then receives the loads of step s + 1 + D
    constexpr bool DEFER = SYN_LT_DEFER && MTW == 4;             // (128-pixel tiles: the 24 operand registers kept across the barrier do not fit into the 128 of two workgroups per CU)
    auto step = [&](int s, auto slot_c) {
        read_a(s);
        if (DEFER) {
            __builtin_amdgcn_sched_barrier(0);
            prod(0, 1);                                          // step s - 1: weights high x activations low (zeros before step 0)
            __builtin_amdgcn_sched_barrier(0);
        }
        read_b(s);
        if (LT_PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LT_LAP(4);
        park((s + 1) & 1, slot_c);
        LT_LAP(2);
        fetch(slot_c);
        fetch_w(slot_c);
        LT_LAP(3);
        if (DEFER) __builtin_amdgcn_sched_barrier(0);
        prod(1, 0);
        prod(0, 0);
        if (!DEFER) prod(0, 1);
        if (DEFER && SYN_LT_PIN) __builtin_amdgcn_sched_barrier(0);      // (else the compiler sinks these MFMAs past the next barrier)
    };
    if (DEFER) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[i][0] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < MTW; ++j) bp[j][1] = u32x4{0u, 0u, 0u, 0u};
    }
    // prologue: steps 0 .. D - 1 into their slots, step 0 parked, step D into the freed slot 0   (steps % D == 0: launcher)
    static_for<D>([&](auto d) { fetch(d); });
    static_for<D>([&](auto d) { fetch_w(d); });
    park(0, std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 0>{});
    fetch_w(std::integral_constant<int, 0>{});
    for (int s0 = 0; s0 < steps; s0 += D)
        static_for<D>([&](auto d) {
            constexpr int dd = decltype(d)::value;
            LT_LAP(0);
            LT_BARRIER();                                        // step s is in LDS; the other half (step s - 1) has been read
            LT_LAP(1);
            step(s0 + dd, std::integral_constant<int, (dd + 1) % D>{});
        });
    if (DEFER) prod(0, 1);                                       // the last step's third product
    LT_LAP(0);
    if (LT_PROF && blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 5) && steps >= 16)
        printf("lt<%d> M %d N %d steps %d wave %d: mfma %llu barrier %llu park %llu fetch %llu dsread %llu  (shader cycles per step)\n", MTW, M, N, steps, wave,
               pt_[0] / steps, pt_[1] / steps, pt_[2] / steps, pt_[3] / steps, pt_[4] / steps);
    const int mw = m0 + wm * (MTW * 16), nw = n0 + wn * 64;
    if (mw >= M) return;
    conv_epilogue<MTW, 4>(acc, scale, shift, inv_s, residual, out, M, N, mw, nw, r16, g, act, fmt, stat);
}

// =====================================================================================
// conv_lp_kernel (round 6): conv_lt_kernel's tile and fragment images, as a PIPELINE.  What conv_lt_kernel's phase profile shows (LT_PROF, layer 3's
// conv2: fragment reads 1150 + parking 400 + load issue 600 + matrix issue 900 + barrier 90 = 3190 cycles per step against 1536 on the matrix pipe)
// is eight waves in lockstep: LDS bandwidth (128 KB of fragment reads + 48 KB of parking writes per step and CU) and the vector memory
// pipe (48 KB per step) are busy while the matrix pipe idles, and the other way round.  Deferring a third of a step's MFMAs past the
// barrier changed nothing (100.3 vs 100.7 us): an in-order wave that is issuing MFMAs issues no memory instruction, so nothing new is in
// flight under them.  Memory work must be ISSUED before the MFMAs that cover it and must not need registers they use:
//   * both operands go global -> LDS directly (buffer_load ... lds, 1 KB per wave instruction into a lane-linear image): no register
//     ring, no parking writes, no vector instructions but the address arithmetic; three stages of 48 / 32 KB, a stage is refilled for step
//     s + 3 right after the barrier that ends step s - 1's reads of it: loads two steps ahead, counted vmcnt (4 / 6 per step stay in flight);
//   * the fragments of step s + 1 are read into a SECOND register set (the 48 registers of the ring pay for it) right after the barrier
//     of step s, before its 48 MFMAs: the reads run under them; one barrier per step as before.
// Activation fragment images: FRAG as conv_lt_kernel (a plane per instruction); !FRAG: whole 128-byte lines (8 pixels per instruction,
// lane = pixel l >> 3, 16-byte chunk (l & 7) ^ (pixel & 6) -- the swizzle sits in the per-lane GLOBAL address, the LDS image stays
// lane-linear), fragment entry (pixel r, k-group g, piece p) at 16-byte slot 64 (r >> 3) + 8 (r & 7) + ((4 p + g) ^ (r & 6)): conflict-free
// in the read lane groups.  Same K order and product order as conv_lt_kernel / conv_h2s_kernel: the same bits.
// =====================================================================================
// DUAL (round 6): a SECOND K segment behind the first -- a 1x1 convolution of another tensor (its own resolution, stride and width) onto the same
// output grid, accumulated into the same registers: out = act(W [a ; x] + shift).  With the BatchNorm scales folded into the weights that is a
// bottleneck's conv3 AND its stride-2 downsample branch in one GEMM (resnet_backbone.py:122-134, :127-128): the branch's fp32 tensor (134 MB written
// and read back in layer3.0) never exists and its launch is gone.  W3 then holds steps + steps2 k32 steps per channel tile.
struct DualSeg {
    const float *in2 = nullptr;     // second input [B, Hin2, Hin2, Cin2], pair format
    int Hin2 = 0, stride2 = 1, Cin2 = 0;
    unsigned in2_bytes = 0;
};
template <int MTW, bool FRAG, bool DUAL = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_lp_kernel(const float *__restrict__ in, const unsigned *__restrict__ W3, const float *__restrict__ scale,
                    const float *__restrict__ shift, const float *__restrict__ residual, float *__restrict__ out, int M, int Hin,
                    int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act, int n_tiles, int m_tiles,
                    float *__restrict__ stat, unsigned in_bytes, int fmt, DualSeg d2 = DualSeg{}) {
#if __HIP_DEVICE_COMPILE__      // (the HOST pass silently drops the kernel's launch stub when it has to instantiate this body -- the generic lambdas over device builtins; it only needs the signature)
    constexpr int PT = 4 * MTW, U = MTW / 2, NS = 3;
    static_assert(MTW == 2 || MTW == 4, "128 | 256 pixels per workgroup");
    constexpr int BF_DW = PT * 512, AF_DW = 8 * 512, ST_DW = BF_DW + AF_DW;      // one stage: activation | weight fragments
    constexpr int G = 2 * U + 2;                                                 // LDS-direct loads per wave and step
    __shared__ __attribute__((aligned(16))) unsigned sm[NS * ST_DW];             // (ONE shared object: a second one makes the compiler drain vmcnt before every ds_read)
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt_idx = q % n_tiles;
    const int mt_idx = (q / n_tiles) * 8 + xcd;                  // the channel tiles of one pixel tile share an XCD's L2
    if (mt_idx >= m_tiles) return;                               // (workgroup-uniform)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = mt_idx * (PT * 16);
    const int n0 = nt_idx * 128;
    const int KCH = Cin >> 5, steps1 = KH * KW * KCH, steps = steps1 + (DUAL ? d2.Cin2 >> 5 : 0);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int sp = lane >> 3, sc = lane & 7;
    int py[U][2], px[U][2];                                       // [u][h]; FRAG: h = 0 only
    unsigned pbase[U][2], pbase2[DUAL ? U : 1][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < (FRAG ? 1 : 2); ++h) {
            int m = m0 + (wave * U + u) * 16 + (FRAG ? r16 : 8 * h + sp);
            m = m < M ? m : M - 1;                               // (rows past the end: clamped loads, no stores)
            const int hw = Hout * Hout;
            const int pbi = m / hw;
            const int r = m - pbi * hw;
            py[u][h] = (r / Hout) * stride - pad;
            px[u][h] = (r % Hout) * stride - pad;
            const int chunk = FRAG ? g : (sc ^ (sp & 6));
            pbase[u][h] = (unsigned)(((pbi * Hin + py[u][h]) * Hin + px[u][h]) * Cin + 4 * chunk) * 4u;     // (wraps for padding rows: only used when the tap is inside)
            if constexpr (DUAL)                                  // the second segment's pixel: (oy stride2, ox stride2) of its own image, no padding
                pbase2[u][h] = (unsigned)(((pbi * d2.Hin2 + (r / Hout) * d2.stride2) * d2.Hin2 + (r % Hout) * d2.stride2) * d2.Cin2 + 4 * chunk) * 4u;
        }
    int roff[2];                                                  // dword offset inside a tile of this lane's fragment entry of piece p
#pragma unroll
    for (int p = 0; p < 2; ++p) roff[p] = FRAG ? p * 256 + lane * 4 : (r16 >> 3) * 256 + ((r16 & 7) * 8 + ((4 * p + g) ^ (r16 & 6))) * 4;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(DUAL ? d2.in2 : in), 0, (int)(DUAL ? d2.in2_bytes : in_bytes), 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(W3), 0, 0x7fffffff, 0x00027000);
    const float inv_s = __builtin_bit_cast(float, W3[(size_t)(N / 16) * steps * 512 + 1]);
    f32x4 acc[MTW][4];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = z4;
    const unsigned wbase = (unsigned)((n0 / 16 + wave) * steps) * 2048u;        // this wave's channel tile: both pieces of a step (bytes)
    const unsigned l16 = lane * 16;
    bool seg2 = false;                                            // DUAL: the step the fetch pointer is at lies in the second segment (uniform)
    // the fetch pointer walks the steps in order: (tap row, tap column, k32 chunk) as running scalars; past the last step it stays there
    // (the last loads re-fetch it into stages nobody reads).  A tap outside the image: an offset past the end of the tensor = zeros.
    int f_s = 0, f_kc = 0, f_kx = 0, f_ky = 0, w_s = 0;
    unsigned goff[U][2];                                          // byte offsets of the step the fetch pointer is at
    unsigned gw = 0;
    auto prepare = [&]() {                                       // the addresses of the fetch pointer's step; the pointer moves on
        const unsigned dlt = (unsigned)((f_ky * Hin + f_kx) * Cin + f_kc * 32) * 4u;
        if constexpr (DUAL) seg2 = f_s >= steps1;
        const unsigned dlt2 = DUAL ? (unsigned)(f_s - steps1) * 128u : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < (FRAG ? 1 : 2); ++h) {
                const int iy = py[u][h] + f_ky, ix = px[u][h] + f_kx;
                const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;
                goff[u][h] = ok ? pbase[u][h] + dlt : 0x80000000u;
                if constexpr (DUAL) goff[u][h] = seg2 ? pbase2[u][h] + dlt2 : goff[u][h];
            }
        const int adv = f_s + 1 < steps ? 1 : 0;
        f_s += adv;
        const int adv1 = DUAL ? (f_s < steps1 ? adv : 0) : adv;   // (the tap walk belongs to the first segment)
        f_kc += adv1;
        const int c1 = f_kc == KCH ? 1 : 0;
        f_kc = c1 ? 0 : f_kc;
        f_kx += c1;
        const int c2 = f_kx == KW ? 1 : 0;
        f_kx = c2 ? 0 : f_kx;
        f_ky += c2;
        gw = __builtin_amdgcn_readfirstlane(wbase + w_s * 2048);      // (uniform by construction; said so, or the prologue gets a waterfall loop)
        w_s += w_s + 1 < steps ? 1 : 0;
    };
    auto issue_k = [&](int so, auto k_c) {                       // load k of the G of a step into the stage at dword offset so
        constexpr int K = decltype(k_c)::value;
        if constexpr (K < 2 * U) {
            constexpr int u = K / 2, e = K % 2;                  // e: FRAG piece | row-shaped: pixel half
            const __amdgpu_buffer_rsrc_t rs = DUAL && seg2 ? rs_in2 : rs_in;      // (a scalar select of the descriptor: no branch around a load)
            if (FRAG) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sm + so + ((wave * U + u) * 2 + e) * 256), 16, goff[u][0], 64 * e, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sm + so + ((wave * U + u) * 2 + e) * 256), 16, goff[u][e], 0, 0, 0);
        } else {
            constexpr int pz = K - 2 * U;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sm + so + BF_DW + (wave * 2 + pz) * 256), 16, l16, gw + pz * 1024, 0, 0);
        }
    };
    auto issue = [&](int so) { prepare(); static_for<G>([&](auto k) { issue_k(so, k); }); };
    auto mm = [](u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    };
    u32x4 wa[2][4][2], bp[2][MTW][2];                             // two fragment sets: step parity
    constexpr int NR = 8 + 2 * MTW, NM = 12 * MTW;                // fragment reads / MFMAs of a wave and step
    auto read_k = [&](int so, auto set_c, auto k_c) {            // read k of the NR of a step: weights low, activations high, weights high, activations low
        constexpr int S = decltype(set_c)::value, K = decltype(k_c)::value;
        const unsigned *b = sm + so;
        if constexpr (K < 4) wa[S][K][1] = *(const u32x4 *)(b + BF_DW + ((wn * 4 + K) * 2 + 1) * 256 + lane * 4);
        else if constexpr (K < 4 + MTW) bp[S][K - 4][0] = *(const u32x4 *)(b + (wm * MTW + K - 4) * 512 + roff[0]);
        else if constexpr (K < 8 + MTW) wa[S][K - 4 - MTW][0] = *(const u32x4 *)(b + BF_DW + ((wn * 4 + K - 4 - MTW) * 2 + 0) * 256 + lane * 4);
        else bp[S][K - 8 - MTW][1] = *(const u32x4 *)(b + (wm * MTW + K - 8 - MTW) * 512 + roff[1]);
    };
    auto mfma_k = [&](auto set_c, auto k_c) {                    // MFMA k of the NM of a step: product-major, (w lo, a hi), (w hi, a hi), (w hi, a lo): every kernel's order
        constexpr int S = decltype(set_c)::value, K = decltype(k_c)::value;
        constexpr int t = K / (4 * MTW), j = (K % (4 * MTW)) / 4, i = K % 4;
        constexpr int pa[3] = {1, 0, 0}, pbk[3] = {0, 0, 1};
        acc[j][i] = mm(wa[S][i][pa[t]], bp[S][j][pbk[t]], acc[j][i]);
    };
    // A step, in G chunks: NM / G MFMAs of step s (registers of set SET_THIS), then ONE load of step s + 3 and NR / G fragment reads of step
    // s + 1 -- the loads cost an in-order wave 60 - 180 cycles of issue each (MI355X_MICROARCH: LDS-DMA issue cost); in a block in front of
    // the MFMAs they left the matrix pipe idle, behind 8 queued MFMAs each they are covered (SYN_LP_INTERLEAVE=0: the block form).
    // Before it: the stage of step s + 1 has landed (own loads: counted vmcnt -- G stay in flight -- then everybody's: barrier), this wave's
    // fragments of step s are in registers (lgkmcnt 0: every wave is done with the stage of step s before the barrier -> refill it).
#ifndef SYN_LP_INTERLEAVE
#define SYN_LP_INTERLEAVE 1
#endif
    auto step = [&](int o_this, int o_next, auto set_this, auto set_next) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(G) : "memory");
        prepare();
        if (SYN_LP_INTERLEAVE) {
            static_for<G>([&](auto c) {
                constexpr int C = decltype(c)::value;
                __builtin_amdgcn_sched_barrier(0);
                static_for<NM / G>([&](auto k) { mfma_k(set_this, std::integral_constant<int, C * (NM / G) + decltype(k)::value>{}); });
                __builtin_amdgcn_sched_barrier(0);
                issue_k(o_this, c);
                static_for<(C + 1) * NR / G - C * NR / G>([&](auto k) { read_k(o_next, set_next, std::integral_constant<int, C * NR / G + decltype(k)::value>{}); });
            });
            __builtin_amdgcn_sched_barrier(0);
        } else {
            static_for<G>([&](auto k) { issue_k(o_this, k); });
            static_for<NR>([&](auto k) { read_k(o_next, set_next, k); });
            static_for<NM>([&](auto k) { mfma_k(set_this, k); });
        }
    };
    int o0 = 0, o1 = ST_DW, o2 = 2 * ST_DW;
    issue(o0);
    issue(o1);
    issue(o2);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G) : "memory");    // stage 0 has landed
    static_for<NR>([&](auto k) { read_k(o0, std::integral_constant<int, 0>{}, k); });
    for (int s0 = 0; s0 < steps; s0 += 2) {                      // (steps is even: launcher)
        step(o0, o1, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        step(o1, o2, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        const int t = o2; o2 = o1; o1 = o0; o0 = t;              // stages of steps s + 2, s + 3, s + 4
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the clamped leftovers: nothing may land in LDS after the workgroup is gone)
    const int mw = m0 + wm * (MTW * 16), nw = n0 + wn * 64;
    if (mw >= M) return;
    conv_epilogue<MTW, 4>(acc, scale, shift, inv_s, residual, out, M, N, mw, nw, r16, g, act, fmt, stat);
#endif
}

template <int MTW>
static void launch_conv_lt_t(const float *in, const unsigned *W3, const float *scale, const float *shift, const float *residual,
                             float *out, int M, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act,
                             hipStream_t s, float *stat, int fmt) {
    const int n_tiles = N / 128;
    const int m_tiles = (M + 64 * MTW - 1) / (64 * MTW);
    const int grid = ((m_tiles + 7) / 8) * n_tiles * 8;
    const unsigned in_bytes = (unsigned)((size_t)(M / (Hout * Hout)) * Hin * Hin * Cin * 4);
    static const int shape = (int)test_knob("lt_stage", -1);      // A/B knob: 0 / 1 forces the row- / plane-shaped staging loads
    static const int glds = (int)test_knob("lt_glds", 1);         // A/B knob: 0 = conv_lt_kernel everywhere, 1 = conv_lp_kernel for 256-pixel tiles, 2 = for both
    const bool frag = shape < 0 ? KH * KW > 1 : shape != 0;       // plane-shaped loads where the taps re-read the activations from L2
    // 128-pixel tiles: three 32 KB stages are one workgroup per CU, which the short-K convolutions (conv3, downsample: their time is the epilogue)
    // pay for -- layer 3 conv3 89 -> 100 us -- and the long-K ones gain from (layer 4 conv1 / conv2 53 -> 48, 100 -> 93)
    const bool lp = MTW == 4 ? glds >= 1 : (glds >= 2 || (glds == 1 && KH * KW * (Cin / 32) >= 32));
#define SYN_LT_LAUNCH(K, F) K<MTW, F><<<grid, 512, 0, s>>>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, n_tiles, m_tiles, stat, in_bytes, fmt)
    if (lp) { if (frag) SYN_LT_LAUNCH(conv_lp_kernel, true); else SYN_LT_LAUNCH(conv_lp_kernel, false); }
    else { if (frag) SYN_LT_LAUNCH(conv_lt_kernel, true); else SYN_LT_LAUNCH(conv_lt_kernel, false); }
#undef SYN_LT_LAUNCH
}

// conv3 + stride-2 downsample branch of a bottleneck as ONE GEMM (conv_lp_kernel DUAL): a = T2 [B, H, H, C1] (1x1), x = the block input [B, Hin2, Hin2, Cin2] at
// stride2; W3d = the folded weights' fragments, [N/16][C1/32 + Cin2/32][2][64][4], {S, 1/S}; ones / shift: [N].  false: shape not served (the caller launches the two).
bool launch_conv_dual(const float *a, const float *x, const unsigned *W3d, const float *ones, const float *shift, float *out, int B, int H, int C1,
                      int Hin2, int stride2, int Cin2, int N, int act, hipStream_t s, float *stat, int fmt) {
    const int M = B * H * H, steps = (C1 + Cin2) / 32;
    if (C1 % 32 || Cin2 % 32 || N % 128 || steps % 2 || (long)((M + 255) / 256) * (N / 128) < 256) return false;
    if ((size_t)B * H * H * C1 * 4 >= (1ull << 31) || (size_t)B * Hin2 * Hin2 * Cin2 * 4 >= (1ull << 31) || (size_t)N * (C1 + Cin2) * 4 >= (1ull << 31)) return false;
    const int n_tiles = N / 128, m_tiles = (M + 255) / 256, grid = ((m_tiles + 7) / 8) * n_tiles * 8;
    DualSeg d2;
    d2.in2 = x; d2.Hin2 = Hin2; d2.stride2 = stride2; d2.Cin2 = Cin2; d2.in2_bytes = (unsigned)((size_t)B * Hin2 * Hin2 * Cin2 * 4);
    conv_lp_kernel<4, false, true><<<grid, 512, 0, s>>>(a, W3d, ones, shift, nullptr, out, M, H, H, C1, N, 1, 1, 1, 0, act, n_tiles, m_tiles, stat,
                                                        (unsigned)((size_t)B * H * H * C1 * 4), fmt, d2);
    return true;
}
// the (K, N1) combinations conv_c3f_kernel is instantiated for (launch_conv_c3f)
bool conv_c3f_supported(int K, int N3, int N1) { return !(N3 % 64 || N3 > 512) && ((K == 64 && (N1 == 64 || N1 == 128)) || (K == 128 && N1 == 128)); }

template <int MT, int NT>
static void launch_conv_f16x2_t(const float *in, const unsigned *W3, const float *scale, const float *shift, const float *residual,
                              float *out, int M, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act,
                              hipStream_t s, float *stat, int fmt) {
    const int n_tiles = N / (NT * 16);
    const int m_tiles = (M + 4 * MT * 16 - 1) / (4 * MT * 16);
    const int grid = ((m_tiles + 7) / 8) * n_tiles * 8;
    conv_f16x2_kernel<MT, NT><<<grid, 256, 0, s>>>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad,
                                                 act, n_tiles, m_tiles, stat, fmt);
}

// in: pair format; fmt: kOutPair | kResPair (the formats of out / residual).  N % 64 == 0, Cin % 64 == 0 (every ResNet-50 convolution behind the stem).
void launch_conv_f16x2(const float *in, const unsigned *W3, const float *scale, const float *shift, const float *residual,
                     float *out, int B, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act,
                     hipStream_t s, float *stat, int lt, int fmt) {
    const int lt_min_m = lt == 2 ? 1 : 4096;                      // (2 = cross-check mode: 128-pixel tiles on every shape that fits, ragged ones too)
    const int M = B * Hout * Hout;
    const long tiles = ((long)M + 127) / 128 * (N / 64);
    const int steps = KH * KW * (Cin / 32);
    const bool small = (size_t)B * Hin * Hin * Cin * 4 < (1ull << 31);       // 32-bit buffer offsets
    if (lt && N % 128 == 0 && M >= lt_min_m && steps % 2 == 0 && small && (size_t)N * KH * KW * Cin * 4 < (1ull << 31)) {
        // 256-pixel tiles (one 8-wave workgroup per CU) for long K when they still give every CU a workgroup; 128-pixel tiles (two workgroups
        // per CU: one's prologue / epilogue beside the other's steps) for K <= 512 -- conv3 and the downsample branches, whose time is their
        // epilogue (layer 3 conv3: 111 -> 96 us; conv1 / conv2 the other way: 56 -> 60, 106 -> 110)
        if (lt != 2 && steps > 16 && (long)((M + 255) / 256) * (N / 128) >= 256) launch_conv_lt_t<4>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, fmt);
        else launch_conv_lt_t<2>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, fmt);
        return;
    }
    if (steps % 2 == 0 && small) {
        // (64 pixels per wave or 128 channels per workgroup need > 256 registers = one wave per SIMD: 10.4 / 9.4 ms against 8.3)
        // 32-pixel wave tiles from 2048 workgroup tiles on: at 1024 (layer 3's conv1 / conv2: 256 x 4) they are 1.33 rounds of the 768 workgroups
        // the chip holds, the 16-pixel configuration's 2048 are 2.67 (B = 512: 6.96 -> 6.85 ms; from 4097 on instead: 7.06)
        if (tiles >= 2048) launch_conv_h2s_t<2, 4, 2>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, fmt);
        else launch_conv_h2s_t<1, 4, 2>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, fmt);
        return;
    }
    if (tiles >= 1024) launch_conv_f16x2_t<2, 4>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, fmt);
    else launch_conv_f16x2_t<1, 4>(in, W3, scale, shift, residual, out, M, Hin, Hout, Cin, N, KH, KW, stride, pad, act, s, stat, fmt);
}

// =====================================================================================
// conv3 + BN + identity + ReLU of a bottleneck FUSED with the NEXT bottleneck's conv1 + BN + ReLU (resnet_backbone.py:122-134 of
// block k, :116-118 of block k+1).  The convolutions of layer 1 are bound by memory throughput, not by the matrix pipe (counters in
// 56 % of the wave cycles in s_waitcnt, deeper prefetch buys nothing): conv3 writes the 256-channel block output, the next
// conv1 reads all of it back as its GEMM operand.  Here a workgroup owns 128 pixels and ALL output channels of conv3, in chunks of 32 / 64:
//   * the conv2 output of its pixels (K = 64 / 128 channels, pair format) is loaded once as ready-made B operands and stays in registers;
//   * per chunk: the chunk's output channels of conv3 (weights through a double-buffered LDS chunk, as in conv_h2s_kernel), BN, + identity,
//     ReLU, the split into the two fp16 pieces -- which are BOTH what is stored (the block output in the pair format) and, still in
//     registers, the B operand of the next block's conv1: in pair order the lane's eight channels of a 32-channel block are k-group g of
//     that k32 step, so conv1's weights are its ordinary fragments, step-major (W1f[k32 step][tile][piece 2][lane][4]); conv1 accumulates
//     over the chunks in registers;
//   * at the end BN + ReLU of conv1 and the store of the next block's 64 / 128-channel input (pair format).
// The block output is read back only as the next block's identity: 1.18 GB instead of 1.65 GB per bottleneck of layer 1 at B = 512, and
// one launch less.
// =====================================================================================
#ifndef SYN_C3F_L2_MT
#define SYN_C3F_L2_MT 2
#endif
#ifndef SYN_C3F_L1_MT
#define SYN_C3F_L1_MT 2                // 16-pixel tiles per wave in layer 1's fused launches (1: half the registers, three workgroups per CU)
#endif
#ifndef SYN_C3F_L1_TC
#define SYN_C3F_L1_TC 4                // output-channel tiles of conv3 per chunk in layer 1's middle blocks
#endif
// DS (layer1.0: 64 -> 256, stride 1): the block's downsample branch (1x1 conv + BN, resnet_backbone.py:127-128) is evaluated IN the kernel
// instead of being read as `identity`: the block input of the workgroup's pixels (64 channels) stays in registers as a second B operand, the
// downsample weights of a chunk come straight from L2 (natural K order, one tile ahead), and the 256-channel branch -- 472 MB written by a
// launch of its own and read back here at B = 512 -- never exists.
struct DsArgs {
    const float *X;             // block input [M, 64], pair format
    const unsigned *Wd;         // downsample conv's fp16 x2 fragments [N3/16][2][2][64][4], {S, 1/S}
    const float *scale_d, *shift_d;
};
// C2F (round 6, layer 1): the bottleneck's conv2 (3x3, 32 KS3 -> 32 KS3 channels, pad 1) runs IN FRONT, in the same launch: the workgroup's
// 128 pixels x 64 channels are accumulated exactly as conv_h2s_kernel<2, 4, 2> does it (weight chunks of two k32 steps through LDS -- the
// region conv3's and conv1's chunks use afterwards --, activations from T1 by buffer loads one step ahead, out-of-range offset = padding),
// BN + ReLU + split land in the registers conv3 takes its B operand from: T2 (118 MB written and read back per block at B = 512) never exists,
// one launch less, and the matrix-heavy front of one workgroup runs beside the HBM-bound back of the other workgroup of its CU.
template <int KS3, int NT1, int MT, int TC, bool DS = false, bool RP = true /* identity in the pair format (else fp32) */, bool C2F = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_c3f_kernel(const float *__restrict__ T2 /*[M, 32 KS3] pairs (unused with C2F)*/, const unsigned *__restrict__ W3 /*[N3/16][KS3][2][64][4], {S, 1/S}*/,
                     const float *__restrict__ scale3, const float *__restrict__ shift3, const float *__restrict__ identity /*[M, N3], pairs or fp32 (res_pair)*/,
                     float *__restrict__ out /*[M, N3] pairs*/, const unsigned *__restrict__ W1f /*[N3/32][NT1][2][64][4]*/, const float *__restrict__ s1 /*{S, 1/S}*/,
                     const float *__restrict__ scale1, const float *__restrict__ shift1, float *__restrict__ T1n /*[M, 16 NT1] pairs*/, int M, int N3,
                     int m_tiles, float *__restrict__ stat3, float *__restrict__ stat1, DsArgs ds = DsArgs{}, C2Args c2 = C2Args{}) {
    constexpr bool res_pair = RP;
    // TC = output-channel tiles of conv3 per chunk (4: 64 channels = two k32 steps of conv1; 2: 32 channels = one -- half the LDS per chunk
    // for the wider layers)
    constexpr int K = 32 * KS3, N1 = 16 * NT1, CW = 16 * TC, S1 = TC / 2;
    constexpr int W3C_DW = TC * KS3 * 2 * 256, W1C_DW = S1 * NT1 * 2 * 256;     // one chunk of conv3 / conv1 fragments
    constexpr int NP3 = TC * KS3 * 2 / 4, NP1 = S1 * NT1 * 2 / 4;                // fragments per wave and chunk
    static_assert(TC == 2 || TC == 4, "a chunk is one or two k32 steps of conv1");
    static_assert((TC * KS3 * 2) % 4 == 0 && (S1 * NT1 * 2) % 4 == 0, "a quarter of a chunk per wave");
    constexpr int NT2 = 2 * KS3, KSC = 2, CH2_DW = NT2 * KSC * 2 * 256, NPW2 = NT2 * KSC * 2 / 4;     // C2F: conv2's weight chunk (two k32 steps)
    static_assert(!C2F || 2 * CH2_DW <= 2 * W3C_DW + 2 * W1C_DW, "conv2's weight chunks live where conv3's and conv1's do afterwards");
    __shared__ __attribute__((aligned(16))) unsigned sm[2 * W3C_DW + 2 * W1C_DW + 1024];
    unsigned *w3l = sm, *w1l = sm + 2 * W3C_DW;
    float *sc3 = reinterpret_cast<float *>(sm + 2 * W3C_DW + 2 * W1C_DW), *sh3 = sc3 + 512;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int mt_idx = q * 8 + xcd;
    if (mt_idx >= m_tiles) return;                               // (workgroup-uniform)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = (mt_idx * 4 + wave) * (MT * 16);
    const int chunks = N3 / CW;
    const float inv_s3 = __builtin_bit_cast(float, W3[(size_t)(N3 / 16) * KS3 * 512 + 1]), inv_s1 = s1[1];
    for (int i = threadIdx.x; i < N3; i += 256) { sc3[i] = scale3[i] * inv_s3; sh3[i] = shift3[i]; }
    auto mm = [](u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // this wave's pixels (clamped: a wave past the end computes on the last pixel and stores nothing)
    int mp[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + j * 16 + r16;
        mp[j] = m < M ? m : M - 1;
    }
    u32x4 bp[MT][KS3][2];                                         // conv3's B operand: the conv2 output of these pixels as pieces
    if constexpr (C2F) {
        // ---- conv2: conv_h2s_kernel<MT, NT2, 2>'s loop (same K order, same product order: the same sums) ----
        const int Hin = c2.Hin, Hout = c2.Hout, steps2 = 9 * KS3, chunks2 = steps2 / KSC;
        int py[MT], px[MT];
        unsigned pbase[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int hw = Hout * Hout, pbi = mp[j] / hw, r = mp[j] - pbi * hw;
            py[j] = (r / Hout) * c2.stride - 1;
            px[j] = (r % Hout) * c2.stride - 1;
            pbase[j] = (unsigned)(((pbi * Hin + py[j]) * Hin + px[j]) * K + 4 * g) * 4u;      // (wraps for padding rows: only used when the tap is inside)
        }
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c2.T1), 0, (int)c2.in_bytes, 0x00027000);
        const float inv_s2 = __builtin_bit_cast(float, c2.W2[(size_t)NT2 * steps2 * 512 + 1]);
        f32x4 acc2[MT][NT2];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NT2; ++i) acc2[j][i] = z4;
        u32x4 wf[NPW2];
        auto fetch_w2 = [&](int c) {
#pragma unroll
            for (int k = 0; k < NPW2; ++k) {
                const int fi = wave + 4 * k, i = fi / (KSC * 2), ks = (fi / 2) % KSC, p = fi & 1;
                wf[k] = *(const u32x4 *)(c2.W2 + ((size_t)i * steps2 + c * KSC + ks) * 512 + p * 256 + lane * 4);
            }
        };
        auto park_w2 = [&](int buf) {
#pragma unroll
            for (int k = 0; k < NPW2; ++k) *(u32x4 *)&sm[buf * CH2_DW + (wave + 4 * k) * 256 + lane * 4] = wf[k];
        };
        int f_s = 0, f_kc = 0, f_kx = 0, f_ky = 0;               // the fetch pointer walks the steps in order; it stops at the last one
        auto fetch_a = [&](u32x4(&bq)[MT][2]) {
            const unsigned dlt = (unsigned)((f_ky * Hin + f_kx) * K + f_kc * 32) * 4u;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int iy = py[j] + f_ky, ix = px[j] + f_kx;
                const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;
                const unsigned off = ok ? pbase[j] + dlt : 0x80000000u;
                bq[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
                bq[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 64, 0);
            }
            const int adv = f_s + 1 < steps2 ? 1 : 0;
            f_s += adv;
            f_kc += adv;
            const int c1 = f_kc == KS3 ? 1 : 0;
            f_kc = c1 ? 0 : f_kc;
            f_kx += c1;
            const int cc = f_kx == 3 ? 1 : 0;
            f_kx = cc ? 0 : f_kx;
            f_ky += cc;
        };
        // activations TWO steps ahead in a ring of three register sets (first-touch rows of T1 come from HBM / the Infinity Cache: one step of
        // 24 MFMAs per wave does not cover that at two waves per SIMD; this phase has the registers to spare): six steps = three chunks per
        // trip of the loop, so that the ring index is a compile-time constant.  9 KS3 / 2 chunks: a multiple of three for KS3 = 2.
        static_assert((9 * KS3 / KSC) % 3 == 0, "three chunks per trip");
        u32x4 ring[3][MT][2];
        fetch_w2(0);
        fetch_a(ring[0]);
        fetch_a(ring[1]);
        park_w2(0);
        for (int c0 = 0; c0 < chunks2; c0 += 3)
            static_for<3>([&](auto cc_c) {
                constexpr int CC = decltype(cc_c)::value;
                const int c = c0 + CC;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (raw: __syncthreads() would drain the activation loads in flight)
                fetch_w2(c + 1 < chunks2 ? c + 1 : c);
                const unsigned *wc = sm + (c & 1) * CH2_DW + lane * 4;
                static_for<KSC>([&](auto ks_c) {
                    constexpr int ks = decltype(ks_c)::value, SL = (CC * KSC + ks) % 3;
                    fetch_a(ring[(SL + 2) % 3]);                 // step s + 2 (past the end: the last step again, unused)
                    u32x4 wa[NT2][2];
#pragma unroll
                    for (int i = 0; i < NT2; ++i)
#pragma unroll
                        for (int p = 0; p < 2; ++p) wa[i][p] = *(const u32x4 *)(wc + ((i * KSC + ks) * 2 + p) * 256);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        constexpr int pa[3] = {1, 0, 0}, pbk[3] = {0, 0, 1};
#pragma unroll
                        for (int j = 0; j < MT; ++j)
#pragma unroll
                            for (int i = 0; i < NT2; ++i) acc2[j][i] = mm(wa[i][pa[t]], ring[SL][j][pbk[t]], acc2[j][i]);
                    }
                });
                park_w2((c + 1) & 1);
            });
        // BN + ReLU + split: tile pair p of conv2's output = k32 step p of conv3 (pair order)
        float vmax2 = 0.f;
#pragma unroll
        for (int p = 0; p < KS3; ++p) {
            const int n = 32 * p + 8 * g;
            const f32x4 sa = *(const f32x4 *)&c2.scale2[n] * inv_s2, sb = *(const f32x4 *)&c2.scale2[n + 4] * inv_s2;
            const f32x4 ha = *(const f32x4 *)&c2.shift2[n], hb = *(const f32x4 *)&c2.shift2[n + 4];
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                f32x4 v0 = acc2[j][2 * p] * sa + ha, v1 = acc2[j][2 * p + 1] * sb + hb;
#pragma unroll
                for (int t = 0; t < 4; ++t) { v0[t] = fmaxf(v0[t], 0.0f); v1[t] = fmaxf(v1[t], 0.0f); }
                if (m0 + j * 16 + r16 < M) vmax2 = fmaxf(vmax2, fmaxf(fmaxf(fmaxf(v0[0], v0[1]), fmaxf(v0[2], v0[3])), fmaxf(fmaxf(v1[0], v1[1]), fmaxf(v1[2], v1[3]))));
                split8(v0, v1, bp[j][p]);
            }
        }
        if (c2.stat2) range_note(c2.stat2, vmax2);             // (kernel-uniform condition)
        __syncthreads();                                         // every wave is done with conv2's weight chunks: the region becomes conv3's / conv1's
    }

    u32x4 pf3[NP3], pf1[NP1];
    auto fetch_w3 = [&](int c) {
#pragma unroll
        for (int k = 0; k < NP3; ++k) pf3[k] = *(const u32x4 *)(W3 + (size_t)c * W3C_DW + (wave + 4 * k) * 256 + lane * 4);
    };
    auto park_w3 = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NP3; ++k) *(u32x4 *)&w3l[buf * W3C_DW + (wave + 4 * k) * 256 + lane * 4] = pf3[k];
    };
    auto fetch_w1 = [&](int c) {
#pragma unroll
        for (int k = 0; k < NP1; ++k) pf1[k] = *(const u32x4 *)(W1f + (size_t)c * W1C_DW + (wave + 4 * k) * 256 + lane * 4);
    };
    auto park_w1 = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NP1; ++k) *(u32x4 *)&w1l[buf * W1C_DW + (wave + 4 * k) * 256 + lane * 4] = pf1[k];
    };
    fetch_w3(0);
    fetch_w1(0);
    if constexpr (!C2F) {                                         // the conv2 output of this wave's pixels: the pieces as stored
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) {
                const float *p = T2 + (size_t)mp[j] * K + ks * 32 + 4 * g;
                bp[j][ks][0] = *(const u32x4 *)p;
                bp[j][ks][1] = *(const u32x4 *)(p + 16);
            }
    }
    // DS: the block input of the same pixels (K = 64: two k32 steps) and the branch's scale
    u32x4 xd[DS ? MT : 1][2][2];
    float inv_sd = 0.f;
    if constexpr (DS) {
        inv_sd = __builtin_bit_cast(float, ds.Wd[(size_t)(N3 / 16) * 2 * 512 + 1]);
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float *p = ds.X + (size_t)mp[j] * 64 + ks * 32 + 4 * g;
                xd[j][ks][0] = *(const u32x4 *)p;
                xd[j][ks][1] = *(const u32x4 *)(p + 16);
            }
    }
    park_w3(0);
    park_w1(0);
    f32x4 acc1[MT][NT1];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT1; ++i) acc1[j][i] = z4;
    float vmax3 = 0.f;
    // dword offset inside a pixel row of (pixel, tile i of the chunk)'s register quad: see conv_epilogue
    int qo_res[TC], qo_out[TC], qn[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        qo_res[i] = 32 * (i >> 1) + (res_pair ? 4 * g + 16 * (i & 1) : 8 * g + 4 * (i & 1));
        qo_out[i] = 32 * (i >> 1) + 4 * g + 16 * (i & 1);
        qn[i] = 32 * (i >> 1) + 8 * g + 4 * (i & 1);          // the quad's first channel (pair order)
    }

    for (int c = 0; c < chunks; ++c) {
        __syncthreads();                                         // chunk c of both weight sets is in LDS, chunk c-1 is read
        // identity of this chunk: requested now, consumed after the conv3 MFMAs
        f32x4 rsv[MT][TC];
        if constexpr (!DS) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < TC; ++i) rsv[j][i] = *(const f32x4 *)&identity[(size_t)mp[j] * N3 + CW * c + qo_res[i]];
        }
        if (c + 1 < chunks) fetch_w3(c + 1);
        if constexpr (DS) {
            // the downsample branch of this chunk: rsv = BN_d(Wd . x); fragments of tile i + 1 are requested while tile i is on the matrix pipe
            u32x4 wd[2][2][2];
            auto fetch_d = [&](int i, u32x4 (&w)[2][2]) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = 0; p < 2; ++p) w[ks][p] = *(const u32x4 *)(ds.Wd + ((size_t)((TC * c + i) * 2 + ks) * 2 + p) * 256 + lane * 4);
            };
            fetch_d(0, wd[0]);
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                if (i + 1 < TC) fetch_d(i + 1, wd[(i + 1) & 1]);
                const f32x4 scd = *(const f32x4 *)&ds.scale_d[CW * c + qn[i]] * inv_sd, shd = *(const f32x4 *)&ds.shift_d[CW * c + qn[i]];
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    f32x4 a = z4;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        a = mm(wd[i & 1][ks][1], xd[j][ks][0], a);
                        a = mm(wd[i & 1][ks][0], xd[j][ks][1], a);
                        a = mm(wd[i & 1][ks][0], xd[j][ks][0], a);
                    }
                    rsv[j][i] = a * scd + shd;
                }
            }
        }
        // ---- conv3: the CW output channels of this chunk ----
        f32x4 acc3[MT][TC];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < TC; ++i) acc3[j][i] = z4;
        const unsigned *w3c = w3l + (c & 1) * W3C_DW + lane * 4;
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks)
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                u32x4 wa[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) wa[p] = *(const u32x4 *)(w3c + ((i * KS3 + ks) * 2 + p) * 256);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    acc3[j][i] = mm(wa[1], bp[j][ks][0], acc3[j][i]);
                    acc3[j][i] = mm(wa[0], bp[j][ks][1], acc3[j][i]);
                    acc3[j][i] = mm(wa[0], bp[j][ks][0], acc3[j][i]);
                }
            }
        if (c + 1 < chunks) { park_w3((c + 1) & 1); fetch_w1(c + 1); }
        // ---- BN, + identity, ReLU, split: the pieces are the stored block output AND conv1's operand ----
        u32x4 op[MT][S1][2];
#pragma unroll
        for (int sk = 0; sk < S1; ++sk) {
            const f32x4 scv0 = *(const f32x4 *)&sc3[CW * c + qn[2 * sk]], shv0 = *(const f32x4 *)&sh3[CW * c + qn[2 * sk]];
            const f32x4 scv1 = *(const f32x4 *)&sc3[CW * c + qn[2 * sk + 1]], shv1 = *(const f32x4 *)&sh3[CW * c + qn[2 * sk + 1]];
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                f32x4 v0 = acc3[j][2 * sk] * scv0 + shv0, v1 = acc3[j][2 * sk + 1] * scv1 + shv1;
                f32x4 r0, r1;
                if (!DS && res_pair) unpair8(__builtin_bit_cast(u32x4, rsv[j][2 * sk]), __builtin_bit_cast(u32x4, rsv[j][2 * sk + 1]), r0, r1);
                else { r0 = rsv[j][2 * sk]; r1 = rsv[j][2 * sk + 1]; }
                v0 += r0; v1 += r1;
#pragma unroll
                for (int t = 0; t < 4; ++t) { v0[t] = fmaxf(v0[t], 0.0f); v1[t] = fmaxf(v1[t], 0.0f); }
                split8(v0, v1, op[j][sk]);
                asm volatile("" : "+v"(op[j][sk][0]), "+v"(op[j][sk][1]));
                if (m0 + j * 16 + r16 < M) {
                    float *dst = &out[(size_t)(m0 + j * 16 + r16) * N3 + CW * c];
                    *(u32x4 *)(dst + qo_out[2 * sk]) = op[j][sk][0];
                    *(u32x4 *)(dst + qo_out[2 * sk + 1]) = op[j][sk][1];
                    vmax3 = fmaxf(vmax3, fmaxf(fmaxf(fmaxf(v0[0], v0[1]), fmaxf(v0[2], v0[3])), fmaxf(fmaxf(v1[0], v1[1]), fmaxf(v1[2], v1[3]))));
                }
            }
        }
        // ---- conv1 of the next block: K = this chunk's channels (S1 k32 steps) ----
        const unsigned *w1c = w1l + (c & 1) * W1C_DW + lane * 4;
#pragma unroll
        for (int sk = 0; sk < S1; ++sk) {
#pragma unroll
            for (int i = 0; i < NT1; ++i) {
                u32x4 wa[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) wa[p] = *(const u32x4 *)(w1c + ((sk * NT1 + i) * 2 + p) * 256);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    acc1[j][i] = mm(wa[1], op[j][sk][0], acc1[j][i]);
                    acc1[j][i] = mm(wa[0], op[j][sk][1], acc1[j][i]);
                    acc1[j][i] = mm(wa[0], op[j][sk][0], acc1[j][i]);
                }
            }
        }
        if (c + 1 < chunks) park_w1((c + 1) & 1);
    }
    if (stat3) range_note(stat3, vmax3);                 // (kernel-uniform conditions)
    // ---- conv1: BN + ReLU, store the next block's input (pair format) ----
    conv_epilogue<MT, NT1>(acc1, scale1, shift1, inv_s1, nullptr, T1n, M, N1, m0, 0, r16, g, 1, kOutPair, stat1);
}

template <int KS3, int NT1, int MT, int TC>
static void launch_c3f_t(const float *T2, const unsigned *W3, const float *scale3, const float *shift3, const float *identity, float *out,
                         const unsigned *W1f, const float *s1, const float *scale1, const float *shift1, float *T1n, int M, int N3, hipStream_t s,
                         float *stat3, float *stat1, int res_pair, const C2Args *c2) {
    const int m_tiles = (M + 4 * MT * 16 - 1) / (4 * MT * 16), grid = ((m_tiles + 7) / 8) * 8;
#define SYN_C3F_GO(RP, C2F, C2V) conv_c3f_kernel<KS3, NT1, MT, TC, false, RP, C2F><<<grid, 256, 0, s>>>(T2, W3, scale3, shift3, identity, out, W1f, s1, scale1, shift1, T1n, M, N3, m_tiles, stat3, stat1, DsArgs{}, C2V)
    if constexpr (KS3 == 2 || KS3 == 4) {                        // (conv2 in front: the 64- and 128-channel bottlenecks of layers 1 / 2)
        if (c2) { if (res_pair) SYN_C3F_GO(true, true, *c2); else SYN_C3F_GO(false, true, *c2); return; }
    }
    if (res_pair) SYN_C3F_GO(true, false, C2Args{}); else SYN_C3F_GO(false, false, C2Args{});
#undef SYN_C3F_GO
}

// ... with the block's downsample branch evaluated in the kernel (DS above): layer1.0.  Every tensor in the pair format.  c2 != nullptr: conv2 in front (T2 unused).
bool launch_conv_c3f_ds(const float *T2, const unsigned *W3, const float *scale3, const float *shift3, const float *X, const unsigned *Wd,
                        const float *scale_d, const float *shift_d, float *out, const unsigned *W1f, const float *s1, const float *scale1,
                        const float *shift1, float *T1n, int M, int K, int Kd, int N3, int N1, hipStream_t s, float *stat3, float *stat1, const C2Args *c2) {
    if (K != 64 || Kd != 64 || N3 != 256 || N1 != 64) return false;
    const DsArgs ds{X, Wd, scale_d, shift_d};
    constexpr int MT = SYN_C3F_L1_MT;
    const int m_tiles = (M + 64 * MT - 1) / (64 * MT), grid = ((m_tiles + 7) / 8) * 8;
    // (32-channel chunks: with 64 the second B operand spills)
    if (c2) conv_c3f_kernel<2, 4, MT, 2, true, false, true><<<grid, 256, 0, s>>>(T2, W3, scale3, shift3, nullptr, out, W1f, s1, scale1, shift1, T1n, M, N3, m_tiles, stat3, stat1, ds, *c2);
    else conv_c3f_kernel<2, 4, MT, 2, true, false><<<grid, 256, 0, s>>>(T2, W3, scale3, shift3, nullptr, out, W1f, s1, scale1, shift1, T1n, M, N3, m_tiles, stat3, stat1, ds);
    return true;
}

// T2, out, T1n: pair format; identity: pairs (res_pair) or fp32 (a downsample branch computed by a launch of its own).  c2 != nullptr (K == 64 only):
// the bottleneck's conv2 runs in front, in the same launch, from c2->T1 (T2 unused)
bool launch_conv_c3f(const float *T2, const unsigned *W3, const float *scale3, const float *shift3, const float *identity, float *out,
                     const unsigned *W1f, const float *s1 /*device {S, 1/S}*/, const float *scale1, const float *shift1, float *T1n, int M, int K, int N3, int N1,
                     hipStream_t s, float *stat3, float *stat1, int res_pair, const C2Args *c2) {
    if (N3 % 64 || N3 > 512) return false;
    if (c2 && K != 64 && K != 128) return false;
#define SYN_C3F(KS3, NT1, MT, TC) launch_c3f_t<KS3, NT1, MT, TC>(T2, W3, scale3, shift3, identity, out, W1f, s1, scale1, shift1, T1n, M, N3, s, stat3, stat1, res_pair, c2)
    if (K == 64 && N1 == 64) SYN_C3F(2, 4, SYN_C3F_L1_MT, SYN_C3F_L1_TC);      // layer 1
    else if (K == 64 && N1 == 128) SYN_C3F(2, 8, SYN_C3F_L1_MT, 2);        // layer 1 -> layer 2 (32-channel chunks: 48 KB of LDS, < 256 registers)
    else if (K == 128 && N1 == 128) SYN_C3F(4, 8, SYN_C3F_L2_MT, 2);       // layer 2
    else return false;
#undef SYN_C3F
    return true;
}

// =====================================================================================
// ResNet stem: 7x7 stride-2 pad-3 conv 3->64 + BN + ReLU (resnet_backbone.py:168-171), 120 -> 60, NHWC out.
// Thread = (output pixel, 4 channels); the 147x64 filter (36.8 KB) sits in LDS.  ~1.4 % of the network's FLOPs.
// U8 variant fuses the HWC->CHW permute and (x-127.5)/128 like the MobileNetV2 stem.
// =====================================================================================
template <bool U8>
__global__ __launch_bounds__(256) void resnet_stem_kernel(const float *__restrict__ img, const uint8_t *__restrict__ img8,
                                                          const float *__restrict__ w /*[147][64]*/, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, float *__restrict__ out, int npix) {
    __shared__ __attribute__((aligned(16))) float sw[147 * 64];
    for (int i = threadIdx.x; i < 147 * 64 / 4; i += 256) *(f32x4 *)&sw[4 * i] = *(const f32x4 *)&w[4 * i];
    __syncthreads();
    const int c4 = threadIdx.x & 15;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= npix) return;
    const int b = p / 3600, r = p - b * 3600;
    const int oy = r / 60, ox = r - oy * 60;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky) {
        const int iy = 2 * oy - 3 + ky;
        if (iy < 0 || iy >= kImg) continue;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const int ix = 2 * ox - 3 + kx;
            if (ix < 0 || ix >= kImg) continue;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                float v;
                if (U8) v = ((float)img8[((size_t)(b * kImg + iy) * kImg + ix) * 3 + ci] - 127.5f) * 0.0078125f;
                else    v = img[((size_t)(b * 3 + ci) * kImg + iy) * kImg + ix];
                acc += v * *(const f32x4 *)&sw[(ci * 49 + ky * 7 + kx) * 64 + 4 * c4];
            }
        }
    }
    const f32x4 sc = *(const f32x4 *)&scale[4 * c4], sh = *(const f32x4 *)&shift[4 * c4];
    f32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = fmaxf(acc[t] * sc[t] + sh[t], 0.0f);
    *(f32x4 *)&out[(size_t)p * 64 + 4 * c4] = o;
}

void launch_resnet_stem(const float *img, const uint8_t *img8, const float *w, const float *scale, const float *shift,
                        float *out, int B, hipStream_t s) {
    const int npix = B * 3600;
    if (img8) resnet_stem_kernel<true><<<(npix + 15) / 16, 256, 0, s>>>(nullptr, img8, w, scale, shift, out, npix);
    else      resnet_stem_kernel<false><<<(npix + 15) / 16, 256, 0, s>>>(img, nullptr, w, scale, shift, out, npix);
}

// MaxPool2d(kernel 3, stride 2, padding 1) (resnet_backbone.py:172), NHWC fp32 in; padding never wins the max (-inf).  out_pair: the pooled
// tensor in the pair format (an fp16 x2 forward: layer 1's conv1 takes it), else fp32.
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float *__restrict__ in, float *__restrict__ out, long total,
                                                           int Hin, int Hout, int C4, float *__restrict__ stat, int out_pair) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) { if (stat) range_note(stat, 0.f); return; }       // (whole waves take part in the reduction)
    const int c4 = (int)(idx % C4);
    long p = idx / C4;
    const long pix = p;
    const int ox = (int)(p % Hout);
    p /= Hout;
    const int oy = (int)(p % Hout), b = (int)(p / Hout);
    f32x4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if (iy < 0 || iy >= Hin) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (ix < 0 || ix >= Hin) continue;
            const f32x4 v = *(const f32x4 *)&in[((size_t)(b * Hin + iy) * Hin + ix) * (C4 * 4) + 4 * c4];
#pragma unroll
            for (int t = 0; t < 4; ++t) m[t] = fmaxf(m[t], v[t]);
        }
    }
    if (out_pair) {
        u32x2 hi, lo;
        split4(m, hi, lo);
        float *dst = out + (size_t)pix * (C4 * 4) + 32 * (c4 >> 3) + 2 * (c4 & 7);
        *(u32x2 *)dst = hi;
        *(u32x2 *)(dst + 16) = lo;
    } else *(f32x4 *)&out[(size_t)idx * 4] = m;
    if (stat) range_note(stat, fmaxf(fmaxf(fabsf(m[0]), fabsf(m[1])), fmaxf(fabsf(m[2]), fabsf(m[3]))));
}

void launch_maxpool3x3s2(const float *in, float *out, int B, int Hin, int Hout, int C, hipStream_t s, float *stat, int out_pair) {
    const long total = (long)B * Hout * Hout * (C / 4);
    maxpool3x3s2_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(in, out, total, Hin, Hout, C / 4, stat, out_pair);
}

// adaptive_avg_pool2d -> flatten -> linear heads (resnet_backbone.py:236-246): feat [B,P,C] NHWC, Wfc [n_out][C].
__global__ __launch_bounds__(256) void pool_fc_generic_kernel(const float *__restrict__ feat, const float *__restrict__ Wfc,
                                                              const float *__restrict__ bias, float *__restrict__ param,
                                                              float *__restrict__ pool, int P, int C, int n_out, int out_stride,
                                                              const float *__restrict__ stat, int n_stat, unsigned *guard_word) {
    __shared__ __attribute__((aligned(16))) float sp[2048];
    const int b = blockIdx.x;
    // range guard: a tensor that some fp16 x2 convolution split left the fp16 window (kRangeHi / kRangeLo) -> the results of this
    // forward are NOT fp32-class: make that loud (NaN) instead of returning plausible numbers
    float poison = 0.f;
    if (n_stat > 0) {                                  // (kernel-uniform) four threads per tensor, 16 sub-slots each
        const int t = threadIdx.x >> 2, q = threadIdx.x & 3;
        float m = 0.f;
        if (t < n_stat)
            for (int i = 0; i < kRangeSub / 4; ++i) m = fmaxf(m, stat[((size_t)t * kRangeSub + q * (kRangeSub / 4) + i) * kRangeStride]);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        const int bad = t < n_stat && (!(m <= kRangeHi) || !(m >= kRangeLo));
        if (__syncthreads_or(bad)) {
            poison = __builtin_nanf("");
            // ... and tell the host (page-locked mapped word, read at the entry of the handle's next forward: synergy_abi.hip run_resnet50)
            if (guard_word && b == 0 && threadIdx.x == 0) __hip_atomic_store(guard_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const float *f = feat + (size_t)b * P * C;
    const float inv = 1.0f / (float)P;
    // (round 5: 74 -> 37 us at B = 512.  The pooling loop used to be load -> add per pixel, and every head row a chain of eight dependent
    // round trips to L2: four pixels / four rows are now requested together -- the same sums in the same order)
    for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        int p = 0;
        for (; p + 4 <= P; p += 4) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *(const f32x4 *)&f[(size_t)(p + i) * C + 4 * c4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a += v[i];
        }
        for (; p < P; ++p) a += *(const f32x4 *)&f[(size_t)p * C + 4 * c4];
        a *= inv;
        *(f32x4 *)&sp[4 * c4] = a;
        if (pool) *(f32x4 *)&pool[(size_t)b * C + 4 * c4] = a + poison;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o0 = wave; o0 < n_out; o0 += 16) {              // this wave's rows o0, o0 + 4, o0 + 8, o0 + 12 together
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 xv = *(const f32x4 *)&sp[c];
            f32x4 wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = o0 + 4 * i < n_out ? o0 + 4 * i : o0;
                wv[i] = *(const f32x4 *)&Wfc[(size_t)o * C + c];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] += wv[i][0] * xv[0] + wv[i][1] * xv[1] + wv[i][2] * xv[2] + wv[i][3] * xv[3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = a[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            const int o = o0 + 4 * i;
            if (lane == 0 && o < n_out) param[(size_t)b * out_stride + o] = t + bias[o] + poison;
        }
    }
}

void launch_pool_fc_generic(const float *feat, const float *Wfc, const float *bias, float *param, float *pool, int B, int P,
                            int C, int n_out, int out_stride, hipStream_t s, const float *stat, int n_stat, unsigned *guard_word) {
    pool_fc_generic_kernel<<<B, 256, 0, s>>>(feat, Wfc, bias, param, pool, P, C, n_out, out_stride, stat, n_stat, guard_word);
}

// =====================================================================================
// 7x7 / 2 stem on v_mfma_f32_32x32x16_f16 for uint8 crops (round 2), same construction as stem_rm.hip:
//   * raw pixel bytes ARE fp16 numbers, so only the filter is carried as two fp16 pieces (2 MFMAs per k16 step); (2p-255)/256 = p/128 - 255/256 is
//     folded into filter (/128) and shift; zero padding of the normalised image = raw 127.5;
//   * im2col K order: kernel row ky contributes 22 consecutive elements of the image row (one don't-care byte + 7 pixels x 3
//     channels, so that every pair of K slots is a 4-byte aligned LDS dword), 7 x 22 = 154 -> ten k16 steps; lane half h holds
//     slots 8h..8h+7 of a step.  Whether a half's pair of a step lies in the same kernel row as the other half's (+16 bytes) or
//     in the next one (row pointer + 1, -28 bytes) is a compile-time property of the slot, so two per-lane pointer tables
//     (rsame / rsel, rebuilt per output row from scalar ring offsets) give every ds_read_b32 an immediate offset;
//   * a compute wave = (face, 32 of the 64 channels) holds its 80 weight-fragment registers for the whole kernel and walks the
//     60 output rows x 2 column blocks; one service wave per workgroup converts image rows to fp16 into a 16-slot LDS ring.
// BN + ReLU epilogue, NHWC store.  (Was: a direct VALU convolution, 1.44 ms of the 12.4 ms forward at B = 512.)
// =====================================================================================
namespace {
typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
constexpr int kRsRowEl = 384, kRsSlots = 16, kRsPadL = 12;
constexpr unsigned kRsPad = 0x57F8u;          // 127.5 as fp16
__device__ __forceinline__ f32x16s mfma_rs(u32x4s a, u32x4s b, f32x16s c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, a), __builtin_bit_cast(f16x8s, b), c, 0, 0, 0);
}
}  // namespace

// POOL: the 3x3 / 2 max-pool (resnet_backbone.py:172) in the epilogue -- out is then [B,30,30,64] and the 60x60x64 stem output (472 MB at
// B = 512, written and read back by a kernel of its own before) never exists.  The two column blocks of a row become lanes j <-> column
// 30 c + j - 1 (30 columns + one neighbour each side, overlapping by two columns instead of exchanging them), so the horizontal 3-max is
// two whole-wave lane shifts; vertically a running max over the rows 2 py - 1, 2 py, 2 py + 1; post-ReLU values are >= 0, so the
// out-of-image neighbours of the pool (its -inf padding) may be zeros.  stat: range-guard slot of the pooled tensor.
template <int U, bool POOL = false>
__global__ __launch_bounds__((2 * U + 1) * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resnet_stem_mfma_kernel(const uint8_t *__restrict__ img /*[B,120,120,3]*/, const unsigned *__restrict__ As3 /*[2][10][2][64][4]*/,
                             const float *__restrict__ s_shift /*[64] folded*/, float *__restrict__ out /*[B,60,60,64]*/, int B,
                             float *__restrict__ stat = nullptr, int out_pair = 0 /*POOL: the pooled tensor in the pair format*/) {
    constexpr int UNIT_DW = (kRsSlots + 1) * kRsRowEl / 2, NT = (2 * U + 1) * 64;
    __shared__ __attribute__((aligned(16))) unsigned smem[U * UNIT_DW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < U * UNIT_DW; i += NT) smem[i] = kRsPad | (kRsPad << 16);      // padding row, side padding, tails
    __syncthreads();

    if (wave_wg == 2 * U) {
        // ---- service wave: image rows -> bf16 -> ring (row iy -> slot iy & 15) ----
        constexpr int PER_ROW = kImg * 3 / 4, TOTAL = U * 2 * PER_ROW, ITER = (TOTAL + 63) / 64;
        auto stage_rows = [&](int fb, int iy0) {
            const uint8_t *fbase = img + (size_t)fb * kImg * kImg * 3;
            unsigned v[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int i = lane + 64 * it;
                const int u = i / (2 * PER_ROW), r = (i / PER_ROW) % 2, d = i % PER_ROW;
                const int iy = iy0 + r;
                const unsigned off = (unsigned)((u * kImg + iy) * kImg * 3 + 4 * d);
                v[it] = 0u;
                if (i < TOTAL && fb + u < B && iy < kImg) v[it] = *reinterpret_cast<const unsigned *>(fbase + off);
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int i = lane + 64 * it;
                const int u = i / (2 * PER_ROW), r = (i / PER_ROW) % 2, d = i % PER_ROW;
                const int iy = iy0 + r;
                if (i < TOTAL && iy < kImg) {
                    u32x2s o;
                    o[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz((float)(v[it] & 0xff), (float)((v[it] >> 8) & 0xff)));
                    o[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz((float)((v[it] >> 16) & 0xff), (float)(v[it] >> 24)));
                    *reinterpret_cast<u32x2s *>(smem + u * UNIT_DW + (iy & (kRsSlots - 1)) * (kRsRowEl / 2) + kRsPadL / 2 + 2 * d) = o;
                }
            }
        };
        for (int fb = blockIdx.x * U; fb < B; fb += gridDim.x * U) {
            stage_rows(fb, 0);
            stage_rows(fb, 2);
            __syncthreads();                                   // (P) image rows 0..3
            for (int oy = 0; oy < 60; ++oy) {
                stage_rows(fb, 2 * oy + 4);                    // what output row oy+1 adds
                __syncthreads();
            }
        }
        return;
    }

    // ---- compute wave: channels 32G .. 32G+31 of face fb + uw ----
    const int uw = wave_wg >> 1, G = wave_wg & 1;
    const int j = lane & 31, h = lane >> 5;
    u32x4s as[10][2];
#pragma unroll
    for (int s = 0; s < 10; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) as[s][p] = *(const u32x4s *)(As3 + ((size_t)(G * 10 + s) * 2 + p) * 256 + lane * 4);
    const float Ss = s_shift[64], inv_ss = s_shift[65];
    f32x4 sh4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sh4[q] = *(const f32x4 *)&s_shift[32 * G + 8 * q + 4 * h] * Ss;
    const char *ring = reinterpret_cast<const char *>(smem + uw * UNIT_DW);

    float vmaxp = 0.f;
    for (int fb = blockIdx.x * U; fb < B; fb += gridDim.x * U) {
        const int f = fb + uw;
        f32x16s prevhm[2], runmax[2];                          // POOL: horizontal maxima of the previous odd row / running maximum of the open pooled row, per column block
        if (POOL) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) { prevhm[c][r] = 0.f; runmax[c][r] = 0.f; }
        }
        __syncthreads();                                       // (P)
        for (int oy = 0; oy < 60; ++oy) {
            // byte offsets of the seven image rows 2oy-3 .. 2oy+3 inside the ring (the padding row for rows outside the image)
            int rowoff[8];
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int iy = 2 * oy - 3 + r;
                rowoff[r] = ((unsigned)iy < (unsigned)kImg ? (iy & (kRsSlots - 1)) : kRsSlots) * (kRsRowEl * 2);
            }
            rowoff[7] = rowoff[6];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = POOL ? 30 * c + j - 1 : 32 * c + j;
                const bool col_ok = (unsigned)col < 60u;
                const int lanebase = (6 * (col < 0 ? 0 : (col < 60 ? col : 59)) + 2) * 2;             // first byte of this lane's 22-element runs
                f32x16s e;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < 4; ++t) e[4 * q + t] = sh4[q][t];
#pragma unroll
                for (int s = 0; s < 10; ++s) {
                    u32x4s xb;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kk0 = 16 * s + 2 * t, r0 = kk0 / 22, m0 = kk0 % 22;          // compile-time
                        // half 0 reads (r0, m0); half 1 reads 8 slots further: same row (+16 bytes) or the next row (-28 bytes)
                        const int a0 = rowoff[r0] + 2 * m0;
                        const int a1 = m0 <= 12 ? rowoff[r0] + 2 * m0 + 16 : rowoff[r0 + 1] + 2 * m0 - 28;
                        xb[t] = *reinterpret_cast<const unsigned *>(ring + lanebase + (h ? a1 : a0));
                    }
                    e = mfma_rs(as[s][1], xb, e);
                    e = mfma_rs(as[s][0], xb, e);
                }
                if constexpr (!POOL) {
                    if (col_ok && f < B) {
                        float *dst = out + ((size_t)(f * 60 + oy) * 60 + col) * 64 + 32 * G + 4 * h;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(f32x4 *)(dst + 8 * q) = (f32x4){fmaxf(e[4 * q], 0.f), fmaxf(e[4 * q + 1], 0.f), fmaxf(e[4 * q + 2], 0.f), fmaxf(e[4 * q + 3], 0.f)} * inv_ss;
                    }
                } else {
                    // ReLU (x 1 / Ss), zero outside the image, horizontal 3-max through whole-wave lane shifts (lanes 0 / 31 of a half are the
                    // overlap columns: what the shifts bring them from the other half is never used)
                    const float keep = col_ok ? inv_ss : 0.f;
                    f32x16s hm;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = fmaxf(e[r], 0.f) * keep;
                        const float l = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
                        const float rt = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
                        hm[r] = fmaxf(fmaxf(l, v), rt);
                    }
                    if (!(oy & 1)) {                           // even stem row 2 py: with the odd row above it
#pragma unroll
                        for (int r = 0; r < 16; ++r) runmax[c][r] = fmaxf(prevhm[c][r], hm[r]);
                    } else {                                   // odd stem row 2 py + 1 closes pooled row py
                        const int py = oy >> 1, px = col >> 1;
                        const bool st = (j & 1) && j <= 29 && f < B;       // even stem columns 30 c + j - 1, j = 1, 3, .. 29
                        float *dst = out + ((size_t)(f * 30 + py) * 30 + px) * 64 + 32 * G + 4 * h;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = {fmaxf(runmax[c][4 * q], hm[4 * q]), fmaxf(runmax[c][4 * q + 1], hm[4 * q + 1]),
                                             fmaxf(runmax[c][4 * q + 2], hm[4 * q + 2]), fmaxf(runmax[c][4 * q + 3], hm[4 * q + 3])};
                            if (st) {
                                if (out_pair) {                // channels 32 G + 8 q + 4 h + 0..3: half a k-group of chunk G -- 8 bytes of each piece
                                    u32x2 hi, lo;
                                    split4(v, hi, lo);
                                    float *dp = dst - 4 * h + 4 * q + 2 * h;     // pixel row + 32 G + (4 q + 2 h) dwords
                                    *(u32x2 *)dp = hi;
                                    *(u32x2 *)(dp + 16) = lo;
                                } else *(f32x4 *)(dst + 8 * q) = v;
                                vmaxp = fmaxf(vmaxp, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) prevhm[c][r] = hm[r];
                    }
                }
            }
            __syncthreads();
        }
    }
    if (POOL && stat) range_note(stat, vmaxp);
}

bool launch_resnet_stem_mfma(const uint8_t *img8, const unsigned *As3, const float *s_shift, float *out, int B, hipStream_t s, int pool, float *stat, int out_pair) {
    if (!img8 || !As3) return false;
    constexpr int U = 2;
    const int wgs = (B + U - 1) / U;
    if (pool) resnet_stem_mfma_kernel<U, true><<<wgs < 256 ? wgs : 256, (2 * U + 1) * 64, 0, s>>>(img8, As3, s_shift, out, B, stat, out_pair);
    else resnet_stem_mfma_kernel<U><<<wgs < 256 ? wgs : 256, (2 * U + 1) * 64, 0, s>>>(img8, As3, s_shift, out, B);
    return true;
}

}  // namespace syn
