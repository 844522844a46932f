// gfx950 (MI355X / CDNA4) kernels for the MobileNetV2 backbone of SynergyNet's inference path
// (reference backbone_nets/mobilenetv2_backbone.py:33-74,104-189).  fp32 in, fp32 accumulate:
// the parity bar is 1e-4 relative through 53 layers, so the pointwise convolutions run on the
// exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak), not bf16.
//
// Activation layout: NHWC fp32, act[b][y][x][c]  (a 1x1 conv is then a row-major GEMM over
// M = B*H*W pixels; depthwise runs channels across lanes with float4 coalesced loads).
#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu6f(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }

// =====================================================================================
// Stem: 3x3 stride-2 conv 3->32 + BN + ReLU6 (features.0, mobilenetv2_backbone.py:129).
// Thread = (output pixel, 4 output channels); a block covers 32 consecutive pixels so each
// wave writes 8 pixels x 128 B = 1 KiB contiguous NHWC output.  The 27x32 filter sits in LDS.
// U8 variant fuses HWC->CHW and (x-127.5)/128 (synergy3DMM.py:189-192).
// =====================================================================================
template <bool U8>
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ img, const uint8_t *__restrict__ img8,
                                                   const float *__restrict__ w, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, float *__restrict__ out, int npix) {
    __shared__ __attribute__((aligned(16))) float sw[27 * 32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) sw[i] = w[i];
    __syncthreads();
    const int g = threadIdx.x & 7;
    const int p = blockIdx.x * 32 + (threadIdx.x >> 3);
    if (p >= npix) return;
    const int b = p / 3600, r = p - b * 3600;
    const int oy = r / 60, ox = r - oy * 60;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            const bool ok = (iy >= 0) & (iy < kImg) & (ix >= 0) & (ix < kImg);
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                float v = 0.f;
                if (ok) {
                    if (U8) v = ((float)img8[((size_t)(b * kImg + iy) * kImg + ix) * 3 + ci] - 127.5f) * 0.0078125f;
                    else    v = img[((size_t)(b * 3 + ci) * kImg + iy) * kImg + ix];
                }
                const f32x4 wv = *(const f32x4 *)&sw[(ci * 9 + ky * 3 + kx) * 32 + 4 * g];
                acc += v * wv;
            }
        }
    }
    const f32x4 sc = *(const f32x4 *)&scale[4 * g];
    const f32x4 sh = *(const f32x4 *)&shift[4 * g];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = relu6f(acc[j] * sc[j] + sh[j]);
    *(f32x4 *)&out[(size_t)p * 32 + 4 * g] = o;
}

void launch_stem(const float *img, const uint8_t *img8, const float *w, const float *scale, const float *shift,
                 float *out, int B, hipStream_t s) {
    const int npix = B * 3600;
    const int grid = (npix + 31) / 32;
    if (img8) stem_kernel<true><<<grid, 256, 0, s>>>(nullptr, img8, w, scale, shift, out, npix);
    else      stem_kernel<false><<<grid, 256, 0, s>>>(img, nullptr, w, scale, shift, out, npix);
}

// =====================================================================================
// Pointwise 1x1 conv = GEMM on v_mfma_f32_16x16x4_f32, fused BN scale/shift (+ReLU6)(+residual).
//
//   C[m][n] = epi( sum_k A[m][k] * W[n][k] )
//
// Operand roles are swapped w.r.t. the textbook GEMM so the epilogue is lane-local and wide:
//   MFMA "A" operand (rows i) = 16 output channels n,   lane l holds W[n0 + (l&15)][k]
//   MFMA "B" operand (cols j) = 16 pixels m,            lane l holds A[m0 + (l&15)][k]
//   D: lane l owns column j = l&15 (one pixel) and rows i = 4*(l>>4)+r (4 consecutive channels)
// so every accumulator is a float4 of 4 consecutive output channels of one pixel: the BN
// scale/shift, ReLU6, residual add and the NHWC store are all per-lane float4 operations.
//
// K is walked 16 at a time: each lane fetches ONE float4 (k0+4g .. k0+4g+3, g = l>>4) per operand
// row and feeds element s of it to MFMA step s -- hardware k-slot g of step s is logical
// k = k0+4g+s for both operands, so the sum is complete (only its order differs).
// No LDS, no barriers: fp32 MFMA issues one 16x16x4 per 32 cycles per SIMD, so operand traffic is
// tiny (8 x 1 KiB loads per 64 MFMAs for a 64x64 wave tile) and comes from L1/L2.
// A wave owns an (MT*16 pixels) x (NT*16 channels) tile; 4 waves of a block stack along M.
// blockIdx -> (m tile, n tile) keeps all n tiles of one m tile on one XCD (block b runs on XCD b%8),
// so the activation rows they share are served by that XCD's L2.
// =====================================================================================
template <int MT, int NT>
__global__ __launch_bounds__(256) void pointwise_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        const float *__restrict__ residual, float *__restrict__ C,
                                                        int M, int K, int Kpad, int N, int n_tiles, int m_tiles, int relu6) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt_idx = q % n_tiles;
    const int mt_idx = (q / n_tiles) * 8 + xcd;
    if (mt_idx >= m_tiles) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = (mt_idx * 4 + wave) * (MT * 16);
    const int n0 = nt_idx * (NT * 16);
    if (m0 >= M) return;

    const float *wp[NT];
    const float *ap[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wp[i] = W + (size_t)(n0 + i * 16 + r16) * Kpad + 4 * g;   // W is padded: always in bounds
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        int m = m0 + j * 16 + r16;
        m = m < M ? m : M - 1;                                                             // clamp tail rows (stores are masked)
        ap[j] = A + (size_t)m * K + 4 * g;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 wf[NT], af[MT], wn[NT], an[MT];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NT; ++i) wf[i] = *(const f32x4 *)(wp[i]);
#pragma unroll
    for (int j = 0; j < MT; ++j) af[j] = (4 * g < K) ? *(const f32x4 *)(ap[j]) : zero;

    for (int k0 = 0; k0 < Kpad; k0 += 16) {
        const int k1 = k0 + 16;
        if (k1 < Kpad) {
#pragma unroll
            for (int i = 0; i < NT; ++i) wn[i] = *(const f32x4 *)(wp[i] + k1);
            const bool ok = (k1 + 4 * g) < K;                                             // K % 8 == 0: a float4 is all-in or all-out
#pragma unroll
            for (int j = 0; j < MT; ++j) an[j] = ok ? *(const f32x4 *)(ap[j] + k1) : zero;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][s], af[j][s], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[i] = wn[i];
#pragma unroll
        for (int j = 0; j < MT; ++j) af[j] = an[j];
    }

#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = n0 + i * 16 + 4 * g;
        if (n >= N) continue;                                                              // N % 4 == 0
        const f32x4 sc = *(const f32x4 *)&scale[n];
        const f32x4 sh = *(const f32x4 *)&shift[n];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + j * 16 + r16;
            if (m >= M) continue;
            f32x4 v = acc[j][i] * sc + sh;
            if (relu6) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = relu6f(v[t]);
            }
            if (residual) v += *(const f32x4 *)&residual[(size_t)m * N + n];              // x + conv(x) (:70-72)
            *(f32x4 *)&C[(size_t)m * N + n] = v;
        }
    }
}

template <int MT, int NT>
static void launch_pw_t(const float *A, const float *W, const float *scale, const float *shift, const float *residual,
                        float *C, int M, int K, int Kpad, int N, int relu6, hipStream_t s) {
    const int n_tiles = (N + NT * 16 - 1) / (NT * 16);
    const int m_tiles = (M + 4 * MT * 16 - 1) / (4 * MT * 16);
    const int m_tiles8 = (m_tiles + 7) / 8;
    const int grid = m_tiles8 * n_tiles * 8;
    pointwise_kernel<MT, NT><<<grid, 256, 0, s>>>(A, W, scale, shift, residual, C, M, K, Kpad, N, n_tiles, m_tiles, relu6);
}

void launch_pointwise(const float *A, const float *W, const float *scale, const float *shift, const float *residual,
                      float *C, int M, int K, int Kpad, int N, int relu6, hipStream_t s) {
    // channel-tile width: whole N when it is small, else the widest of {64,48,32} that wastes least
    int NT;
    if (N <= 16) NT = 1;
    else if (N <= 32) NT = 2;
    else if (N % 64 == 0) NT = 4;
    else if (N % 48 == 0) NT = 3;
    else if (N % 32 == 0) NT = 2;
    else NT = 4;
    // pixel-tile height: 64 rows per wave when there is enough work to fill 256 CUs, else smaller
    const long tiles64 = ((long)M + 255) / 256 * ((N + NT * 16 - 1) / (NT * 16));
    int MT = tiles64 >= 2048 ? 4 : (tiles64 >= 512 ? 2 : 1);
#define SYN_PW(mt, nt) launch_pw_t<mt, nt>(A, W, scale, shift, residual, C, M, K, Kpad, N, relu6, s)
#define SYN_PW_N(mt) do { switch (NT) { case 1: SYN_PW(mt, 1); break; case 2: SYN_PW(mt, 2); break; \
                                        case 3: SYN_PW(mt, 3); break; default: SYN_PW(mt, 4); } } while (0)
    if (MT == 4) SYN_PW_N(4); else if (MT == 2) SYN_PW_N(2); else SYN_PW_N(1);
#undef SYN_PW_N
#undef SYN_PW
}

// =====================================================================================
// Depthwise 3x3 (pad 1, stride 1|2) + BN + ReLU6 (mobilenetv2_backbone.py:62), NHWC.
// Thread = (output pixel, 4 channels): consecutive lanes take consecutive channel quads of one
// pixel, so every one of the 9 taps is a fully coalesced float4 row segment; the 3x3 re-reads
// hit L1/L2.  ~1 flop/byte: bandwidth bound by construction.
// =====================================================================================
__global__ __launch_bounds__(256) void depthwise_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        float *__restrict__ out, long total, int Hin, int Hout, int C4,
                                                        int stride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    long p = idx / C4;
    const int ox = (int)(p % Hout);
    p /= Hout;
    const int oy = (int)(p % Hout);
    const int b = (int)(p / Hout);
    const int C = C4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - 1 + ky;
        if (iy < 0 || iy >= Hin) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride - 1 + kx;
            if (ix < 0 || ix >= Hin) continue;
            const f32x4 v = *(const f32x4 *)&in[((size_t)(b * Hin + iy) * Hin + ix) * C + 4 * c4];
            const f32x4 wv = *(const f32x4 *)&w[(ky * 3 + kx) * C + 4 * c4];
            acc += v * wv;
        }
    }
    const f32x4 sc = *(const f32x4 *)&scale[4 * c4];
    const f32x4 sh = *(const f32x4 *)&shift[4 * c4];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = relu6f(acc[j] * sc[j] + sh[j]);
    *(f32x4 *)&out[(size_t)idx * 4] = o;
}

void launch_depthwise(const float *in, const float *w, const float *scale, const float *shift, float *out, int B,
                      int Hin, int Hout, int C, int stride, hipStream_t s) {
    const long total = (long)B * Hout * Hout * (C / 4);
    const int grid = (int)((total + 255) / 256);
    depthwise_kernel<<<grid, 256, 0, s>>>(in, w, scale, shift, out, total, Hin, Hout, C / 4, stride);
}

// =====================================================================================
// adaptive_avg_pool2d(4x4 -> 1) + classifier_{ori,shape,exp} + cat (mobilenetv2_backbone.py:179-188).
// One block per face: pooled 1280-vector into LDS, then 62 dot products of length 1280, one wave
// per output with a 64-lane shuffle reduction.
// =====================================================================================
__global__ __launch_bounds__(256) void pool_fc_kernel(const float *__restrict__ feat, const float *__restrict__ Wfc,
                                                      const float *__restrict__ bias, float *__restrict__ param,
                                                      float *__restrict__ pool) {
    __shared__ __attribute__((aligned(16))) float sp[kPool];
    const int b = blockIdx.x;
    const float *f = feat + (size_t)b * 16 * kPool;
    for (int c4 = threadIdx.x; c4 < kPool / 4; c4 += 256) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 16; ++p) a += *(const f32x4 *)&f[p * kPool + 4 * c4];
        a *= 0.0625f;
        *(f32x4 *)&sp[4 * c4] = a;
        if (pool) *(f32x4 *)&pool[(size_t)b * kPool + 4 * c4] = a;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < kParam; o += 4) {
        const float *wr = Wfc + (size_t)o * kPool;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < kPool / 256; ++i) {
            const int c = (i * 64 + lane) * 4;
            const f32x4 wv = *(const f32x4 *)&wr[c];
            const f32x4 xv = *(const f32x4 *)&sp[c];
            a += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) param[(size_t)b * kParam + o] = a + bias[o];
    }
}

void launch_pool_fc(const float *feat, const float *Wfc, const float *bias, float *param, float *pool, int B,
                    hipStream_t s) {
    pool_fc_kernel<<<B, 256, 0, s>>>(feat, Wfc, bias, param, pool);
}

}  // namespace syn
