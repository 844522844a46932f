// Fused network head for gfx950: features.0 (3x3 s2 conv 3->32 + BN + ReLU6) and features.1
// (depthwise 3x3 + BN + ReLU6, then linear 1x1 32->16 + BN; t = 1 block without expand conv) in ONE
// kernel (reference mobilenetv2_backbone.py:129, :58-66).  The 60x60x32 stem output -- the largest
// activation of the network, 460 KB per face -- never leaves the CU: a workgroup reads the 25x25x3
// image patch behind a 10x10 output tile, builds the 12x12x32 stem tile in LDS, runs the depthwise
// stage LDS->LDS and the 32->16 projection on the fp32 MFMA, and stores 10x10x16 (NHWC).
// HBM traffic per face: 43 KB of uint8 crop in (+halo re-reads from L2), 230 KB out.
// The uint8 variant also folds the HWC->CHW permute and (x-127.5)/128 (synergy3DMM.py:189-192).
#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int T = 10;                 // output tile edge (60 = 6 tiles)
constexpr int FT = T + 2;             // stem-output tile edge incl. the depthwise halo
constexpr int IT = 2 * FT + 1;        // image tile edge (stride-2 3x3 receptive field) = 25
constexpr int ITS = IT + 1;           // padded image row
constexpr int ES = 36;                // LDS row stride of the 32-channel tiles
constexpr int PIN = FT * FT;          // 144 stem pixels
constexpr int POUT = T * T, POUTP = 112;
constexpr int NTH = 256;
__device__ __forceinline__ float r6(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }
}  // namespace

template <bool U8>
__global__ __launch_bounds__(NTH) void stem_block1_kernel(
    const float *__restrict__ img, const uint8_t *__restrict__ img8, const float *__restrict__ w0 /*[27][32]*/,
    const float *__restrict__ s0, const float *__restrict__ b0, const float *__restrict__ wd /*[9][32]*/,
    const float *__restrict__ sd, const float *__restrict__ bd, const float *__restrict__ wp /*Wpk[1][2][64][4]*/,
    const float *__restrict__ sp, const float *__restrict__ bp, float *__restrict__ Y, int B) {
    __shared__ __attribute__((aligned(16))) float im[3 * IT * ITS];
    __shared__ __attribute__((aligned(16))) float Es[PIN * ES];
    __shared__ __attribute__((aligned(16))) float Ds[POUTP * ES];
    __shared__ __attribute__((aligned(16))) float W0[27 * 32 + 64];     // stem filter | scale | shift
    __shared__ __attribute__((aligned(16))) float WD[9 * 32 + 64];      // depthwise filter | scale | shift
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % 6;
    bid /= 6;
    const int ty = bid % 6;
    const int f = bid / 6;
    const int oy0 = ty * T, ox0 = tx * T;            // first output pixel (60x60 grid)
    const int fy0 = oy0 - 1, fx0 = ox0 - 1;          // stem-output coords of Es pixel (0,0)
    const int iy0 = 2 * fy0 - 1, ix0 = 2 * fx0 - 1;  // image coords of im pixel (0,0)

    for (int i = tid; i < 27 * 32 / 4; i += NTH) *(f32x4 *)&W0[4 * i] = *(const f32x4 *)&w0[4 * i];
    if (tid < 8) { *(f32x4 *)&W0[864 + 4 * tid] = *(const f32x4 *)&s0[4 * tid]; *(f32x4 *)&W0[896 + 4 * tid] = *(const f32x4 *)&b0[4 * tid]; }
    for (int i = tid; i < 9 * 32 / 4; i += NTH) *(f32x4 *)&WD[4 * i] = *(const f32x4 *)&wd[4 * i];
    if (tid >= 64 && tid < 72) { const int t = tid - 64; *(f32x4 *)&WD[288 + 4 * t] = *(const f32x4 *)&sd[4 * t]; *(f32x4 *)&WD[320 + 4 * t] = *(const f32x4 *)&bd[4 * t]; }
    // image patch -> LDS as normalised fp32 planes; zero outside the image (conv padding = 1)
    for (int i = tid; i < IT * IT; i += NTH) {
        const int ly = i / IT, lx = i % IT;
        const int iy = iy0 + ly, ix = ix0 + lx;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (iy >= 0 && iy < kImg && ix >= 0 && ix < kImg) {
            if (U8) {
                const uint8_t *p = img8 + ((size_t)(f * kImg + iy) * kImg + ix) * 3;
                v0 = ((float)p[0] - 127.5f) * 0.0078125f;
                v1 = ((float)p[1] - 127.5f) * 0.0078125f;
                v2 = ((float)p[2] - 127.5f) * 0.0078125f;
            } else {
                const float *p = img + ((size_t)f * 3 * kImg + iy) * kImg + ix;
                v0 = p[0]; v1 = p[kImg * kImg]; v2 = p[2 * kImg * kImg];
            }
        }
        im[ly * ITS + lx] = v0;
        im[IT * ITS + ly * ITS + lx] = v1;
        im[2 * IT * ITS + ly * ITS + lx] = v2;
    }
    for (int i = tid; i < (POUTP - POUT) * ES; i += NTH) Ds[POUT * ES + i] = 0.f;
    __syncthreads();

    // ---- stem conv: thread = (stem pixel, 4 channels) ----
    for (int it = tid; it < PIN * 8; it += NTH) {
        const int c4 = it & 7, p = it >> 3;
        const int ly = p / FT, lx = p % FT;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = im[ci * IT * ITS + (2 * ly + ky) * ITS + 2 * lx + kx];
                    a += v * *(const f32x4 *)&W0[(ci * 9 + ky * 3 + kx) * 32 + 4 * c4];
                }
        const f32x4 sc = *(const f32x4 *)&W0[864 + 4 * c4], sh = *(const f32x4 *)&W0[896 + 4 * c4];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = r6(a[j] * sc[j] + sh[j]);
        *(f32x4 *)&Es[p * ES + 4 * c4] = o;
    }
    __syncthreads();
    // ---- depthwise 3x3 s1 on the stem tile (zero padding = taps outside the 60x60 map contribute 0) ----
    for (int it = tid; it < POUT * 8; it += NTH) {
        const int c4 = it & 7, po = it >> 3;
        const int oyl = po / T, oxl = po % T;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int fy = fy0 + oyl + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int fx = fx0 + oxl + kx;
                const bool ok = fy >= 0 && fy < 60 && fx >= 0 && fx < 60;
                f32x4 e = *(const f32x4 *)&Es[((oyl + ky) * FT + oxl + kx) * ES + 4 * c4];
                const f32x4 w = *(const f32x4 *)&WD[(ky * 3 + kx) * 32 + 4 * c4];
                if (!ok) e = (f32x4){0.f, 0.f, 0.f, 0.f};
                a += e * w;
            }
        }
        const f32x4 sc = *(const f32x4 *)&WD[288 + 4 * c4], sh = *(const f32x4 *)&WD[320 + 4 * c4];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = r6(a[j] * sc[j] + sh[j]);
        *(f32x4 *)&Ds[po * ES + 4 * c4] = o;
    }
    __syncthreads();
    // ---- linear 1x1 32 -> 16 on the MFMA: 7 pixel tiles over 4 waves ----
    const int r16 = lane & 15, g = lane >> 4;
    const f32x4 a0 = *(const f32x4 *)(wp + lane * 4), a1 = *(const f32x4 *)(wp + 256 + lane * 4);
    const f32x4 sc = *(const f32x4 *)&sp[4 * g], sh = *(const f32x4 *)&bp[4 * g];
    for (int pt = wave; pt < POUTP / 16; pt += 4) {
        const f32x4 b0v = *(const f32x4 *)&Ds[(pt * 16 + r16) * ES + 4 * g];
        const f32x4 b1v = *(const f32x4 *)&Ds[(pt * 16 + r16) * ES + 16 + 4 * g];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0v[s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b1v[s], acc, 0, 0, 0);
        const int po = pt * 16 + r16;
        if (po < POUT) {
            const int oy = oy0 + po / T, ox = ox0 + po % T;
            *(f32x4 *)&Y[((size_t)(f * 60 + oy) * 60 + ox) * 16 + 4 * g] = acc * sc + sh;
        }
    }
}

void launch_stem_block1(const float *img, const uint8_t *img8, const float *w0, const float *s0, const float *b0,
                        const float *wd, const float *sd, const float *bd, const float *wp, const float *sp,
                        const float *bp, float *Y, int B, hipStream_t s) {
    const int grid = B * 36;
    if (img8) stem_block1_kernel<true><<<grid, NTH, 0, s>>>(nullptr, img8, w0, s0, b0, wd, sd, bd, wp, sp, bp, Y, B);
    else      stem_block1_kernel<false><<<grid, NTH, 0, s>>>(img, nullptr, w0, s0, b0, wd, sd, bd, wp, sp, bp, Y, B);
}

}  // namespace syn
