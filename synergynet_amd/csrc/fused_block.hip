// Fused MobileNetV2 inverted-residual block for gfx950 (reference mobilenetv2_backbone.py:45-74):
//
//   y = [x +] BN(pw_project( ReLU6(BN(dw3x3( ReLU6(BN(pw_expand(x))) ))) ))
//
// in ONE kernel.  The 6x-expanded activations (up to 1.38 MB per face at 60x60) never leave the CU:
// a workgroup owns a spatial tile of output pixels (of NF faces), keeps the input tile (with its 3x3
// halo) in LDS and walks the hidden channels HC at a time:
//
//   stage 1  expand   E[pix_in ][HC] = ReLU6(BN(Xs[pix_in][CIN] . We^T))    fp32 MFMA 16x16x4 -> LDS
//   stage 2  dw 3x3   D[pix_out][HC] = ReLU6(BN(dw(E)))                       VALU, LDS -> LDS
//   stage 3  project  acc[pix_out][COUT] += D . Wp[:, chunk]^T                fp32 MFMA, accumulators in VGPRs
//
// and finishes with BN (+ residual, read back from the LDS input tile) and one NHWC store.  HBM
// traffic per block is input tile + output tile; everything else is LDS / L2-resident weights.
//
// MFMA operand convention (same as pointwise_kernel): MFMA "A" rows = 16 output channels (weights),
// MFMA "B" cols = 16 pixels (activations from LDS); a lane owns 4 consecutive channels of one pixel.
// Weights are pre-packed by the host in lane order  Wpk[n_tile][k_chunk][lane][4]  with
//   value = W[n = 16*n_tile + (lane&15)][k = 16*k_chunk + 4*(lane>>4) + s]
// so each weight fetch of a wave is one fully coalesced 1 KiB load.
#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu6_(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }
__device__ __forceinline__ f32x4 relu6_(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = relu6_(v[i]);
    return r;
}

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int rup(int a, int b) { return cdiv(a, b) * b; }

template <int CIN_, int HID_, int COUT_, int HIN_, int S_, bool RES_, int TH_, int TW_, int NF_, int HC_, int NW_,
          int EPB_, int WN_, int WP_>
struct BlockCfg {
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_, HIN = HIN_, S = S_, TH = TH_, TW = TW_, NF = NF_,
                         HC = HC_, NW = NW_, EPB = EPB_, WN = WN_, WP = WP_;
    static constexpr bool RES = RES_;
    static constexpr int HOUT = S == 2 ? (HIN + 1) / 2 : HIN;
    static constexpr int TILES_Y = cdiv(HOUT, TH), TILES_X = cdiv(HOUT, TW);
    // a tile that covers the whole image needs no halo ring: everything outside is zero padding
    static constexpr bool WHOLE = (TILES_Y == 1 && TILES_X == 1);
    static constexpr int IH = WHOLE ? HIN : (TH - 1) * S + 3, IW = WHOLE ? HIN : (TW - 1) * S + 3;   // input tile
    static constexpr int PIN = NF * IH * IW, PINP = rup(PIN, 16);
    static constexpr int POUT = NF * TH * TW, POUTP = rup(POUT, 16);
    static constexpr int CINP = rup(CIN, 16), COUTP = rup(COUT, 16);
    static constexpr int KCH = CINP / 16;                                        // expand k-chunks
    static constexpr int XS = CINP + 4, ES = HC + 4;                             // LDS row strides (floats)
    static constexpr int NT_E = HC / 16, PT_IN = PINP / 16, PG = cdiv(PT_IN, EPB), JOBS = NT_E * PG;
    static constexpr int NT_O = COUTP / 16, PT_O = POUTP / 16;
    static constexpr int AN = cdiv(NT_O, WN), AP = cdiv(PT_O, WP);
    static constexpr int XS_FLOATS = PINP * XS, ES_FLOATS = PINP * ES, DS_FLOATS = POUTP * ES;
    static constexpr int LDS_FLOATS = XS_FLOATS + ES_FLOATS + DS_FLOATS;
    static_assert(HID % HC == 0 && HC % 16 == 0, "hidden chunking");
    static_assert(WN * WP == NW, "wave grid");
    static_assert(!RES || (S == 1 && CIN == COUT), "residual only on stride-1 same-width blocks");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

template <class C>
__global__ __launch_bounds__(C::NW * 64) void fused_block_kernel(
    const float *__restrict__ X, const float *__restrict__ We, const float *__restrict__ e_scale,
    const float *__restrict__ e_shift, const float *__restrict__ Wd, const float *__restrict__ d_scale,
    const float *__restrict__ d_shift, const float *__restrict__ Wp, const float *__restrict__ p_scale,
    const float *__restrict__ p_shift, float *__restrict__ Y, int B) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    float *Xs = smem, *Es = smem + C::XS_FLOATS, *Ds = Es + C::ES_FLOATS;
    constexpr int NT = C::NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;

    int bid = blockIdx.x;
    const int tx = bid % C::TILES_X;
    bid /= C::TILES_X;
    const int ty = bid % C::TILES_Y;
    const int f0 = (bid / C::TILES_Y) * C::NF;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int iy0 = C::WHOLE ? 0 : oy0 * C::S - 1, ix0 = C::WHOLE ? 0 : ox0 * C::S - 1;   // image coords of tile pixel (0,0)

    // ---- stage 0: input tile (halo included, zero outside the image / past CIN) -> LDS ----
    for (int it = tid; it < C::PINP * (C::CINP / 4); it += NT) {
        const int c4 = it % (C::CINP / 4), p = it / (C::CINP / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p < C::PIN && 4 * c4 < C::CIN) {
            const int i = p / (C::IH * C::IW), r = p % (C::IH * C::IW);
            const int iy = iy0 + r / C::IW, ix = ix0 + r % C::IW;
            const int f = f0 + i;
            if (f < B && iy >= 0 && iy < C::HIN && ix >= 0 && ix < C::HIN)
                v = *(const f32x4 *)&X[((size_t)(f * C::HIN + iy) * C::HIN + ix) * C::CIN + 4 * c4];
        }
        *(f32x4 *)&Xs[p * C::XS + 4 * c4] = v;
    }
    if (C::POUTP > C::POUT)
        for (int it = tid; it < (C::POUTP - C::POUT) * C::ES; it += NT) Ds[C::POUT * C::ES + it] = 0.f;

    f32x4 acc[C::AN][C::AP];
#pragma unroll
    for (int i = 0; i < C::AN; ++i)
#pragma unroll
        for (int j = 0; j < C::AP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wn = wave % C::WN, wp = wave / C::WN;
    __syncthreads();

    for (int hc0 = 0; hc0 < C::HID; hc0 += C::HC) {
        // ---- stage 1: expand 1x1 + BN + ReLU6 for every pixel of the input tile ----
        for (int job = wave; job < C::JOBS; job += C::NW) {
            const int nt = job % C::NT_E, pg = job / C::NT_E;
            f32x4 ea[C::EPB];
#pragma unroll
            for (int q = 0; q < C::EPB; ++q) ea[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *wa = We + ((size_t)(hc0 / 16 + nt) * C::KCH) * 256 + lane * 4;
#pragma unroll
            for (int kc = 0; kc < C::KCH; ++kc) {
                const f32x4 a = *(const f32x4 *)(wa + kc * 256);
#pragma unroll
                for (int q = 0; q < C::EPB; ++q) {
                    const int pt = pg * C::EPB + q;
                    if (pt < C::PT_IN) {
                        const f32x4 b = *(const f32x4 *)&Xs[(pt * 16 + r16) * C::XS + kc * 16 + 4 * g];
#pragma unroll
                        for (int s = 0; s < 4; ++s) ea[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], ea[q], 0, 0, 0);
                    }
                }
            }
            const int ch = hc0 + nt * 16 + 4 * g;
            const f32x4 sc = *(const f32x4 *)&e_scale[ch];
            const f32x4 sh = *(const f32x4 *)&e_shift[ch];
#pragma unroll
            for (int q = 0; q < C::EPB; ++q) {
                const int pt = pg * C::EPB + q;
                if (pt < C::PT_IN) *(f32x4 *)&Es[(pt * 16 + r16) * C::ES + nt * 16 + 4 * g] = relu6_(ea[q] * sc + sh);
            }
        }
        __syncthreads();
        // ---- stage 2: depthwise 3x3 + BN + ReLU6 (zero padding = skip taps outside the image) ----
        for (int it = tid; it < C::POUT * (C::HC / 4); it += NT) {
            const int c4 = it % (C::HC / 4), po = it / (C::HC / 4);
            const int i = po / (C::TH * C::TW), r = po % (C::TH * C::TW);
            const int oyl = r / C::TW, oxl = r % C::TW;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = (oy0 + oyl) * C::S - 1 + ky, ly = iy - iy0;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = (ox0 + oxl) * C::S - 1 + kx, lx = ix - ix0;
                    if (iy >= 0 && iy < C::HIN && ix >= 0 && ix < C::HIN) {
                        const f32x4 e = *(const f32x4 *)&Es[((i * C::IH + ly) * C::IW + lx) * C::ES + 4 * c4];
                        const f32x4 w = *(const f32x4 *)&Wd[(ky * 3 + kx) * C::HID + hc0 + 4 * c4];
                        a += e * w;
                    }
                }
            }
            const f32x4 sc = *(const f32x4 *)&d_scale[hc0 + 4 * c4];
            const f32x4 sh = *(const f32x4 *)&d_shift[hc0 + 4 * c4];
            *(f32x4 *)&Ds[po * C::ES + 4 * c4] = relu6_(a * sc + sh);
        }
        __syncthreads();
        // ---- stage 3: project 1x1, K = this hidden chunk, accumulators stay in registers ----
#pragma unroll
        for (int kc = 0; kc < C::HC / 16; ++kc) {
            f32x4 a[C::AN], b[C::AP];
#pragma unroll
            for (int i = 0; i < C::AN; ++i) {
                const int nt = wn + i * C::WN;
                if (nt < C::NT_O) a[i] = *(const f32x4 *)(Wp + ((size_t)nt * (C::HID / 16) + hc0 / 16 + kc) * 256 + lane * 4);
            }
#pragma unroll
            for (int j = 0; j < C::AP; ++j) {
                const int pt = wp + j * C::WP;
                if (pt < C::PT_O) b[j] = *(const f32x4 *)&Ds[(pt * 16 + r16) * C::ES + kc * 16 + 4 * g];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < C::AN; ++i)
#pragma unroll
                    for (int j = 0; j < C::AP; ++j)
                        if (wn + i * C::WN < C::NT_O && wp + j * C::WP < C::PT_O)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        // no barrier here: the next stage 1 only writes Es (its readers finished before the barrier above),
        // and Ds is rewritten only after the next barrier, which every wave reaches after its stage 3.
    }

    // ---- epilogue: BN (+ residual from the LDS input tile) and NHWC store ----
#pragma unroll
    for (int i = 0; i < C::AN; ++i) {
        const int nt = wn + i * C::WN;
        const int n = nt * 16 + 4 * g;
        if (nt >= C::NT_O || n >= C::COUT) continue;
        const f32x4 sc = *(const f32x4 *)&p_scale[n];
        const f32x4 sh = *(const f32x4 *)&p_shift[n];
#pragma unroll
        for (int j = 0; j < C::AP; ++j) {
            const int pt = wp + j * C::WP;
            const int po = pt * 16 + r16;
            if (pt >= C::PT_O || po >= C::POUT) continue;
            const int fi = po / (C::TH * C::TW), r = po % (C::TH * C::TW);
            const int oyl = r / C::TW, oxl = r % C::TW;
            const int f = f0 + fi, oy = oy0 + oyl, ox = ox0 + oxl;
            if (f >= B || oy >= C::HOUT || ox >= C::HOUT) continue;
            f32x4 v = acc[i][j] * sc + sh;
            if (C::RES) v += *(const f32x4 *)&Xs[((fi * C::IH + (oy - iy0)) * C::IW + (ox - ix0)) * C::XS + n];   // x + conv(x)
            *(f32x4 *)&Y[((size_t)(f * C::HOUT + oy) * C::HOUT + ox) * C::COUT + n] = v;
        }
    }
}

template <class C>
static void launch_cfg(const FusedBlockArgs &a, int B, hipStream_t s) {
    const int groups = (B + C::NF - 1) / C::NF;
    const int grid = groups * C::TILES_Y * C::TILES_X;
    fused_block_kernel<C><<<grid, C::NW * 64, 0, s>>>(a.X, a.We, a.e_scale, a.e_shift, a.Wd, a.d_scale, a.d_shift, a.Wp,
                                                       a.p_scale, a.p_shift, a.Y, B);
}

//                      CIN  HID COUT HIN S  RES    TH  TW NF  HC NW EPB WN WP
using Cfg2 = BlockCfg<  16,  96,  24, 60, 2, false,  6,  6, 1, 32, 4, 3, 2, 2>;   // features.2   60 -> 30
using Cfg3 = BlockCfg<  24, 144,  24, 30, 1, true,  10, 10, 1, 48, 4, 3, 2, 2>;   // features.3   30
using Cfg4 = BlockCfg<  24, 144,  32, 30, 2, false,  5,  5, 1, 48, 4, 2, 2, 2>;   // features.4   30 -> 15
using Cfg5 = BlockCfg<  32, 192,  32, 15, 1, true,  15, 15, 1, 32, 8, 4, 2, 4>;   // features.5,6 15
using Cfg7 = BlockCfg<  32, 192,  64, 15, 2, false,  8,  8, 1, 32, 4, 4, 4, 1>;   // features.7   15 -> 8
using Cfg8 = BlockCfg<  64, 384,  64,  8, 1, true,   8,  8, 1, 64, 4, 4, 4, 1>;   // features.8-10
using Cfg11 = BlockCfg< 64, 384,  96,  8, 1, false,  8,  8, 1, 64, 4, 4, 2, 2>;   // features.11
using Cfg12 = BlockCfg< 96, 576,  96,  8, 1, true,   8,  8, 1, 64, 4, 4, 2, 2>;   // features.12,13
using Cfg14 = BlockCfg< 96, 576, 160,  8, 2, false,  4,  4, 2, 64, 4, 4, 4, 1>;   // features.14  8 -> 4
using Cfg15 = BlockCfg<160, 960, 160,  4, 1, true,   4,  4, 4, 64, 4, 4, 2, 2>;   // features.15,16
using Cfg17 = BlockCfg<160, 960, 320,  4, 1, false,  4,  4, 4, 64, 4, 4, 4, 1>;   // features.17

bool launch_fused_block(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    switch (feature) {
        case 2: launch_cfg<Cfg2>(a, B, s); return true;
        case 3: launch_cfg<Cfg3>(a, B, s); return true;
        case 4: launch_cfg<Cfg4>(a, B, s); return true;
        case 5: case 6: launch_cfg<Cfg5>(a, B, s); return true;
        case 7: launch_cfg<Cfg7>(a, B, s); return true;
        case 8: case 9: case 10: launch_cfg<Cfg8>(a, B, s); return true;
        case 11: launch_cfg<Cfg11>(a, B, s); return true;
        case 12: case 13: launch_cfg<Cfg12>(a, B, s); return true;
        case 14: launch_cfg<Cfg14>(a, B, s); return true;
        case 15: case 16: launch_cfg<Cfg15>(a, B, s); return true;
        case 17: launch_cfg<Cfg17>(a, B, s); return true;
        default: return false;
    }
}

}  // namespace syn
