// Register-resident fused inverted-residual block for the 4x4 MobileNetV2 blocks (features.15-17: 160 -> 960 -> 160 | 320).
// Reference: backbone_nets/mobilenetv2_backbone.py:45-74 (InvertedResidual.forward), :33-42 (ConvBNReLU).
//
// Same idea as fused_block_lb.hip (hidden activations never leave registers, fp16 x2 operands, three partial products per MAC),
// re-cut for a block whose weights (1.2 / 1.8 MB as fragments) dwarf its activations (10 KB per face):
//
//   * one WAVE = one face = one 16-column block of v_mfma_f32_16x16x32_f16 (lane column n = 4y + x): the vertical neighbours of a
//     pixel are row_shr:4 / row_shl:4 inside the 16-lane DPP row (zero fill = the image border), the horizontal ones
//     row_shr:1 / row_shl:1 with the filter column zeroed where the shift crosses x = 0 | 3;
//   * the block input of the face lives in REGISTERS as pre-split B fragments (5 k32 steps x 2 pieces = 40 registers), the
//     accumulators of all output tiles too (40 | 80): the wave walks the 30 hidden groups alone, no partial sums to exchange;
//   * the WEIGHTS go through LDS: the four waves of a workgroup walk the groups in lockstep and share one fetch of each group's
//     fragments + constants -- one contiguous 44 | 64 KB run of the packed blob, a quarter per wave, fetched into registers while the
//     current group computes and written to the other half of a double buffer at its end; one barrier per group.  (Each wave
//     fetching its own fragments would pull 1.2 MB through L2 per face.)
//   * one wave per SIMD (512 registers are not needed, but 4 faces per CU is what B = 1024 offers; matrix and vector instructions
//     of co-resident waves do not overlap anyway: tools/ubench/mfma_valu_kinds.hip).
// Scales as in fused_block_lb.hip: X16 = 16 x;  D = 16 Se (We x + shift);  E = med3(D, 0, 96 Se);  O16 = 16 dshift + sum (taps / Se) E;
// B = med3(O16, 0, 96);  acc = 16 Sp (Wp o);  y = acc / (16 Sp) + pshift (+ x).
#include "syn_internal.h"

#include <cstdio>
#include <cstdlib>

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ void split2q(float x0, float x1, unsigned &a, unsigned &b) {
    const f16x2 ah = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    const float r0 = x0 - (float)ah[0], r1 = x1 - (float)ah[1];
    a = __builtin_bit_cast(unsigned, ah);
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ void split2w(float x0, float x1, u32x4 (&pc)[2], int d) {
    unsigned a, b;
    split2q(x0, x1, a, b);
    pc[0][d] = a; pc[1][d] = b;
}
__device__ __forceinline__ f32x4 mfmaq(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (by value: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index -- clang 19 / ROCm 7.2)
template <int CTRL>
__device__ __forceinline__ float dppq(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ f32x2 dppq2(f32x2 v) {
    f32x2 r;
    r[0] = dppq<CTRL>(v[0]);
    r[1] = dppq<CTRL>(v[1]);
    return r;
}
constexpr int kShr1 = 0x111, kShl1 = 0x101, kShr4 = 0x114, kShl4 = 0x104;     // lane n <- n-1 | n+1 | n-4 | n+4 of its 16-lane row, else 0
}  // namespace

template <int CIN_, int HID_, int COUT_, bool RES_>
struct Lb4Cfg {
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_;
    static constexpr bool RES = RES_;
    static constexpr int KE = CIN / 32, NG = HID / 32, MT = COUT / 16;
    static constexpr int WE_DW = 2 * KE * 2 * 256, WP_DW = MT * 2 * 256, TB_DW = 2 * 256;   // a group's expand | project fragments | constants (12 x 32 floats + pad)
    static constexpr int NKB = ((WE_DW + WP_DW + TB_DW) / 256 + 3) / 4 * 4;               // 1 KB pieces per group, padded to four per wave... (lb4_group_dwords)
    static constexpr int GRP_DW = NKB * 256;
    static constexpr int LDS_DW = 2 * GRP_DW;
    static_assert(CIN % 32 == 0 && HID % 32 == 0 && COUT % 16 == 0, "k32 steps, groups, output tiles");
    static_assert(!RES || CIN == COUT, "residual only on same-width blocks");
    static_assert(LDS_DW * 4 <= 160 * 1024, "LDS budget");
};

template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void fused_block_lb4_kernel(const float *__restrict__ X, const unsigned *__restrict__ Glb /*[NG][NKB][64][4]: We | Wp | table per group*/,
                            const float *__restrict__ p_shift, float *__restrict__ Y, int B) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DW];
    constexpr int KE = C::KE, MT = C::MT, CIN = C::CIN, COUT = C::COUT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.x * 4 + wave;
    const bool real = f < B;
    const int fc = real ? f : B - 1;
    const int n = lane & 15, g = lane >> 4;
    const unsigned l4 = lane * 4, g4 = g * 4;

    // every wave copies every fourth 1 KB piece of a group's run through registers: fetched at the start of the previous group,
    // written to the other half of the double buffer at its end.  (LDS-DMA -- global_load_lds_dwordx4, no registers -- delivers
    // ~25 GB/s per CU whoever issues it: 44 KB per group took 1.8 us, longer than the group's arithmetic.)
    constexpr int NPW = C::NKB / 4;
    u32x4 pf[NPW];
    auto fetch = [&](int G) __attribute__((always_inline)) {
        const unsigned *src = Glb + (size_t)G * C::GRP_DW + l4;
#pragma unroll
        for (int i = 0; i < NPW; ++i) pf[i] = *(const u32x4 *)(src + (wave + 4 * i) * 256);
    };
    auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) *(u32x4 *)&smem[buf * C::GRP_DW + (wave + 4 * i) * 256 + l4] = pf[i];
    };
    fetch(0);

    // ---- block input of this face -> pre-split B fragments in registers (x 16) ----
    u32x4 Xr[KE][2];
    {
        f32x4 xv[KE][2];
#pragma unroll
        for (int kc = 0; kc < KE; ++kc) {
            const float *src = X + ((size_t)fc * 16 + n) * CIN + 32 * kc + 8 * g;
            xv[kc][0] = *(const f32x4 *)src;
            xv[kc][1] = *(const f32x4 *)(src + 4);
        }
#pragma unroll
        for (int kc = 0; kc < KE; ++kc) {
            f32x4 a = xv[kc][0] * 16.0f, b = xv[kc][1] * 16.0f;
            if (!real) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }
            split2w(a[0], a[1], Xr[kc], 0);
            split2w(a[2], a[3], Xr[kc], 1);
            split2w(b[0], b[1], Xr[kc], 2);
            split2w(b[2], b[3], Xr[kc], 3);
        }
    }
    const float mL = (n & 3) != 0 ? 1.f : 0.f, mR = (n & 3) != 3 ? 1.f : 0.f;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float c6e = 0.f, inv_p = 0.f;
    park(0);

    for (int G = 0; G < C::NG; ++G) {
        __syncthreads();                                 // every wave has written its pieces of group G and is done reading group G-1
        if (G + 1 < C::NG) fetch(G + 1);
        const unsigned *We = smem + (G & 1) * C::GRP_DW, *Wp = We + C::WE_DW;
        const float *Tb = reinterpret_cast<const float *>(Wp + C::WP_DW);
        if (G == 0) { c6e = Tb[11 * 32]; inv_p = Tb[11 * 32 + 1]; }
        // One wave per SIMD: nobody covers an LDS round trip, and a fragment read placed next to its MFMA costs ~150 cycles each.
        // All fragments of the group are read up front instead (registers are plentiful at one wave per SIMD).
        u32x4 Ae[2][KE][2], Ap[MT][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kc = 0; kc < KE; ++kc)
#pragma unroll
                for (int p = 0; p < 2; ++p) Ae[t][kc][p] = *(const u32x4 *)&We[((t * KE + kc) * 2 + p) * 256 + l4];
        constexpr int MT0 = MT / 2;                      // (the second half of the project fragments is read after the depthwise: 256 architectural registers)
#pragma unroll
        for (int mt = 0; mt < MT0; ++mt)
#pragma unroll
            for (int p = 0; p < 2; ++p) Ap[mt][p] = *(const u32x4 *)&Wp[(mt * 2 + p) * 256 + l4];
        // ---- expand 1x1 + BN shift: D[t], channels 32 G + 16 t + 4 g + i of pixel n ----
        f32x4 D[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) D[t] = *(const f32x4 *)&Tb[10 * 32 + 16 * t + g4];
#pragma unroll
        for (int kc = 0; kc < KE; ++kc) {
            // three partial products, smallest first; the two tiles interleave as independent chains
            D[0] = mfmaq(Ae[0][kc][1], Xr[kc][0], D[0]); D[1] = mfmaq(Ae[1][kc][1], Xr[kc][0], D[1]);
            D[0] = mfmaq(Ae[0][kc][0], Xr[kc][1], D[0]); D[1] = mfmaq(Ae[1][kc][0], Xr[kc][1], D[1]);
            D[0] = mfmaq(Ae[0][kc][0], Xr[kc][0], D[0]); D[1] = mfmaq(Ae[1][kc][0], Xr[kc][0], D[1]);
        }
#pragma unroll
        for (int mt = 0; mt < MT0; ++mt)
#pragma unroll
            for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(Ap[mt][p]));      // (keeps these reads up here instead of next to their MFMAs)
        // ---- ReLU6, depthwise 3x3 + BN shift + ReLU6 on the registers, split in place into the B operand of the project step ----
        u32x4 Bd[2];
#pragma unroll
        for (int th = 0; th < 4; ++th) {
            const int t = th >> 1, hf = th & 1;
            const int c0 = 16 * t + 2 * hf;             // + 4 g per lane group
            f32x2 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x2 *)&Tb[k * 32 + c0 + g4];
            const f32x2 dsh = *(const f32x2 *)&Tb[9 * 32 + c0 + g4];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { w[3 * dy] *= mL; w[3 * dy + 2] *= mR; }
            f32x2 E;
            E[0] = __builtin_amdgcn_fmed3f(D[t][2 * hf], 0.0f, c6e);
            E[1] = __builtin_amdgcn_fmed3f(D[t][2 * hf + 1], 0.0f, c6e);
            // horizontal first (two lane shifts), then the three row sums shifted vertically (two more): 8 DPP moves per channel pair
            // instead of 16.  (Summation order differs from the other kernels': dx inside dy inside the vertical sum.)
            const f32x2 l = dppq2<kShr1>(E), rt = dppq2<kShl1>(E);
            f32x2 H[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                H[dy] = l * w[3 * dy];
                H[dy] += E * w[3 * dy + 1];
                H[dy] += rt * w[3 * dy + 2];
            }
            f32x2 O = dsh + dppq2<kShr4>(H[0]);          // kernel row 0 applies to the input row above: take it from lane n - 4
            O += H[1];
            O += dppq2<kShl4>(H[2]);
            split2w(__builtin_amdgcn_fmed3f(O[0], 0.0f, 96.0f), __builtin_amdgcn_fmed3f(O[1], 0.0f, 96.0f), Bd, th);
        }
        // ---- project 1x1, K = this group ----
#pragma unroll
        for (int mt = MT0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < 2; ++p) Ap[mt][p] = *(const u32x4 *)&Wp[(mt * 2 + p) * 256 + l4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = mfmaq(Ap[mt][1], Bd[0], acc[mt]);
            acc[mt] = mfmaq(Ap[mt][0], Bd[1], acc[mt]);
            acc[mt] = mfmaq(Ap[mt][0], Bd[0], acc[mt]);
        }
        if (G + 1 < C::NG) park((G + 1) & 1);
    }

    // ---- rescale, BN shift, residual, NHWC store: lane (n, g) holds channels 16 mt + 4 g .. + 3 of pixel n ----
    if (!real) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int nch = 16 * mt + g4;
        const size_t at = ((size_t)f * 16 + n) * COUT + nch;
        f32x4 v = acc[mt] * inv_p + *(const f32x4 *)&p_shift[nch];
        if (C::RES) v += *(const f32x4 *)&X[at];
        *(f32x4 *)&Y[at] = v;
    }
}

template <class C>
static void launch_lb4(const FusedBlockArgs &a, int B, hipStream_t s) {
    fused_block_lb4_kernel<C><<<(B + 3) / 4, 256, 0, s>>>(a.X, a.Glb, a.p_shift, a.Y, B);
}

using Q15 = Lb4Cfg<160, 960, 160, true>;      // features.15, 16
using Q17 = Lb4Cfg<160, 960, 320, false>;     // features.17

bool launch_fused_block_lb4(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.Glb || a.prof) return false;
    static const int min_b = getenv("SYN_LB4_MIN") ? atoi(getenv("SYN_LB4_MIN")) : 768;     // fewer faces: not enough workgroups of four
    if (B < min_b) return false;
    switch (feature) {
        case 15: case 16: launch_lb4<Q15>(a, B, s); return true;
        case 17: if (getenv("SYN_LB4_17")) { launch_lb4<Q17>(a, B, s); return true; } return false;    // (measured: 88 vs 80 us for the tiled kernel)
        default: return false;
    }
}

}  // namespace syn
