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
//     accumulators too: the waves walk the 30 hidden groups with no partial sums to exchange;
//   * the WEIGHTS go through LDS: the eight waves of a workgroup (four faces) walk the groups in lockstep and share one fetch of each group's
//     fragments + constants -- one contiguous 48 | 64 KB run of the packed blob, an eighth per wave, fetched into registers while the
//     current group computes and written to the other half of a double buffer at its end; one barrier per group.  (Each wave
//     fetching its own fragments would pull 1.2 MB through L2 per face.)
//   * TWO waves per face (B = 1024 offers four faces per CU, and a lone wave per SIMD issues vector instructions at half the
//     rate): wave (face, t) owns hidden tile t through expand and depthwise and half of the output tiles in the project; the
//     two halves of the project operand cross through LDS (16 bytes per lane, one extra barrier per group).
// Scales as in fused_block_lb.hip: X16 = 16 x;  D = 16 Se (We x + shift);  E = med3(D, 0, 96 Se);  O16 = 16 dshift + sum (taps / Se) E;
// B = med3(O16, 0, 96);  acc = 16 Sp (Wp o);  y = acc / (16 Sp) + pshift (+ x).
#include "syn_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ void split2q(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
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
    static constexpr int KE = CIN / 32, NG = HID / 32, MT = COUT / 16, MTW = MT / 2;         // MTW: output tiles per wave
    static constexpr int WE_DW = 2 * KE * 2 * 256, WP_DW = MT * 2 * 256, TB_DW = 2 * 256;   // a group's expand | project fragments | constants (12 x 32 floats + pad)
    static constexpr int NKB = ((WE_DW + WP_DW + TB_DW) / 256 + 7) / 8 * 8;               // 1 KB pieces per group, padded to one more round of the 8 waves (lb4_group_dwords)
    static constexpr int GRP_DW = NKB * 256;
    static constexpr int XCH_DW = 4 * 2 * 64 * 4;                                         // depthwise outputs exchanged between the two waves of a face
    static constexpr int LDS_DW = 2 * GRP_DW + XCH_DW;
    static_assert(CIN % 32 == 0 && HID % 32 == 0 && COUT % 32 == 0, "k32 steps, groups, two halves of the output tiles");
    static_assert(!RES || CIN == COUT, "residual only on same-width blocks");
    static_assert(LDS_DW * 4 <= 160 * 1024, "LDS budget");
};

// Workgroup = 4 faces x 2 waves: wave (face, t) owns hidden tile t (16 of the group's 32 channels) through expand and depthwise,
// hands its half of the project operand to its partner through LDS, and accumulates half of the output tiles.  Two waves per SIMD
// issue vector instructions at 2.4 cycles each instead of a lone wave's 5 (the depthwise is the larger part of a group).
//
// One block = one STAGE; features.15-17 run as a chain of three stages in one launch (fused_chain_lb4_kernel): the block output goes
// to the next stage as pre-split fragments through the weight buffer half the last group has just left, the residual of the next block
// stays in the accumulator's registers, and the next stage's first weight group is fetched during the last group of this one --
// no kernel boundary (10-16 us each), no global round trip of the 4x4 activations.
struct Lb4NoNext { static constexpr int NKB = 0; };
struct Lb4StageArgs {
    const float *X;          // block input (global): a FIRST stage's input and residual
    const unsigned *Glb;     // [NG][NKB][64][4]: We | Wp | table per group
    const float *p_shift;
    float *Y;                // block output (global): written by the last stage only
    float *part = nullptr;   // hidden-sliced schedule: partial sums [slice][B][16][COUT] (raw accumulators)
    int g0 = 0, g1 = 0;      // hidden groups [g0, g1) of this workgroup's slice (g1 = 0: all); slice = blockIdx.y
};

// SYN_LB4_ABL: TIMING-ONLY ablations (wrong results; tools/build_variant.sh): 1 no weight fetch from L2 | 2 no park into LDS | 4 no exchange
// barrier | 8 no project-fragment LDS reads | 16 no depthwise arithmetic | 32 no group barrier
#ifndef SYN_LB4_ABL
#define SYN_LB4_ABL 0
#endif
template <int N>
__device__ __forceinline__ void lb4_fetch(u32x4 *pf, const unsigned *src /* + 4 lane */, int wave) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (SYN_LB4_ABL & 1) { pf[i] = (u32x4){(unsigned)wave, 1u, 2u, 3u}; asm volatile("" : "+v"(pf[i])); }
        else pf[i] = *(const u32x4 *)(src + (wave + 8 * i) * 256);
    }
}
template <int N>
__device__ __forceinline__ void lb4_park(const u32x4 *pf, unsigned *dst /* + 4 lane */, int wave) {
    if (SYN_LB4_ABL & 2) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("" :: "v"(pf[i]));
        return;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) *(u32x4 *)&dst[(wave + 8 * i) * 256] = pf[i];
}

// GRPL: dwords per half of the LDS weight double buffer (>= the group run of every stage of the kernel)
// PARTIAL (small batches): the workgroup (blockIdx.x = four faces, blockIdx.y = slice) walks only the hidden groups of its slice and
// stores its raw accumulators; lb4_reduce_kernel adds the slices in fixed order, rescales, adds BN shift and residual.  A batch of 128
// faces is then 32 x 6 workgroups of five groups each instead of 32 workgroups walking all thirty.
template <class C, class CN, bool FIRST, int GRPL, bool PARTIAL = false>
__device__ __forceinline__ void lb4_stage(unsigned *smem, const Lb4StageArgs &sa, const unsigned *GlbNext, int B, u32x4 (&Xr)[5][2], f32x4 (&vres)[5]) {
    constexpr int KE = C::KE, MTW = C::MTW, CIN = C::CIN, COUT = C::COUT;
    constexpr bool HANDOFF = !__is_same(CN, void);
    static_assert(KE == 5 && C::NG % 2 == 0, "160 input channels; the double buffer's parity carries over to the next stage");
    const float *__restrict__ X = sa.X, *__restrict__ p_shift = sa.p_shift;
    const unsigned *__restrict__ Glb = sa.Glb;
    float *__restrict__ Y = sa.Y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = wave >> 1, t = wave & 1;
    const int f = blockIdx.x * 4 + fl;
    const bool real = f < B;
    const int fc = real ? f : B - 1;
    const int n = lane & 15, g = lane >> 4;
    const unsigned l4 = lane * 4, g4 = g * 4;
    unsigned *Xch = smem + 2 * GRPL;

    // every wave copies every eighth 1 KB piece of a group's run through registers: fetched at the start of the previous group,
    // written to the other half of the double buffer at its end.  (LDS-DMA -- global_load_lds_dwordx4, no registers -- delivers
    // ~25 GB/s per CU whoever issues it: 44 KB per group took 1.8 us, longer than the group's arithmetic.)
    constexpr int NPW = C::NKB / 8;
    using CNX = typename std::conditional<HANDOFF, CN, Lb4NoNext>::type;
    constexpr int NPWN = CNX::NKB / 8;
    u32x4 pf[NPW > NPWN ? NPW : NPWN];
    const int gsl = PARTIAL ? (sa.g1 - sa.g0) : C::NG;   // groups per slice (all slices the same)
    const int gb = PARTIAL ? (int)blockIdx.y * gsl : 0, ge = gb + gsl;
    if (FIRST) {
        lb4_fetch<NPW>(pf, Glb + (size_t)gb * C::GRP_DW + l4, wave);
        // ---- block input of this face -> pre-split B fragments in registers (x 16; both waves of the face hold it) ----
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
    f32x4 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float c6e = 0.f, inv_p = 0.f;
    if (FIRST) lb4_park<NPW>(pf, smem + l4, wave);

    {   // ReLU6 ceiling of the scaled expand output, accumulator -> output: constants of the block, kept in group 0's table
        const float *t0 = reinterpret_cast<const float *>(Glb + C::WE_DW + C::WP_DW);
        c6e = t0[11 * 32]; inv_p = t0[11 * 32 + 1];
    }
    for (int G = gb; G < ge; ++G) {
        if (!(SYN_LB4_ABL & 32)) __syncthreads();        // every wave has written its pieces of group G and is done with group G-1
        if (G + 1 < ge) lb4_fetch<NPW>(pf, Glb + (size_t)(G + 1) * C::GRP_DW + l4, wave);
        else if (HANDOFF) lb4_fetch<NPWN>(pf, GlbNext + l4, wave);          // the next block's first group
        const unsigned *We = smem + ((G - gb) & 1) * GRPL, *Wp = We + C::WE_DW;
        const float *Tb = reinterpret_cast<const float *>(Wp + C::WP_DW);
#ifndef SYN_LB4_V1
        __builtin_amdgcn_sched_barrier(0);               // the next group's fetch stays in front of this group's LDS reads
#endif
        // fragments of this wave.  Round 5: the eight waves read 22 KB each here, the LDS delivers 128 bytes per cycle, and the expand chain
        // used to wait for ALL of it (the project fragments were pinned in front of it): ~1400 cycles per group in which nothing else ran.
        // Now only the expand fragments and the group's constants stand in front of the expand chain; the project fragments are requested
        // behind it and land during the depthwise arithmetic, which leaves the LDS idle.  (SYN_LB4_V1: the old order, for A/B runs.)
        u32x4 Ae[KE][2], Ap[MTW][2];
        // (the LDS returns in order: the accumulator's start value first, then the fragments in the order the chain consumes them, then the filter)
        f32x4 D = *(const f32x4 *)&Tb[10 * 32 + 16 * t + g4];
#pragma unroll
        for (int kc = 0; kc < KE; ++kc)
#pragma unroll
            for (int p = 1; p >= 0; --p) Ae[kc][p] = *(const u32x4 *)&We[((t * KE + kc) * 2 + p) * 256 + l4];
        constexpr int MTW0 = MTW > 5 ? MTW / 2 : MTW;    // (320 output channels: the second half of the project fragments after the exchange)
#ifdef SYN_LB4_V1
#pragma unroll
        for (int i = 0; i < MTW0; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) Ap[i][p] = *(const u32x4 *)&Wp[((t * MTW + i) * 2 + p) * 256 + l4];
#else
        // depthwise filter, its BN shift: two channels x two halves per lane group, requested with the expand fragments
        f32x2 wt[2][9], dsht[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wt[hf][k] = *(const f32x2 *)&Tb[k * 32 + 16 * t + 2 * hf + g4];
            dsht[hf] = *(const f32x2 *)&Tb[9 * 32 + 16 * t + 2 * hf + g4];
        }
#endif
        // ---- expand 1x1 + BN shift: D = channels 32 G + 16 t + 4 g + i of pixel n (three partial products, smallest first) ----
#ifndef SYN_LB4_V1
        __builtin_amdgcn_sched_barrier(0);               // every read above is issued before the first matrix instruction (the LDS returns in order: counted waits)
#endif
#pragma unroll
        for (int kc = 0; kc < KE; ++kc) {
            D = mfmaq(Ae[kc][1], Xr[kc][0], D);
            D = mfmaq(Ae[kc][0], Xr[kc][1], D);
            D = mfmaq(Ae[kc][0], Xr[kc][0], D);
        }
#ifdef SYN_LB4_V1
#pragma unroll
        for (int i = 0; i < MTW0; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(Ap[i][p]));      // (keeps these reads up here instead of next to their MFMAs)
#else
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MTW0; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (SYN_LB4_ABL & 8) Ap[i][p] = Ae[i % KE][p];
                else Ap[i][p] = *(const u32x4 *)&Wp[((t * MTW + i) * 2 + p) * 256 + l4];
            }
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- ReLU6, depthwise 3x3 + BN shift + ReLU6 on the registers, split into this wave's half of the project operand ----
        u32x4 own;                                      // {piece 0 dwords hf 0, 1 | piece 1 dwords hf 0, 1}
        if (SYN_LB4_ABL & 16) own = __builtin_bit_cast(u32x4, D);
        else
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#ifdef SYN_LB4_V1
            const int c0 = 16 * t + 2 * hf;             // + 4 g per lane group
            f32x2 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x2 *)&Tb[k * 32 + c0 + g4];
            const f32x2 dsh = *(const f32x2 *)&Tb[9 * 32 + c0 + g4];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { w[3 * dy] *= mL; w[3 * dy + 2] *= mR; }
#else
            const f32x2 (&w)[9] = wt[hf];
            const f32x2 dsh = dsht[hf];
#endif
            f32x2 E;
            // ReLU6 as clamp modifiers (fused_block_lb.hip): E = relu6 / 6 = clamp(D / (96 Se)); the last add of the output clamps too
            E[0] = __builtin_amdgcn_fmed3f(D[2 * hf] * c6e, 0.0f, 1.0f);
            E[1] = __builtin_amdgcn_fmed3f(D[2 * hf + 1] * c6e, 0.0f, 1.0f);
            // horizontal first (two lane shifts), then the three row sums shifted vertically (two more): 8 DPP moves per channel pair
            // instead of 16.  (Summation order differs from the other kernels': dx inside dy inside the vertical sum.)
#ifdef SYN_LB4_V1
            const f32x2 l = dppq2<kShr1>(E), rt = dppq2<kShl1>(E);
#else
            // the image border in x: the shifted VALUE is zeroed (two multiplies) instead of six filter entries -- the same products, bit for bit
            const f32x2 l = dppq2<kShr1>(E) * mL, rt = dppq2<kShl1>(E) * mR;
#endif
            f32x2 H[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                H[dy] = l * w[3 * dy];
                H[dy] += E * w[3 * dy + 1];
                H[dy] += rt * w[3 * dy + 2];
            }
            f32x2 O = dsh + dppq2<kShr4>(H[0]);          // kernel row 0 applies to the input row above: take it from lane n - 4
            O += H[1];
            {
                const f32x2 h2 = dppq2<kShl4>(H[2]);
                asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(O) : "v"(O), "v"(h2));
            }
            unsigned a, b;
            split2q(O[0], O[1], a, b);
            own[hf] = a; own[2 + hf] = b;
        }
        // ---- the partner's half: K slots 0-3 of a lane group are tile 0's channels, 4-7 tile 1's ----
        *(u32x4 *)&Xch[((fl * 2 + t) * 64 + lane) * 4] = own;
        if (!(SYN_LB4_ABL & 4)) __syncthreads();
        const u32x4 oth = *(const u32x4 *)&Xch[((fl * 2 + (1 - t)) * 64 + lane) * 4];
#pragma unroll
        for (int i = MTW0; i < MTW; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) Ap[i][p] = *(const u32x4 *)&Wp[((t * MTW + i) * 2 + p) * 256 + l4];
        u32x4 Bd[2];
        Bd[0] = t == 0 ? (u32x4){own[0], own[1], oth[0], oth[1]} : (u32x4){oth[0], oth[1], own[0], own[1]};
        Bd[1] = t == 0 ? (u32x4){own[2], own[3], oth[2], oth[3]} : (u32x4){oth[2], oth[3], own[2], own[3]};
        // ---- project 1x1, K = this group, this wave's half of the output tiles ----
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            acc[i] = mfmaq(Ap[i][1], Bd[0], acc[i]);
            acc[i] = mfmaq(Ap[i][0], Bd[1], acc[i]);
            acc[i] = mfmaq(Ap[i][0], Bd[0], acc[i]);
        }
        if (G + 1 < ge) lb4_park<NPW>(pf, smem + ((G + 1 - gb) & 1) * GRPL + l4, wave);
        else if (HANDOFF) lb4_park<NPWN>(pf, smem + l4, wave);               // (NG is even: the next block starts in half 0 again)
    }
    if constexpr (PARTIAL) {                             // raw accumulators of this slice -> [slice][B][16][COUT]
        if (real) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
                *(f32x4 *)&sa.part[(((size_t)blockIdx.y * B + f) * 16 + n) * COUT + 16 * (t * MTW + i) + g4] = acc[i];
        }
        return;
    }

    // ---- rescale, BN shift, residual: lane (n, g) holds channels 16 mt + 4 g .. + 3 of pixel n.  The last stage stores NHWC; the
    //      others keep the result as the next block's residual and hand it over as B fragments ----
    f32x4 vout[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int nch = 16 * (t * MTW + i) + g4;
        const size_t at = ((size_t)fc * 16 + n) * COUT + nch;
        f32x4 v = acc[i] * inv_p + *(const f32x4 *)&p_shift[nch];
        if constexpr (C::RES) {
            if (FIRST) v += *(const f32x4 *)&X[at];
            else v += vres[i];
        }
        if (!HANDOFF && real) *(f32x4 *)&Y[at] = v;
        vout[i] = v;
    }
    if constexpr (HANDOFF) {
        static_assert(CN::CIN == COUT && MTW == 5, "the next block takes this block's output: five output tiles per wave");
        unsigned *HO = smem + GRPL;                      // half 1: the last group's weights, no longer read once every wave is here
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            vres[i] = vout[i];
            // channels 16 mt + 4 g .. + 3 of pixel n: k32 step mt >> 1, lane group 2 (mt & 1) + (g >> 1), dwords 2 (g & 1), + 1
            const int mt = t * MTW + i, kc = mt >> 1, lt = (2 * (mt & 1) + (g >> 1)) * 16 + n, dw = 2 * (g & 1);
            const f32x4 v = real ? vout[i] * 16.0f : (f32x4){0.f, 0.f, 0.f, 0.f};
            unsigned a0, b0, a1, b1;
            split2q(v[0], v[1], a0, b0);
            split2q(v[2], v[3], a1, b1);
            *(u32x2 *)&HO[((fl * KE + kc) * 2 + 0) * 256 + lt * 4 + dw] = (u32x2){a0, a1};
            *(u32x2 *)&HO[((fl * KE + kc) * 2 + 1) * 256 + lt * 4 + dw] = (u32x2){b0, b1};
        }
        __syncthreads();
#pragma unroll
        for (int kc = 0; kc < KE; ++kc)
#pragma unroll
            for (int p = 0; p < 2; ++p) Xr[kc][p] = *(const u32x4 *)&HO[((fl * KE + kc) * 2 + p) * 256 + l4];
        // (half 1 is rewritten at the end of the next block's first group, two barriers from here)
    }
}

template <class C, bool PARTIAL = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_block_lb4_kernel(Lb4StageArgs sa, int B) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DW];
    u32x4 Xr[5][2];
    f32x4 vres[5];
    lb4_stage<C, void, true, C::GRP_DW, PARTIAL>(smem, sa, nullptr, B, Xr, vres);
}

// y = (slice 0 + slice 1 + ... in this order) / (16 Sp) + BN shift (+ x): one thread per four channels of a pixel
template <class C>
__global__ __launch_bounds__(256) void lb4_reduce_kernel(const float *__restrict__ part, int S, const unsigned *__restrict__ Glb,
                                                         const float *__restrict__ p_shift, const float *__restrict__ X,
                                                         float *__restrict__ Y, int B) {
    constexpr int C4 = C::COUT / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)B * 16 * C4;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const float inv_p = reinterpret_cast<const float *>(Glb + C::WE_DW + C::WP_DW)[11 * 32 + 1];
    f32x4 a = *(const f32x4 *)&part[(size_t)idx * 4];
    for (int sl = 1; sl < S; ++sl) a += *(const f32x4 *)&part[((size_t)sl * total + idx) * 4];
    f32x4 v = a * inv_p + *(const f32x4 *)&p_shift[4 * c4];
    if (C::RES) v += *(const f32x4 *)&X[(size_t)idx * 4];
    *(f32x4 *)&Y[(size_t)idx * 4] = v;
}

using Q15 = Lb4Cfg<160, 960, 160, true>;      // features.15, 16
using Q17 = Lb4Cfg<160, 960, 320, false>;     // features.17

// features.15, 16, 17 of four faces in one launch
constexpr int kChain4Grp = Q17::GRP_DW > Q15::GRP_DW ? Q17::GRP_DW : Q15::GRP_DW;
constexpr int kChain4LdsDw = 2 * kChain4Grp + Q15::XCH_DW;
static_assert(kChain4LdsDw * 4 <= 160 * 1024 && 4 * 5 * 2 * 256 <= kChain4Grp, "LDS budget; the handed-over fragments of four faces fit one buffer half");
struct Lb4ChainArgs { Lb4StageArgs s[3]; };

// L2 warm-up of the three weight runs at kernel start (syn_internal.h l2_touch): measured SLOWER here -- 167.4 against 163.3 us, interleaved on one
// box (gpurun_out/r5c3), while the same touch gains 4.5 us in the features.7-14 chain: this kernel's groups are paced by the LDS (224 KB through
// it per group) and its two barriers, not by where the weight lines come from.  Off; SYN_LB4_L2_TOUCH=1 builds it for A/B runs.
#ifndef SYN_LB4_L2_TOUCH
#define SYN_LB4_L2_TOUCH 0
#endif
#define SYN_L2_TOUCH SYN_LB4_L2_TOUCH
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_chain_lb4_kernel(Lb4ChainArgs ca, int B) {
    __shared__ __attribute__((aligned(16))) unsigned smem[kChain4LdsDw];
    u32x4 Xr[5][2];
    f32x4 vres[5];
    // the three blocks' weight runs (1.2 + 1.2 + 1.8 MB) into this XCD's L2 before the first group is needed (syn_internal.h l2_touch)
    unsigned sink = 0;
    if (SYN_L2_TOUCH) {
        const unsigned gi = (blockIdx.x >> 3) * 512u + threadIdx.x, nth = ((gridDim.x + 7) >> 3) * 512u;
        l2_touch(ca.s[0].Glb, Q15::NG * Q15::GRP_DW * 4u, gi, nth, sink);
        l2_touch(ca.s[1].Glb, Q15::NG * Q15::GRP_DW * 4u, gi, nth, sink);
        l2_touch(ca.s[2].Glb, Q17::NG * Q17::GRP_DW * 4u, gi, nth, sink);
    }
    lb4_stage<Q15, Q15, true, kChain4Grp>(smem, ca.s[0], ca.s[1].Glb, B, Xr, vres);
    lb4_stage<Q15, Q17, false, kChain4Grp>(smem, ca.s[1], ca.s[2].Glb, B, Xr, vres);
    lb4_stage<Q17, void, false, kChain4Grp>(smem, ca.s[2], nullptr, B, Xr, vres);
    if (SYN_L2_TOUCH) l2_touch_done(sink);
}

template <class C>
static void launch_lb4(const FusedBlockArgs &a, int B, hipStream_t s) {
    fused_block_lb4_kernel<C><<<(B + 3) / 4, 512, 0, s>>>(Lb4StageArgs{a.X, a.Glb, a.p_shift, a.Y}, B);
}

// small batches: S slices of the hidden groups per four faces (S divides the 30 groups; >= ~192 workgroups), partial sums in `scratch`
template <class C>
static bool launch_lb4_sliced(const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.scratch) return false;
    const int wg = (B + 3) / 4;
    static const int divs[] = {2, 3, 5, 6, 10, 15, 30};
    // the fewest slices that give 192 workgroups (more slices lose: 320 / 480 workgroups cost the landmarks-only step +21 / +18 us at B = 128,
    // +25 / +30 at B = 256 -- every slice streams its own partial tensor through the reduce launch)
    int S = 30;
    for (int d : divs) if (wg * d >= 192) { S = d; break; }
    if ((size_t)S * B * 16 * C::COUT > a.scratch_floats) return false;
    Lb4StageArgs sa{a.X, a.Glb, a.p_shift, a.Y};
    sa.part = a.scratch; sa.g0 = 0; sa.g1 = C::NG / S;
    fused_block_lb4_kernel<C, true><<<dim3(wg, S), 512, 0, s>>>(sa, B);
    const long total = (long)B * 16 * (C::COUT / 4);
    lb4_reduce_kernel<C><<<(int)((total + 255) / 256), 256, 0, s>>>(a.scratch, S, a.Glb, a.p_shift, a.X, a.Y, B);
    return true;
}

// features.17 of a small batch: the slices only -- the tail's staging adds them (head_kernel.hip, SIN).  B < 513: the two-face tails.
bool launch_lb4_sliced17_deferred(const FusedBlockArgs &a, int B, hipStream_t s, HeadSliced *out) {
    static const bool on = test_knob("head_sliced_in", 1) != 0;
    static const int wide_min = (int)test_knob("head_wide_min", 513);
    if (!on || !out || !a.Glb || a.prof || !a.scratch || B >= wide_min || B >= 576) return false;
    using C = Q17;
    const int wg = (B + 3) / 4;
    static const int divs[] = {2, 3, 5, 6, 10, 15, 30};
    int S = 30;
    for (int d : divs) if (wg * d >= 192) { S = d; break; }
    // only with two or three slices (>= 253 faces): the tail's staging pays a round of loads per pair of slices -- landmarks-only step, deferred against
    // the reduce launch (ms): B = 512 0.610 / 0.616, 256 0.3744 / 0.3773, 128 (six slices) 0.2837 / 0.2845, 64 0.2575 / 0.2567, 1 (thirty) 0.2167 / 0.2068
    if (S > 3) return false;
    if ((size_t)S * B * 16 * C::COUT > a.scratch_floats) return false;
    Lb4StageArgs sa{a.X, a.Glb, a.p_shift, a.Y};
    sa.part = a.scratch; sa.g0 = 0; sa.g1 = C::NG / S;
    fused_block_lb4_kernel<C, true><<<dim3(wg, S), 512, 0, s>>>(sa, B);
    *out = HeadSliced{a.scratch, S, reinterpret_cast<const float *>(a.Glb + C::WE_DW + C::WP_DW) + 11 * 32 + 1, a.p_shift};
    return true;
}

static int lb4_min_batch() {
    constexpr int min_b = 576;     // fewer faces: hidden-sliced (B = 512: 123 us for the three blocks, 640: 209; the chain: 167 whatever the batch up to 1024)
    return min_b;
}

// a[i] = the arguments of features.(15 + i); false: launch them one by one
bool launch_fused_chain_lb4(const FusedBlockArgs *a, int B, hipStream_t s) {
    static const int chain = (int)test_knob("lb4_chain", 1);
    if (!chain || B < lb4_min_batch()) return false;
    Lb4ChainArgs ca;
    for (int i = 0; i < 3; ++i) {
        if (!a[i].Glb || a[i].prof) return false;
        ca.s[i] = Lb4StageArgs{a[i].X, a[i].Glb, a[i].p_shift, a[i].Y};
    }
    fused_chain_lb4_kernel<<<(B + 3) / 4, 512, 0, s>>>(ca, B);
    return true;
}

bool launch_fused_block_lb4(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.Glb || a.prof) return false;
    if (B < lb4_min_batch()) {
        // (round 4: hidden-sliced from ONE face on -- thirty one-group workgroups per face quad and a reduce launch take ~15 us per block
        // where the tiled kernel's own sliced schedule, the choice below 32 faces until then, took 43-45: B = 1 0.333 -> ms)
        switch (feature) {
            case 15: case 16: return launch_lb4_sliced<Q15>(a, B, s);
            case 17: return launch_lb4_sliced<Q17>(a, B, s);
            default: return false;
        }
    }
    switch (feature) {
        case 15: case 16: launch_lb4<Q15>(a, B, s); return true;
        case 17: launch_lb4<Q17>(a, B, s); return true;
        default: return false;
    }
}

}  // namespace syn
