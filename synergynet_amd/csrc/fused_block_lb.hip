// Register-resident fused inverted-residual blocks for the 8x8 MobileNetV2 blocks (features.8-13), the two stride-2 blocks around them
// (features.7: 15x15 -> 8x8, lb7_stage; features.14: 8x8 -> 4x4, LbCfg::S2) and the CHAIN kernel that runs features.7-14 of a face in one
// launch (fused_chain_lb_kernel: a block = one stage, the block output handed to the next stage as MFMA fragments through LDS); for
// small batches the same stage hidden-sliced over several workgroups (PARTIAL + lb_reduce_kernel).
// Reference: backbone_nets/mobilenetv2_backbone.py:45-74 (InvertedResidual.forward), :33-42 (ConvBNReLU).
//
// The tiled kernel (fused_block_f16.hip) walks the hidden width in chunks behind two workgroup barriers per chunk and moves
// every hidden activation through LDS twice; with two 4-wave workgroups per CU the matrix pipe is busy ~30 % of the time
// (profiles/r2/stage_profile_b1024.txt).  Here ONE WAVE carries a whole face through a hidden group of 32 channels without
// leaving registers, and a face is shared by two waves that split the hidden groups between them (even / odd):
//
//   * pixel layout: the 64 pixels of a face are four 16-column blocks of v_mfma_f32_16x16x32_bf16; block r, lane column
//     n = l & 15 is pixel (y = r + 4 (n >> 3), x = n & 7).  The vertical neighbours of a pixel are then the SAME LANE of
//     blocks r-1 / r+1 -- except row 3 <-> row 4, which is a shift by 8 lanes inside a 16-lane DPP row (row_shr:8 / row_shl:8,
//     whose zero fill is exactly the image border) -- and the horizontal ones are row_shr:1 / row_shl:1, with the filter
//     column zeroed on the lanes where that shift crosses x = 0 | 7;
//   * both GEMMs run on v_mfma_f32_16x16x32_f16 with every fp32 operand carried as TWO fp16 pieces, x = a + b (a = fp16(x),
//     b = fp16(x - a), both toward zero: 22 significant bits) and three products a a + a b + b a (b b <= 2^-22 dropped): fp16 x fp16
//     is exact in fp32 and the accumulation is fp32, so the result is fp32-class at HALF the matrix instructions of the 3-way
//     bf16 split (6 products).  fp16's narrow exponent is met by power-of-two operand scales folded into the constants
//     (synergy_abi.hip): activations x 16, weights per layer to max |w| in [2^13, 2^14);
//   * expand: D[tile t of 16 channels][block r] = 16 Se (shift + We . X); the block input is staged ONCE per face as pre-split B
//     fragments in LDS, the weights stream from L2 straight into registers (buffer loads: one address register);
//   * depthwise 3x3 + BN shift + ReLU6 on the D registers (same tap order as the other kernels; filter / Se and 16 x shift come
//     pre-scaled), split into fp16 pieces in place: lane group g = l >> 4 holds channels 4g..4g+3 of both tiles = the 8 K slots of
//     ONE k32 step of the project GEMM (the host packs the project weights in that K order);
//   * project: acc[out tile][block] += Wp[:, group] . D, accumulators in registers across all groups of the wave;
//   * the two waves of a face exchange half of their partial sums through LDS at the end (stream 0 + stream 1, fixed order),
//     rescale, add the BN shift and the residual and store NHWC.  Barriers: one after staging, two at the end.
//   * MFMA and VALU instructions of the two waves of a SIMD do not overlap (tools/ubench/mfma_valu_kinds.hip: t = t_mfma + t_valu
//     for every instruction kind), so the cost of a group is the SUM of its matrix and vector instruction time: halving the
//     products and shrinking the split (6 instead of 11 instructions per pair) is what pays, not occupancy.
#include "syn_internal.h"

#include <cstdio>
#include <cstdlib>

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
// x0, x1 -> packed fp16 pieces a (high) and b (low), x = a + b to 22 bits
__device__ __forceinline__ void split2h(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
// ... into component d of the two piece vectors
__device__ __forceinline__ void split2v(float x0, float x1, u32x4 (&pc)[2], int d) {
    unsigned a, b;
    split2h(x0, x1, a, b);
    pc[0][d] = a; pc[1][d] = b;
}
__device__ __forceinline__ f32x4 mfmal(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the three partial products (smallest first) of four independent accumulators, round-robin
__device__ __forceinline__ void mac3x4(const u32x4 (&a0)[2], const u32x4 (&b0)[2], f32x4 &c0, const u32x4 (&a1)[2], const u32x4 (&b1)[2], f32x4 &c1,
                                       const u32x4 (&a2)[2], const u32x4 (&b2)[2], f32x4 &c2, const u32x4 (&a3)[2], const u32x4 (&b3)[2], f32x4 &c3) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        c0 = mfmal(a0[PA[j]], b0[PB[j]], c0);
        c1 = mfmal(a1[PA[j]], b1[PB[j]], c1);
        c2 = mfmal(a2[PA[j]], b2[PB[j]], c2);
        c3 = mfmal(a3[PA[j]], b3[PB[j]], c3);
    }
}
__device__ __forceinline__ u32x4 bload4(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
__device__ __forceinline__ f32x4 bload4f(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
// (by value: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index -- clang 19 / ROCm 7.2)
template <int CTRL>
__device__ __forceinline__ float dpp1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ f32x2 dpp2(f32x2 v) {
    f32x2 r;
    r[0] = dpp1<CTRL>(v[0]);
    r[1] = dpp1<CTRL>(v[1]);
    return r;
}
// ReLU6 without instructions of its own (round 4, tools/ubench/valu_issue.hip: v_med3_f32 costs 1.85 ns of a SIMD, an fp32 multiply 1.25, a
// packed FMA 2.2): activations are carried as relu6(x) / 6 in [0, 1] = what the `clamp` output modifier leaves -- on the multiply that
// rescales the expand accumulator (compiler-made: a vector instruction that reads a matrix result needs wait states only the compiler
// inserts) and on the LAST packed FMA of a depthwise output (inline assembly: its operands are vector results).  The 6 rides on the
// constants: expand multiplier 1 / (96 Se), the plain depthwise filter, depthwise shift / 6, output rescale 6 / Sp (synergy_abi.hip).
__device__ __forceinline__ float relu01(float d, float m) { return __builtin_amdgcn_fmed3f(d * m, 0.0f, 1.0f); }
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
constexpr int kRowShr1 = 0x111, kRowShl1 = 0x101, kRowShr8 = 0x118, kRowShl8 = 0x108;   // lane n <- n-1 | n+1 | n-8 | n+8 of its 16-lane row, else 0
constexpr int kRowShr4 = 0x114, kRowShr5 = 0x115;
}  // namespace

template <int CIN_, int HID_, int COUT_, bool RES_, int EPF_, int PPF_, int FPW_ = 2, bool S2_ = false, int NS_ = 2>
struct LbCfg {
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_, FPW = FPW_;
    static constexpr int PPF = PPF_;                     // output tiles the project fragments are fetched ahead (1 | 2)
    static constexpr bool TLATE = false;                 // (true: the next group's table is fetched after the project instead of across it: 8 registers less, was needed with three bf16 pieces)
    static constexpr int EPF = EPF_;                     // k32 steps of the NEXT group's expand fragments fetched during the project (1 | KE)
    static constexpr bool RES = RES_;
    static constexpr bool S2 = S2_;                      // stride-2 depthwise (8x8 -> 4x4, features.14): see the pixel order below
    static constexpr int NB = S2 ? 1 : 4;                // 16-pixel blocks of the block OUTPUT
    static constexpr int PIXO = NB * 16;
    static constexpr int KE = CIN / 32;                  // k32 steps of the expand GEMM
    static constexpr int NG = HID / 32;                  // hidden groups
    static constexpr int MT = COUT / 16;                 // output channel tiles
    static constexpr int NS = NS_;                       // waves per face (hidden groups s, s + NS, ...).  4 (round 4): the small-batch chain --
                                                         // one face per workgroup, four streams, all partial sums through LDS (below)
    static constexpr int NW = FPW * NS, NT = NW * 64;
    static constexpr int XF_DW = KE * 4 * 2 * 256;       // block input of one face as fragments [KE][block 4][piece 2][lane 64][4 dwords]
    static constexpr int NSX = NS == 8 ? 4 : NS;         // streams that keep output tiles (eight streams: 4 .. 7 first add theirs into 0 .. 3)
    static constexpr int KT = (MT + NSX - 1) / NSX;      // output tiles a keeper wave keeps (mt % NSX == its stream)
    static constexpr int RED_DW = (NS == 2 ? 1 : NSX) * MT * NB * 256;  // exchange buffer of one face: [stream 2][MT / 2][block NB][lane 64][4]; NS > 2: [stream 4][MT][NB][lane][4]
    static constexpr int TB_DW = 12 * 32;                // per wave: depthwise filter [9][32] | depthwise shift | expand shift | (pad) of its current group
    static constexpr int FACE_DW = XF_DW > RED_DW ? XF_DW : RED_DW;      // the exchange buffer reuses the fragments of its face
    static constexpr int LDS_DW = FPW * FACE_DW + NW * TB_DW;
    static_assert(CIN % 32 == 0 && HID % 64 == 0 && COUT % 32 == 0, "k32 steps, two streams, two halves of the output tiles");
    static_assert(NS == 2 || (NS == 8 && FPW == 1 && (HID / 32) >= NS), "eight streams: the one-face-per-workgroup schedule of small batches (the four-stream form of round 4 is retired)");
    static_assert(EPF == 1 || EPF == KE, "expand prefetch depth");
    static_assert(!RES || (CIN == COUT && !S2), "residual only on same-width stride-1 blocks");
    static_assert((FPW == 4 || NS >= 4 ? 1 : 2) * LDS_DW * 4 <= 160 * 1024, "one 8-wave or two 4-wave workgroups per CU");
};

#ifndef SYN_LB_SHADOW
#define SYN_LB_SHADOW 0               // 1: channel tile 1's expand MFMAs inside depthwise passes 0 / 1 (in-wave MFMA shadow: the round-6 experiment,
                                     //    measured 252.9 against 249.0 us for the features.7-14 chain -- profiles/r6/mfma_shadow_experiment.txt); 0: rounds 2-5
#endif
#ifndef SYN_LB_SHADOW_PIN
#define SYN_LB_SHADOW_PIN 7           // vector instructions pinned behind each shadowed MFMA (sched_group_barrier); 0: the compiler's own order
#endif
#ifndef SYN_LB_DW2
#define SYN_LB_DW2 1                // 0: lane shifts on the input rows (rounds 2-4), for A/B runs
#endif
// compiler fence between the phases of a hidden group: without it every load of a group is hoisted to the top of the loop body
// and unchained arithmetic floats across the scheduling barriers (~370 registers live)
#define SYNL_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// PROF: s_memtime sums of every wave per phase {staging, expand, depthwise, project, exchange + store} into prof[0..4], the number
// of waves into prof[7] (syn_debug_profile_block)
#define SYNL_LAP(i) do { if (PROF) { tn = __builtin_amdgcn_s_memtime(); pt_[i] += tn - tk; tk = tn; } } while (0)

// One block of a face chain.  FIRST: the block input comes from global memory (X); otherwise the previous stage of the chain left it
// in LDS as fragments.  CN != void: the block output is handed to stage CN of the same kernel -- written to global memory as usual
// (the residual source of the next block, read back past the vector cache) AND split into the B fragments of CN's expand GEMM.
// A workgroup walks the stages of a chain at its own pace: no kernel boundary (= device-wide barrier), no launch gap and no global
// round trip of the block input between the blocks (measured 10-16 us per boundary, tools: the `fake` variant in DESIGN 5.9).
struct LbStageArgs {
    const float *X;          // block input (global): staging of a FIRST stage, residual of a RES stage
    const unsigned *Weh;     // [HID/16][KE][2][64][4]
    const float *Tlb;        // [NG][12][32]
    const unsigned *Wlb;     // [NG][MT][2][64][4]
    const float *p_shift;
    float *Y;                // block output (global)
    float *part = nullptr;   // hidden-sliced schedule (small batches): partial sums [slice][B][pixels][COUT] (raw accumulators)
    int gsl = 0;             // ... hidden groups per slice; slice = blockIdx.y
};

// PARTIAL (small batches, single block per launch): the workgroup (blockIdx.x = two faces, blockIdx.y = slice) walks only the hidden groups
// of its slice and stores the raw sums of its two streams; lb_reduce_kernel adds the slices in fixed order, rescales, adds BN shift and residual.
// RED_OFF > 0 (round 5, the eight-wave small-batch chain): the exchange buffer lives RED_OFF dwords behind the fragments instead of aliasing them,
// which makes the barrier in front of the exchange and the one in front of the hand-over unnecessary (two of a stage's six barriers;
// B = 128 landmarks-only step 0.308 -> 0.304 ms, with features.7 in the chain 0.300 -> 0.296: gpurun_out/r5c1/b128ab.txt)
template <class C, class CN, bool FIRST, bool PROF, int FACE_DW, bool STORE = true, bool PARTIAL = false, int RED_OFF = 0>
__device__ __forceinline__ void lb_stage(unsigned *smem, const LbStageArgs &sa, int B, unsigned long long (&pt_)[5], unsigned long long &tk) {
    unsigned long long tn = 0;
    const float *__restrict__ X = sa.X;
    const unsigned *__restrict__ Weh = sa.Weh, *__restrict__ Wlb = sa.Wlb;
    const float *__restrict__ Tlb = sa.Tlb, *__restrict__ p_shift = sa.p_shift;
    float *__restrict__ Y = sa.Y;
    constexpr int KE = C::KE, MT = C::MT, CIN = C::CIN, COUT = C::COUT, NB = C::NB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = wave / C::NS, st = wave % C::NS;
    const int f = blockIdx.x * C::FPW + fl;
    const bool real = f < B;
    const int fc = real ? f : B - 1;
    const int n = lane & 15, g = lane >> 4;
    const unsigned l4 = lane * 4, g4 = g * 4;
    unsigned *Xf = smem + fl * FACE_DW;
    constexpr int BPW = 4 / C::NS;                       // blocks of the block input a wave stages
    // input pixel of (block b, lane column n).  Stride 1: y = b + 4 (n >> 3), x = n & 7.  Stride 2: the blocks are the four parity
    // classes, b = 2 (y & 1) + (x & 1), and n = 4 (y >> 1) + (x >> 1) -- output pixel (oy, ox) = lane 4 oy + ox then takes its nine
    // taps from its own lane (rows 2 oy, 2 oy + 1 / columns 2 ox, 2 ox + 1) and from lanes n - 4 (row 2 oy - 1), n - 1 (column 2 ox - 1)
    // and n - 5: row_shr:4 | 1 | 5, zero fill = the top border, the filter's left column zeroed where ox = 0.
    auto pix_in = [&](int b, int nn) __attribute__((always_inline)) {
        return C::S2 ? 16 * (nn >> 2) + 8 * (b >> 1) + 2 * (nn & 3) + (b & 1) : 32 * (nn >> 3) + (nn & 7) + 8 * b;
    };

    // ---- stage: block input of this face -> pre-split B fragments (this wave: blocks 2 st, 2 st + 1) ----
    if constexpr (FIRST && C::NS == 8) {
        // eight streams: the KE x 4 (k32 step, block) pieces of the block input dealt over the waves
        constexpr int NIT = (KE * 4 + 7) / 8;
        f32x4 xv[NIT][2];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = st + 8 * i, itc = it < KE * 4 ? it : 0;
            const float *src = X + ((size_t)fc * 64 + pix_in(itc & 3, n)) * CIN + 32 * (itc >> 2) + 8 * g;
            xv[i][0] = *(const f32x4 *)src;
            xv[i][1] = *(const f32x4 *)(src + 4);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = st + 8 * i;
            f32x4 a = xv[i][0], b = xv[i][1];
            if (!real) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }
            a *= 16.0f; b *= 16.0f;
            u32x4 pc[2];
            split2v(a[0], a[1], pc, 0);
            split2v(a[2], a[3], pc, 1);
            split2v(b[0], b[1], pc, 2);
            split2v(b[2], b[3], pc, 3);
            if (it < KE * 4) {
#pragma unroll
                for (int p = 0; p < 2; ++p) *(u32x4 *)&Xf[(it * 2 + p) * 256 + lane * 4] = pc[p];      // it = kc * 4 + block
            }
        }
    } else if (FIRST) {
        f32x4 xv[KE][BPW][2];
#pragma unroll
        for (int kc = 0; kc < KE; ++kc)
#pragma unroll
            for (int rr = 0; rr < BPW; ++rr) {
                const float *src = X + ((size_t)fc * 64 + pix_in(BPW * st + rr, n)) * CIN + 32 * kc + 8 * g;
                xv[kc][rr][0] = *(const f32x4 *)src;
                xv[kc][rr][1] = *(const f32x4 *)(src + 4);
            }
#pragma unroll
        for (int kc = 0; kc < KE; ++kc)
#pragma unroll
            for (int rr = 0; rr < BPW; ++rr) {
                f32x4 a = xv[kc][rr][0], b = xv[kc][rr][1];
                if (!real) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }
                a *= 16.0f; b *= 16.0f;
                u32x4 pc[2];
                split2v(a[0], a[1], pc, 0);
                split2v(a[2], a[3], pc, 1);
                split2v(b[0], b[1], pc, 2);
                split2v(b[2], b[3], pc, 3);
#pragma unroll
                for (int p = 0; p < 2; ++p) *(u32x4 *)&Xf[((kc * 4 + BPW * st + rr) * 2 + p) * 256 + lane * 4] = pc[p];
            }
    }
    const float mL = (C::S2 ? (n & 3) : (n & 7)) != 0 ? 1.f : 0.f, mR = (C::S2 || (n & 7) != 7) ? 1.f : 0.f;
    f32x4 acc[MT][NB];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < NB; ++r) acc[mt][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Per-group constants (depthwise filter and the two BN shifts of 32 channels, 1.4 KB) go through a private LDS buffer of the
    // wave: fetched (two 16-byte loads per lane) at the start of the previous group's project, written at its end -- an L2 round
    // trip per use would otherwise stand exposed five times per group with only two waves per SIMD to cover it.
    float *Tb = reinterpret_cast<float *>(smem + C::FPW * FACE_DW + wave * C::TB_DW);
    // weights and tables through buffer loads: ONE address register (16 * lane) for every fragment, the rest is scalar -- flat
    // addressing keeps a 64-bit lane pointer per 4 KB of fragment range alive across the loop (~25 registers)
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(Weh), 0, 0x7fffffff, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(Wlb), 0, 0x7fffffff, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Tlb), 0, 0x7fffffff, 0x00027000);
    const unsigned l16 = lane * 16;
    f32x4 tv[2];
    auto fetch_t = [&](int G) __attribute__((always_inline)) {
        tv[0] = bload4f(rs_t, l16, G * (C::TB_DW * 4));
        tv[1] = bload4f(rs_t, l16 & 511, G * (C::TB_DW * 4) + 1024);            // rows 8..11; lanes 32-63 duplicate lanes 0-31
    };
    auto park_t = [&]() __attribute__((always_inline)) {
        *(f32x4 *)&Tb[l4] = tv[0];
        *(f32x4 *)&Tb[256 + (l4 & 127)] = tv[1];
    };
    // Weight fragments are fetched ahead of their use (expand: the next k32 step, and EPF steps of the next group during the
    // project; project: two output tiles ahead) and compiler fences keep every load of a group from being hoisted to its top.
    const float c6e = Tlb[11 * 32], inv_p = Tlb[11 * 32 + 1];       // ReLU6 ceiling of the scaled expand output; accumulator -> output
    u32x4 Ae[C::EPF == 1 ? 2 : KE][2][2];
    auto fetch_e = [&](int G, int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                Ae[kc % (C::EPF == 1 ? 2 : KE)][t][p] = bload4(rs_e, l16, G * (2 * KE * 2048) + ((t * KE + kc) * 2 + p) * 1024);
    };
    const int gb = PARTIAL ? (int)blockIdx.y * sa.gsl : 0, gend = PARTIAL ? gb + sa.gsl : C::NG;     // this workgroup's hidden groups
    fetch_t(gb + st);
#pragma unroll
    for (int kc = 0; kc < C::EPF; ++kc) fetch_e(gb + st, kc);
    park_t();
    __syncthreads();
    SYNL_LAP(0);

    for (int G = gb + st; G < gend; G += C::NS) {
        // ---- expand 1x1 (bf16 x3) + BN shift + ReLU6: D[t][r], channels 32 G + 16 t + 4 g + i of pixel (r, n) ----
        f32x4 D[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 es = *(const f32x4 *)&Tb[10 * 32 + 16 * t + g4];
#pragma unroll
            for (int r = 0; r < 4; ++r) D[t][r] = es;
        }
        // SYN_LB_SHADOW (round 6, VERDICT r5 #3 -- the in-wave interleave experiment): channel tile 0 of the group is expanded first (all k32 steps,
        // four blocks = four accumulator chains); tile 1's 12 MFMAs per k32 step ride INSIDE depthwise passes 0 and 1 -- which read tile 0 only --
        // so that this wave's vector instructions issue in the shadow of its own matrix instructions (tools/ubench/mfma_interleave.hip: two
        // plain VALU hide behind a 16-cycle MFMA inside one wave).  No register of another group is needed; the block-input fragments are read
        // twice.  Stages with EPF == KE == 2 (features.8-11).
        constexpr bool SHADOW = SYN_LB_SHADOW && C::EPF == KE && KE == 2 && !C::S2;
        auto expand_unit = [&](int kc, int t) __attribute__((always_inline)) {       // one k32 step of one channel tile: 12 MFMAs on D[t][0..3]
            u32x4 Bx[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int p = 0; p < 2; ++p) Bx[r][p] = *(const u32x4 *)&Xf[((kc * 4 + r) * 2 + p) * 256 + lane * 4];
            mac3x4(Ae[kc][t], Bx[0], D[t][0], Ae[kc][t], Bx[1], D[t][1], Ae[kc][t], Bx[2], D[t][2], Ae[kc][t], Bx[3], D[t][3]);
        };
        if constexpr (SHADOW) {
#pragma unroll
            for (int kc = 0; kc < KE; ++kc) { expand_unit(kc, 0); SYNL_FENCE(); }
        } else {
#pragma unroll
        for (int kc = 0; kc < KE; ++kc) {
            if (kc + C::EPF < KE) fetch_e(G, kc + C::EPF);
            constexpr int SL = C::EPF == 1 ? 2 : KE;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {        // two blocks' fragments in flight, four accumulator chains
                u32x4 Bx[2][2];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int p = 0; p < 2; ++p) Bx[rr][p] = *(const u32x4 *)&Xf[((kc * 4 + r + rr) * 2 + p) * 256 + lane * 4];
                mac3x4(Ae[kc % SL][0], Bx[0], D[0][r], Ae[kc % SL][1], Bx[0], D[1][r], Ae[kc % SL][0], Bx[1], D[0][r + 1], Ae[kc % SL][1], Bx[1], D[1][r + 1]);
                SYNL_FENCE();
            }
        }
        }
        // EPF == KE: the next group's expand fragments are requested HERE, into the registers the expand just stopped reading -- the depthwise
        // phase (~1600 cycles) and the project stand between the request and its use instead of the project alone (~800 cycles against an L2
        // round trip of about that: with the loads replaced by constants the chain runs 270 -> 252 us, this placement gets 6 of those 18)
        // (SHADOW: after depthwise pass 1, when tile 1 has stopped reading them)
        if (!SHADOW && C::EPF == KE && G + C::NS < gend) {
#pragma unroll
            for (int kc = 0; kc < C::EPF; ++kc) fetch_e(G + C::NS, kc);
        }
        SYNL_FENCE();
        SYNL_LAP(1);
        // ---- depthwise 3x3 + BN shift + ReLU6, split in place into the B operand of the project step ----
        constexpr int RING = C::S2 ? 4 : C::PPF + 1;   // project fragment slots (stride 2: output tiles go in pairs)
        u32x4 Ap[RING][2];
        auto fetch_p = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 2; ++p) Ap[mt % RING][p] = bload4(rs_p, l16, G * (MT * 2048) + (mt * 2 + p) * 1024);
        };
        u32x4 Bd[NB][2];
        // two channels (one packed K dword) at a time: 18 filter registers live instead of 36
#pragma unroll
        for (int th = 0; th < 4; ++th) {
            const int t = th >> 1, hf = th & 1;
            if (th == 3) fetch_p(0);                    // first project fragments: in flight behind the last depthwise pass
            if constexpr (SHADOW) {
                if (th < 2) expand_unit(th, 1);        // tile 1, k32 step th: its MFMAs interleave with this pass's vector work (pattern below)
                if (th == 2 && G + C::NS < gend) {
#pragma unroll
                    for (int kc = 0; kc < C::EPF; ++kc) fetch_e(G + C::NS, kc);
                }
            }
            const int c0 = 16 * t + 2 * hf;             // + 4 g per lane group
            f32x2 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x2 *)&Tb[k * 32 + c0 + g4];
            const f32x2 dsh = *(const f32x2 *)&Tb[9 * 32 + c0 + g4];
            if (C::S2 || !SYN_LB_DW2) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) { w[3 * dy] *= mL; w[3 * dy + 2] *= mR; }
            }
            f32x2 E[4], O[NB];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                E[r][0] = relu01(D[t][r][2 * hf], c6e);
                E[r][1] = relu01(D[t][r][2 * hf + 1], c6e);
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) O[r] = dsh;
            if constexpr (C::S2) {
                // blocks E[2 py + px]; taps in the filter's own order (dy, dx), the shifted ones from the odd-row / odd-column classes
                O[0] += dpp2<kRowShr5>(E[3]) * w[0];
                O[0] += dpp2<kRowShr4>(E[2]) * w[1];
                O[0] += dpp2<kRowShr4>(E[3]) * w[2];
                asm volatile("" : "+v"(O[0]));
                O[0] += dpp2<kRowShr1>(E[1]) * w[3];
                O[0] += E[0] * w[4];
                O[0] += E[1] * w[5];
                asm volatile("" : "+v"(O[0]));
                O[0] += dpp2<kRowShr1>(E[3]) * w[6];
                O[0] += E[2] * w[7];
                O[0] = pk_fma_clamp01(E[3], w[8], O[0]);
            } else if (SYN_LB_DW2) {
            // Round 5: the lane shifts move from the INPUT rows to the OUTPUT rows.  A lane shift commutes with the per-channel filter weight, so
            //   out[r] = B + mL L(A) + mR R(C),   A / B / C = sum over dy of (left / centre / right filter column) x in[r + dy - 1]
            // needs two shifts per output row (4 rows) where shifting every input row needs two per input row (6 rows: the block's four and the two
            // halo rows): 20 instead of 28 lane shifts per channel pair, and the image border is a multiplier of the shifted SUM (no filter masking).
            // Another summation order than rounds 2-4 (dx outermost): same arithmetic class.
            const f32x2 mL2 = {mL, mL}, mR2 = {mR, mR};
            f32x2 rowT = E[3], rowB = E[0];
            asm volatile("" : "+v"(rowT), "+v"(rowB));
            rowT = dpp2<kRowShr8>(rowT);                 // image row -1 of the block columns (zero fill = the border)
            rowB = dpp2<kRowShl8>(rowB);                 // image row 4
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 &i0 = r == 0 ? rowT : E[r - 1], &i1 = E[r], &i2 = r == 3 ? rowB : E[r + 1];
                f32x2 A = i0 * w[0], Cc = i0 * w[2], Bc = O[r] + i0 * w[1];
                A += i1 * w[3]; Bc += i1 * w[4]; Cc += i1 * w[5];
                A += i2 * w[6]; Bc += i2 * w[7]; Cc += i2 * w[8];
                Bc += dpp2<kRowShr1>(A) * mL2;
                O[r] = pk_fma_clamp01(dpp2<kRowShl1>(Cc), mR2, Bc);
                asm volatile("" : "+v"(O[r]));           // (one output row at a time: three column sums live, not twelve)
            }
            } else {
            // input rows q = -1 .. 4 of the block rows (row q feeds outputs q - dy, dy = 0..2: ascending dy per output).  The pins
            // chain the rows: unchained arithmetic is otherwise scheduled all rows at once.
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                f32x2 &src = E[q == 0 ? 3 : (q == 5 ? 0 : q - 1)];
                asm volatile("" : "+v"(src));
                const f32x2 c = q == 0 ? dpp2<kRowShr8>(src) : (q == 5 ? dpp2<kRowShl8>(src) : src);
                const f32x2 l = dpp2<kRowShr1>(c), rt = dpp2<kRowShl1>(c);
#pragma unroll
                for (int dy = 2; dy >= 0; --dy) {
                    const int r = q - dy;
                    if (r < 0 || r > 3) continue;
                    O[r] += l * w[3 * dy];
                    O[r] += c * w[3 * dy + 1];
                    // dy == 2 (input row r + 2) is the last kernel row an output row receives: its last FMA clamps
                    if (dy == 2) O[r] = pk_fma_clamp01(rt, w[3 * dy + 2], O[r]);
                    else O[r] += rt * w[3 * dy + 2];
                    asm volatile("" : "+v"(O[r]));
                }
            }
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                split2v(O[r][0], O[r][1], Bd[r], th);
                if (hf) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(Bd[r][p]));
                }
            }
            if constexpr (SHADOW && SYN_LB_SHADOW_PIN > 0) {
                if (th < 2) {                           // 12 x (1 MFMA, SYN_LB_SHADOW_PIN vector instructions): the pass's ~95 VALU spread under the MFMAs
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, SYN_LB_SHADOW_PIN, 0);
                    }
                }
            }
            SYNL_FENCE();
        }
        SYNL_LAP(2);
        // ---- project 1x1 (bf16 x3), K = this group ----
        const bool more = G + C::NS < gend;
#pragma unroll
        for (int i = 1; i < C::PPF; ++i)
            if (i < MT) fetch_p(i);
        if (more) {
            if (!C::TLATE) fetch_t(G + C::NS);
            if (C::EPF != KE) {                                  // (one k32 step of them: the other slot is still being read -- see above for EPF == KE)
#pragma unroll
                for (int kc = 0; kc < C::EPF; ++kc) fetch_e(G + C::NS, kc);
            }
        }
        if constexpr (C::S2) {
            static_assert(!C::S2 || (C::PPF == 2 && MT % 2 == 0), "output tiles in pairs, two tiles fetched ahead");
#pragma unroll
            for (int mt = 0; mt < MT; mt += 2) {         // one 16-pixel block: two output tiles = two accumulator chains
                if (mt + 2 < MT) { fetch_p(mt + 2); fetch_p(mt + 3); }
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    acc[mt][0] = mfmal(Ap[mt % RING][PA[j]], Bd[0][PB[j]], acc[mt][0]);
                    acc[mt + 1][0] = mfmal(Ap[(mt + 1) % RING][PA[j]], Bd[0][PB[j]], acc[mt + 1][0]);
                }
                SYNL_FENCE();
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt + C::PPF < MT) fetch_p(mt + C::PPF);
                mac3x4(Ap[mt % RING], Bd[0], acc[mt][0], Ap[mt % RING], Bd[1], acc[mt][1], Ap[mt % RING], Bd[2], acc[mt][2], Ap[mt % RING], Bd[3], acc[mt][3]);
                SYNL_FENCE();
            }
        }
        if (more) {
            if (C::TLATE) fetch_t(G + C::NS);           // (no registers to carry the table through the project: an exposed L2 round trip per group)
            park_t();
        }
        SYNL_LAP(3);
    }

    // ---- exchange: wave `st` keeps the output tiles mt with (mt & 1) == st and hands the others to its partner ----
    float *Red = reinterpret_cast<float *>(Xf + RED_OFF);
    int le = lane;
    asm volatile("" : "+v"(le));                         // (output addresses are computed here, not carried through the loop)
    const int ne = le & 15, ge = le >> 4, pixe = C::S2 ? ne : 32 * (ne >> 3) + (ne & 7);      // output pixel of (block 0, lane column ne); block r adds 8 r
    // residual and BN shift of this wave's tiles: requested before the barriers, consumed after them (an L2 round trip otherwise
    // stands between the second barrier and the stores).  Inside a chain the residual is what this wave stored one stage ago, into
    // a buffer this CU read two stages ago: the load goes past the vector cache (sc0: miss in the CU's cache, served by the XCD's L2, where the store landed).
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, 0x7fffffff, 0x00027000);
    constexpr int KT = C::KT, NSW = C::NSX;              // (eight streams: four keepers, see below)
    f32x4 rs[KT][NB], psh[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        const int mtk = NSW * i + (st & (NSW - 1));      // the i-th output tile this wave keeps (may not exist when MT % NS != 0)
        const int nch = 16 * (mtk < MT ? mtk : MT - 1) + 4 * ge;
        psh[i] = *(const f32x4 *)&p_shift[nch];
#pragma unroll
        for (int r = 0; r < NB; ++r)
            if (C::RES && !PARTIAL) rs[i][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)(((fc * 64 + pixe + 8 * r) * COUT + nch) * 4), 0, FIRST ? 0 : 1));
    }
    if (RED_OFF == 0) __syncthreads();                   // every wave is done reading the fragments (which the exchange buffer aliases)
    constexpr bool HANDOFF = !__is_same(CN, void);
    f32x4 vout[KT][NB];
    if constexpr (C::NS == 2) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if ((mt & 1) == st) continue;
#pragma unroll
        for (int r = 0; r < NB; ++r) *(f32x4 *)&Red[(((st * (MT / 2) + (mt >> 1)) * NB + r) * 64 + lane) * 4] = acc[mt][r];
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if ((mt & 1) != st) continue;
        const int nch = 16 * mt + 4 * ge;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const f32x4 o = *(const f32x4 *)&Red[((((1 - st) * (MT / 2) + (mt >> 1)) * NB + r) * 64 + lane) * 4];
            f32x4 v = st == 0 ? acc[mt][r] + o : o + acc[mt][r];        // stream 0 + stream 1
            if constexpr (PARTIAL) {
                if (real) *(f32x4 *)&sa.part[(((size_t)blockIdx.y * B + f) * C::PIXO + pixe + 8 * r) * COUT + nch] = v;
                continue;
            }
            v = v * inv_p + psh[mt >> 1];
            if (C::RES) v += rs[mt >> 1][r];
            if (STORE && real) *(f32x4 *)&Y[((size_t)f * C::PIXO + pixe + 8 * r) * COUT + nch] = v;
            if (HANDOFF) vout[mt >> 1][r] = v;
        }
    }
    } else {
        // four streams: every wave publishes the tiles it does not keep, then adds the other three streams' copies of its own tiles in
        // stream order 0 + 1 + 2 + 3 (fixed: results do not depend on the position in the batch)
        static_assert(!PARTIAL, "the four-stream schedule reduces inside the workgroup");
        if constexpr (C::NS == 8) {
            // eight streams: streams 4 .. 7 hand ALL their sums to streams 0 .. 3 first (s + (s + 4), fixed order), which then meet as above
            if (st >= 4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < NB; ++r) *(f32x4 *)&Red[((((st - 4) * MT + mt) * NB + r) * 64 + lane) * 4] = acc[mt][r];
            }
            __syncthreads();
            if (st < 4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < NB; ++r) acc[mt][r] += *(const f32x4 *)&Red[(((st * MT + mt) * NB + r) * 64 + lane) * 4];
            }
            __syncthreads();
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt % NSW == st || st >= NSW) continue;
#pragma unroll
            for (int r = 0; r < NB; ++r) *(f32x4 *)&Red[(((st * MT + mt) * NB + r) * 64 + lane) * 4] = acc[mt][r];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt % NSW != st) continue;
            const int nch = 16 * mt + 4 * ge;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                f32x4 v = st == 0 ? acc[mt][r] : *(const f32x4 *)&Red[(((0 * MT + mt) * NB + r) * 64 + lane) * 4];
#pragma unroll
                for (int so = 1; so < NSW; ++so) v += so == st ? acc[mt][r] : *(const f32x4 *)&Red[(((so * MT + mt) * NB + r) * 64 + lane) * 4];
                v = v * inv_p + psh[mt / NSW];
                if (C::RES) v += rs[mt / NSW][r];
                if (STORE && real) *(f32x4 *)&Y[((size_t)f * C::PIXO + pixe + 8 * r) * COUT + nch] = v;
                if (HANDOFF) vout[mt / NSW][r] = v;
            }
        }
    }
    if constexpr (HANDOFF) {
        // ---- hand the block output to the next stage: x 16, split, into the fragment layout of ITS expand GEMM.  This lane holds
        //      channels 16 mt + 4 ge .. + 3 of pixel (r, ne): k32 step mt >> 1, lane group 2 (mt & 1) + (ge >> 1), dwords 2 (ge & 1), + 1 ----
        static_assert(CN::CIN == COUT && !C::S2, "the next block of the chain takes this block's output (8x8)");
        if (RED_OFF == 0) __syncthreads();               // everybody has read the exchange buffer (it aliases the fragments)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt % NSW != st) continue;
            const int kc = mt >> 1, lg = (2 * (mt & 1) + (ge >> 1)) * 16, dw = 2 * (ge & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 v = real ? vout[mt / NSW][r] * 16.0f : (f32x4){0.f, 0.f, 0.f, 0.f};
                unsigned a0, b0, a1, b1;
                split2h(v[0], v[1], a0, b0);
                split2h(v[2], v[3], a1, b1);
                // this lane's pixel (y = r + 4 (ne >> 3), x = ne & 7) in the next stage's pixel order (stride 2: parity classes)
                const int bn = CN::S2 ? 2 * (r & 1) + (ne & 1) : r;
                const int nn = CN::S2 ? 4 * ((r >> 1) + 2 * (ne >> 3)) + ((ne & 7) >> 1) : ne;
                *(u32x2 *)&Xf[((kc * 4 + bn) * 2 + 0) * 256 + (lg + nn) * 4 + dw] = (u32x2){a0, a1};
                *(u32x2 *)&Xf[((kc * 4 + bn) * 2 + 1) * 256 + (lg + nn) * 4 + dw] = (u32x2){b0, b1};
            }
        }
        // (the barrier after the next stage's prologue publishes the fragments)
    }
    SYNL_LAP(4);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// features.7 (32 -> 192 -> 64, stride 2, 15x15 -> 8x8): the block in front of the chain, on the same machinery.
//
//   * the 15x15 input is padded to 16x16 and cut into its four parity classes c = 2 (y & 1) + (x & 1), each an 8x8 image over
//     (i, j) = (y >> 1, x >> 1) = four 16-column blocks b with lane column n <-> (i = 2 b + (n >> 3), j = n & 7).  Output pixel
//     (oy, ox), block ob = oy >> 1 in the same lane layout, then takes its taps from its OWN lane of the classes (rows 2 oy, 2 oy + 1 /
//     columns 2 ox, 2 ox + 1), from lane n - 1 (column 2 ox - 1: row_shr:1, filter column zeroed at ox = 0) and from the class row above
//     (row 2 oy - 1): lanes 8-15 of a block find it in lanes 0-7 of the same block (row_shr:8), lanes 0-7 in lanes 8-15 of the block
//     before (row_shl:8) -- both zero-filling, so the two shifted values simply add;
//   * the padding row / column 15 must be ZERO after expand + ReLU6: those lanes get a ReLU6 ceiling of 0 (v_med3);
//   * the two waves of a face split the OUTPUT (wave st: output rows 4 st .. 4 st + 3 = blocks 2 st, 2 st + 1) and walk all six hidden
//     groups: nothing to exchange at the end, each wave stores / hands over its half.  Wave 1 also expands block 1 of the two odd-row
//     classes (the row above its first output row): 10 instead of 8 blocks per group.
struct L7 {
    static constexpr int CIN = 32, HID = 192, COUT = 64, NG = 6, MT = 4, H = 15, HO = 8;
    static constexpr int FPW = 2, NW = 4, NT = 256;
    static constexpr int XF_DW = 16 * 2 * 256;           // [class 4][block 4][piece 2][lane 64][4 dwords]
    static constexpr int TB_DW = 12 * 32;
    static constexpr int FACE_DW = XF_DW;
    static constexpr int LDS_DW = FPW * FACE_DW + NW * TB_DW;
};

template <class CN, bool PROF, int FACE_DW>
__device__ __forceinline__ void lb7_stage(unsigned *smem, const LbStageArgs &sa, int B, unsigned long long (&pt_)[5], unsigned long long &tk) {
    unsigned long long tn = 0;
    const float *__restrict__ X = sa.X;
    const float *__restrict__ Tlb = sa.Tlb, *__restrict__ p_shift = sa.p_shift;
    float *__restrict__ Y = sa.Y;
    constexpr int MT = L7::MT, CIN = L7::CIN, COUT = L7::COUT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = wave >> 1, st = wave & 1;
    const int f = blockIdx.x * L7::FPW + fl;
    const bool real = f < B;
    const int fc = real ? f : B - 1;
    const int n = lane & 15, g = lane >> 4;
    const unsigned l4 = lane * 4, g4 = g * 4;
    unsigned *Xf = smem + fl * FACE_DW;

    // ---- stage: the 16 class blocks of this face as pre-split B fragments (x 16); wave st converts the classes of row parity st ----
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {                     // column parity px = hb
        f32x4 xv[4][2];
        bool ok[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int y = 2 * (2 * b + (n >> 3)) + st, x = 2 * (n & 7) + hb;
            ok[b] = real && y < L7::H && x < L7::H;
            const float *src = X + ((size_t)fc * (L7::H * L7::H) + (ok[b] ? y * L7::H + x : 0)) * CIN + 8 * g;
            xv[b][0] = *(const f32x4 *)src;
            xv[b][1] = *(const f32x4 *)(src + 4);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 a = xv[b][0], c = xv[b][1];
            if (!ok[b]) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; c = a; }
            a *= 16.0f; c *= 16.0f;
            u32x4 pc[2];
            split2v(a[0], a[1], pc, 0);
            split2v(a[2], a[3], pc, 1);
            split2v(c[0], c[1], pc, 2);
            split2v(c[2], c[3], pc, 3);
#pragma unroll
            for (int p = 0; p < 2; ++p) *(u32x4 *)&Xf[((((2 * st + hb) * 4 + b) * 2) + p) * 256 + lane * 4] = pc[p];
        }
    }
    const float mL = (n & 7) != 0 ? 1.f : 0.f;
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[mt][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float *Tb = reinterpret_cast<float *>(smem + L7::FPW * FACE_DW + wave * L7::TB_DW);
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(sa.Weh), 0, 0x7fffffff, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(sa.Wlb), 0, 0x7fffffff, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Tlb), 0, 0x7fffffff, 0x00027000);
    const unsigned l16 = lane * 16;
    f32x4 tv[2];
    auto fetch_t = [&](int G) __attribute__((always_inline)) {
        tv[0] = bload4f(rs_t, l16, G * (L7::TB_DW * 4));
        tv[1] = bload4f(rs_t, l16 & 511, G * (L7::TB_DW * 4) + 1024);
    };
    auto park_t = [&]() __attribute__((always_inline)) {
        *(f32x4 *)&Tb[l4] = tv[0];
        *(f32x4 *)&Tb[256 + (l4 & 127)] = tv[1];
    };
    const float c6e = Tlb[11 * 32], inv_p = Tlb[11 * 32 + 1];
    // ReLU6 ceilings: 0 on the lanes that are padding (column 15 = odd-column classes, j = 7; row 15 = odd-row classes, block 3, i = 7)
    const float cJ = (n & 7) != 7 ? c6e : 0.f;
    const float cR0 = (st == 1 && n >= 8) ? 0.f : c6e, cR1 = (st == 1 && n >= 8) ? 0.f : cJ;    // second own block of the odd-row classes
    u32x4 Ae[2][2];
    auto fetch_e = [&](int G) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) Ae[t][p] = bload4(rs_e, l16, G * 4096 + (t * 2 + p) * 1024);
    };
    fetch_t(0);
    fetch_e(0);
    park_t();
    __syncthreads();
    SYNL_LAP(0);

    for (int G = 0; G < L7::NG; ++G) {
        // ---- expand: D[t][class][k]: own blocks b = 2 st + k (k = 0, 1); k = 2 (wave 1, odd-row classes): block 1 ----
        f32x4 D[2][4][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 es = *(const f32x4 *)&Tb[10 * 32 + 16 * t + g4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) D[t][c][k] = es;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 Bx[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int p = 0; p < 2; ++p) Bx[k][p] = *(const u32x4 *)&Xf[(((c * 4 + 2 * st + k) * 2) + p) * 256 + lane * 4];
            mac3x4(Ae[0], Bx[0], D[0][c][0], Ae[1], Bx[0], D[1][c][0], Ae[0], Bx[1], D[0][c][1], Ae[1], Bx[1], D[1][c][1]);
            SYNL_FENCE();
        }
        if (st == 1) {
            u32x4 Bx[2][2];
#pragma unroll
            for (int c = 2; c < 4; ++c)
#pragma unroll
                for (int p = 0; p < 2; ++p) Bx[c - 2][p] = *(const u32x4 *)&Xf[(((c * 4 + 1) * 2) + p) * 256 + lane * 4];
            mac3x4(Ae[0], Bx[0], D[0][2][2], Ae[1], Bx[0], D[1][2][2], Ae[0], Bx[1], D[0][3][2], Ae[1], Bx[1], D[1][3][2]);
            SYNL_FENCE();
        }
        SYNL_LAP(1);
        // ---- depthwise 3x3 stride 2 + BN shift + ReLU6, split in place into the B operand of the project step ----
        u32x4 Ap[3][2];
        auto fetch_p = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 2; ++p) Ap[mt % 3][p] = bload4(rs_p, l16, G * (MT * 2048) + (mt * 2 + p) * 1024);
        };
        u32x4 Bd[2][2];
#pragma unroll
        for (int th = 0; th < 4; ++th) {
            const int t = th >> 1, hf = th & 1;
            if (th == 3) fetch_p(0);
            const int c0 = 16 * t + 2 * hf;
            f32x2 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x2 *)&Tb[k * 32 + c0 + g4];
            const f32x2 dsh = *(const f32x2 *)&Tb[9 * 32 + c0 + g4];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) w[3 * dy] *= mL;
            auto relu = [&](const f32x4 &d, float ceil) __attribute__((always_inline)) {
                f32x2 e;
                e[0] = relu01(d[2 * hf], ceil);
                e[1] = relu01(d[2 * hf + 1], ceil);
                return e;
            };
            f32x2 up2, up3;                              // the odd-row classes' block before this wave's first block (wave 1), lanes 8-15 -> 0-7
            if (st == 1) {
                up2 = dpp2<kRowShl8>(relu(D[t][2][2], c6e));
                up3 = dpp2<kRowShl8>(relu(D[t][3][2], cJ));
            } else {
                up2 = (f32x2){0.f, 0.f}; up3 = up2;      // (row -1: the image border)
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f32x2 E0 = relu(D[t][0][k], c6e), E1 = relu(D[t][1][k], cJ);
                const f32x2 E2 = relu(D[t][2][k], k == 1 ? cR0 : c6e), E3 = relu(D[t][3][k], k == 1 ? cR1 : cJ);
                // the class row above: lanes 8-15 from lanes 0-7 of this block, lanes 0-7 from lanes 8-15 of the block before
                const f32x2 U2 = dpp2<kRowShr8>(E2) + up2, U3 = dpp2<kRowShr8>(E3) + up3;
                f32x2 O = dsh;
                O += dpp2<kRowShr1>(U3) * w[0];
                O += U2 * w[1];
                O += U3 * w[2];
                asm volatile("" : "+v"(O));
                O += dpp2<kRowShr1>(E1) * w[3];
                O += E0 * w[4];
                O += E1 * w[5];
                asm volatile("" : "+v"(O));
                O += dpp2<kRowShr1>(E3) * w[6];
                O += E2 * w[7];
                O = pk_fma_clamp01(E3, w[8], O);
                split2v(O[0], O[1], Bd[k], th);
                if (k == 0) { up2 = dpp2<kRowShl8>(E2); up3 = dpp2<kRowShl8>(E3); }
            }
            SYNL_FENCE();
        }
        SYNL_LAP(2);
        // ---- project 1x1, K = this group ----
        const bool more = G + 1 < L7::NG;
        fetch_p(1);
        if (more) { fetch_t(G + 1); fetch_e(G + 1); }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt + 2 < MT) fetch_p(mt + 2);
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                acc[mt][0] = mfmal(Ap[mt % 3][PA[j]], Bd[0][PB[j]], acc[mt][0]);
                acc[mt][1] = mfmal(Ap[mt % 3][PA[j]], Bd[1][PB[j]], acc[mt][1]);
            }
            SYNL_FENCE();
        }
        if (more) park_t();
        SYNL_LAP(3);
    }

    // ---- rescale, BN shift (no residual: the block changes width and stride), NHWC store and / or hand-over ----
    int le = lane;
    asm volatile("" : "+v"(le));
    const int ne = le & 15, ge = le >> 4, ox = ne & 7;
    constexpr bool HANDOFF = !__is_same(CN, void);
    if constexpr (HANDOFF) __syncthreads();              // every wave is done reading the class fragments (the hand-over overwrites them)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int nch = 16 * mt + 4 * ge;
        const f32x4 psh = *(const f32x4 *)&p_shift[nch];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int oy = 4 * st + 2 * k + (ne >> 3);
            const f32x4 v = acc[mt][k] * inv_p + psh;
            if (real) *(f32x4 *)&Y[((size_t)f * 64 + oy * 8 + ox) * COUT + nch] = v;     // (the next block's residual reads it)
            if constexpr (HANDOFF) {
                static_assert(CN::CIN == COUT && !CN::S2, "features.8 takes this block's output");
                // pixel (oy, ox) in the next stage's layout: block oy & 3, lane column 8 (oy >> 2) + ox
                const int kc = mt >> 1, lg = (2 * (mt & 1) + (ge >> 1)) * 16, dw = 2 * (ge & 1);
                const int r = 2 * k + (ne >> 3), nn = 8 * st + ox;
                const f32x4 vs = real ? v * 16.0f : (f32x4){0.f, 0.f, 0.f, 0.f};
                unsigned a0, b0, a1, b1;
                split2h(vs[0], vs[1], a0, b0);
                split2h(vs[2], vs[3], a1, b1);
                *(u32x2 *)&Xf[((kc * 4 + r) * 2 + 0) * 256 + (lg + nn) * 4 + dw] = (u32x2){a0, a1};
                *(u32x2 *)&Xf[((kc * 4 + r) * 2 + 1) * 256 + (lg + nn) * 4 + dw] = (u32x2){b0, b1};
            }
        }
    }
    SYNL_LAP(4);
}

// features.7 as the FIRST stage of the eight-wave small-batch chain (one face per workgroup; round 5: B = 128 0.308 -> 0.300 ms, B = 1 0.226 -> 0.220,
// same tests as the chain without it: test_small_batch_chain_equals_the_block_by_block_schedule, the ragged-batch and threshold tests).
// lb7_stage gives a face two waves that split the OUTPUT rows and walk all six hidden groups -- one wave per SIMD and a critical path of
// six groups of 8-10 block expansions.  Here wave w = (hidden stream hs = w >> 2, output block q = w & 3): a wave owns ONE 16-pixel output
// block (output rows 2 q, 2 q + 1) and every second hidden group (hs, hs + 2, hs + 4).  Per group it expands its own block of the four parity
// classes and, for q >= 1, block q - 1 of the two odd-row classes (the class row above its first output row, lanes 8-15 of that block):
// 4 + 2 block expansions instead of 8-10, three groups instead of six.  The two hidden streams of an output block meet through LDS
// (stream 1 publishes, stream 0 adds -- a fixed order), stream 0 stores and hands the block to features.8's fragment layout.
template <class CN, int FACE_DW>
__device__ __forceinline__ void lb7_stage8(unsigned *smem, const LbStageArgs &sa, int B) {
    const float *__restrict__ X = sa.X;
    const float *__restrict__ Tlb = sa.Tlb, *__restrict__ p_shift = sa.p_shift;
    float *__restrict__ Y = sa.Y;
    constexpr int MT = L7::MT, CIN = L7::CIN, COUT = L7::COUT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave & 3, hs = wave >> 2;
    const int f = blockIdx.x;
    const bool real = f < B;
    const int fc = real ? f : B - 1;
    const int n = lane & 15, g = lane >> 4;
    const unsigned l4 = lane * 4, g4 = g * 4;
    unsigned *Xf = smem;

    // ---- stage: the 16 class blocks of the face as pre-split B fragments (x 16); wave w converts class w & 3, blocks 2 (w >> 2), + 1 ----
    {
        const int c = wave & 3, py = c >> 1, px = c & 1, b0 = 2 * (wave >> 2);
        f32x4 xv[2][2];
        bool ok[2];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int b = b0 + bb;
            const int y = 2 * (2 * b + (n >> 3)) + py, x = 2 * (n & 7) + px;
            ok[bb] = real && y < L7::H && x < L7::H;
            const float *src = X + ((size_t)fc * (L7::H * L7::H) + (ok[bb] ? y * L7::H + x : 0)) * CIN + 8 * g;
            xv[bb][0] = *(const f32x4 *)src;
            xv[bb][1] = *(const f32x4 *)(src + 4);
        }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            f32x4 a = xv[bb][0], d = xv[bb][1];
            if (!ok[bb]) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; d = a; }
            a *= 16.0f; d *= 16.0f;
            u32x4 pc[2];
            split2v(a[0], a[1], pc, 0);
            split2v(a[2], a[3], pc, 1);
            split2v(d[0], d[1], pc, 2);
            split2v(d[2], d[3], pc, 3);
#pragma unroll
            for (int p = 0; p < 2; ++p) *(u32x4 *)&Xf[(((c * 4 + b0 + bb) * 2) + p) * 256 + lane * 4] = pc[p];
        }
    }
    const float mL = (n & 7) != 0 ? 1.f : 0.f;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float *Tb = reinterpret_cast<float *>(smem + FACE_DW + wave * L7::TB_DW);
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(sa.Weh), 0, 0x7fffffff, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(sa.Wlb), 0, 0x7fffffff, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Tlb), 0, 0x7fffffff, 0x00027000);
    const unsigned l16 = lane * 16;
    f32x4 tv[2];
    auto fetch_t = [&](int G) __attribute__((always_inline)) {
        tv[0] = bload4f(rs_t, l16, G * (L7::TB_DW * 4));
        tv[1] = bload4f(rs_t, l16 & 511, G * (L7::TB_DW * 4) + 1024);
    };
    auto park_t = [&]() __attribute__((always_inline)) {
        *(f32x4 *)&Tb[l4] = tv[0];
        *(f32x4 *)&Tb[256 + (l4 & 127)] = tv[1];
    };
    const float c6e = Tlb[11 * 32], inv_p = Tlb[11 * 32 + 1];
    // ReLU6 ceilings: 0 on the padding lanes (column 15 = odd-column classes, j = 7; row 15 = odd-row classes, block 3, i = 7)
    const float cJ = (n & 7) != 7 ? c6e : 0.f;
    const float cR0 = (q == 3 && n >= 8) ? 0.f : c6e, cR1 = (q == 3 && n >= 8) ? 0.f : cJ;
    u32x4 Ae[2][2];
    auto fetch_e = [&](int G) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) Ae[t][p] = bload4(rs_e, l16, G * 4096 + (t * 2 + p) * 1024);
    };
    fetch_t(hs);
    fetch_e(hs);
    park_t();
    __syncthreads();

    for (int G = hs; G < L7::NG; G += 2) {
        // ---- expand: D[t][class] = own block q; Dh[t][class - 2] = block q - 1 of the odd-row classes (q >= 1) ----
        f32x4 D[2][4], Dh[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 es = *(const f32x4 *)&Tb[10 * 32 + 16 * t + g4];
#pragma unroll
            for (int c = 0; c < 4; ++c) D[t][c] = es;
            Dh[t][0] = es; Dh[t][1] = es;
        }
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            u32x4 Bx[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int p = 0; p < 2; ++p) Bx[k][p] = *(const u32x4 *)&Xf[((((c + k) * 4 + q) * 2) + p) * 256 + lane * 4];
            mac3x4(Ae[0], Bx[0], D[0][c], Ae[1], Bx[0], D[1][c], Ae[0], Bx[1], D[0][c + 1], Ae[1], Bx[1], D[1][c + 1]);
            SYNL_FENCE();
        }
        if (q >= 1) {
            u32x4 Bx[2][2];
#pragma unroll
            for (int c = 2; c < 4; ++c)
#pragma unroll
                for (int p = 0; p < 2; ++p) Bx[c - 2][p] = *(const u32x4 *)&Xf[(((c * 4 + q - 1) * 2) + p) * 256 + lane * 4];
            mac3x4(Ae[0], Bx[0], Dh[0][0], Ae[1], Bx[0], Dh[1][0], Ae[0], Bx[1], Dh[0][1], Ae[1], Bx[1], Dh[1][1]);
            SYNL_FENCE();
        }
        // ---- depthwise 3x3 stride 2 + BN shift + ReLU6, split in place into the B operand of the project step ----
        u32x4 Ap[3][2];
        auto fetch_p = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 2; ++p) Ap[mt % 3][p] = bload4(rs_p, l16, G * (MT * 2048) + (mt * 2 + p) * 1024);
        };
        u32x4 Bd[2];
#pragma unroll
        for (int th = 0; th < 4; ++th) {
            const int t = th >> 1, hf = th & 1;
            if (th == 3) fetch_p(0);
            const int c0 = 16 * t + 2 * hf;
            f32x2 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x2 *)&Tb[k * 32 + c0 + g4];
            const f32x2 dsh = *(const f32x2 *)&Tb[9 * 32 + c0 + g4];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) w[3 * dy] *= mL;
            auto relu = [&](const f32x4 &d, float ceil) __attribute__((always_inline)) {
                f32x2 e;
                e[0] = relu01(d[2 * hf], ceil);
                e[1] = relu01(d[2 * hf + 1], ceil);
                return e;
            };
            f32x2 up2, up3;                              // the odd-row classes' row above this block's first row: lanes 8-15 of block q - 1 -> lanes 0-7
            if (q >= 1) {
                up2 = dpp2<kRowShl8>(relu(Dh[t][0], c6e));
                up3 = dpp2<kRowShl8>(relu(Dh[t][1], cJ));
            } else {
                up2 = (f32x2){0.f, 0.f}; up3 = up2;      // (row -1: the image border)
            }
            const f32x2 E0 = relu(D[t][0], c6e), E1 = relu(D[t][1], cJ);
            const f32x2 E2 = relu(D[t][2], cR0), E3 = relu(D[t][3], cR1);
            const f32x2 U2 = dpp2<kRowShr8>(E2) + up2, U3 = dpp2<kRowShr8>(E3) + up3;
            f32x2 O = dsh;
            O += dpp2<kRowShr1>(U3) * w[0];
            O += U2 * w[1];
            O += U3 * w[2];
            asm volatile("" : "+v"(O));
            O += dpp2<kRowShr1>(E1) * w[3];
            O += E0 * w[4];
            O += E1 * w[5];
            asm volatile("" : "+v"(O));
            O += dpp2<kRowShr1>(E3) * w[6];
            O += E2 * w[7];
            O = pk_fma_clamp01(E3, w[8], O);
            split2v(O[0], O[1], Bd, th);
            SYNL_FENCE();
        }
        // ---- project 1x1, K = this group ----
        const bool more = G + 2 < L7::NG;
        fetch_p(1);
        if (more) { fetch_t(G + 2); fetch_e(G + 2); }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt + 2 < MT) fetch_p(mt + 2);
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[mt] = mfmal(Ap[mt % 3][PA[j]], Bd[PB[j]], acc[mt]);
            SYNL_FENCE();
        }
        if (more) park_t();
    }

    // ---- the two hidden streams of an output block meet (stream 0 + stream 1), then rescale, BN shift, NHWC store and hand-over ----
    float *Ex = reinterpret_cast<float *>(Xf);
    int le = lane;
    asm volatile("" : "+v"(le));
    const int ne = le & 15, ge = le >> 4, ox = ne & 7, oy = 2 * q + (ne >> 3);
    __syncthreads();                                     // every wave is done reading the class fragments
    if (hs == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *(f32x4 *)&Ex[((q * MT + mt) * 64 + lane) * 4] = acc[mt];
    }
    __syncthreads();
    f32x4 v[MT];
    if (hs == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int nch = 16 * mt + 4 * ge;
            v[mt] = (acc[mt] + *(const f32x4 *)&Ex[((q * MT + mt) * 64 + lane) * 4]) * inv_p + *(const f32x4 *)&p_shift[nch];
            if (real) *(f32x4 *)&Y[((size_t)f * 64 + oy * 8 + ox) * COUT + nch] = v[mt];     // (the next block's residual reads it)
        }
    }
    static_assert(CN::CIN == COUT && !CN::S2, "features.8 takes this block's output");
    __syncthreads();                                     // the exchange buffer has been read (the hand-over overwrites it)
    if (hs == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // pixel (oy, ox) in the next stage's layout: block oy & 3, lane column 8 (oy >> 2) + ox
            const int kc = mt >> 1, lg = (2 * (mt & 1) + (ge >> 1)) * 16, dw = 2 * (ge & 1);
            const int r = oy & 3, nn = 8 * (oy >> 2) + ox;
            const f32x4 vs = real ? v[mt] * 16.0f : (f32x4){0.f, 0.f, 0.f, 0.f};
            unsigned a0, b0, a1, b1;
            split2h(vs[0], vs[1], a0, b0);
            split2h(vs[2], vs[3], a1, b1);
            *(u32x2 *)&Xf[((kc * 4 + r) * 2 + 0) * 256 + (lg + nn) * 4 + dw] = (u32x2){a0, a1};
            *(u32x2 *)&Xf[((kc * 4 + r) * 2 + 1) * 256 + (lg + nn) * 4 + dw] = (u32x2){b0, b1};
        }
    }
    // (the barrier after the next stage's prologue publishes the fragments)
}

template <bool PROF = false>
__global__ __launch_bounds__(L7::NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_block_lb7_kernel(LbStageArgs sa, int B, unsigned long long *prof = nullptr) {
    unsigned long long pt_[5] = {0, 0, 0, 0, 0}, tk = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
    __shared__ __attribute__((aligned(16))) unsigned smem[L7::LDS_DW];
    lb7_stage<void, PROF, L7::FACE_DW>(smem, sa, B, pt_, tk);
    if (PROF && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&prof[i], pt_[i]);
        atomicAdd(&prof[7], 1ull);
    }
}

template <class C, bool PROF = false, bool PARTIAL = false>
__global__ __launch_bounds__(C::NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_block_lb_kernel(LbStageArgs sa, int B, unsigned long long *prof = nullptr) {
    unsigned long long pt_[5] = {0, 0, 0, 0, 0}, tk = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DW];
    lb_stage<C, void, true, PROF, C::FACE_DW, true, PARTIAL>(smem, sa, B, pt_, tk);
    if (PROF && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&prof[i], pt_[i]);
        atomicAdd(&prof[7], 1ull);
    }
}

//                    CIN  HID COUT  RES  EPF PPF
#ifndef SYN_L8_PPF
#define SYN_L8_PPF 3
#endif
#ifndef SYN_L11_PPF
#define SYN_L11_PPF 3
#endif
using L8 = LbCfg<      64, 384,  64, true,  2, SYN_L8_PPF>;     // features.8-10
using L11 = LbCfg<     64, 384,  96, false, 2, SYN_L11_PPF>;     // features.11
using L12 = LbCfg<     96, 576,  96, true,  1, 3>;     // features.12, 13
using L14 = LbCfg<     96, 576, 160, false, 1, 2, 2, true>;   // features.14 (stride 2: 8x8 -> 4x4)

// features.7 .. 14 of a face in ONE launch: 7 (32 -> 192 -> 64, stride 2, 15x15 -> 8x8), 8, 9, 10 (64 -> 384 -> 64, residual),
// 11 (64 -> 384 -> 96), 12, 13 (96 -> 576 -> 96, residual), 14 (96 -> 576 -> 160, stride 2; features.13's output then never goes to
// global memory).  s[i] = features.(7 + i); WITH7 = false starts at features.8, WITH14 = false stops after features.13.
struct LbChainArgs { LbStageArgs s[8]; };
constexpr int kChainFaceDw = L7::FACE_DW;
static_assert(L8::FACE_DW <= kChainFaceDw && L11::FACE_DW <= kChainFaceDw && L12::FACE_DW <= kChainFaceDw && L14::FACE_DW <= kChainFaceDw,
              "every stage reuses the fragment buffers of the first");
constexpr int kChainLdsDw = L8::FPW * kChainFaceDw + L8::NW * L8::TB_DW;
static_assert(2 * kChainLdsDw * 4 <= 160 * 1024, "two workgroups per CU");

#ifndef SYN_L2_TOUCH
#define SYN_L2_TOUCH 1              // 0: no L2 warm-up of the weight fragments (A/B)
#endif
// one stage's expand fragments, project fragments and per-group tables (the layouts fetch_e / fetch_p / fetch_t walk)
template <int NG, int KE, int MT>
__device__ __forceinline__ void lb_touch_stage(const LbStageArgs &sa, unsigned gi, unsigned nth, unsigned &sink) {
    l2_touch(sa.Weh, NG * KE * 4096u, gi, nth, sink);
    l2_touch(sa.Wlb, NG * MT * 2048u, gi, nth, sink);
    l2_touch(sa.Tlb, NG * 12u * 32u * 4u, gi, nth, sink);
}
template <bool WITH7, bool WITH14>
__global__ __launch_bounds__(L8::NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_chain_lb_kernel(LbChainArgs ca, int B) {
    unsigned long long pt_[5] = {0, 0, 0, 0, 0}, tk = 0;
#ifndef SYN_LB_LDS_PAD
#define SYN_LB_LDS_PAD 0              // experiment knob: extra LDS dwords (22000: one workgroup per CU = ONE wave per SIMD)
#endif
    __shared__ __attribute__((aligned(16))) unsigned smem[kChainLdsDw + SYN_LB_LDS_PAD];
    // every stage's weights (2.6 MB for features.7-14) into this XCD's L2 before the walk starts (syn_internal.h l2_touch): the waves fetch their
    // fragments one group ahead, which covers an L2 hit but not the miss the first CU of an XCD takes on every group of a cold run
    // (B = 1024, interleaved on one box: 267.0 against 271.6 us without, gpurun_out/r5c3)
    unsigned sink = 0;
    if (SYN_L2_TOUCH) {
        const unsigned gi = (blockIdx.x >> 3) * (unsigned)L8::NT + threadIdx.x, nth = ((gridDim.x + 7) >> 3) * (unsigned)L8::NT;
        if (WITH7) lb_touch_stage<L7::NG, 1, L7::MT>(ca.s[0], gi, nth, sink);
        lb_touch_stage<L8::NG, L8::KE, L8::MT>(ca.s[1], gi, nth, sink);
        lb_touch_stage<L8::NG, L8::KE, L8::MT>(ca.s[2], gi, nth, sink);
        lb_touch_stage<L8::NG, L8::KE, L8::MT>(ca.s[3], gi, nth, sink);
        lb_touch_stage<L11::NG, L11::KE, L11::MT>(ca.s[4], gi, nth, sink);
        lb_touch_stage<L12::NG, L12::KE, L12::MT>(ca.s[5], gi, nth, sink);
        lb_touch_stage<L12::NG, L12::KE, L12::MT>(ca.s[6], gi, nth, sink);
        if (WITH14) lb_touch_stage<L14::NG, L14::KE, L14::MT>(ca.s[7], gi, nth, sink);
    }
    if constexpr (WITH7) {
        lb7_stage<L8, false, kChainFaceDw>(smem, ca.s[0], B, pt_, tk);
        lb_stage<L8, L8, false, false, kChainFaceDw>(smem, ca.s[1], B, pt_, tk);
    } else
        lb_stage<L8, L8, true, false, kChainFaceDw>(smem, ca.s[1], B, pt_, tk);
    lb_stage<L8, L8, false, false, kChainFaceDw>(smem, ca.s[2], B, pt_, tk);
    lb_stage<L8, L11, false, false, kChainFaceDw>(smem, ca.s[3], B, pt_, tk);
    lb_stage<L11, L12, false, false, kChainFaceDw>(smem, ca.s[4], B, pt_, tk);
    lb_stage<L12, L12, false, false, kChainFaceDw>(smem, ca.s[5], B, pt_, tk);
    if constexpr (WITH14) {
        lb_stage<L12, L14, false, false, kChainFaceDw, false>(smem, ca.s[6], B, pt_, tk);
        lb_stage<L14, void, false, false, kChainFaceDw>(smem, ca.s[7], B, pt_, tk);
    } else
        lb_stage<L12, void, false, false, kChainFaceDw>(smem, ca.s[6], B, pt_, tk);
    if (SYN_L2_TOUCH) l2_touch_done(sink);
}

// Small batches (round 4; BASELINE configs[1] is 128 faces): features.8 .. 14 (round 5: 7 .. 14) as ONE launch with ONE face per workgroup.  Below
// ~400 faces the chain above leaves most CUs empty and its workgroups walk 12-18 hidden groups per block with two waves, so those batches ran
// the blocks one launch each, hidden-sliced over workgroups (PARTIAL) with a reduce launch behind every block: 14 launches, 135 us at B = 128.
// Several streams per face shorten a face's critical path, their partial sums meet in LDS (no global round trip, no reduce kernel), and the
// block output stays on chip as the next stage's fragments.  (The first form had FOUR waves per face, one per SIMD; retired in round 5.)
constexpr int cmax4(int a, int b, int c, int d) { return (a > b ? a : b) > (c > d ? c : d) ? (a > b ? a : b) : (c > d ? c : d); }

// EIGHT waves per face (two per SIMD, 256 registers): the four-stream kernel has one wave per SIMD, and a lone wave stalls on every dependent
// step of a hidden group (~3.4 us per group against ~2.4 per SIMD in the two-waves-per-SIMD chain of large batches); with eight streams a
// SIMD's two waves cover each other's round trips, a wave walks 1-3 groups per block instead of 3-5, and the partial sums meet in two
// levels (streams 4 .. 7 into 0 .. 3, then as above).  Landmarks-only step, four -> eight streams (ms, interleaved on one box): B = 1 0.244 -> 0.225,
// 8 0.265 -> 0.245, 64 0.300 -> 0.281, 128 0.330 -> 0.310, 256 0.429 -> 0.409.
using L8e = LbCfg<     64, 384,  64, true,  2, SYN_L8_PPF, 1, false, 8>;
using L11e = LbCfg<    64, 384,  96, false, 2, SYN_L11_PPF, 1, false, 8>;
using L12e = LbCfg<    96, 576,  96, true,  1, 3, 1, false, 8>;
using L14e = LbCfg<    96, 576, 160, false, 1, 2, 1, true, 8>;
constexpr int kChainFaceDwE = cmax4(L8e::FACE_DW, L11e::FACE_DW, L12e::FACE_DW, L14e::FACE_DW);
constexpr int kChainLdsDwE = L8e::FPW * kChainFaceDwE + L8e::NW * L8e::TB_DW;
static_assert(kChainLdsDwE * 4 <= 160 * 1024, "one workgroup per CU");

static_assert(L7::XF_DW <= kChainFaceDwE, "features.7's class fragments fit the chain's face buffer");
#ifndef SYN_SMALL8_SPLIT
#define SYN_SMALL8_SPLIT 1             // fragments | exchange buffer side by side, four barriers per stage instead of six (0: the aliased buffer of round 4, for A/B runs)
#endif
constexpr int kChainXfDwE = cmax4(L7::XF_DW, L8e::XF_DW, L12e::XF_DW, L14e::XF_DW);
constexpr int kChainRedDwE = cmax4(L8e::RED_DW, L11e::RED_DW, L12e::RED_DW, L14e::RED_DW);
constexpr int kRedOffE = SYN_SMALL8_SPLIT ? kChainXfDwE : 0;
constexpr int kFaceDwE = SYN_SMALL8_SPLIT ? kChainXfDwE + kChainRedDwE : kChainFaceDwE;
constexpr int kLdsDwE = kFaceDwE + L8e::NW * L8e::TB_DW;
static_assert(kLdsDwE * 4 <= 160 * 1024, "one workgroup per CU");
template <bool WITH7 = false>
__global__ __launch_bounds__(L8e::NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_chain_lb_small8_kernel(LbChainArgs ca, int B) {
    unsigned long long pt_[5] = {0, 0, 0, 0, 0}, tk = 0;
    __shared__ __attribute__((aligned(16))) unsigned smem[kLdsDwE];
    if constexpr (WITH7) {
        lb7_stage8<L8e, kFaceDwE>(smem, ca.s[0], B);
        lb_stage<L8e, L8e, false, false, kFaceDwE, true, false, kRedOffE>(smem, ca.s[1], B, pt_, tk);
    } else
    lb_stage<L8e, L8e, true, false, kFaceDwE, true, false, kRedOffE>(smem, ca.s[1], B, pt_, tk);
    lb_stage<L8e, L8e, false, false, kFaceDwE, true, false, kRedOffE>(smem, ca.s[2], B, pt_, tk);
    lb_stage<L8e, L11e, false, false, kFaceDwE, true, false, kRedOffE>(smem, ca.s[3], B, pt_, tk);
    lb_stage<L11e, L12e, false, false, kFaceDwE, true, false, kRedOffE>(smem, ca.s[4], B, pt_, tk);
    lb_stage<L12e, L12e, false, false, kFaceDwE, true, false, kRedOffE>(smem, ca.s[5], B, pt_, tk);
    lb_stage<L12e, L14e, false, false, kFaceDwE, false, false, kRedOffE>(smem, ca.s[6], B, pt_, tk);
    lb_stage<L14e, void, false, false, kFaceDwE, true, false, kRedOffE>(smem, ca.s[7], B, pt_, tk);
}

// y = (slice 0 + slice 1 + ... in this order) / (16 Sp) + BN shift (+ x): one thread per four channels of a pixel
template <class C>
__global__ __launch_bounds__(256) void lb_reduce_kernel(const float *__restrict__ part, int S, const float *__restrict__ Tlb,
                                                        const float *__restrict__ p_shift, const float *__restrict__ X,
                                                        float *__restrict__ Y, int B) {
    constexpr int C4 = C::COUT / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)B * C::PIXO * C4;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const float inv_p = Tlb[11 * 32 + 1];
    f32x4 a = *(const f32x4 *)&part[(size_t)idx * 4];
    for (int sl = 1; sl < S; ++sl) a += *(const f32x4 *)&part[((size_t)sl * total + idx) * 4];
    f32x4 v = a * inv_p + *(const f32x4 *)&p_shift[4 * c4];
    if (C::RES) v += *(const f32x4 *)&X[(size_t)idx * 4];
    *(f32x4 *)&Y[(size_t)idx * 4] = v;
}

// small batches: S slices of the hidden groups per two faces (>= ~256 workgroups, at least two groups per slice), partial sums in `scratch`
template <class C>
static bool launch_lb_sliced(const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.scratch || a.prof) return false;
    const int wg = (B + C::FPW - 1) / C::FPW;
    int S = 0;
    for (int d = 2; d <= C::NG / 2; ++d)
        if (C::NG % d == 0) { S = d; if (wg * d >= 256) break; }
    if (S < 2 || (size_t)S * B * C::PIXO * C::COUT > a.scratch_floats) return false;
    LbStageArgs sa{a.X, a.Alb_e, a.Tlb, a.Alb_p, a.p_shift, a.Y};
    sa.part = a.scratch; sa.gsl = C::NG / S;
    fused_block_lb_kernel<C, false, true><<<dim3(wg, S), C::NT, 0, s>>>(sa, B);
    const long total = (long)B * C::PIXO * (C::COUT / 4);
    lb_reduce_kernel<C><<<(int)((total + 255) / 256), 256, 0, s>>>(a.scratch, S, a.Tlb, a.p_shift, a.X, a.Y, B);
    return true;
}

template <class C>
static void launch_lb(const FusedBlockArgs &a, int B, hipStream_t s) {
    const int grid = (B + C::FPW - 1) / C::FPW;
    const LbStageArgs sa{a.X, a.Alb_e, a.Tlb, a.Alb_p, a.p_shift, a.Y};
    if (a.prof) fused_block_lb_kernel<C, true><<<grid, C::NT, 0, s>>>(sa, B, a.prof);
    else fused_block_lb_kernel<C><<<grid, C::NT, 0, s>>>(sa, B);
}

// The chain launch: mode 0 = off (one launch per block), 1 = features.8-13, 2 = features.8-14, 3 = features.7-14 (SYN_LB_CHAIN; default 3)
constexpr int kChainMin = 384;          // (us, blocks one by one -> chain: B = 384 239 -> 195, 512 253 -> 201, 640 367 -> 259; B = 256 198 -> 184 but a slower step)
constexpr int kSmallChainMax = 256;     // the one-face-per-workgroup chain: one round of workgroups
static bool small_f7() {      // features.7 as the first stage of the eight-wave small-batch chain (SYN_SMALL_F7=0: its own launch, as in round 4)
    static const bool on = test_knob("small_f7", 1) != 0;
    return on;
}
int lb_chain_mode(int B, bool small) {
    static const int chain = (int)test_knob("lb_chain", 3);
    if (chain <= 0) return 0;
    if (B < kChainMin) return (small && B <= kSmallChainMax) ? (small_f7() ? 3 : 2) : 0;      // (2 = features.8 .. 14: launch_fused_chain_lb picks the small-batch kernel)
    return chain > 3 ? 3 : chain;
}
// a[i] = the arguments of features.(first + i), first = 7 | 8, first + n_blocks - 1 = 13 | 14; false: not applicable
bool launch_fused_chain_lb(const FusedBlockArgs *a, int first, int n_blocks, int B, hipStream_t s) {
    const int last = first + n_blocks - 1;
    if ((first != 7 && first != 8) || (last != 13 && last != 14) || (first == 7 && last != 14)) return false;
    LbChainArgs ca;
    for (int i = 0; i < n_blocks; ++i) {
        if (!a[i].Alb_e || !a[i].Alb_p || !a[i].Tlb || a[i].prof) return false;
        ca.s[first - 7 + i] = LbStageArgs{a[i].X, a[i].Alb_e, a[i].Tlb, a[i].Alb_p, a[i].p_shift, a[i].Y};
    }
    if (first == 8) ca.s[0] = ca.s[1];
    if (last == 13) ca.s[7] = ca.s[6];
    if (B < kChainMin) {
        if (last != 14 || (first == 7 && !small_f7())) return false;
        if (first == 7) { fused_chain_lb_small8_kernel<true><<<B, L8e::NT, 0, s>>>(ca, B); return true; }
        fused_chain_lb_small8_kernel<false><<<B, L8e::NT, 0, s>>>(ca, B);
        return true;
    }
    const int grid = (B + L8::FPW - 1) / L8::FPW;
    if (first == 7) fused_chain_lb_kernel<true, true><<<grid, L8::NT, 0, s>>>(ca, B);
    else if (last == 14) fused_chain_lb_kernel<false, true><<<grid, L8::NT, 0, s>>>(ca, B);
    else fused_chain_lb_kernel<false, false><<<grid, L8::NT, 0, s>>>(ca, B);
    return true;
}

// below: too few workgroups to put two on every CU (the hidden-sliced schedule, or the tiled kernel, is faster)
constexpr int kLbMinBatch = 768;

bool launch_fused_block_lb(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.Alb_e || !a.Alb_p || !a.Tlb) return false;
    if (B < kLbMinBatch) {
        // small batches: hidden-sliced.  Measured (us, tiled -> sliced): B = 128: features.12 / 13 / 14
        // 42 / 40 / 25 -> 25 / 29 / 21, features.8-11 20-22 -> 17-20; B = 512: 47 / 46 / 30 -> 42 / 42 / 31, but features.8-11 24-28 -> 27-32:
        // the narrow blocks only up to 192 faces.
        constexpr int sl_min = 32, narrow_max = 192;
        if (B < sl_min || feature < 8 || (feature < 12 && B > narrow_max)) return false;
        switch (feature) {
            case 8: case 9: case 10: return launch_lb_sliced<L8>(a, B, s);
            case 11: return launch_lb_sliced<L11>(a, B, s);
            case 12: case 13: return launch_lb_sliced<L12>(a, B, s);
            case 14: return launch_lb_sliced<L14>(a, B, s);
            default: return false;
        }
    }
    switch (feature) {
        case 7: {
            const LbStageArgs sa{a.X, a.Alb_e, a.Tlb, a.Alb_p, a.p_shift, a.Y};
            const int grid = (B + L7::FPW - 1) / L7::FPW;
            if (a.prof) fused_block_lb7_kernel<true><<<grid, L7::NT, 0, s>>>(sa, B, a.prof);
            else fused_block_lb7_kernel<false><<<grid, L7::NT, 0, s>>>(sa, B);
            return true;
        }
        case 8: case 9: case 10: launch_lb<L8>(a, B, s); return true;
        case 11: launch_lb<L11>(a, B, s); return true;
        case 12: case 13: launch_lb<L12>(a, B, s); return true;
        case 14: launch_lb<L14>(a, B, s); return true;
        default: return false;
    }
}

}  // namespace syn
