// Row-marching fused inverted-residual block for the EARLY MobileNetV2 blocks (features.2-4), second generation.
// Reference: backbone_nets/mobilenetv2_backbone.py:45-74 (InvertedResidual.forward), :33-42 (ConvBNReLU).
//
// The tiled kernel (fused_block_early.hip) moves every activation through LDS twice (expand -> Es -> depthwise -> D planes ->
// project) behind two workgroup barriers per hidden chunk: LDS-array cycles and barrier stalls are ~2/3 of its time
// (profiles/r2).  Here the hidden activations never leave registers:
//
//   * one WAVE owns 32 hidden channels ("group") of one face and marches down the image row by row;
//   * expand runs on v_mfma_f32_32x32x16_f16 (two fp16 pieces per operand, x = a + b to 22 bits, three partial products -- half the
//     matrix instructions of the exact 3-way bf16 split used before; power-of-two operand scales, see below) with the hidden channels as the
//     A rows and ONE IMAGE ROW as the 32 B columns: lane (j = l&31, h = l>>5) then holds 16 hidden channels
//     {0-3, 8-11, 16-19, 24-27} + 4h of pixel j -- the horizontal neighbours of a pixel are the neighbouring LANES
//     (v_mov_dpp wave_shr:1 / wave_shl:1), the vertical ones are earlier / later rows of the same lane;
//   * the depthwise 3x3 is "scattered": a fresh expand row adds its three kernel rows into the accumulators of the (up to)
//     three output rows it touches, so only accumulators are live, never a window of expanded rows;
//   * a finished depthwise row is ReLU6'ed, split into its two fp16 pieces IN PLACE -- with K-slot (h, e) := register
//     8s+e the D layout of one MFMA IS the B operand of the next (the host packs the project weights in that K order) --
//     and projected (6 MFMAs) to a partial sum over this wave's 32 hidden channels;
//   * the only LDS traffic is: the block-input row as pre-split B fragments, the per-wave partial sums and broadcast reads
//     of the depthwise filter; ONE barrier per input row;
//   * per unit one SERVICE wave does everything that is not the hidden pipeline: it loads the next block-input row, splits it
//     into the fp16 fragments all compute waves read, and reduces the partial sums of a finished output row in fixed wave order
//     (+ BN shift, + residual) into the NHWC store.  The compute waves run straight-line code; the service waves sit on the
//     SIMDs that hold one compute wave fewer (waves go to SIMDs cyclically), which evens out the 5-groups-on-4-SIMDs split.
//
//   stride 2: an input row is two column blocks, U = odd columns (-1, 1, 3, ...) and V = even columns, so that output pixel x
//   needs U[x], V[x] (same lane) and U[x+1] (wave_shl:1); features.4 (15-wide output) puts TWO faces side by side in a block.
//   Zero padding costs nothing: out-of-image lanes get ReLU6 ceiling 0 (v_med3), missing rows are simply not added.
#include "syn_internal.h"

#include <cstdlib>

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int cdivr(int a, int b) { return (a + b - 1) / b; }

// two floats -> packed fp16 pieces a (high) and b (low), x = a + b to 22 significant bits (both conversions toward zero)
__device__ __forceinline__ void split2r(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ f32x16 mfma32r(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the three partial products a a + a b + b a (b b <= 2^-22 dropped), smallest terms first
__device__ __forceinline__ f32x16 mac3r(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 c) {
    c = mfma32r(a[1], b[0], c);
    c = mfma32r(a[0], b[1], c);
    c = mfma32r(a[0], b[0], c);
    return c;
}
// lane j <- lane j-1 / lane j+1 across the whole wave (lanes 0 / 32 resp. 31 / 63 receive a neighbour half's or no value:
// they are padding columns whose results are never stored)
__device__ __forceinline__ float from_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
}  // namespace

template <int CIN_, int HID_, int COUT_, int H_, int S_, int NF_, bool RES_, int WPE_, int U_, bool LEAN_ = false, int WREG_ = 0, int NBD_ = 1, bool SVC2_ = false>
struct RmCfg {
    // SVC2 (round 5, features.5/6): TWO service waves per unit -- one stages the block-input rows, the other reduces the partial sums and stores --
    // so that the 2 x 6 compute + 4 service waves of a workgroup fill the four SIMDs evenly (3 + 1 each).  With one service wave per unit the
    // two of them sat on SIMD 0 and 1 beside three compute waves each, and every row step ended when THOSE SIMDs were done.
    static constexpr bool SVC2 = SVC2_;
    static constexpr int NSVW = SVC2_ ? 2 * U_ : U_;           // service waves of a workgroup
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_, H = H_, S = S_, NF = NF_, WPE = WPE_, U = U_;
    static constexpr bool RES = RES_;
    // LEAN (features.5/6: six hidden groups, two units = 14 waves want 4 waves per SIMD, i.e. <= 128 registers, and 160 KB of LDS):
    // the project weight fragments live in LDS instead of 24 registers, and the partial sums have ONE slot -- a second barrier per
    // step, which the compute waves pass only after the service wave has read the previous row's sums (it arrives long before).
    static constexpr bool LEAN = LEAN_;
    // WREG (stride 1): register quads whose depthwise filter (9 x 4 values per lane) stays in registers for the whole kernel instead of
    // being re-read from LDS every row step (10 of the 48 ds_read_b128 of a step per quad; the LDS arrays are busy 55-60 % of these kernels)
    static constexpr int WREG = WREG_;
    static constexpr int KS = cdivr(CIN, 16);            // k16 steps of the expand GEMM
    static constexpr int NG = cdivr(HID, 32);            // hidden groups of 32 channels = compute waves of a unit
    static constexpr int HIDP = NG * 32;
    static constexpr int NQ = COUT / 8;                  // valid register quads of the 32-row project tile
    static constexpr int NB = S == 1 ? 1 : 2;            // column blocks per input row
    static constexpr int HO = S == 2 ? H / 2 : H;
    // NBD > 1 (small batches): a unit is a BAND of HB output rows of its face(s), so that B faces give NBD x B / NF units to spread over the
    // chip.  A band marches the input rows its outputs touch -- stride 1: r0 - 1 .. r0 + HB (HB + 2 steps, padded to the row ring's multiple
    // of 3), stride 2: the row pairs r0 - 1 .. r0 + HB - 1 (2 HB + 2 steps) -- rows outside the image expand to zeros (ReLU6 ceiling 0, as the
    // padding columns do); every band runs the same number of steps (the units of a workgroup share its barriers).
    static constexpr int NBD = NBD_, HB = HO / NBD;
    static constexpr int BSTEPS = S == 1 ? cdivr(HB + 2, 3) * 3 : 2 * (HB + 1);
    static_assert(HO % NBD == 0 && (NBD == 1 || !LEAN_), "bands: equal heights; not implemented for the one-slot (LEAN) sums");
    static_assert(!SVC2_ || NBD_ == 1, "two service waves per unit: implemented for the whole-face march");
    // A workgroup carries U independent units (a unit = NF faces marching together) on one barrier: NG compute waves each
    // (wave ids 0 .. U*NG-1) and one service wave each (ids U*NG ..).  One big workgroup, not several small ones: the hardware
    // reserves ceil(waves / 4) wave slots on EVERY SIMD per workgroup, so two 5-wave workgroups never share a CU at 3 waves
    // per SIMD.
    static constexpr int NW = NG, NCW = U * NG, NT = (NCW + NSVW) * 64;
    static constexpr int FR = NB * KS;                   // block-input fragments per input row
    static constexpr int XP_DW = FR * 2 * 256, PART_DW = NW * NQ * 256;
    static constexpr int PSLOTS = LEAN ? 1 : 2;
    static constexpr int UNIT_DW = 2 * XP_DW + PSLOTS * PART_DW;
    static constexpr int APL_DW = LEAN ? NG * 2 * 2 * 256 : 0;
    static constexpr int LDS_DW = U * UNIT_DW + 11 * HIDP + 32 + APL_DW;     // per unit: X fragments x2 | partial sums;  filter 9 rows + depthwise shift | expand shift | project shift | (LEAN) project fragments
    static_assert(COUT % 8 == 0 && COUT <= 32, "project tile");
    static_assert(S == 1 || H % 2 == 0, "stride-2 blocks have even input sizes");
    static_assert(S == 1 ? (NF == 1 ? H + 2 <= 32 : 2 * H + 2 <= 32) : (NF == 1 ? H / 2 + 1 <= 32 : H / 2 + 1 <= 16), "one image row per 32-lane block");
    static_assert(!RES || (S == 1 && CIN == COUT), "residual only on stride-1 same-width blocks");
    static_assert(!LEAN || S == 1, "LEAN is implemented for the stride-1 march");
    static_assert(WREG == 0 || (S == 1 && WREG <= 4), "the register-held filter is implemented for the stride-1 march");
    static_assert(S == 2 || H % 3 == 0, "row ring unrolled by 3");
    static_assert(NT <= 1024 && LDS_DW * 4 <= 160 * 1024, "workgroup size / LDS budget");
};

#define SYNR_LAP(i) do { if (PROF) { tn = __builtin_amdgcn_s_memtime(); pt_[i] += tn - tk; tk = tn; } } while (0)
// PROF: barrier that also books the time since the previous barrier as this wave's busy time
#define SYNR_BARRIER() do { if (PROF) { const unsigned long long tb_ = __builtin_amdgcn_s_memtime(); busy_ += tb_ - tw_; __syncthreads(); tw_ = __builtin_amdgcn_s_memtime(); } else __syncthreads(); } while (0)

// one block (the whole kernel body; `smem` = the workgroup's C::LDS_DW dwords).  Service waves return from here at the end of their loop.
template <class C, bool PROF>
__device__ __forceinline__ void rm_block(unsigned *smem, const float *__restrict__ X, const unsigned *__restrict__ Ae3 /*[NG][KS][2][64][4]*/,
                           const unsigned *__restrict__ Ap3 /*[NG][2][2][64][4]*/, const float *__restrict__ e_shift,
                           const float *__restrict__ Wd /*[9][HID] scaled*/, const float *__restrict__ d_shift,
                           const float *__restrict__ p_shift, float *__restrict__ Y, int B, int ub_begin, int ub_end, int ub_step /*this workgroup's units: ub_begin, + ub_step, ... < ub_end (each round: C::U units)*/,
                           const float *__restrict__ scl_e /*{S, 1/S, 6 S} of the expand weights*/, const float *__restrict__ scl_p,
                           unsigned long long *prof) {
    // PROF: s_memtime sums of compute wave 0 per phase {(unused), expand, (unused), depthwise, finalize, barrier wait,
    // whole workgroup lifetime} and the number of row steps (syn_debug_profile_block)
    unsigned long long pt_[7] = {0, 0, 0, 0, 0, 0, 0}, tk = PROF ? __builtin_amdgcn_s_memtime() : 0ull, tn = 0, nsteps = 0;
    const unsigned long long t_begin = tk;
    unsigned long long busy_ = 0, tw_ = tk;
    constexpr int H = C::H, HO = C::HO, NW = C::NW, NT = C::NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = wave_wg >= C::NCW;
    const int uw = service ? (wave_wg - C::NCW) % C::U : wave_wg / NW;        // unit inside the workgroup
    // service roles (SVC2: wave NCW + u stages the rows of unit u, wave NCW + U + u reduces its sums; otherwise one wave does both)
    const bool svc_in = !C::SVC2 || wave_wg - C::NCW < C::U, svc_out = !C::SVC2 || wave_wg - C::NCW >= C::U;
    const int wave = service ? 0 : wave_wg % NW;                              // hidden group inside the unit (compute waves)
    unsigned *Xp = smem + uw * C::UNIT_DW;                                    // per unit: [2][FR][2][64][4]
    float *Part = reinterpret_cast<float *>(Xp + 2 * C::XP_DW);               // per unit: [2][NW][NQ][64][4]
    float *Filt = reinterpret_cast<float *>(smem + C::U * C::UNIT_DW);        // [9][HIDP] + row 9 = depthwise BN shift, shared
    float *Esh = Filt + 10 * C::HIDP, *Psh = Esh + C::HIDP;                   // [HIDP], [32]
    unsigned *ApL = reinterpret_cast<unsigned *>(Psh + 32);                   // LEAN: [NG][2][2][64][4] project fragments
    constexpr int DSH = 9 * C::HIDP;
    const int j = lane & 31, h = lane >> 5;
    const int cb = wave * 32 + 4 * h;               // hidden channel of register quad q: cb + 8q .. +3
    // HID % 32 == 16: the last group has 16 real channels = register quads 0, 1 (rows 0-15 of its A tile); quads 2, 3 are zero
    // padding -- their depthwise taps, splits and the second project k-step are skipped (wave-uniform), which makes that wave
    // light; with the cyclic wave -> SIMD placement the light waves are the third compute wave of their SIMD.
    const int nq_live = (C::HID % 32 == 16 && wave == NW - 1) ? 2 : 4;

    // lane geometry.  Input block b: lane j carries column icol[b] of face (unit*NF + ia); output rows: column ocol of face oa.
    int ia, icol[C::NB], oa, ocol;
    if (C::S == 1 && C::NF == 2) {
        // two H-wide faces in one block sharing their zero-padding columns: lane 0 | face 0 columns 0..H-1 | lane H+1 | face 1 | (lane 32 = the
        // other half's lane 0): every neighbour of an image column is either an image column of the same face or a zero lane
        ia = j > C::H; icol[0] = (j == 0 || j == C::H + 1) ? -1 : j - 1 - (C::H + 1) * ia; oa = ia; ocol = icol[0];
    } else if (C::S == 1) { ia = 0; icol[0] = j - 1; oa = 0; ocol = j - 1; }
    else if (C::NF == 1) { ia = 0; icol[0] = 2 * j - 1; icol[C::NB - 1] = 2 * j; oa = 0; ocol = j; }
    else { ia = j >> 4; icol[0] = 2 * (j & 15) - 1; icol[C::NB - 1] = 2 * (j & 15); oa = j >> 4; ocol = j & 15; }
    // Round 5: what the first row step needs from global memory is requested BEFORE the constants are staged (another round trip to L2 / HBM
    // behind a barrier): the compute waves' weight fragments, and -- as touches whose data is dropped -- the first two input rows of the first unit
    // of the row-staging service waves (their real loads then hit the cache).  SYN_RM_EARLY=0: the old order.
#ifndef SYN_RM_EARLY
#define SYN_RM_EARLY 1
#endif
    u32x4 ae[C::KS][2], ap[2][2];
    unsigned early_sink = 0;
    if (SYN_RM_EARLY) {
        if (!service) {
#pragma unroll
            for (int s = 0; s < C::KS; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) ae[s][p] = *(const u32x4 *)(Ae3 + ((size_t)(wave * C::KS + s) * 2 + p) * 256 + lane * 4);
            if (!C::LEAN)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) ap[s][p] = *(const u32x4 *)(Ap3 + ((size_t)(wave * 2 + s) * 2 + p) * 256 + lane * 4);
        } else if (svc_in && ub_begin < ub_end) {
            const int unit = ub_begin + uw, fu = C::NBD > 1 ? unit / C::NBD : unit, r0 = C::NBD > 1 ? (unit - fu * C::NBD) * C::HB : 0;
            const int f_in = fu * C::NF + ia, y0 = C::NBD > 1 ? (C::S == 1 ? r0 - 1 : 2 * (r0 - 1)) : 0;
#pragma unroll
            for (int b = 0; b < C::NB; ++b)
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const int y = y0 + yy;
                    if ((unsigned)icol[b] < (unsigned)H && f_in < B && (unsigned)y < (unsigned)H) {
                        const float *src = X + ((size_t)(f_in * H + y) * H + icol[b]) * C::CIN + 8 * h;
                        asm volatile("global_load_dword %0, %1, off" : "+v"(early_sink) : "v"(src) : "memory");
                    }
                }
        }
    }
    // power-of-two operand scales of the fp16 pieces (synergy_abi.hip): the expand accumulators start at Se x shift, ReLU6 clamps at
    // 6 Se and the depthwise filter carries 1 / Se; the project sums are rescaled by 1 / Sp where the service wave reduces them
    //
    // ReLU6 without instructions of its own (round 5; the register-resident blocks and the stem since round 4): both activations are carried as
    // relu6(x) / 6 in [0, 1] = what the `clamp` output modifier leaves -- on the multiply that rescales the expand accumulator (1 / (6 Se), 0 on
    // padding lanes) and on the LAST fused multiply-add a depthwise output receives.  The 6 rides on the constants: plain depthwise filter,
    // depthwise shift / 6, output rescale 6 / Sp.  Per row step and wave 32 v_med3 (4 issue cycles each) become 16 multiplies (2.7); SYN_RM_MED3=1: the old form.
#ifndef SYN_RM_MED3
#define SYN_RM_MED3 0
#endif
    constexpr bool CLAMP = !SYN_RM_MED3;
    const float Se = scl_e[0], inv_se = scl_e[1], c6e = CLAMP ? scl_e[1] * (1.0f / 6.0f) : scl_e[2], inv_sp = CLAMP ? scl_p[1] * 6.0f : scl_p[1];
    const float f_scale = CLAMP ? 1.0f : inv_se, d_scale = CLAMP ? 1.0f / 6.0f : 1.0f;
    for (int i = tid; i < 9 * C::HIDP; i += NT) { const int c = i % C::HIDP; Filt[i] = c < C::HID ? Wd[(i / C::HIDP) * C::HID + c] * f_scale : 0.f; }
    for (int i = tid; i < C::HIDP; i += NT) { Filt[DSH + i] = i < C::HID ? d_shift[i] * d_scale : 0.f; Esh[i] = i < C::HID ? e_shift[i] * Se : 0.f; }
    if (tid < 32) Psh[tid] = tid < C::COUT ? p_shift[tid] : 0.f;
    if (C::LEAN)
        for (int i = tid; i < C::APL_DW / 4; i += NT) *(u32x4 *)&ApL[4 * i] = *(const u32x4 *)&Ap3[4 * i];

    __syncthreads();
    // the touches are inline assembly the compiler's wait-count bookkeeping does not know, and a barrier alone drains nothing: wait for them
    // explicitly before early_sink's register may be reused (service waves that stage no constants issue no tracked load a
    // compiler wait would cover).  By now they have long returned; the compute waves' fragment loads are consumed right behind anyway.
    l2_touch_done(early_sink);

    if (service) {
        // The service waves are the youngest of their SIMD and would be served last by the issue arbiter, yet every compute wave
        // waits for them at the row barrier: raise their priority (their work is a fraction of a compute wave's).
        __builtin_amdgcn_s_setprio(3);
        // =====================================================================================================================
        // service wave of unit uw: block-input rows -> fp16 x2 fragments in LDS; finished output rows: partial sums -> NHWC row
        // =====================================================================================================================
        for (int ub = ub_begin; ub < ub_end; ub += ub_step) {
            const int unit = ub + uw;
            const int fu = C::NBD > 1 ? unit / C::NBD : unit, r0 = C::NBD > 1 ? (unit - fu * C::NBD) * C::HB : 0;     // face unit, first output row of the band
            const int f_in = fu * C::NF + ia, f_out = fu * C::NF + oa;
            const bool out_ok = (unsigned)ocol < (unsigned)HO && f_out < B;
            f32x4 xr[C::FR][2];
            // fragment b*KS + s = channels 16s + 8h .. +7 of the pixels of block b
            auto load_row = [&](int y) {
#pragma unroll
                for (int fr = 0; fr < C::FR; ++fr) {
                    const int b = fr / C::KS, sk = fr % C::KS, c0 = 16 * sk + 8 * h;
                    xr[fr][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; xr[fr][1] = xr[fr][0];
                    if (c0 + 8 <= C::CIN && (unsigned)icol[b] < (unsigned)H && f_in < B && (C::NBD == 1 || (unsigned)y < (unsigned)H)) {
                        const float *src = X + ((size_t)(f_in * H + y) * H + icol[b]) * C::CIN + c0;
                        xr[fr][0] = *(const f32x4 *)src; xr[fr][1] = *(const f32x4 *)(src + 4);
                    }
                }
            };
            auto store_row = [&](int slot) {
#pragma unroll
                for (int fr = 0; fr < C::FR; ++fr) {
                    u32x4 pc[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        unsigned a0, b0, a1, b1;
                        split2r(xr[fr][t][0], xr[fr][t][1], a0, b0);
                        split2r(xr[fr][t][2], xr[fr][t][3], a1, b1);
                        pc[0][2 * t] = a0; pc[0][2 * t + 1] = a1; pc[1][2 * t] = b0; pc[1][2 * t + 1] = b1;
                    }
                    unsigned *dst = Xp + (size_t)slot * C::XP_DW + fr * 512 + lane * 4;
#pragma unroll
                    for (int p = 0; p < 2; ++p) *(u32x4 *)(dst + p * 256) = pc[p];
                }
            };
            // Global latencies get a whole row step: row y+2 and the residual of output row y-1 are requested in step y and
            // consumed in step y+1.
            f32x4 res[C::RES ? C::NQ : 1];
            auto load_res = [&](int yo) {
                if (C::RES) {
#pragma unroll
                    for (int q = 0; q < C::NQ; ++q) {
                        res[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (out_ok && yo >= 0 && yo < HO) res[q] = *(const f32x4 *)(X + ((size_t)(f_out * H + yo) * H + ocol) * C::CIN + 8 * q + 4 * h);
                    }
                }
            };
            // partial sums of all compute waves in fixed order + BN shift (+ residual, already in `res`) -> NHWC row
            auto reduce_row = [&](int yo, int pslot) {
#pragma unroll
                for (int q = 0; q < C::NQ; ++q) {
                    const float *src = Part + (size_t)pslot * C::PART_DW + q * 256 + lane * 4;
                    f32x4 v = *(const f32x4 *)src;
#pragma unroll
                    for (int w = 1; w < NW; ++w) v += *(const f32x4 *)(src + (size_t)w * C::NQ * 256);
                    v = v * inv_sp + *(const f32x4 *)&Psh[8 * q + 4 * h];
                    if (C::RES) v += res[q];
                    if (out_ok) *(f32x4 *)(Y + ((size_t)(f_out * HO + yo) * HO + ocol) * C::COUT + 8 * q + 4 * h) = v;
                }
            };
            if (C::NBD > 1) {
                // band: local step k marches input row gy0 + k; the sums finalized in step k are reduced in step k + 1
                constexpr int NS = C::BSTEPS, HB = C::HB;
                const int gy0 = C::S == 1 ? r0 - 1 : 2 * (r0 - 1);
                load_row(gy0);
                store_row(0);
                load_row(gy0 + 1);
                SYNR_BARRIER();                              // (P)
                for (int k = 0; k < NS; ++k) {
                    if (C::S == 1) {
                        if (k >= 3 && k < HB + 3) reduce_row(r0 + k - 3, (k - 1) & 1);       // output row r0 + k - 3 was finalized in step k - 1
                        if (k >= 2) load_res(r0 + k - 2);
                    } else {
                        if (!(k & 1) && k >= 4) reduce_row(r0 + (k >> 1) - 2, ((k >> 1) - 1) & 1);      // pair k/2 - 1 finalized output row r0 - 1 + (k/2 - 1)
                    }
                    if (k + 1 < NS) store_row((k + 1) & 1);
                    if (k + 2 < NS) load_row(gy0 + k + 2);
                    SYNR_BARRIER();
                }
                if (C::S == 2) reduce_row(r0 + HB - 1, HB & 1);
                else if (NS == HB + 2) reduce_row(r0 + HB - 1, (NS - 1) & 1);
                continue;
            }
            if (svc_in) { load_row(0); store_row(0); load_row(1); }     // (row 1 in flight across the barrier)
            SYNR_BARRIER();                                  // (P) row 0 is in slot 0
            if (C::S == 1) {
                if (svc_out) load_res(0);
                constexpr int PM = C::PSLOTS - 1;             // partial-sum slot of output row r: r & PM
                for (int y = 0; y < H; ++y) {
                    if (svc_out && y >= 2) reduce_row(y - 2, y & PM);    // completed by the barrier that ended step y-1; the slot is rewritten in step y+1 (LEAN: y, after the barrier below)
                    if (C::LEAN) SYNR_BARRIER();              // the compute waves write this step's sums only after it
                    if (svc_out) load_res(y - 1);             // for the next step's reduction
                    if (svc_in && y + 1 < H) store_row((y + 1) & 1);    // row y+1 (requested a step ago); slot (y+1)&1 was last read in step y-1
                    if (svc_in && y + 2 < H) load_row(y + 2);
                    SYNR_BARRIER();
                }
                if (svc_out) reduce_row(H - 2, (H - 2) & PM);
                if (C::LEAN) SYNR_BARRIER();
                if (svc_out) load_res(H - 1);
                SYNR_BARRIER();                              // the compute waves finalized the last row
                if (svc_out) reduce_row(H - 1, (H - 1) & PM);
            } else {
                for (int y = 0; y < H; ++y) {
                    if (svc_out && !(y & 1) && y >= 2) reduce_row((y >> 1) - 1, ((y >> 1) - 1) & 1);     // completed by the barrier that ended odd step y-1
                    if (svc_in && y + 1 < H) store_row((y + 1) & 1);
                    if (svc_in && y + 2 < H) load_row(y + 2);
                    SYNR_BARRIER();
                }
                if (svc_out) reduce_row(HO - 1, (HO - 1) & 1);
            }
        }
        if (PROF && lane == 0) atomicAdd(&prof[8 + wave_wg], busy_);
        return;
    }

    // =========================================================================================================================
    // compute wave: hidden group `wave` of unit uw
    // =========================================================================================================================
    // this wave's weight fragments stay in registers for the whole (persistent) kernel
    if (!SYN_RM_EARLY) {
#pragma unroll
        for (int s = 0; s < C::KS; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p) ae[s][p] = *(const u32x4 *)(Ae3 + ((size_t)(wave * C::KS + s) * 2 + p) * 256 + lane * 4);
        if (!C::LEAN)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) ap[s][p] = *(const u32x4 *)(Ap3 + ((size_t)(wave * 2 + s) * 2 + p) * 256 + lane * 4);
    }

    f32x4 wreg[C::WREG > 0 ? C::WREG : 1][9], wbase[C::WREG > 0 ? C::WREG : 1];
#pragma unroll
    for (int q = 0; q < C::WREG; ++q) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wreg[q][k] = *(const f32x4 *)(Filt + cb + 8 * q + k * C::HIDP);
        wbase[q] = *(const f32x4 *)(Filt + cb + 8 * q + DSH);
    }
    // every wave of the workgroup runs the same number of rounds (and barriers); a unit past the end computes on zeros and
    // stores nothing (its faces are >= B)
    for (int ub = ub_begin; ub < ub_end; ub += ub_step) {
        const int unit = ub + uw;
        const int fu = C::NBD > 1 ? unit / C::NBD : unit, r0 = C::NBD > 1 ? (unit - fu * C::NBD) * C::HB : 0;
        const int f_in = fu * C::NF + ia;
        float ehi[C::NB];                         // ReLU6 ceiling of the expanded pixel: 6 inside the image, 0 on padding lanes (clamp form: the multiplier 1 / (6 Se) | 0)
        float ehs[C::NB];                         // ... of the row being expanded (bands: 0 for the rows above / below the image)
#pragma unroll
        for (int b = 0; b < C::NB; ++b) ehs[b] = ehi[b] = ((unsigned)icol[b] < (unsigned)H && f_in < B) ? c6e : 0.0f;
        auto row_ceiling = [&](int gy) {
#pragma unroll
            for (int b = 0; b < C::NB; ++b) ehs[b] = (unsigned)gy < (unsigned)H ? ehi[b] : 0.0f;
        };

        // ---- expand one block of the row in slot `slot`: 16 hidden channels per lane, BN shift, ReLU6 (0 on padding lanes) ----
        auto expand = [&](int slot, int b, f32x16 &e, int cbo) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sh = *(const f32x4 *)&Esh[cbo + 8 * q];
#pragma unroll
                for (int t = 0; t < 4; ++t) e[4 * q + t] = sh[t];
            }
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                u32x4 xb[2];
                const unsigned *src = Xp + (size_t)slot * C::XP_DW + (b * C::KS + s) * 512 + lane * 4;
#pragma unroll
                for (int p = 0; p < 2; ++p) xb[p] = *(const u32x4 *)(src + p * 256);
                e = mac3r(ae[s], xb, e);
            }
            // (compiler-made clamp: a vector instruction that reads a matrix result needs wait states only the compiler inserts)
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = CLAMP ? __builtin_amdgcn_fmed3f(e[r] * ehs[b], 0.0f, 1.0f) : __builtin_amdgcn_fmed3f(e[r], 0.0f, ehs[b]);
        };
        // ---- finished depthwise row -> ReLU6 -> fp16 x2 pieces (in place: register 8s+e = K slot e of step s) -> project
        //      partial over this wave's 32 hidden channels -> LDS ----
        auto finalize = [&](f32x16 &d, int pslot, bool raw = false) {        // raw: an output row whose last kernel row does not exist (clamp form: not clamped yet)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (2 * s >= nq_live) break;                // padded half of the last group: its pieces and weights are zero
                u32x4 db[2];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v0 = d[8 * s + 2 * t], v1 = d[8 * s + 2 * t + 1];
                    if (!CLAMP) { v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, 6.0f); v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, 6.0f); }
                    else if (raw) { v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, 1.0f); v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, 1.0f); }
                    unsigned ha, hb;
                    split2r(v0, v1, ha, hb);
                    db[0][t] = ha; db[1][t] = hb;
                }
                if (C::LEAN) {
                    u32x4 apl[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) apl[p] = *(const u32x4 *)(ApL + ((size_t)(wave * 2 + s) * 2 + p) * 256 + lane * 4);
                    acc = mac3r(apl, db, acc);
                } else {
                    acc = mac3r(ap[s], db, acc);
                }
            }
            float *dst = Part + (size_t)pslot * C::PART_DW + (size_t)wave * C::NQ * 256 + lane * 4;
#pragma unroll
            for (int q = 0; q < C::NQ; ++q) *(f32x4 *)(dst + q * 256) = (f32x4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        };
        // three taps of one kernel row into one accumulator quad; `init`: the accumulator starts at the BN shift (Filt row 9)
        // `last` (clamp form): this kernel row completes the output row -- its last multiply-add clamps to [0, 1]
        auto taps3 = [&](f32x16 &d, int q, const float *wq, int ky, const f32x4 &l4, const f32x4 &c4, const f32x4 &r4, bool init, bool last = false) {
            const f32x4 w0 = *(const f32x4 *)(wq + (3 * ky + 0) * C::HIDP), w1 = *(const f32x4 *)(wq + (3 * ky + 1) * C::HIDP),
                        w2 = *(const f32x4 *)(wq + (3 * ky + 2) * C::HIDP);
            f32x4 base;
            if (init) base = *(const f32x4 *)(wq + DSH);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b0 = init ? base[t] : d[4 * q + t];
                const float v = __builtin_fmaf(r4[t], w2[t], __builtin_fmaf(c4[t], w1[t], __builtin_fmaf(l4[t], w0[t], b0)));
                d[4 * q + t] = CLAMP && last ? __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f) : v;
            }
            // pin the update here: otherwise the compiler sinks these FMAs to where the accumulator is next read (the following
            // row's step) and keeps their operands + filter quads alive across the barrier instead
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d[4 * q + t]));
        };
        // the same with the filter quads in registers (wr[3 * ky + kx])
        auto taps3r = [&](f32x16 &d, int q, const f32x4 *wr, const f32x4 &base, int ky, const f32x4 &l4, const f32x4 &c4, const f32x4 &r4, bool init, bool last = false) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b0 = init ? base[t] : d[4 * q + t];
                const float v = __builtin_fmaf(r4[t], wr[3 * ky + 2][t], __builtin_fmaf(c4[t], wr[3 * ky + 1][t], __builtin_fmaf(l4[t], wr[3 * ky][t], b0)));
                d[4 * q + t] = CLAMP && last ? __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f) : v;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d[4 * q + t]));
        };
        // two taps (U blocks of the stride-2 layout: this lane and its right neighbour) / one tap (V blocks)
        auto taps2 = [&](f32x16 &d, int q, const float *wq, int ta, int tb, const f32x4 &c4, const f32x4 &r4, bool init) {
            const f32x4 wa = *(const f32x4 *)(wq + ta * C::HIDP), wb = *(const f32x4 *)(wq + tb * C::HIDP);
            f32x4 base;
            if (init) base = *(const f32x4 *)(wq + DSH);
#pragma unroll
            for (int t = 0; t < 4; ++t) d[4 * q + t] = __builtin_fmaf(r4[t], wb[t], __builtin_fmaf(c4[t], wa[t], init ? base[t] : d[4 * q + t]));
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d[4 * q + t]));
        };
        auto taps1 = [&](f32x16 &d, int q, const float *wq, int ta, const f32x16 &e, bool last = false) {
            const f32x4 wa = *(const f32x4 *)(wq + ta * C::HIDP);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = __builtin_fmaf(e[4 * q + t], wa[t], d[4 * q + t]);
                d[4 * q + t] = CLAMP && last ? __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f) : v;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d[4 * q + t]));
        };
        // the filter / shift reads are loop invariant; an opaque base keeps the compiler from hoisting 150 registers' worth of them
        // out of the row loop (and spilling them)
        auto opaque_cb = [&]() { int c = cb; asm volatile("" : "+v"(c)); return c; };

        SYNR_BARRIER();                                      // (P) row 0 is in slot 0

        if (C::S == 1) {
            f32x16 d0, d1, d2;
            {
                const int cbo = opaque_cb();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 sh = *(const f32x4 *)(Filt + cbo + 8 * q + DSH);      // output row 0 starts at the BN shift
#pragma unroll
                    for (int t = 0; t < 4; ++t) { d0[4 * q + t] = sh[t]; d1[4 * q + t] = 0.f; d2[4 * q + t] = 0.f; }
                }
            }
            // input row y -> kernel row 2 of output row y-1 (dm), row 1 of y (dc), row 0 of y+1 (dn)
            // (xslot: fragment slot of the row; fin / pslot: finalize dm into that partial-sum slot)
            auto step = [&](int xslot, bool fin, int pslot, f32x16 &dm, f32x16 &dc, f32x16 &dn) {
                SYNR_LAP(5);
                const int cbo = opaque_cb();
                f32x16 e;
                expand(xslot, 0, e, cbo);
                SYNR_LAP(1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= nq_live) break;
                    f32x4 c4, l4, r4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { c4[t] = e[4 * q + t]; l4[t] = from_left(c4[t]); r4[t] = from_right(c4[t]); }
                    const float *wq = Filt + cbo + 8 * q;
                    if (q < C::WREG) {
                        taps3r(dn, q, wreg[q < C::WREG ? q : 0], wbase[q < C::WREG ? q : 0], 0, l4, c4, r4, true);
                        taps3r(dc, q, wreg[q < C::WREG ? q : 0], wbase[q < C::WREG ? q : 0], 1, l4, c4, r4, false);
                        taps3r(dm, q, wreg[q < C::WREG ? q : 0], wbase[q < C::WREG ? q : 0], 2, l4, c4, r4, false, true);
                    } else {
                        taps3(dn, q, wq, 0, l4, c4, r4, true);
                        taps3(dc, q, wq, 1, l4, c4, r4, false);
                        taps3(dm, q, wq, 2, l4, c4, r4, false, true);
                    }
                    __builtin_amdgcn_sched_barrier(0);          // one register quad at a time
                }
                SYNR_LAP(3);
                if (C::LEAN) SYNR_BARRIER();                   // the service wave has read the previous row's sums
                if (fin) finalize(dm, pslot);
                SYNR_LAP(4);
                SYNR_BARRIER();
                nsteps += 1;
            };
            if (C::NBD > 1) {
                // band: step k marches input row r0 - 1 + k and completes output row r0 + k - 2 (the accumulators of rows r0 - 2, r0 - 1 are
                // never finalized; padded steps k >= HB + 2 complete nothing)
                auto bstep = [&](int k, f32x16 &dm, f32x16 &dc, f32x16 &dn) {
                    row_ceiling(r0 - 1 + k);
                    step(k & 1, k >= 2 && k < C::HB + 2, k & 1, dm, dc, dn);
                };
                for (int k = 0; k < C::BSTEPS; k += 3) {
                    bstep(k, d2, d0, d1);
                    bstep(k + 1, d0, d1, d2);
                    bstep(k + 2, d1, d2, d0);
                }
            } else {
                for (int y = 0; y < H; y += 3) {
                    step(y & 1, y >= 1, (y - 1) & (C::PSLOTS - 1), d2, d0, d1);
                    step((y + 1) & 1, true, y & (C::PSLOTS - 1), d0, d1, d2);
                    step((y + 2) & 1, true, (y + 1) & (C::PSLOTS - 1), d1, d2, d0);
                }
                // the last output row has no input row below it: complete as it is ((H-1) % 3 == 2 -> d2)
                if (C::LEAN) SYNR_BARRIER();
                finalize(d2, (H - 1) & (C::PSLOTS - 1), true);
                SYNR_BARRIER();
            }
        } else {
            f32x16 dcur, dnext;
            {
                const int cbo = opaque_cb();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 sh = *(const f32x4 *)(Filt + cbo + 8 * q + DSH);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { dcur[4 * q + t] = sh[t]; dnext[4 * q + t] = 0.f; }
                }
            }
            // pairs of input rows (2 yo, 2 yo + 1); bands: yo = r0 - 1 .. r0 + HB - 1, the first pair only starts output row r0
            const int np = C::NBD > 1 ? C::HB + 1 : HO, p0 = C::NBD > 1 ? r0 - 1 : 0;
            for (int lp = 0; lp < np; ++lp) {
                const int yo = p0 + lp;                      // (even rows travel in fragment slot 0, odd rows in slot 1)
                // ---- even input row 2yo: kernel row 1 of output row yo ----
                {
                    if (C::NBD > 1) row_ceiling(2 * yo);
                    SYNR_LAP(5);
                    const int cbo = opaque_cb();
                    f32x16 e;
                    expand(0, 0, e, cbo);                                    // U: columns 2x-1 (tap 3) and, from the right lane, 2x+1 (tap 5)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q >= nq_live) break;
                        f32x4 c4, r4;
#pragma unroll
                        for (int t = 0; t < 4; ++t) { c4[t] = e[4 * q + t]; r4[t] = from_right(c4[t]); }
                        taps2(dcur, q, Filt + cbo + 8 * q, 3, 5, c4, r4, false);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    expand(0, 1, e, cbo);                                    // V: column 2x (tap 4)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q >= nq_live) break;
                        taps1(dcur, q, Filt + cbo + 8 * q, 4, e);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    SYNR_LAP(3);
                    SYNR_BARRIER();
                    nsteps += 1;
                }
                // ---- odd input row 2yo+1: kernel row 2 of output row yo, kernel row 0 of output row yo+1 ----
                {
                    if (C::NBD > 1) row_ceiling(2 * yo + 1);
                    SYNR_LAP(5);
                    const int cbo = opaque_cb();
                    f32x16 e;
                    expand(1, 0, e, cbo);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q >= nq_live) break;
                        f32x4 c4, r4;
#pragma unroll
                        for (int t = 0; t < 4; ++t) { c4[t] = e[4 * q + t]; r4[t] = from_right(c4[t]); }
                        taps2(dcur, q, Filt + cbo + 8 * q, 6, 8, c4, r4, false);
                        taps2(dnext, q, Filt + cbo + 8 * q, 0, 2, c4, r4, true);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    expand(1, 1, e, cbo);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q >= nq_live) break;
                        taps1(dcur, q, Filt + cbo + 8 * q, 7, e, true);
                        taps1(dnext, q, Filt + cbo + 8 * q, 1, e);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    SYNR_LAP(3);
                    if (C::NBD == 1 || lp >= 1) finalize(dcur, lp & 1);
                    SYNR_LAP(4);
                    dcur = dnext;
                    SYNR_BARRIER();
                    nsteps += 1;
                }
            }
        }
    }
    if (PROF && lane == 0) atomicAdd(&prof[8 + wave_wg], busy_);
    if (PROF && tid == 0) {
        pt_[6] = __builtin_amdgcn_s_memtime() - t_begin;          // whole lifetime of the workgroup (column "epilog" of tools/stage_profile.py)
        for (int i = 0; i < 7; ++i) atomicAdd(&prof[i], pt_[i]);
        atomicAdd(&prof[7], nsteps);
    }
}

template <class C, bool PROF = false>
__global__ __launch_bounds__(C::NT) __attribute__((amdgpu_waves_per_eu(C::WPE, C::WPE)))
void fused_block_rm_kernel(const float *__restrict__ X, const unsigned *__restrict__ Ae3, const unsigned *__restrict__ Ap3, const float *__restrict__ e_shift,
                           const float *__restrict__ Wd, const float *__restrict__ d_shift, const float *__restrict__ p_shift, float *__restrict__ Y, int B,
                           int n_units, const float *__restrict__ scl_e, const float *__restrict__ scl_p, unsigned long long *prof = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DW];
    rm_block<C, PROF>(smem, X, Ae3, Ap3, e_shift, Wd, d_shift, p_shift, Y, B, blockIdx.x * C::U, n_units, gridDim.x * C::U, scl_e, scl_p, prof);
}

// TWO consecutive blocks in one launch (round 5: features.5 + 6, features.3 + 4).  A workgroup marches its units through the first
// block, then the SAME units through the second: unit u of the second block reads exactly what unit u of the first one wrote (the same
// faces), so nothing crosses workgroups and the kernel boundary between the two launches -- the drain of one grid, the dispatch of the
// next, a second prologue on an empty chip -- becomes one workgroup barrier.  The first block's output still goes through global memory
// (it is the second block's residual too): stored by this CU's service waves, read back by them past the barrier (workgroup-scope
// release / acquire: one CU, one vector cache, write-through).
struct RmStageArgs {
    const float *X; const unsigned *Ae3, *Ap3; const float *e_shift, *Wd, *d_shift, *p_shift; float *Y; const float *scl_e, *scl_p;
};
// CA, CB: the two blocks' configurations (same workgroup size and register budget).  A workgroup takes GROUPS of FG faces -- FG = what one round of
// either configuration covers at most -- through block A (one or two rounds of its units), then through block B.
template <class CA, class CB>
__global__ __launch_bounds__(CA::NT) __attribute__((amdgpu_waves_per_eu(CA::WPE, CA::WPE)))
void fused_pair_rm_kernel(RmStageArgs a, RmStageArgs b, int B) {
    static_assert(CA::NT == CB::NT && CA::WPE == CB::WPE && CA::NBD == 1 && CB::NBD == 1, "one workgroup shape for both blocks; whole-face marches");
    constexpr int FA = CA::U * CA::NF, FB = CB::U * CB::NF, FG = FA > FB ? FA : FB;       // faces per round of A / of B / per group
    static_assert(FG % FA == 0 && FG % FB == 0, "a group is whole rounds of both blocks");
    __shared__ __attribute__((aligned(16))) unsigned smem[CA::LDS_DW > CB::LDS_DW ? CA::LDS_DW : CB::LDS_DW];
    const int nua = (B + CA::NF - 1) / CA::NF, nub = (B + CB::NF - 1) / CB::NF;
    {   // one group per workgroup (not persistent: a loop over groups around the two inlined blocks kept both blocks' lane geometry alive and spilled)
        const int gq = blockIdx.x;
        const int ua = gq * (FG / CA::NF), ub = gq * (FG / CB::NF);                        // first unit of the group in A's / B's unit numbering
        rm_block<CA, false>(smem, a.X, a.Ae3, a.Ap3, a.e_shift, a.Wd, a.d_shift, a.p_shift, a.Y, B, ua, ua + FG / CA::NF < nua ? ua + FG / CA::NF : nua, CA::U,
                            a.scl_e, a.scl_p, nullptr);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's stores of the first block are out of the CU
        __syncthreads();                                            // ... and so are everybody's; nobody still reads the first block's LDS
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        rm_block<CB, false>(smem, b.X, b.Ae3, b.Ap3, b.e_shift, b.Wd, b.d_shift, b.p_shift, b.Y, B, ub, ub + FG / CB::NF < nub ? ub + FG / CB::NF : nub, CB::U,
                            b.scl_e, b.scl_p, nullptr);
    }
}

template <class C>
static void launch_rm(const FusedBlockArgs &a, int B, hipStream_t s, int wgs_per_cu) {
    const int n_units = (B + C::NF - 1) / C::NF * C::NBD;
    const int cap = 256 * wgs_per_cu, wgs = (n_units + C::U - 1) / C::U;  // persistent: as many workgroups as the CUs hold at once
    const int grid = wgs < cap ? wgs : cap;
    if (a.prof)
        fused_block_rm_kernel<C, true><<<grid, C::NT, 0, s>>>(a.X, a.Arm_e, a.Arm_p, a.e_shift, a.Wd, a.d_shift, a.p_shift, a.Y, B, n_units, a.scl_e, a.scl_p, a.prof);
    else
        fused_block_rm_kernel<C><<<grid, C::NT, 0, s>>>(a.X, a.Arm_e, a.Arm_p, a.e_shift, a.Wd, a.d_shift, a.p_shift, a.Y, B, n_units, a.scl_e, a.scl_p);
}

// Units per workgroup: as many as fit a CU, but a batch must still fill the chip -- the kernels are persistent over units, so with
// fewer units than 256 workgroups x U the time of a launch stops shrinking with the batch (one round of a workgroup is ~35 us per
// unit row-march).  Smaller batches take a smaller U; below the last threshold the spatially tiled kernels (fused_block_early.hip),
// which split a face over many workgroups, are faster and the launcher declines.
//                       CIN  HID COUT  H  S NF  RES   waves/SIMD  units/workgroup
template <int U> using R2 = RmCfg< 16,  96,  24, 60, 2, 1, false, (U == 4 ? 4 : 3), U>;    // features.2   60 -> 30      U x (3 + 1) waves
#ifndef SYN_R3_WREG
#define SYN_R3_WREG 1
#endif
template <int U> using R3 = RmCfg< 24, 144,  24, 30, 1, 1, true,  3, U, false, SYN_R3_WREG>;                   // features.3   30            U x (5 + 1) waves
template <int U> using R4 = RmCfg< 24, 144,  32, 30, 2, 2, false, 3, U>;                   // features.4   30 -> 15      U x (5 + 1) waves, two faces per unit
#ifndef SYN_R5_SVC2
#define SYN_R5_SVC2 1
#endif
template <int U> using R5 = RmCfg< 32, 192,  32, 15, 1, 2, true,  4, U, true, 0, 1, SYN_R5_SVC2 != 0>;    // features.5/6 15     U x (6 + 2) waves, two faces per unit, 4 per SIMD
// small batches: NBD row bands per face (RmCfg::NBD).  A band march has a floor of its own -- its steps are a dependent chain of ~1.3-2 us
// each whatever the batch (features.2: 12 steps + prologue = 16-18 us at B = 1, features.3: 25) -- against 12 us for the tiled kernels, so the
// bands pay in a window: features.2 from ~40 faces (B = 64 / 128 / 256: 20 / 24 / 38 us against 24 / 36 / 57 tiled; six bands of five output
// rows, one unit per workgroup, three workgroups per CU), features.3 from ~96 (B = 128: 28 against 34; three bands of ten rows, two units per
// workgroup).  features.4 (two faces per unit: 21 against 23 us at B = 128, 34 against 32 at 256) and the other band counts / unit counts
// tried (features.2: 3 or 5 bands 30 / 26 us, two units 29; features.3: 2 bands 29, 5 bands 38) are not instantiated.  tools/band_ab.sh.
template <int U, int NBD> using R2b = RmCfg< 16,  96,  24, 60, 2, 1, false, 3, U, false, 0, NBD>;
template <int U, int NBD> using R3b = RmCfg< 24, 144,  24, 30, 1, 1, true,  3, U, false, SYN_R3_WREG, NBD>;
constexpr int kBand2Min = 40, kBand3Min = 96;

// Two consecutive blocks in one launch (B >= 513: the configurations launch_fused_block_rm would pick for either): features.5 + 6 (first = 5) and
// features.3 + 4 (first = 3).  false: launch them one by one
template <class CA, class CB>
static void launch_pair(const FusedBlockArgs &a, const FusedBlockArgs &b, int B, hipStream_t s) {
    constexpr int FA = CA::U * CA::NF, FB = CB::U * CB::NF, FG = FA > FB ? FA : FB;
    const int n_groups = (B + FG - 1) / FG;             // one workgroup per group of FG faces (256 resident at a time)
    fused_pair_rm_kernel<CA, CB><<<n_groups, CA::NT, 0, s>>>(RmStageArgs{a.X, a.Arm_e, a.Arm_p, a.e_shift, a.Wd, a.d_shift, a.p_shift, a.Y, a.scl_e, a.scl_p},
                                                        RmStageArgs{b.X, b.Arm_e, b.Arm_p, b.e_shift, b.Wd, b.d_shift, b.p_shift, b.Y, b.scl_e, b.scl_p}, B);
}
bool launch_fused_pair_rm(int first, const FusedBlockArgs &a, const FusedBlockArgs &b, int B, hipStream_t s) {
    static const bool on56 = test_knob("rm_pair56", 1) != 0, on34 = test_knob("rm_pair34", 1) != 0;
    if (B < 513 || a.prof || b.prof) return false;
    if (!a.Arm_e || !a.Arm_p || !a.scl_e || !a.scl_p || !b.Arm_e || !b.Arm_p || !b.scl_e || !b.scl_p) return false;
    if (first == 5 && on56) { launch_pair<R5<2>, R5<2>>(a, b, B, s); return true; }
    // features.3 + 4: one round of workgroups only (B <= 1024) -- a group of four faces is two rounds of features.3 and one of features.4 (~150 us), and
    // a partly filled LAST round of such workgroups costs more than the boundary saves (B = 2307: 430 against 388 us; B = 1024: 153 against 160)
    if (first == 3 && on34 && (B + 3) / 4 <= 256) { launch_pair<R3<2>, R4<2>>(a, b, B, s); return true; }
    return false;
}

bool launch_fused_block_rm(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.Arm_e || !a.Arm_p || !a.scl_e || !a.scl_p) return false;
    // faces at which a configuration starts to pay (measured, tools/perlaunch.py --batch N).  The kernels are
    // persistent over 256 workgroups: 513 faces is where the smaller configuration needs a second round of workgroups (B = 640, us: features.2
    // 123 -> 91, features.4 74 -> 52, features.5/6 51 / 49 -> 43 / 42 with the larger one; at B = 512 the smaller one wins: 66 / 42 / 36 vs 88 / 50 / 41)
    static const bool bands2 = test_knob("rm_band2", 1) != 0, bands3 = test_knob("rm_band3", 1) != 0;     // (0: the tiled kernels below the thresholds, as before round 4)
    switch (feature) {
        case 2:
            if (B >= 513) { launch_rm<R2<4>>(a, B, s, 1); return true; }
            if (B >= 352) { launch_rm<R2<2>>(a, B, s, 1); return true; }
            if (bands2 && B >= kBand2Min) { launch_rm<R2b<1, 6>>(a, B, s, 3); return true; }
            return false;
        case 3:
            if (B >= 448) { launch_rm<R3<2>>(a, B, s, 1); return true; }
            if (B >= 200) { launch_rm<R3<1>>(a, B, s, 1); return true; }
            if (bands3 && B >= kBand3Min) { launch_rm<R3b<2, 3>>(a, B, s, 1); return true; }
            return false;
        case 4:
            if (B >= 513) { launch_rm<R4<2>>(a, B, s, 1); return true; }
            if (B >= 480) { launch_rm<R4<1>>(a, B, s, 1); return true; }
            return false;
        case 5:
        case 6:
            if (B >= 513) { launch_rm<R5<2>>(a, B, s, 1); return true; }
            return false;
        default: return false;
    }
}

}  // namespace syn
