// Row-marching network head for uint8 crops: features.0 (3x3 s2 conv 3->32 + BN + ReLU6) and features.1 (depthwise 3x3 + BN +
// ReLU6, linear 1x1 32->16 + BN) with the 60x60x32 stem output living in registers only.
// Reference: backbone_nets/mobilenetv2_backbone.py:129 (features.0), :58-66 (t = 1 block), synergy3DMM.py:189-192 (the
// HWC->CHW permute and (x-127.5)/128 are folded in).  Same dataflow as fused_block_rm.hip:
//
//   * a compute wave owns one 30-column half of one face (32 lanes = 30 output columns + one neighbour column each side: the
//     two halves overlap by two stem columns instead of exchanging them) and marches down the 60 rows;
//   * the stem convolution is an im2col GEMM on v_mfma_f32_32x32x16_f16: A = the 32 stem channels x 27 taps (two k16 steps),
//     B = the raw pixel bytes as fp16 -- every integer 0..255 IS an fp16 number, so only the filter is carried as two fp16
//     pieces (22 bits, scaled by a power of two; 2 MFMAs per step).  The normalisation is folded: (2p-255)/256 = p/128 - 255/256, i.e. the packed filter is w/128
//     (exact) and the accumulator starts at shift - 255/256 * sum(w); zero padding (0 in normalised space) is the raw value
//     127.5, also exact in fp16.  The LDS row ring holds a pixel as FOUR fp16 (R, G, B, -), so a lane's window of three pixels
//     starts on a 16-byte boundary and is read by aligned ds_read_b128 / b64 (round 2-3: packed RGB, a lane's nine taps started at
//     an odd fp16 and the compiler's merged 16-byte reads were misaligned -- 20 of the kernel's 130 us, tools: the `ral` variant
//     of DESIGN 7).  K is then 9 pixel quads of 4 slots (the fourth has zero weight): lane half 0 carries kernel row 0 and the
//     first two pixels of row 1, half 1 kernel row 2 and the last pixel of row 1 -- three k16 steps (the third half empty);
//   * depthwise 3x3 scattered into three row accumulators, neighbours through wave_shr / wave_shl, filter from LDS
//     (broadcast reads); ReLU6 -> in-place fp16 x2 split = B operand of the 32->16 projection (rows 16..31 of its A tile are
//     zero), BN shift, NHWC store straight from the compute wave -- one hidden group, so there is no partial-sum exchange;
//   * service waves keep the image rows of all units flowing: buffer loads one stage ahead (counted vmcnt, no branches) -> fp16 -> LDS
//     row ring (8 slots per unit), one barrier per output row.  Four of them since round 5, one per SIMD, a quarter of a stage each
//     (StemRmCfg::NSV: a single one made SIMD 0 the slowest SIMD of every step).
// fp32 crops (forward_test): the F32 instantiation below (round 4).  Small batches: row bands of a face from 112 faces on (StemRmCfg::NBD),
// the tiled kernel of stem_block1.hip below that (and for fp32 crops below 480).
#include "syn_internal.h"

#include <cstdlib>

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
// two floats -> packed fp16 pieces a (high) and b (low), x = a + b to 22 significant bits
__device__ __forceinline__ void split2s(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ f32x16 mfma32s(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float left_of(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float right_of(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
// ReLU6 without an instruction of its own (tools/ubench/valu_issue.hip: v_med3_f32 costs 1.85 ns of a SIMD, a plain fp32 multiply or FMA
// 1.25): activations are carried as relu6(x) / 6 in [0, 1], which is what the `clamp` output modifier produces -- on the multiply that
// rescales the expand accumulator, and on the LAST FMA of a depthwise accumulator; the 6 rides on constants that are applied anyway.
// (written as med3(x, 0, 1) of the product: the compiler folds that into the producer's clamp bit -- and, unlike for inline assembly,
// keeps the wait states a vector instruction needs behind the matrix instruction whose result it reads)
__device__ __forceinline__ float mul_clamp01(float a, float b) { return __builtin_amdgcn_fmed3f(a * b, 0.0f, 1.0f); }
__device__ __forceinline__ float fma_clamp01(float a, float b, float c) { return __builtin_amdgcn_fmed3f(__builtin_fmaf(a, b, c), 0.0f, 1.0f); }
constexpr int kImgW = 120, kHid = 60;
constexpr int kRowDw = 244;               // dwords per image-row slot: 122 pixels x (R, G, B, -) fp16; pixel t = image column t - 1 (t = 0: left padding)
constexpr int kSlots = 8;                 // image-row ring per unit
constexpr unsigned kPadF16 = 0x57F8u;     // 127.5 as fp16: the raw value of a zero in normalised space
}  // namespace

// F32 (round 4): normalised fp32 NCHW crops (forward_test, reference synergy3DMM.py:151-154) through the same march.  Arbitrary floats
// need BOTH operands as two fp16 pieces (three products per k16 step instead of two): the service wave splits every value once and the
// ring holds two planes per image row, high pieces | low pieces, in the same (R, G, B, -) pixel layout; the filter is the plain BN-folded
// one (nothing to fold the normalisation into) and the padding value is 0.  Domain: |x| < 6e4 (the reference's own normalisation gives
// [-1, 1]); replaces the spatially tiled bf16 x3 kernel of stem_block1.hip for batches that fill the chip.
// NBD > 1 (small batches, uint8 crops): a workgroup round is ONE BAND of HB output rows of its U faces, so that B faces give NBD x B / U
// rounds to spread over the chip (fused_block_rm.hip RmCfg::NBD is the same idea).  A band marches the stem rows r0 - 1 .. r0 + HB (HB + 2
// steps, padded to the accumulator ring's multiple of 3); stem rows above / below the map contribute zeros (the ReLU6 multiplier of the
// step is 0, as on the padding columns).  Step l reads image rows 2 (r0 - 1 + l) - 1 .. + 1 = local stages l and l + 1 (stage j = image rows
// 2 (r0 - 2 + j), + 1), so the service wave runs TWO stages ahead of the compute waves: BSTG = BSTEPS + 2 stages and barriers per round on
// both sides.
template <int U_, int WPE_, bool F32_ = false, int NBD_ = 1>
struct StemRmCfg {
    static constexpr int U = U_, WPE = WPE_;                                  // faces (units) per workgroup
    static constexpr bool F32 = F32_;
    static constexpr int NBD = NBD_, HB = 60 / NBD;
    static constexpr int BSTEPS = (HB + 2 + 2) / 3 * 3, BSTG = BSTEPS + 2;
    static_assert(60 % NBD == 0 && (NBD == 1 || (!F32_ && BSTG % 2 == 0)), "bands: equal heights, uint8 crops, stages in pairs");
    // service waves.  Round 5: FOUR, one per SIMD (wave ids NCW .. NCW + 3 follow the 2 U compute waves cyclically), each converting a quarter of a
    // stage.  One service wave (two for F32) sat on SIMD 0 (and 1) beside that SIMD's two compute waves: ~600 of its ~3600 issue cycles per row
    // step against ~3000 on the other SIMDs, and every step ends at a workgroup barrier -- the whole chip ran at SIMD 0's pace.
#ifndef SYN_STEM_NSV
#define SYN_STEM_NSV 4
#endif
    static constexpr int NSV = SYN_STEM_NSV > 0 ? SYN_STEM_NSV : (F32 ? 2 : 1);
    static constexpr int NCW = 2 * U, NT = (NCW + NSV) * 64;       // compute waves (face, half) + the service wave(s)
    static constexpr int SLOT_DW = (F32 ? 2 : 1) * kRowDw;         // one image row: plane of high pieces | (F32) plane of low pieces
    static constexpr int UNIT_DW = (kSlots + 1) * SLOT_DW;         // ring + one all-padding row (image row -1)
    static constexpr int LDS_DW = U * UNIT_DW + 10 * 32 + 32 + 32; // + depthwise filter 9x32 | depthwise shift | stem shift | project shift
    static_assert(NT <= 1024 && LDS_DW * 4 <= 160 * 1024, "workgroup size / LDS budget");
};

template <class C>
__global__ __launch_bounds__(C::NT) __attribute__((amdgpu_waves_per_eu(C::WPE, 3)))
void stem_rm_kernel(const uint8_t *__restrict__ img /*uint8 [B,120,120,3] | F32: float [B,3,120,120]*/, const unsigned *__restrict__ As3 /*[3][2][64][4]*/,
                    const float *__restrict__ s_shift /*[32] folded, then {S, 1/S, 6 S} of the stem filter*/, const float *__restrict__ Wd /*[9][32] scaled*/,
                    const float *__restrict__ d_shift, const unsigned *__restrict__ Ap3 /*[1][2][2][64][4]*/,
                    const float *__restrict__ p_shift /*[16]*/, const float *__restrict__ scl_p, float *__restrict__ Y /*[B,60,60,16]*/, int B) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = wave_wg >= C::NCW;
    float *Filt = reinterpret_cast<float *>(smem + C::U * C::UNIT_DW);       // [9][32] + row 9 = depthwise shift
    float *Ssh = Filt + 10 * 32, *Psh = Ssh + 32;
    constexpr int DSH = 9 * 32;
    // power-of-two scales of the fp16 weight pieces: stem accumulators start at Ss x shift, ReLU6 clamps at 6 Ss, the depthwise filter
    // carries 1 / Ss; the projection starts at Sp x shift and is rescaled before the store
    // stem output e' = relu6(.) / 6 = clamp(acc / (6 Ss)); depthwise d' = relu6(d) / 6 = clamp(sum e' Wd + shift / 6): the filter is the
    // plain one; the projection contracts d' and its accumulator (started at Sp shift / 6) is rescaled by 6 / Sp before the store
    const float Ss = s_shift[32], c6s = s_shift[34], Sp = scl_p[0], inv_sp6 = 6.0f * scl_p[1];
    const float inv_c6s = 1.0f / c6s;
    for (int i = tid; i < 9 * 32; i += C::NT) Filt[i] = Wd[i];
    if (tid < 32) { Filt[DSH + tid] = d_shift[tid] * (1.0f / 6.0f); Ssh[tid] = s_shift[tid] * Ss; Psh[tid] = tid < 16 ? p_shift[tid] * Sp * (1.0f / 6.0f) : 0.f; }
    // padding row (image row -1), the left padding pixel, the fourth element of every pixel and the tails of every ring slot: 127.5
    // everywhere, then the service wave only ever rewrites pixels 1 .. 120
    for (int i = tid; i < C::U * C::UNIT_DW; i += C::NT) smem[i] = C::F32 ? 0u : kPadF16 | (kPadF16 << 16);
    __syncthreads();

    if (service) {
        __builtin_amdgcn_s_setprio(3);          // every compute wave waits for this wave at the row barrier
        // ---- image rows -> fp16 -> ring: row iy of unit u lands in slot iy & 7 ----
        // A "stage" = image rows 2k, 2k+1 of the U faces of a group (k = 0 .. 59; stage k is what compute step k adds).  The
        // loads of stage k+1 are ISSUED before stage k is converted, so a global round trip has a whole row step to land instead
        // of standing in every step (round 3: 12 conditional loads, then s_waitcnt vmcnt(0) -- the load latency WAS the step).
        // For the compiler to emit counted waits the loop body has no vector-memory branches: buffer loads whose out-of-range lanes
        // (faces past the batch, the surplus lanes of the last 64-lane round) return zeros, and the slot offsets are immediates
        // (four stages per loop iteration).
        if constexpr (C::F32) {
            // a work item = one pixel: R, G, B from the three planes of the NCHW crop (dword loads, 256 contiguous bytes per instruction);
            // lane stride 8 bytes in the row slot -> conflict-free ds_write_b64 (see the uint8 service wave below)
            constexpr int TOTAL = C::U * 2 * kImgW, ITER = (TOTAL + 64 * C::NSV - 1) / (64 * C::NSV);
            constexpr unsigned ROW_B = kImgW * 4, PLANE_B = kImgW * ROW_B, FACE_B = 3 * PLANE_B;
            const int sv = wave_wg - C::NCW;                               // the service waves deal the items 64 at a time
            unsigned gofs[ITER], lofs[ITER];
            bool live[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int i = lane + 64 * (it * C::NSV + sv);
                const int u = i / (2 * kImgW), r = (i / kImgW) % 2, px = i % kImgW;
                live[it] = i < TOTAL;
                gofs[it] = live[it] ? (unsigned)(u * FACE_B + r * ROW_B + 4 * px) : 0x80000000u;
                lofs[it] = (unsigned)(u * C::UNIT_DW + r * C::SLOT_DW + 2 + 2 * px);       // pixel t = 1 + px of the row slot
            }
            int ifb = blockIdx.x * C::U, ik = 0;
            struct Px4 { float c[3]; };
            auto issue = [&](Px4 (&v)[ITER]) {
                // records = the group's faces inside the batch (the range check is on the lane offset, which names the face; plane and row ride
                // in the scalar offset and stay inside that face)
                long long left = ((long long)B - ifb) * (long long)FACE_B;
                if (left > (long long)(C::U * FACE_B)) left = C::U * FACE_B;
                const int nrec = left > 0 ? (int)left : 0;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img) + (size_t)ifb * FACE_B, 0, nrec, 0x00027000);
#pragma unroll
                for (int it = 0; it < ITER; ++it)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)
                        v[it].c[ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, gofs[it], ch * PLANE_B + ik * 2 * ROW_B, 0));
                if (++ik == kHid) { ik = 0; ifb += gridDim.x * C::U; }
            };
            auto consume = [&](const Px4 (&v)[ITER], int slot) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    u32x2 *dst = reinterpret_cast<u32x2 *>(smem + lofs[it] + slot * C::SLOT_DW);
                    unsigned a_rg, b_rg, a_b, b_b;
                    split2s(v[it].c[0], v[it].c[1], a_rg, b_rg);
                    split2s(v[it].c[2], 0.0f, a_b, b_b);
                    if (live[it]) { dst[0] = (u32x2){a_rg, a_b}; dst[kRowDw / 2] = (u32x2){b_rg, b_b}; }
                }
            };
            Px4 va[ITER], vb[ITER];
            issue(va);
            for (int fb = blockIdx.x * C::U; fb < B; fb += gridDim.x * C::U) {
                for (int k = 0; k < kHid; k += 4) {
                    issue(vb); __builtin_amdgcn_sched_barrier(0); consume(va, 0); __syncthreads();
                    issue(va); __builtin_amdgcn_sched_barrier(0); consume(vb, 2); __syncthreads();
                    issue(vb); __builtin_amdgcn_sched_barrier(0); consume(va, 4); __syncthreads();
                    issue(va); __builtin_amdgcn_sched_barrier(0); consume(vb, 6); __syncthreads();
                }
                __syncthreads();
            }
            return;
        } else {
        // a work item = ONE pixel: lane stride 8 bytes in the row slot, so the 64 ds_write_b64 of an instruction fall into distinct banks
        // (items of eight pixels, 64 bytes apart, put sixteen lanes on every bank: the service wave's stores alone kept the LDS busy a
        // fifth of a row step).  A pixel's three bytes start anywhere in a dword: the dword that holds the first byte and the next one
        // (two loads: the second one of the very last pixel of a batch is out of range and reads 0), v_alignbyte by the lane's own shift.
        constexpr int TOTAL = C::U * 2 * kImgW, ITER = (TOTAL + 64 * C::NSV - 1) / (64 * C::NSV);
        constexpr unsigned FACE_B = kImgW * kImgW * 3, ROW2_B = 2 * kImgW * 3;
        const int sv = wave_wg - C::NCW;                                   // the service waves deal the items 64 at a time
        unsigned gofs[ITER], lofs[ITER];
        bool live[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = lane + 64 * (it * C::NSV + sv);
            const int u = i / (2 * kImgW), r = (i / kImgW) % 2, px = i % kImgW;
            live[it] = i < TOTAL;
            gofs[it] = live[it] ? (unsigned)(u * FACE_B + r * (kImgW * 3) + ((3 * px) & ~3)) : 0x80000000u;
            lofs[it] = (unsigned)(u * C::UNIT_DW + r * C::SLOT_DW + 2 + 2 * px);       // pixel t = 1 + px of the row slot
        }
        const unsigned sft = (3u * (lane & 3)) & 3u;               // (3 px) & 3 with px = (lane + 64 it) % 120: 64 and 120 are multiples of 4
        int ifb = blockIdx.x * C::U, ik = 0;                       // the stage `issue` requests next
        struct Px8 { unsigned lo, hi; };
        auto issue = [&](Px8 (&v)[ITER]) {
            // base = row 2 ik of face ifb; records = what is left of the group's faces inside the batch (<= 0: everything reads as zero)
            long long left = ((long long)B - ifb) * (long long)FACE_B;
            if (left > (long long)(C::U * FACE_B)) left = C::U * FACE_B;
            left -= (long long)ik * ROW2_B;
            const int nrec = left > 0 ? (int)left : 0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t *>(img) + ((size_t)ifb * FACE_B + (size_t)ik * ROW2_B), 0, nrec, 0x00027000);
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                v[it].lo = __builtin_amdgcn_raw_buffer_load_b32(rs, gofs[it], 0, 0);
                v[it].hi = __builtin_amdgcn_raw_buffer_load_b32(rs, gofs[it] + 4, 0, 0);
            }
            if (++ik == kHid) { ik = 0; ifb += gridDim.x * C::U; }
        };
        auto consume = [&](const Px8 (&v)[ITER], int slot /*compile-time after unrolling*/) {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                // bytes (R, G, B) -> (R, G), (B, 0) as fp16, exactly: 0x64pp is the fp16 number 1024 + p (ulp 1 there); minus 1024 leaves p
                const unsigned T = __builtin_amdgcn_alignbyte(v[it].hi, v[it].lo, sft);
                const unsigned rg = __builtin_amdgcn_perm(0x64646464u, T, 0x04010400u);
                const unsigned bx = __builtin_amdgcn_perm(0x64646464u, T, 0x040c0402u);
                const f16x2 k1024 = {(_Float16)1024.0f, (_Float16)1024.0f};
                u32x2 o;
                o[0] = __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, rg) - k1024);
                o[1] = __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, bx) - k1024);
                if (live[it]) *reinterpret_cast<u32x2 *>(smem + lofs[it] + slot * C::SLOT_DW) = o;
            }
        };
        Px8 va[ITER], vb[ITER];
        if constexpr (C::NBD > 1) {
            const int n_rounds = (B + C::U - 1) / C::U * C::NBD;
            int iw = blockIdx.x, ij = 0;                       // round / stage `issue_b` requests next
            auto issue_b = [&](Px8 (&v)[ITER]) {
                const int fg = iw / C::NBD, band = iw - fg * C::NBD, fb0 = fg * C::U;
                const int kk = band * C::HB - 2 + ij;          // global stage = image rows 2 kk, 2 kk + 1 (outside the image: nothing is read, zeros arrive)
                const int kp = kk < 0 ? 0 : kk;
                long long left = ((long long)B - fb0) * (long long)FACE_B;
                if (left > (long long)(C::U * FACE_B)) left = C::U * FACE_B;
                left -= (long long)kp * ROW2_B;
                const int nrec = (left > 0 && kk >= 0 && iw < n_rounds) ? (int)left : 0;
                const size_t base = iw < n_rounds ? (size_t)fb0 * FACE_B + (size_t)kp * ROW2_B : 0;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img) + base, 0, nrec, 0x00027000);
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    v[it].lo = __builtin_amdgcn_raw_buffer_load_b32(rs, gofs[it], 0, 0);
                    v[it].hi = __builtin_amdgcn_raw_buffer_load_b32(rs, gofs[it] + 4, 0, 0);
                }
                if (++ij == C::BSTG) { ij = 0; iw += gridDim.x; }
            };
            auto consume_b = [&](const Px8 (&v)[ITER], int slot) {
                unsigned *ringb = smem + slot * C::SLOT_DW;
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const unsigned T = __builtin_amdgcn_alignbyte(v[it].hi, v[it].lo, sft);
                    const unsigned rg = __builtin_amdgcn_perm(0x64646464u, T, 0x04010400u);
                    const unsigned bx = __builtin_amdgcn_perm(0x64646464u, T, 0x040c0402u);
                    const f16x2 k1024 = {(_Float16)1024.0f, (_Float16)1024.0f};
                    u32x2 o;
                    o[0] = __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, rg) - k1024);
                    o[1] = __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, bx) - k1024);
                    if (live[it]) *reinterpret_cast<u32x2 *>(ringb + lofs[it]) = o;
                }
            };
            issue_b(va);
            for (int w = blockIdx.x; w < n_rounds; w += gridDim.x)
                for (int j = 0; j < C::BSTG; j += 2) {         // local stage j -> ring slots (2 j) & 7, + 1
                    issue_b(vb); __builtin_amdgcn_sched_barrier(0); consume_b(va, (2 * j) & 7); __syncthreads();
                    issue_b(va); __builtin_amdgcn_sched_barrier(0); consume_b(vb, (2 * j + 2) & 7); __syncthreads();
                }
            return;
        }
        issue(va);
        for (int fb = blockIdx.x * C::U; fb < B; fb += gridDim.x * C::U) {
            for (int k = 0; k < kHid; k += 4) {            // barrier (P), then the barriers that end compute steps 0 .. 58
                issue(vb); __builtin_amdgcn_sched_barrier(0); consume(va, 0); __syncthreads();
                issue(va); __builtin_amdgcn_sched_barrier(0); consume(vb, 2); __syncthreads();
                issue(vb); __builtin_amdgcn_sched_barrier(0); consume(va, 4); __syncthreads();
                issue(va); __builtin_amdgcn_sched_barrier(0); consume(vb, 6); __syncthreads();   // k + 3 == 59: the next group's stage 0
            }
            __syncthreads();                                   // ends compute step 59
        }
        return;
        }
    }

    // ---- compute wave: half c of face (fb + uw) ----
    const int uw = wave_wg >> 1, c = wave_wg & 1;
    const int j = lane & 31, h = lane >> 5;
    const int hc = 30 * c + j - 1;                                 // stem / hidden column of this lane
    const bool col_ok = (unsigned)hc < (unsigned)kHid;
    const bool out_lane = j >= 1 && j <= 30;
    const unsigned *ring = smem + uw * C::UNIT_DW;
    // dword offset of this lane's window inside a row slot: pixels t = 2 hc, 2 hc + 1, 2 hc + 2 (image columns 2 hc - 1 ..) = bytes
    // 16 hc .. 16 hc + 23 (clamped for the out-of-image lanes, whose result is forced to zero anyway)
    const int run0 = 4 * (col_ok ? hc : 0);
    u32x4 as[3][2], ap[2][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) as[s][p] = *(const u32x4 *)(As3 + ((size_t)s * 2 + p) * 256 + lane * 4);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) ap[s][p] = *(const u32x4 *)(Ap3 + ((size_t)s * 2 + p) * 256 + lane * 4);
    const int cb = 4 * h;
    auto opaque_cb = [&]() { int v = cb; asm volatile("" : "+v"(v)); return v; };

    const int n_rounds = C::NBD > 1 ? (B + C::U - 1) / C::U * C::NBD : 0;
    for (int fbw = C::NBD > 1 ? blockIdx.x : blockIdx.x * C::U; fbw < (C::NBD > 1 ? n_rounds : B); fbw += C::NBD > 1 ? gridDim.x : gridDim.x * C::U) {
        const int fb = C::NBD > 1 ? (fbw / C::NBD) * C::U : fbw;                        // first face of the round
        const int band_r0 = C::NBD > 1 ? (fbw - (fbw / C::NBD) * C::NBD) * C::HB : 0;   // first output row of the band
        const int f = fb + uw;
        const float emul_f = (col_ok && f < B) ? inv_c6s : 0.0f;      // 0 on the out-of-image lanes: the depthwise zero padding
        float emul = emul_f;                                          // ... of the row being computed (bands: 0 above / below the map)
        const bool st_ok = out_lane && f < B;
        const int yofs = (30 * c + j - 1) * 16 + 4 * h;             // (column, channel quad) inside an output row; < 2^31 elements per face

        auto finalize = [&](f32x16 &d, int oy, bool clamped = true) {
            const int cbo = opaque_cb();
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sh = *(const f32x4 *)&Psh[(cbo + 8 * q) & 31];          // rows >= 16 are padding (never stored)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[4 * q + t] = q < 2 ? sh[t] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 db[2];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // (clamped by the FMA that completed the row; the last image row has no row below it to do so)
                    const float v0 = clamped ? d[8 * s + 2 * t] : __builtin_amdgcn_fmed3f(d[8 * s + 2 * t], 0.0f, 1.0f);
                    const float v1 = clamped ? d[8 * s + 2 * t + 1] : __builtin_amdgcn_fmed3f(d[8 * s + 2 * t + 1], 0.0f, 1.0f);
                    unsigned ha, hb;
                    split2s(v0, v1, ha, hb);
                    db[0][t] = ha; db[1][t] = hb;
                }
                acc = mfma32s(ap[s][1], db[0], acc);
                acc = mfma32s(ap[s][0], db[1], acc);
                acc = mfma32s(ap[s][0], db[0], acc);
            }
            if (st_ok) {
                float *dst = Y + ((size_t)f * kHid + oy) * kHid * 16 + yofs;
                *(f32x4 *)dst = (f32x4){acc[0], acc[1], acc[2], acc[3]} * inv_sp6;             // channels 4h .. 4h+3
                *(f32x4 *)(dst + 8) = (f32x4){acc[4], acc[5], acc[6], acc[7]} * inv_sp6;       // channels 8 + 4h ..
            }
        };
        auto taps3 = [&](f32x16 &d, int q, const float *wq, int ky, const f32x4 &l4, const f32x4 &c4, const f32x4 &r4, bool init, bool last = false) {
            const f32x4 w0 = *(const f32x4 *)(wq + (3 * ky + 0) * 32), w1 = *(const f32x4 *)(wq + (3 * ky + 1) * 32), w2 = *(const f32x4 *)(wq + (3 * ky + 2) * 32);
            f32x4 base;
            if (init) base = *(const f32x4 *)(wq + DSH);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b0 = init ? base[t] : d[4 * q + t];
                const float p2 = __builtin_fmaf(c4[t], w1[t], __builtin_fmaf(l4[t], w0[t], b0));
                d[4 * q + t] = last ? fma_clamp01(r4[t], w2[t], p2) : __builtin_fmaf(r4[t], w2[t], p2);     // last: the row is complete -> ReLU6 (as [0, 1])
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d[4 * q + t]));
        };

        f32x16 d0, d1, d2;
        {
            const int cbo = opaque_cb();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sh = *(const f32x4 *)(Filt + cbo + 8 * q + DSH);
#pragma unroll
                for (int t = 0; t < 4; ++t) { d0[4 * q + t] = sh[t]; d1[4 * q + t] = 0.f; d2[4 * q + t] = 0.f; }
            }
        }
        if (C::NBD > 1) __syncthreads();                       // (bands: the service wave is two stages ahead)
        __syncthreads();                                       // (P)

        // stem row hy -> kernel row 2 of output row hy-1 (dm), row 1 of hy (dc), row 0 of hy+1 (dn);  sl0: ring slot of image row 2hy-1
        // (2hy, 2hy+1 follow it); fin: output row hy-1 is complete and wanted
        auto step = [&](int hy, int sl0, bool fin, f32x16 &dm, f32x16 &dc, f32x16 &dn) {
            const int cbo = opaque_cb();
            // ---- im2col B operand: image rows 2hy-1 (kernel row 0), 2hy, 2hy+1 from the ring; row -1 = the padding row ----
            const unsigned *r0 = ring + (hy == 0 ? kSlots : (sl0 & (kSlots - 1))) * C::SLOT_DW + run0;
            const unsigned *r1 = ring + ((sl0 + 1) & (kSlots - 1)) * C::SLOT_DW + run0;
            const unsigned *r2 = ring + ((sl0 + 2) & (kSlots - 1)) * C::SLOT_DW + run0;
            // lane half 0: kernel row 0 (three pixels) + pixels 0, 1 of kernel row 1; half 1: kernel row 2 + pixel 2 of row 1 (and one
            // pixel past the window under zero weights).  K slots: step 0 = pixels 0, 1 of pa, step 1 = pixel 2 of pa | first pixel of
            // pb, step 2 = second pixel of pb | (zero weights)
            const unsigned *pa = h ? r2 : r0, *pb = h ? r1 + 4 : r1;
            u32x4 xb[C::F32 ? 2 : 1][3];
#pragma unroll
            for (int pl = 0; pl < (C::F32 ? 2 : 1); ++pl) {
                const u32x4 wa = *(const u32x4 *)(pa + pl * kRowDw), wb = *(const u32x4 *)(pb + pl * kRowDw);
                const u32x2 wc = *(const u32x2 *)(pa + pl * kRowDw + 4);
                xb[pl][0] = wa;
                xb[pl][1] = (u32x4){wc[0], wc[1], wb[0], wb[1]};
                xb[pl][2] = (u32x4){wb[2], wb[3], wb[2], wb[3]};
            }
            f32x16 e;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sh = *(const f32x4 *)&Ssh[cbo + 8 * q];
#pragma unroll
                for (int t = 0; t < 4; ++t) e[4 * q + t] = sh[t];
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                e = mfma32s(as[s][1], xb[0][s], e);
                if (C::F32) e = mfma32s(as[s][0], xb[1][s], e);
                e = mfma32s(as[s][0], xb[0][s], e);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = mul_clamp01(e[r], emul);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 c4, l4, r4;
#pragma unroll
                for (int t = 0; t < 4; ++t) { c4[t] = e[4 * q + t]; l4[t] = left_of(c4[t]); r4[t] = right_of(c4[t]); }
                const float *wq = Filt + cbo + 8 * q;
                taps3(dn, q, wq, 0, l4, c4, r4, true);
                taps3(dc, q, wq, 1, l4, c4, r4, false);
                taps3(dm, q, wq, 2, l4, c4, r4, false, true);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (fin) finalize(dm, hy - 1);
            __syncthreads();
        };
        if (C::NBD > 1) {
            // band: step l computes stem row r0 - 1 + l from local stages l (its odd image row) and l + 1 and completes output row r0 + l - 2;
            // the map's last row is completed (and clamped) by the all-zero stem row 60
            auto bstep = [&](int l, f32x16 &dm, f32x16 &dc, f32x16 &dn) {
                const int hy = band_r0 - 1 + l;
                emul = (unsigned)hy < (unsigned)kHid ? emul_f : 0.0f;
                step(hy, 2 * l + 1, l >= 2 && l < C::HB + 2, dm, dc, dn);
            };
            for (int l = 0; l < C::BSTEPS; l += 3) {
                bstep(l, d2, d0, d1);
                bstep(l + 1, d0, d1, d2);
                bstep(l + 2, d1, d2, d0);
            }
        } else {
            for (int hy = 0; hy < kHid; hy += 3) {
                step(hy, 2 * hy - 1, hy >= 1, d2, d0, d1);
                step(hy + 1, 2 * hy + 1, true, d0, d1, d2);
                step(hy + 2, 2 * hy + 3, true, d1, d2, d0);
            }
            finalize(d2, kHid - 1, false);       // (60 - 1) % 3 == 2
        }
    }
}

template <class C>
static void launch_stem_cfg(const void *img, const unsigned *As3, const float *s_shift, const float *Wd, const float *d_shift,
                            const unsigned *Ap3, const float *p_shift, const float *scl_p, float *Y, int B, hipStream_t s) {
    const int wgs = (B + C::U - 1) / C::U * C::NBD;
    const int grid = wgs < 256 ? wgs : 256;
    stem_rm_kernel<C><<<grid, C::NT, 0, s>>>(static_cast<const uint8_t *>(img), As3, s_shift, Wd, d_shift, Ap3, p_shift, scl_p, Y, B);
}

// like fused_block_rm.hip: persistent over faces, so small batches take fewer faces per workgroup and, below the last
// threshold, the spatially tiled kernel (stem_block1.hip)
constexpr int kStemMin4 = 513;      // (two faces per workgroup need a second round of workgroups from here on: B = 640 163 -> 122 us)
constexpr int kStemMin2 = 480;
constexpr int kStemBandMin = 112;   // row bands of a face below kStemMin2 (StemRmCfg::NBD)

bool launch_stem_rm(const uint8_t *img8, const unsigned *As3, const float *s_shift, const float *Wd, const float *d_shift,
                    const unsigned *Ap3, const float *p_shift, const float *scl_p, float *Y, int B, hipStream_t s) {
    if (!img8 || !As3 || !Ap3 || !scl_p) return false;
    if (reinterpret_cast<uintptr_t>(img8) & 7) return false;       // the service wave fetches eight-byte pieces of the image rows
    if (B >= kStemMin4) { launch_stem_cfg<StemRmCfg<4, 2>>(img8, As3, s_shift, Wd, d_shift, Ap3, p_shift, scl_p, Y, B, s); return true; }
    if (B >= kStemMin2) { launch_stem_cfg<StemRmCfg<2, 2>>(img8, As3, s_shift, Wd, d_shift, Ap3, p_shift, scl_p, Y, B, s); return true; }
    // six bands of ten output rows, four faces per workgroup (12 steps + 2 lead stages per round; step time of configs[1] - tiled stem, us:
    // B = 64 +6, 96 -1, 128 -6, 256 -12, 400 -20; four bands, or two faces per workgroup: +3 ... -3 at B = 128; tools/band_ab.sh)
    static const bool bands = test_knob("stem_band", 1) != 0;
    if (bands && B >= kStemBandMin) { launch_stem_cfg<StemRmCfg<4, 2, false, 6>>(img8, As3, s_shift, Wd, d_shift, Ap3, p_shift, scl_p, Y, B, s); return true; }
    return false;
}

// fp32 NCHW crops (As3 / s_shift: the second constant set of rm_stem_dwords, filter not divided by 128, plain BN shift)
bool launch_stem_rm_f32(const float *img, const unsigned *As3, const float *s_shift, const float *Wd, const float *d_shift,
                        const unsigned *Ap3, const float *p_shift, const float *scl_p, float *Y, int B, hipStream_t s) {
    if (!img || !As3 || !Ap3 || !scl_p) return false;
    if (reinterpret_cast<uintptr_t>(img) & 15) return false;       // sixteen-byte loads of four pixels of a plane
    if (B >= kStemMin4) { launch_stem_cfg<StemRmCfg<4, 2, true>>(img, As3, s_shift, Wd, d_shift, Ap3, p_shift, scl_p, Y, B, s); return true; }
    if (B >= kStemMin2) { launch_stem_cfg<StemRmCfg<2, 2, true>>(img, As3, s_shift, Wd, d_shift, Ap3, p_shift, scl_p, Y, B, s); return true; }
    return false;
}

}  // namespace syn
