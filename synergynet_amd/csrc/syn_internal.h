// Internal launcher interface between the C-ABI host code (synergy_abi.hip) and the
// gfx950 kernels (backbone_kernels.hip, recon_kernels.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace syn {

#ifdef __HIPCC__
// L2 warm-up ("touch") of a weight run a kernel is going to stream later (round 5).  The kernels of features.7-17 and the head walk 1.6-4 MB of
// weight fragments group by group, fetched ONE group (~1.5 us) ahead; every forward finds them cold (1.4 GB of activations went through
// the 4 MB L2 of each XCD since the last one), so the first CU of an XCD to reach a group pays a miss into the Infinity Cache / HBM of
// about that long, and the CUs of an XCD -- started together -- all wait for the same line fills: timing-only builds of the features.15-17
// chain without its weight fetch ran 21 of 155 us faster (gpurun_out/r5c2).  Here every thread of the launch issues one dword load per
// 128-byte line of its share of the run, long before the run is needed: the misses overlap each other instead of standing in 90 groups.
//   gi / nth: this thread's index among, and the number of, the threads of the launch that run on ITS XCD (workgroup b runs on XCD b % 8).
//   sink: ONE register that receives every touched dword (never read).  It must stay allocated until the loads have returned, which is
//   what l2_touch_done() at the end of the kernel is for.  The loads are inline assembly on purpose: the compiler's wait-count
//   bookkeeping does not know them, so no later s_waitcnt is computed FOR them; vector memory returns in order, hence every wait the
//   compiler inserts for a younger load of its own still covers them (waits can only be longer than needed, never too short).
__device__ __forceinline__ void l2_touch(const void *base, unsigned bytes, unsigned gi, unsigned nth, unsigned &sink) {
    for (unsigned ofs = gi * 128u; ofs < bytes; ofs += nth * 128u) {
        const char *p = static_cast<const char *>(base) + ofs;
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
    }
}
__device__ __forceinline__ void l2_touch_done(unsigned &sink) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) :: "memory"); }
#endif

// Test / A-B knobs of the launchers (never needed in production: every default is the measured best).  ONE environment variable,
//   SYNERGY_HIP_TEST_KNOBS = "name=value,name=value,..."   (integers; 0x.. hexadecimal accepted)
// read once per process (synergy_abi.hip); rounds 2-5 had 18 separate SYN_* variables.  Names: lb_chain, lb4_chain, small_f7, head_sliced_in,
// head_wide_min, f16_pair56, rm_pair56, rm_pair34, rm_band2, rm_band3, stem_band, ablate_stem, recon_no_pk, recon_pk_wpg, recon_wgs, recon_prof,
// lt_stage, lt_glds, resnet_exact_mask.
long long test_knob(const char *name, long long dflt);
bool test_knob_set(const char *name);

constexpr int kImg = 120;           // utils/params.py:34
constexpr int kParam = 62;
constexpr int kPool = 1280;
constexpr int kBasisK = 52;         // 40 shape + 10 expression + 1 (mean u, alpha = 1) + 1 zero pad
constexpr int kVertTile = 32;        // vertices per reconstruction tile (one 32x32 MFMA column block)

// ---- backbone ---------------------------------------------------------------------
// Activations are NHWC fp32: act[b][y][x][c].

// 3x3 stride-2 stem conv + BN scale/shift + ReLU6: NCHW fp32 (or HWC uint8) -> NHWC [B,60,60,32].
void launch_stem(const float *img_nchw, const uint8_t *img_hwc_u8, const float *w27x32,
                 const float *scale, const float *shift, float *out, int B, hipStream_t s);

// Pointwise 1x1 conv as an fp32-MFMA GEMM with fused BN scale/shift (+ReLU6) (+residual):
//   C[m][n] = epi(sum_k A[m][k] * W[n][k]),  A [M,K], W [Npad,Kpad] zero padded, C [M,N].
void launch_pointwise(const float *A, const float *W, const float *scale, const float *shift,
                      const float *residual, float *C, int M, int K, int Kpad, int N,
                      int relu6, hipStream_t s);

// Depthwise 3x3 (pad 1, stride 1|2) + BN scale/shift + ReLU6, NHWC.
void launch_depthwise(const float *in, const float *w9xC, const float *scale, const float *shift,
                      float *out, int B, int Hin, int Hout, int C, int stride, hipStream_t s);

// Global average pool over 4x4 + the three linear heads (one [62,1280] GEMV per face).
void launch_pool_fc(const float *feat /*[B,16,1280]*/, const float *Wfc /*[64,1280]*/,
                    const float *bias /*[64]*/, float *param /*[B,62]*/, float *pool /*nullable*/,
                    int B, hipStream_t s);

// Fused inverted-residual block (expand 1x1 -> dw 3x3 -> project 1x1 [+ residual]) for
// .features[2..17]; weights in MFMA lane order (see fused_block.hip).  Returns false if `feature`
// has no fused configuration.
struct FusedBlockArgs {
    const float *X;                                   // block input  NHWC
    const float *We, *e_scale, *e_shift;              // expand: Wpk[HID/16][CINP/16][64][4], [HID], [HID]
    const float *Wd, *d_scale, *d_shift;              // depthwise: [9][HID], [HID], [HID]
    const float *Wp, *p_scale, *p_shift;              // project: Wpk[COUTP/16][HID/16][64][4], [COUTP], [COUTP]
    float *Y;                                         // block output NHWC
    unsigned long long *prof = nullptr;               // debug: 8 device counters (per-stage s_memtime sums)
    const unsigned *We3 = nullptr, *Wp3 = nullptr;    // split weights: fp16 x2 for features.5-17 (fused_block_f16.hip), bf16 x3 for features.2-4 (fused_block_early.hip), or null
    const float *scl_e = nullptr, *scl_p = nullptr;   // features.5-17: {S, 1/S, 6 S} of the expand / project weights (device)
    const unsigned *Arm_e = nullptr, *Arm_p = nullptr;  // features.2-4: weight fragments of the row-marching kernel (fused_block_rm.hip), or null
    const unsigned *Alb_p = nullptr;                  // features.8-13: project fragments of the register-resident kernel (fused_block_lb.hip), or null
    const unsigned *Alb_e = nullptr;                  // ... its expand fragments
    const float *Tlb = nullptr;                       // ... and its per-group constants table
    const unsigned *Glb = nullptr;                    // features.15-17: per-group runs of fused_block_lb4.hip, or null
    float *scratch = nullptr;                         // workspace for schedules that keep partial sums (hidden-sliced small batches), or null
    size_t scratch_floats = 0;
};
bool launch_fused_block(int feature, const FusedBlockArgs &a, int B, hipStream_t s);
// early blocks (features.2-4) on the bf16 matrix pipe (fused_block_early.hip).  Their hidden width is walked in chunks of
// early_block_hc(HID) channels, each zero padded to a multiple of 32 in the project GEMM's K: the host packs Wp3 that way.
constexpr int early_block_hc(int hid) { return hid % 32 == 0 ? 32 : 48; }
bool launch_fused_block_early(int feature, const FusedBlockArgs &a, int B, hipStream_t s);
// early blocks, row-marching schedule (fused_block_rm.hip): hidden activations stay in registers, one wave per 32 hidden channels.
// Weight fragments for v_mfma_f32_32x32x16_f16, two fp16 pieces per weight (scaled by the layer's power of two, scl_e / scl_p above),
// lane (i = l&31, hh = l>>5) holds 8 K slots, [group][k16 step][piece 2][lane 64][4 dwords]:
//   expand  Arm_e: row i = hidden channel 32g + i, slot e of step s = input channel 16s + 8hh + e        (ceil(CIN/16) steps)
//   project Arm_p: row i = output channel i,       slot e of step s = hidden channel 32g + 16s + 8(e>>2) + 4hh + (e&3)   (2 steps)
// -- the project K order is the register order in which the expand / depthwise stage leaves a lane's 16 channels.
constexpr int rm_expand_dwords(int cin, int hid) { return ((hid + 31) / 32) * ((cin + 15) / 16) * 512; }
constexpr int rm_project_dwords(int hid) { return ((hid + 31) / 32) * 2 * 512; }
bool launch_fused_block_rm(int feature, const FusedBlockArgs &a, int B, hipStream_t s);
// features.3 + 4 / features.5 + 6 (first = 3 | 5) in ONE launch of the row-marching kernel; false: not applicable (launch them one by one)
bool launch_fused_pair_rm(int first, const FusedBlockArgs &a, const FusedBlockArgs &b, int B, hipStream_t s);
// features.5 + 6 of a small batch (<= 256 faces) in ONE launch of the whole-image tiled kernel (fused_block_f16.hip); false: not applicable
bool launch_fused_pair_f16(const FusedBlockArgs &a5, const FusedBlockArgs &a6, int B, hipStream_t s);
// 8x8 blocks (features.8-13), register-resident schedule (fused_block_lb.hip), both GEMMs on v_mfma_f32_16x16x32_f16 with every
// operand as TWO fp16 pieces (x = a + b, 22 significant bits; three products a a, a b, b a) and power-of-two operand scaling
// (synergy_abi.hip pack_backbone_mbv2).  Fragments [..][piece 2][lane 64][4 dwords], lane (m = l&15, kg = l>>4):
//   Alb_e [hidden tile HID/16][k32 step CIN/32]: row = hidden channel 16 nt + m, K slot e = input channel 32 kc + 8 kg + e
//   Alb_p [group HID/32][out tile COUT/16]:      row = output channel 16 mt + m, K slot e = hidden channel 32 G + (e < 4 ? 4 kg + e : 16 + 4 kg + e - 4)
//     -- the order in which the expand / depthwise stage leaves a lane group's eight channels;
//   Tlb [group][12][32] floats: rows 0-8 depthwise filter / Se (tap-major), 9: 16 x depthwise BN shift, 10: 16 Se x expand BN shift,
//     row 11 of group 0: {96 Se, 1 / (16 Sp)}.
constexpr int lb_project_dwords(int hid, int cout) { return (hid / 32) * (cout / 16) * 512; }
constexpr int lb_expand_dwords(int cin, int hid) { return (hid / 16) * (cin / 32) * 512; }
constexpr int lb_table_floats(int hid) { return (hid / 32) * 12 * 32; }
bool launch_fused_block_lb(int feature, const FusedBlockArgs &a, int B, hipStream_t s);
// features.first .. first + n - 1 of every face in ONE launch (a[i]: features.(first + i)).  lb_chain_mode: 3 = features.7-14,
// 2 = features.8-14, 1 = features.8-13, 0 = batch too small / SYN_LB_CHAIN=0 (launch them one by one)
int lb_chain_mode(int B, bool small = true);       // small: batches below the chain threshold may take the one-face-per-workgroup chain (features.8-14, mode 2)
bool launch_fused_chain_lb(const FusedBlockArgs *a, int first, int n_blocks, int B, hipStream_t s);
// 4x4 blocks (features.15-17), fused_block_lb4.hip: the same fragments and constants, but one contiguous run per hidden group
// Glb [group HID/32]: Alb_e fragments of the group's two hidden tiles [tile 2][k32 step][piece 2][64][4] | Alb_p fragments
// [out tile][piece 2][64][4] | Tlb rows [12][32] floats + 128 floats of padding  -- what one LDS-DMA burst copies.
constexpr int lb4_group_dwords(int cin, int cout) { return (2 * (cin / 32) * 512 + (cout / 16) * 512 + 512 + 2047) / 2048 * 2048; }     // padded to 8 KB
bool launch_fused_block_lb4(int feature, const FusedBlockArgs &a, int B, hipStream_t s);
// features.15-17 of every face in ONE launch (a[i]: features.(15 + i)); false = not applicable, launch them one by one
bool launch_fused_chain_lb4(const FusedBlockArgs *a, int B, hipStream_t s);
// same block with both GEMMs on the bf16 matrix pipe through the exact 3-way operand split (features.5-17)
bool launch_fused_block_f16(int feature, const FusedBlockArgs &a, int B, hipStream_t s);

// features.0 + features.1 fused (stem_block1.hip): image -> NHWC [B,60,60,16].
void launch_stem_block1(const float *img_nchw, const uint8_t *img_hwc_u8, const float *w0, const unsigned *w0b3, const float *s0,
                        const float *b0, const float *wd, const float *sd, const float *bd, const float *wp_pk,
                        const float *sp, const float *bp, float *Y, int B, hipStream_t s);

// features.0 + features.1 from uint8 crops, row-marching schedule (stem_rm.hip).  As3: stem filter / 128, scaled by the power of
// two S and split into two fp16 pieces, as fragments of v_mfma_f32_32x32x16_f16, [k16 step 2][piece 2][lane 64][4 dwords], lane
// (i = stem channel, hh = l>>5) K slot q = 8s + e:
//   hh = 0: q < 9 -> tap (ky 0, m = q), 9 <= q < 14 -> (ky 1, m = q - 9);   hh = 1: q < 9 -> (ky 2, m = q), 9 <= q < 13 -> (ky 1, m = q - 4)
//   with m = 3*kx + ci (the byte order of a pixel triple in the HWC image); the other slots are zero.
// s_shift [32]: BN shift - 255/256 * sum of the (scaled) filter, then {S, 1/S, 6 S}; Ap3: the 32->16 projection in rm_project order
// (1 group), scl_p its scales.
constexpr int rm_stem_set_dwords() { return 3 * 2 * 256 + 32 + 4; }      // fragments [3][2][64][4] | shift [32] | {S, 1/S, 6 S, -}
constexpr int rm_stem_dwords() { return 2 * rm_stem_set_dwords(); }      // the uint8 set (filter / 128, shift with the folded -255/256 sum) | the fp32 set (plain)
bool launch_stem_rm(const uint8_t *img8, const unsigned *As3, const float *s_shift, const float *Wd, const float *d_shift,
                    const unsigned *Ap3, const float *p_shift, const float *scl_p, float *Y, int B, hipStream_t s);
// the same march for normalised fp32 NCHW crops [B,3,120,120] (forward_test): As3 / s_shift = the second constant set
bool launch_stem_rm_f32(const float *img, const unsigned *As3, const float *s_shift, const float *Wd, const float *d_shift,
                        const unsigned *Ap3, const float *p_shift, const float *scl_p, float *Y, int B, hipStream_t s);

// features.18 + global average pool + the three heads fused (head_kernel.hip): NHWC [B,4,4,320] -> param [B,62].
void launch_head(const float *X, const float *Wpk /*[80][20][64][4]*/, const float *scale, const float *shift,
                 const float *Wfc /*[64,1280]*/, const float *bfc, float *param, float *pool /*nullable*/, int B,
                 hipStream_t s);

// same tail with the GEMM on the bf16 matrix pipe via an exact 3-way bf16 split of fp32 operands (head_kernel.hip)
// features.17's output still as the hidden-slice partial sums of the small-batch schedule (fused_block_lb4.hip): the tail adds them while it
// stages its input tile -- lb4_reduce_kernel's arithmetic and order -- instead of a reduce launch in front of it (round 5, batches below 513 faces)
struct HeadSliced {
    const float *part;       // [S][B][16][320] raw accumulators
    int S;
    const float *inv_p;      // accumulator -> output scale (one float, device)
    const float *p_shift;    // [320] BN shift of features.17's projection
};
void launch_head_f16x2(const float *X, const unsigned *Wb3 /*[80][10][3][64][4]*/, const float *shift, const float *Wfc,
                        const float *bfc, float *param, float *pool, float *scratch /*[B,1280]*/, int B, hipStream_t s,
                       const HeadSliced *sliced_in = nullptr);
// features.17 hidden-sliced WITHOUT its reduce launch: fills `out` for launch_head_f16x2; false: not applicable (run the block as usual)
bool launch_lb4_sliced17_deferred(const FusedBlockArgs &a, int B, hipStream_t s, HeadSliced *out);

// ---- on-device crop + Lanczos-4 resize (preproc_kernels.hip) ----
void launch_crop_resize(const uint8_t *frame, int H, int W, const int *box, const int *xofs, const short *xcoef,
                        const int *yofs, const short *ycoef, uint8_t *out, int B, hipStream_t s,
                        const long long *foff = nullptr, const int *fdim = nullptr, const int *fidx = nullptr);   // foff: faces of several frames (frame + foff[fidx[b]], fdim[2 f] x fdim[2 f + 1])

// ---- ResNet-50 variant (resnet_kernels.hip) ----
// implicit-GEMM conv, NHWC: W [Npad][KH*KW*Cin] (tap-major), act 0 none / 1 ReLU after the optional residual add
// ResNet-50 activation formats (resnet_kernels.hip "pair format"): inside an fp16 x2 forward every tensor a convolution consumes is stored by
// its producer as the two fp16 pieces of the consumer's MFMA operands ([pixel][32-channel chunk][32 high halves | 32 low halves]); `fmt` is a
// bit set: kFmtOutPair = the output is written in that format, kFmtResPair = the residual is read in it (else fp32 NHWC).
constexpr int kFmtOutPair = 1, kFmtResPair = 2;
// exact fp32-MFMA convolution; in_pair: the input is in the pair format (an fp16 x2 forward), else fp32.  N % 64 == 0.
void launch_conv(const float *in, const float *W, const float *scale, const float *shift, const float *residual, float *out,
                 int B, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act, hipStream_t s,
                 float *stat = nullptr /* range-guard slot of the output, see launch_conv_f16x2 */, int in_pair = 0, int fmt = 0);
// same conv on the fp16 matrix pipe (two-piece operand split, three products): W3 [N/16][taps*Cin/32][2][64][4] dwords, output channels in pair order,
// Cin % 64 == 0, N % 64 == 0; the input is in the pair format
// stat (nullable): one float of the per-forward range-guard array -- max |output| is folded into it (resnet_kernels.hip range_note)
void launch_conv_f16x2(const float *in, const unsigned *W3, const float *scale, const float *shift, const float *residual,
                     float *out, int B, int Hin, int Hout, int Cin, int N, int KH, int KW, int stride, int pad, int act,
                     hipStream_t s, float *stat = nullptr, int lt = 0 /* 1: the LDS-tiled kernel where its shape fits (N % 128 == 0, M >= 4096); 2: ... 128-pixel tiles only */,
                     int fmt = kFmtOutPair);
// window of a tensor's max |x| inside which the fp16 x2 split of an UNSCALED activation keeps >= 15 bits relative to that maximum:
// above kRangeHi v_cvt_pkrtz_f16_f32 saturates, below kRangeLo even the largest element's low piece is a 4-bit subnormal
constexpr float kRangeHi = 6.0e4f, kRangeLo = 9.765625e-4f;      // 2^-10
// layout of the status array: tensor t owns kRangeSub sub-slots, kRangeStride floats (256 B) apart: stat[(t * kRangeSub + sub) * kRangeStride];
// launchers receive the tensor's base pointer
constexpr int kRangeSub = 64, kRangeStride = 64;
// conv3 + BN + identity + ReLU of a bottleneck fused with the next bottleneck's conv1 + BN + ReLU (resnet_kernels.hip conv_c3f_kernel).
// W3: conv3's fp16 x2 fragments as launch_conv_f16x2 takes them; W1f: the next conv1's fragments step-major [Cin/32][N1/16][piece 2][lane 64][4 dwords]
// (natural K order: in pair order conv3's accumulators are conv1's operand).  T2 / out / T1n in the pair format, identity in it (res_pair) or fp32.
// false: this (K, N1) combination is not instantiated -- launch the two separately.
// conv2 of the bottleneck in front of the fused launch (round 6, 64-channel bottlenecks = layer 1): see resnet_kernels.hip C2Args
struct C2Args {
    const float *T1 = nullptr;      // conv2 input [B, Hin, Hin, 64], pair format
    const unsigned *W2 = nullptr;   // conv2's fp16 x2 fragments as launch_conv_f16x2 takes them
    const float *scale2 = nullptr, *shift2 = nullptr;
    int Hin = 0, Hout = 0, stride = 1;
    unsigned in_bytes = 0;
    float *stat2 = nullptr;         // range-guard slot of conv2's output
};
bool launch_conv_c3f(const float *T2, const unsigned *W3, const float *scale3, const float *shift3, const float *identity, float *out,
                     const unsigned *W1f, const float *s1 /*device {S, 1/S}*/, const float *scale1, const float *shift1, float *T1n, int M, int K, int N3, int N1,
                     hipStream_t s, float *stat3, float *stat1, int res_pair, const C2Args *c2 = nullptr);
// ... with the block's downsample branch (1x1 conv + BN on the block input X [M, Kd]) evaluated in the same kernel instead of read as identity
// (layer1.0: Kd = 64, stride 1); Wd: the downsample conv's fragments as launch_conv_f16x2 takes them.  false: not this shape.
bool launch_conv_c3f_ds(const float *T2, const unsigned *W3, const float *scale3, const float *shift3, const float *X, const unsigned *Wd,
                        const float *scale_d, const float *shift_d, float *out, const unsigned *W1f, const float *s1, const float *scale1,
                        const float *shift1, float *T1n, int M, int K, int Kd, int N3, int N1, hipStream_t s, float *stat3, float *stat1, const C2Args *c2 = nullptr);
// conv3 + the stride-2 downsample branch of a bottleneck as one GEMM over [T2 ; x] with the BatchNorm scales folded into the weights (resnet_kernels.hip DUAL)
bool launch_conv_dual(const float *a, const float *x, const unsigned *W3d, const float *ones, const float *shift, float *out, int B, int H, int C1,
                      int Hin2, int stride2, int Cin2, int N, int act, hipStream_t s, float *stat, int fmt);
bool conv_c3f_supported(int K, int N3, int N1);
void launch_resnet_stem(const float *img_nchw, const uint8_t *img_hwc_u8, const float *w147x64, const float *scale,
                        const float *shift, float *out /*[B,60,60,64]*/, int B, hipStream_t s);
// 7x7 stem on the fp16 matrix instructions (uint8 crops): As3 [group 2][k16 step 10][piece 2][lane 64][4 dwords], lane (i = channel 32G + i,
// hh) K slot kk = 16s + 8hh + e: kernel row ky = kk / 22, m = kk % 22 (m = 0: don't-care byte, else kx = (m-1)/3, ci = (m-1)%3), kk >= 154
// zero; weights / 128 x S (power of two) as two fp16 pieces; s_shift [64] = BN shift - 255/256 * sum of the scaled filter, then {S, 1/S}
constexpr int rn_stem_dwords() { return 2 * 10 * 2 * 256 + 64 + 4; }
// pool != 0: the 3x3 / 2 max-pool in the epilogue, out = [B,30,30,64] (stat: range-guard slot of the pooled tensor, nullable)
bool launch_resnet_stem_mfma(const uint8_t *img8, const unsigned *As3, const float *s_shift, float *out, int B, hipStream_t s, int pool = 0,
                             float *stat = nullptr, int out_pair = 0 /* pool: the pooled tensor in the pair format */);
void launch_maxpool3x3s2(const float *in, float *out, int B, int Hin, int Hout, int C, hipStream_t s, float *stat = nullptr, int out_pair = 0);
void launch_pool_fc_generic(const float *feat, const float *Wfc, const float *bias, float *param, float *pool, int B, int P,
                            int C, int n_out, int out_stride, hipStream_t s, const float *stat = nullptr, int n_stat = 0,
                            unsigned *guard_word = nullptr /* host-mapped word: set to 1 by a forward the range guard poisoned */);

// ---- reconstruction -----------------------------------------------------------------
// basis: pre-packed per 32-vertex tile in MFMA-operand lane order (see recon_kernels.hip):
//   Bp[tile][coord x|y|z][chunk][lane][4], K = 52: 0..39 shape, 40..49 expression,
//   50 = mean u (alpha 1), 51 = 0;  nvp = n_vert rounded up to 32.
// rec: [B,64] float scratch for the per-face records (alpha[52] | M[9] | T[3]).
void launch_reconstruct(const float *param, const float *mean62, const float *std62,
                        const float *basis, int n_vert, int nvp, const float *roi,
                        int transform, float *out, int pitch /*floats between rows of out, >= n_vert*/, int B, hipStream_t s, float *rec);

// Same contraction on v_mfma_f32_32x32x16_f16 (two fp16 pieces per operand, three partial products).
//   basis3: per (32-vertex tile, coord) kBasisF16 dwords: [k16 step 3][piece 2][lane 64][4 dwords] for k = 0..47
//           (lane (j = l&31 vertex, hh = l>>5) holds k = 16*step + 8*hh + e), column k scaled by its own power of two 2^e_k, then one
//           more [lane 64][4] fragment: a fourth k16 step whose slots carry the split partial products of columns 48, 49 and the
//           mean (recon_prep_f16_kernel).  mean62 / std62 here are the COLUMN-SCALED de-whitening constants (entries 12..61 x 2^-e_k,
//           mean62[62] = 2^-e_u, the coefficient of the scaled mean shape): synergy_abi.hip pack_basis.
//   rec3:   per 32-face tile kRecTileF16 dwords: the alpha pieces (x the face's power of two Sa) in the same lane order (faces), the
//           matching fourth-step fragment, then 32 x 16 fp32 records M[9] / Sa | T[3] | 0 x 4.
constexpr int kBasisF16 = (3 * 2 + 1) * 256;
constexpr int kRecTileF16 = (3 * 2 + 1) * 256 + 32 * 16;
constexpr int kRecFloatsPerFace = 96;       // workspace share per face for the records of either layout (+ kRecSlack in total)
constexpr int kRecSlack = 4096;
void launch_reconstruct_f16(const float *param, const float *mean62, const float *std62, const unsigned *basis3, int n_vert,
                           int nvp, const float *roi, int transform, float *out, int pitch, int pad_writable, int B, hipStream_t s, float *rec3,
                           hipEvent_t *marks = nullptr);

// ---- mesh consumers (render_kernels.hip): Sim3DR.get_normal / RenderPipeline / Sim3DR.rasterize / cv2.addWeighted ----
// vertices: F meshes, planar = 1 -> [F,3,nver] (the layout syn_reconstruct writes), 0 -> [F,nver,3] (the reference's)
void launch_mesh_normals(const float *vertices, const int *tri, const int *adj_off, const int *adj_tri, float *tri_normal,
                         float *normal /*[F,nver,3]*/, unsigned *mm /*[F,6] scratch: per-axis min/max keys*/, int F, int nver,
                         int ntri, int planar, hipStream_t s);
void launch_mesh_lighting(const float *vertices, const float *normal, const unsigned *mm, const float *cfg /*16 floats, device*/,
                          float *light /*[F,nver,3]*/, int F, int nver, int planar, hipStream_t s);
void launch_rasterize(const float *vertices, const int *tri, const float *colors /*[F,nver,c]*/, unsigned long long *zkey /*[h*w]*/,
                      unsigned char *image /*[h,w,c] in place*/, int F, int nver, int ntri, int h, int w, int c, int planar,
                      int reverse, hipStream_t s);
void launch_add_weighted(const unsigned char *a, float alpha, const unsigned char *b, float beta, unsigned char *out, size_t n,
                         hipStream_t s);

// ---- AFLW2000-3D landmark error (eval_kernels.hip) ----
void launch_nme(const float *fit, const float *gt, const float *roi, float *nme, int N, hipStream_t s);

// ---- FaceBoxes detector (detector_kernels.hip) ----
void launch_det_preproc(const unsigned char *frame, int H, int W, float *out, int Ho, int Wo, hipStream_t s);
void launch_det_conv(const float *in, const float *Wt, const float *shift, float *out, int Hi, int Wi, int cs_in, int ci0, int Cin, int Ho,
                     int Wo, int cs_out, int co0, int Cout, int K, int stride, int pad, int act, hipStream_t s);
void launch_det_pool(const float *in, float *out, int Hi, int Wi, int C, int Ho, int Wo, int stride, int is_max, hipStream_t s);
void launch_det_decode(const float *loc, const float *conf, int P, int Hn, int Wn, int H4, int W4, int H5, int W5, int H6, int W6,
                       float scale, float thr, float *cand, int *n_cand, int max_cand, float *boxes_out, float *scores_out,
                       hipStream_t s);
void launch_det_nms(const float *cand, const int *n_cand, int max_cand, int top_k, float nms_thr, int keep_top_k, float *dets, int *n_out,
                    hipStream_t s);
int det_sort_capacity();

void launch_pose(const float *param, const float *mean62, const float *std62, const float *roi,
                 double *angles /*nullable together with t3d*/, float *t3d, float *pmat /*nullable [B,3,4]*/, int B, hipStream_t s);
// landmarks + pose in one launch (recon_kernels.hip lmk_pose_kernel): fp32 landmark tiles, plain fp32 multiply-adds
void launch_lmk_pose(const float *param, const float *mean62, const float *std62, const float *basis_lmk, int n_lmk, int nlp, const float *roi,
                     int transform, float *lmk, double *angles, float *t3d, int B, hipStream_t s);

}  // namespace syn
