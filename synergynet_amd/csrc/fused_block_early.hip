// Fused inverted-residual block for the EARLY MobileNetV2 blocks (features.2-4: 60x60 / 30x30 maps, 16-32 channels in,
// 96-144 hidden) with both 1x1 GEMMs on the fp16 matrix instructions at fp32-equivalent accuracy (two fp16 pieces per operand, three partial products: fused_block_f16.hip;
// see fused_block_f16.hip).  Reference: backbone_nets/mobilenetv2_backbone.py:33-70 (InvertedResidual.forward).
//
// These blocks are spatially tiled (halo ring recomputed) and their weights are tiny, so the dataflow differs from the
// late blocks:
//   * a workgroup is PERSISTENT; all weights of the block live in LDS, already split and in MFMA lane order;
//   * every wave OWNS pixel tiles of the input tile: it loads their channels straight from global memory into
//     registers (next tile prefetched while the current one computes), splits them into the two fp16 pieces ONCE per
//     tile and keeps them as the MFMA "B" operand for every hidden channel tile of every chunk -- the block input
//     never touches LDS and the expand stage issues no LDS operand read besides the (wave-broadcast) weight fragments;
//   * expand  E = ReLU6(X . We^T + b)   -> LDS fp32  (out-of-image halo pixels written as 0: the ReLU ceiling trick)
//     depthwise D = ReLU6(dw3x3(E) + b)  -> LDS, split into two fp16 planes on the way out
//     project  acc += D . Wp[:, chunk]^T -> fp32 accumulators in registers (K of a chunk padded to a multiple of 32)
//   With the fp32-input MFMA gone, the vector pipe only carries the depthwise FMAs, the splits and the epilogues; the
//   matrix pipe runs the GEMMs concurrently (fp32-input MFMAs were 50-85% of these blocks' time before).
#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int cdive(int a, int b) { return (a + b - 1) / b; }
constexpr int rupe(int a, int b) { return cdive(a, b) * b; }
// two floats -> packed fp16 pieces a (high) and b (low), x = a + b to 22 significant bits (fused_block_f16.hip)
__device__ __forceinline__ void split2e(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ f32x4 mfmae(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the three partial products, smallest terms first
__device__ __forceinline__ f32x4 mac3e(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x4 c) {
    c = mfmae(a[1], b[0], c);
    c = mfmae(a[0], b[1], c);
    c = mfmae(a[0], b[0], c);
    return c;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfmae32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
}  // namespace

#define SYNE_LAP(i) do { if (PROF) { tn = __builtin_amdgcn_s_memtime(); pt_[i] += tn - tk; tk = tn; } } while (0)

template <int CIN_, int HID_, int COUT_, int HIN_, int S_, bool RES_, int TH_, int TW_, int NW_, int NG_, int WN_, int WP_>
struct EarlyCfg {
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_, HIN = HIN_, S = S_, TH = TH_, TW = TW_, NW = NW_, NG = NG_, WN = WN_, WP = WP_;
    static constexpr int GW = NW / NG, GT = GW * 64;            // waves / threads of one group (a group works on its own tile)
    static constexpr bool RES = RES_;
    static constexpr int HC = early_block_hc(HID_);             // hidden chunk (shared with the host packer)
    static constexpr int HCP = rupe(HC, 32), KP = HCP / 32;     // project K per chunk, padded to k32 steps
    static constexpr int NCH = HID / HC;
    static constexpr int NT = NW * 64;
    static constexpr int HOUT = S == 2 ? (HIN + 1) / 2 : HIN;
    static constexpr int TILES_Y = cdive(HOUT, TH), TILES_X = cdive(HOUT, TW);
    static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    static constexpr int PIN = IH * IW, PINP = rupe(PIN, CIN_ == 16 ? 32 : 16), PT_IN = PINP / 16;
    static constexpr int POUT = TH * TW, POUTP = rupe(POUT, 16), PT_O = POUTP / 16;
    // CIN = 16: the expand GEMM runs on v_mfma_f32_32x32x16_f16 -- K = 16 is exactly one step (the 16x16x32 form would
    // spend half of its K on zeros): 32-pixel x 32-channel tiles, lane (j = l&31, h = l>>5) holds k = 8h .. 8h+7
    static constexpr bool K16 = (CIN == 16);
    static constexpr int PXT = K16 ? 32 : 16;                  // pixels per expand tile
    static constexpr int PT_X = PINP / PXT;
    static constexpr int PPW = cdive(PT_X, GW);                // input pixel tiles owned by a wave
    // leftover pixel tiles of the last round (PT_X % GW of them): when few (features.3: 9 tiles on 8 waves) one wave -- and its
    // SIMD -- would carry a whole extra tile per chunk; instead its hidden-channel tiles are dealt to different waves
    static constexpr int LEFT = PT_X - (PPW - 1) * GW;
    static constexpr bool SPLIT = CIN_ != 16 && HC / 16 >= 2 && PPW >= 2 && LEFT * (HC / 16) < GW;
    static constexpr int COUTP = rupe(COUT, 16), NT_O = COUTP / 16, NT_E = HC / 16;
    static constexpr int AN = cdive(NT_O, WN), AP = cdive(PT_O, WP);
    static constexpr int ES = HC + 4, DSD = HCP / 2 + 4, DPL = POUTP * DSD;
    // depthwise stage mapping: thread = (channel quad, output column, row segment)
    static constexpr int C4N = HC / 4;
    static constexpr int RS_ = GT / (C4N * TW);
    static constexpr int RS = RS_ < 1 ? 1 : (RS_ > TH ? TH : RS_);
    static constexpr int RPS = cdive(TH, RS), DW_THREADS = C4N * TW * cdive(TH, RPS);
    // LDS carve (dwords)
    static constexpr int ES_DW = PINP * ES, DB_DW = 2 * DPL;
    static constexpr int WE_DW = (CIN_ == 16 ? HID / 32 : HID / 16) * 512, WP_DW = NT_O * NCH * KP * 512, WD_DW = 11 * HID;
    static constexpr int LDS_DWORDS = NG * (ES_DW + DB_DW) + WE_DW + WP_DW + WD_DW + HID;
    static_assert(CIN <= 32 && CIN % 8 == 0, "one k32 step of expand, whole 8-channel lane groups");
    static_assert(HID % HC == 0 && HC % 16 == 0, "hidden chunking");
    static_assert(CIN_ != 16 || HC % 32 == 0, "32-channel expand tiles");
    static_assert(NW % NG == 0 && WN * WP == GW, "wave grid of a group");
    static_assert(!RES || (S == 1 && CIN == COUT), "residual only on stride-1 same-width blocks");
    static_assert(HOUT % TH == 0 && HOUT % TW == 0, "whole tiles");
    static_assert(LDS_DWORDS * 4 <= 160 * 1024, "LDS budget");
};

template <class C, bool PROF = false>
__global__ __launch_bounds__(C::NW * 64) __attribute__((amdgpu_waves_per_eu(cdive(C::NW, 4), cdive(C::NW, 4))))
void fused_block_early_kernel(
    const float *__restrict__ X, const unsigned *__restrict__ We3 /*[HID/16][2][64][4]*/, const float *__restrict__ e_shift,
    const float *__restrict__ Wd /*[9][HID] scaled*/, const float *__restrict__ d_shift,
    const unsigned *__restrict__ Wp3 /*[NT_O][NCH*KP][2][64][4]*/, const float *__restrict__ p_shift,
    float *__restrict__ Y, int B, int total_tiles, const float *__restrict__ scl_e /*{S, 1/S, 6 S} of the expand weights*/,
    const float *__restrict__ scl_p, unsigned long long *prof = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DWORDS];
    constexpr int NT = C::NT, GT = C::GT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / C::GW, gw = wave % C::GW, gtid = tid % GT;     // group, wave / thread inside the group
    float *Es = reinterpret_cast<float *>(smem) + grp * C::ES_DW;         // per group: [PINP][ES] fp32
    unsigned *Db = smem + C::NG * C::ES_DW + grp * C::DB_DW;              // per group: 3 planes [POUTP][DSD]
    unsigned *Wle = smem + C::NG * (C::ES_DW + C::DB_DW), *Wlp = Wle + C::WE_DW;
    float *Wds = reinterpret_cast<float *>(Wlp + C::WP_DW);               // [11][HID]: 9 taps | (unused) | shift
    float *Ebn = Wds + C::WD_DW;                                          // [HID] expand BN shift
    const int r16 = lane & 15, g = lane >> 4;
    const int xl = C::K16 ? (lane & 31) : r16, xg = C::K16 ? (lane >> 5) : g;   // expand operand: pixel in tile, channel octet
    const int wn = gw % C::WN, wp = gw / C::WN;
    // SPLIT: wave gw in [1, LEFT*NT_E] expands hidden tile (gw-1) % NT_E of leftover pixel tile (gw-1) / NT_E
    const bool xvalid = C::SPLIT && gw >= 1 && gw <= C::LEFT * C::NT_E;
    const int xq = xvalid ? (gw - 1) / C::NT_E : 0, xnt = xvalid ? (gw - 1) % C::NT_E : -1;
    auto slot_pt = [&](int i) { return (C::SPLIT && i == C::PPW - 1) ? (C::PPW - 1) * C::GW + xq : gw + i * C::GW; };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // power-of-two scales of the fp16 weight pieces (fused_block_f16.hip): expand accumulators start at Se x shift, ReLU6 clamps at
    // 6 Se, the depthwise filter carries 1 / Se; project accumulators start at Sp x shift and are rescaled before the residual add
    const float Se = scl_e[0], inv_se = scl_e[1], c6e = scl_e[2], Sp = scl_p[0], inv_sp = scl_p[1];
    unsigned long long pt_[7] = {0, 0, 0, 0, 0, 0, 0}, tk = PROF ? __builtin_amdgcn_s_memtime() : 0ull, tn = 0, ntiles_done = 0;

    // this wave's input pixel tiles of tile `tile` -> registers (8 channels 8g..8g+7 of pixel pt*16 + r16; zero outside
    // the image, past CIN and past the tile)
    f32x4 xr[C::PPW][2];
    auto load_x = [&](int tile) {
        const int tx = tile % C::TILES_X, ty = (tile / C::TILES_X) % C::TILES_Y, f = tile / (C::TILES_X * C::TILES_Y);
        const int iy0 = ty * C::TH * C::S - 1, ix0 = tx * C::TW * C::S - 1;
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) {
            const int p = slot_pt(i) * C::PXT + xl;
            const int iy = iy0 + p / C::IW, ix = ix0 + p % C::IW;
            xr[i][0] = z4; xr[i][1] = z4;
            if ((!C::SPLIT || i < C::PPW - 1 || xvalid) && p < C::PIN && 8 * xg < C::CIN && (unsigned)iy < (unsigned)C::HIN && (unsigned)ix < (unsigned)C::HIN) {
                const float *src = &X[((size_t)(f * C::HIN + iy) * C::HIN + ix) * C::CIN + 8 * xg];
                xr[i][0] = *(const f32x4 *)src;
                xr[i][1] = *(const f32x4 *)(src + 4);
            }
        }
    };

    // groups take alternate tiles; every wave of the workgroup runs the same number of iterations (and barriers): a
    // group without a tile in the last round skips the work only
    const int stride = gridDim.x * C::NG;
    const int n_iter = (total_tiles - (int)blockIdx.x * C::NG + stride - 1) / stride;
    int tile = blockIdx.x * C::NG + grp;
    if (tile < total_tiles) load_x(tile);
    // ---- once per (persistent) workgroup: all weights of the block -> LDS; the D planes start as zeros ----
    for (int i = tid; i < C::WE_DW / 4; i += NT) *(u32x4 *)&Wle[4 * i] = *(const u32x4 *)&We3[4 * i];
    for (int i = tid; i < C::WP_DW / 4; i += NT) *(u32x4 *)&Wlp[4 * i] = *(const u32x4 *)&Wp3[4 * i];
    for (int i = tid; i < 11 * C::HID / 4; i += NT) {
        const int row = i / (C::HID / 4), c4 = i % (C::HID / 4);
        f32x4 v = z4;
        if (row < 9) v = *(const f32x4 *)&Wd[row * C::HID + 4 * c4] * inv_se;
        else if (row == 10) v = *(const f32x4 *)&d_shift[4 * c4];
        *(f32x4 *)&Wds[row * C::HID + 4 * c4] = v;
    }
    for (int i = tid; i < C::HID / 4; i += NT) *(f32x4 *)&Ebn[4 * i] = *(const f32x4 *)&e_shift[4 * i] * Se;
    for (int i = gtid; i < C::DB_DW / 4; i += GT) *(u32x4 *)&Db[4 * i] = (u32x4){0u, 0u, 0u, 0u};  // pad rows / pad K columns stay 0
    f32x4 psh[C::AN];
#pragma unroll
    for (int i = 0; i < C::AN; ++i) {
        const int n = (wn + i * C::WN) * 16 + 4 * g;
        psh[i] = n < C::COUTP ? *(const f32x4 *)&p_shift[n] * Sp : z4;
    }
    __syncthreads();
    // the project weight fragments of this wave's output-channel tiles are the same for every tile: registers, not LDS reads
    u32x4 pa[C::AN][C::NCH][C::KP][2];
#pragma unroll
    for (int i = 0; i < C::AN; ++i) {
        int nt = wn + i * C::WN;
        nt = nt < C::NT_O ? nt : 0;
#pragma unroll
        for (int c = 0; c < C::NCH; ++c)
#pragma unroll
            for (int kc = 0; kc < C::KP; ++kc)
#pragma unroll
                for (int p = 0; p < 2; ++p) pa[i][c][kc][p] = *(const u32x4 *)(Wlp + ((size_t)(nt * C::NCH + c) * C::KP + kc) * 512 + p * 256 + lane * 4);
    }
    // Two groups run the same barrier-separated stage sequence [project(c-1) expand(c)] | [depthwise(c)] | ... one stage
    // apart: while one group's waves feed the matrix pipe (expand / project), the other group's waves on the same SIMDs
    // run the depthwise stage on the VALU.  (s_barrier only counts arrivals, so the groups may sit at different barriers.)
    if (C::NG == 2 && grp == 1) __syncthreads();

    for (int it = 0; it < n_iter; ++it, tile += stride) {
        const bool live = tile < total_tiles;
        const int tx = tile % C::TILES_X, ty = (tile / C::TILES_X) % C::TILES_Y, f = tile / (C::TILES_X * C::TILES_Y);
        const int oy0 = ty * C::TH, ox0 = tx * C::TW;
        const int iy0 = oy0 * C::S - 1, ix0 = ox0 * C::S - 1;             // image coords of input-tile pixel (0,0)

        // ---- stage 0: split this wave's pixel tiles into fp16 pieces (registers); prefetch the next tile ----
        u32x4 xb[C::PPW][2];
        float ehi[C::PPW];             // ReLU6 ceiling of the hidden pixel: 6 inside the image, 0 on the zero-padding ring
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = xr[i][h];
                unsigned a0, b0, a1, b1;
                split2e(v[0], v[1], a0, b0);
                split2e(v[2], v[3], a1, b1);
                xb[i][0][2 * h] = a0; xb[i][0][2 * h + 1] = a1;
                xb[i][1][2 * h] = b0; xb[i][1][2 * h + 1] = b1;
            }
            const int p = slot_pt(i) * C::PXT + xl;
            const int iy = iy0 + p / C::IW, ix = ix0 + p % C::IW;
            ehi[i] = ((unsigned)iy < (unsigned)C::HIN && (unsigned)ix < (unsigned)C::HIN) ? c6e : 0.0f;
        }
        if (tile + stride < total_tiles) load_x(tile + stride);
        SYNE_LAP(0);

        f32x4 acc[C::AN][C::AP];
#pragma unroll
        for (int i = 0; i < C::AN; ++i)
#pragma unroll
            for (int j = 0; j < C::AP; ++j) acc[i][j] = psh[i];        // BN shift = accumulator start
        f32x4 resv[C::RES ? C::AN : 1][C::RES ? C::AP : 1];

#pragma unroll
        for (int c = 0; c < C::NCH; ++c) {
            const int hc0 = c * C::HC;
            // ---- stage 1: expand 1x1 (fp16 x2) + BN shift + ReLU6 -> Es (fp32); operands: LDS weights x registers ----
            if (C::K16) {
#pragma unroll
                for (int nt = 0; nt < C::HC / 32; ++nt) {
                    u32x4 a[2];
                    const unsigned *wa = Wle + (size_t)(hc0 / 32 + nt) * 512 + lane * 4;
#pragma unroll
                    for (int p = 0; p < 2; ++p) a[p] = *(const u32x4 *)(wa + p * 256);
                    // D rows (channels) of register r: (r&3) + 8*(r>>2) + 4*xg  -> four float4 groups of consecutive channels
                    // (fetching these fragments one stage ahead, across the project stage, measured 3 % slower)
                    f32x4 sh4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) sh4[q] = *(const f32x4 *)&Ebn[hc0 + nt * 32 + 8 * q + 4 * xg];
#pragma unroll
                    for (int i = 0; i < C::PPW; ++i) {
                        const int pt = gw + i * C::GW;
                        if (pt >= C::PT_X || !live) break;              // wave-uniform
                        f32x16 e;
#pragma unroll
                        for (int r = 0; r < 16; ++r) e[r] = sh4[r >> 2][r & 3];
                        e = mfmae32(a[1], xb[i][0], e);
                        e = mfmae32(a[0], xb[i][1], e);
                        e = mfmae32(a[0], xb[i][0], e);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 ev;
#pragma unroll
                            for (int t = 0; t < 4; ++t) ev[t] = __builtin_amdgcn_fmed3f(e[4 * q + t], 0.0f, ehi[i]);
                            *(f32x4 *)&Es[(pt * 32 + xl) * C::ES + nt * 32 + 8 * q + 4 * xg] = ev;
                        }
                    }
                }
            } else {
            // weight fragments + BN shift of hidden tile nt+1 are read from LDS while tile nt runs on the matrix pipe
            u32x4 aq[2][2];
            f32x4 shq[2];
            auto ldw = [&](int nt, u32x4(&a)[2], f32x4 &sh) {
                const unsigned *wa = Wle + (size_t)(hc0 / 16 + nt) * 512 + lane * 4;
#pragma unroll
                for (int p = 0; p < 2; ++p) a[p] = *(const u32x4 *)(wa + p * 256);
                sh = *(const f32x4 *)&Ebn[hc0 + nt * 16 + 4 * g];
            };
            ldw(0, aq[0], shq[0]);
#pragma unroll
            for (int nt = 0; nt < C::NT_E; ++nt) {
                if (nt + 1 < C::NT_E) ldw(nt + 1, aq[(nt + 1) & 1], shq[(nt + 1) & 1]);
                const u32x4(&a)[2] = aq[nt & 1];
                const f32x4 sh = shq[nt & 1];
#pragma unroll
                for (int i = 0; i < C::PPW; ++i) {
                    const int pt = slot_pt(i);
                    if (pt >= C::PT_IN || !live) break;                 // wave-uniform
                    if (C::SPLIT && i == C::PPW - 1 && nt != xnt) continue;     // leftover tile: only this wave's share
                    const f32x4 e = mac3e(a, xb[i], sh);
                    f32x4 ev;
#pragma unroll
                    for (int q = 0; q < 4; ++q) ev[q] = __builtin_amdgcn_fmed3f(e[q], 0.0f, ehi[i]);
                    *(f32x4 *)&Es[(pt * 16 + r16) * C::ES + nt * 16 + 4 * g] = ev;
                }
            }
            }
            SYNE_LAP(1);
            __syncthreads();
            SYNE_LAP(2);
            // ---- stage 2: depthwise 3x3 + BN shift + ReLU6 (fp32 VALU), output split into fp16 x2 planes ----
            // thread = (channel quad, output column, row segment); 3-row sliding window.  Zero padding needs no code:
            // hidden pixels outside the image were written as zeros by stage 1.
            for (int t = live ? gtid : C::DW_THREADS; t < C::DW_THREADS; t += GT) {
                const int c4 = t % C::C4N, q = t / C::C4N;
                const int oxl = q % C::TW, seg = q / C::TW;
                f32x4 w[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) w[k] = *(const f32x4 *)&Wds[k * C::HID + hc0 + 4 * c4];
                const f32x4 sh = *(const f32x4 *)&Wds[10 * C::HID + hc0 + 4 * c4];
                const float *ebase = Es + (oxl * C::S) * C::ES + 4 * c4;
                f32x4 rb[3][3];
                auto load_row = [&](int ly, f32x4(&dst)[3]) {
                    const float *er = ebase + ly * C::IW * C::ES;
                    dst[0] = *(const f32x4 *)(er);
                    dst[1] = *(const f32x4 *)(er + C::ES);
                    dst[2] = *(const f32x4 *)(er + 2 * C::ES);
                };
#pragma unroll
                for (int r = 0; r < C::RPS; ++r) {
                    const int oyl = seg * C::RPS + r;
                    if (oyl >= C::TH) break;
                    const int lyb = oyl * C::S;                          // input-tile row of tap ky = 0
                    if (r == 0) {
                        load_row(lyb, rb[0]); load_row(lyb + 1, rb[1]); load_row(lyb + 2, rb[2]);
                    } else if (C::S == 1) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) { rb[0][k] = rb[1][k]; rb[1][k] = rb[2][k]; }
                        load_row(lyb + 2, rb[2]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k) rb[0][k] = rb[2][k];
                        load_row(lyb + 1, rb[1]); load_row(lyb + 2, rb[2]);
                    }
                    f32x4 a = sh;
                    a += rb[0][0] * w[0]; a += rb[0][1] * w[1]; a += rb[0][2] * w[2];
                    a += rb[1][0] * w[3]; a += rb[1][1] * w[4]; a += rb[1][2] * w[5];
                    a += rb[2][0] * w[6]; a += rb[2][1] * w[7]; a += rb[2][2] * w[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = __builtin_amdgcn_fmed3f(a[e], 0.0f, 6.0f);
                    unsigned a0, b0, a1, b1;
                    split2e(a[0], a[1], a0, b0);
                    split2e(a[2], a[3], a1, b1);
                    const int dofs = (oyl * C::TW + oxl) * C::DSD + 2 * c4;
                    *(u32x2 *)&Db[0 * C::DPL + dofs] = (u32x2){a0, a1};
                    *(u32x2 *)&Db[1 * C::DPL + dofs] = (u32x2){b0, b1};
                }
            }
            SYNE_LAP(3);
            __syncthreads();
            SYNE_LAP(4);
            // residual: fetched (L2-hot block input) while the last project stage runs, and before any store of this tile --
            // vector memory retires in order, a load issued between two stores would wait for the first store's acknowledgement
            if (C::RES && c == C::NCH - 1) {
#pragma unroll
                for (int i = 0; i < C::AN; ++i)
#pragma unroll
                    for (int j = 0; j < C::AP; ++j) {
                        // branch-free (indices clamped into the tile; the epilogue skips what is not real): with control flow
                        // around a load the compiler's wait for it degrades to vmcnt(0), i.e. to waiting for the stores too
                        int n = (wn + i * C::WN) * 16 + 4 * g, po = (wp + j * C::WP) * 16 + r16;
                        n = n + 3 < C::COUT ? n : 0;
                        po = po < C::POUT ? po : 0;
                        const int fc = live ? f : 0;
                        resv[i][j] = *(const f32x4 *)&X[((size_t)(fc * C::HOUT + oy0 + po / C::TW) * C::HOUT + ox0 + po % C::TW) * C::COUT + n];
                    }
            }
            // ---- stage 3: project 1x1 (fp16 x2), K = this hidden chunk (zero padded to k32 steps) ----
            if (live)
#pragma unroll
            for (int kc = 0; kc < C::KP; ++kc) {
                u32x4 b[C::AP][2];
#pragma unroll
                for (int j = 0; j < C::AP; ++j) {
                    const int pt = wp + j * C::WP;
                    const int row = ((pt < C::PT_O ? pt : 0) * 16 + r16) * C::DSD + kc * 16 + 4 * g;
#pragma unroll
                    for (int p = 0; p < 2; ++p) b[j][p] = *(const u32x4 *)&Db[p * C::DPL + row];
                }
#pragma unroll
                for (int i = 0; i < C::AN; ++i)
#pragma unroll
                    for (int j = 0; j < C::AP; ++j) acc[i][j] = mac3e(pa[i][c][kc], b[j], acc[i][j]);
            }
            SYNE_LAP(5);
            // no barrier: the next stage 1 only writes Es (its readers finished before the barrier above); the D planes
            // are rewritten only after the next barrier, which every wave reaches after this stage
        }

        // ---- epilogue: rescale, (+ residual) and NHWC store ----
#pragma unroll
        for (int i = 0; i < C::AN; ++i)
#pragma unroll
            for (int j = 0; j < C::AP; ++j) acc[i][j] *= inv_sp;
        if (C::RES) {                   // all residual adds first: every load is consumed before the first store is issued
#pragma unroll
            for (int i = 0; i < C::AN; ++i)
#pragma unroll
                for (int j = 0; j < C::AP; ++j) {
                    acc[i][j] += resv[i][j];                                // x + conv(x): same shape, same index
                    asm volatile("" : "+v"(acc[i][j]));                      // (keeps the add from being sunk into the store's branch)
                }
        }
#pragma unroll
        for (int i = 0; i < C::AN; ++i) {
            const int nt = wn + i * C::WN;
            const int n = nt * 16 + 4 * g;
            if (nt >= C::NT_O || n >= C::COUT) continue;
#pragma unroll
            for (int j = 0; j < C::AP; ++j) {
                const int pt = wp + j * C::WP;
                const int po = pt * 16 + r16;
                if (pt >= C::PT_O || po >= C::POUT || !live) continue;
                const int oy = oy0 + po / C::TW, ox = ox0 + po % C::TW;
                f32x4 v = acc[i][j];
                const size_t o = ((size_t)(f * C::HOUT + oy) * C::HOUT + ox) * C::COUT + n;
                *(f32x4 *)&Y[o] = v;
            }
        }
        ntiles_done += live ? 1 : 0;
        SYNE_LAP(6);
    }
    if (C::NG == 2 && grp == 0) __syncthreads();
    if (PROF && gtid == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&prof[i], pt_[i]);
        atomicAdd(&prof[7], ntiles_done);
    }
}

template <class C>
static void launch_early(const FusedBlockArgs &a, int B, hipStream_t s) {
    const int total = B * C::TILES_X * C::TILES_Y;
    const int wgs = (total + C::NG - 1) / C::NG;
    const int grid = wgs < 256 ? wgs : 256;            // persistent: one workgroup per CU
    if (a.prof)
        fused_block_early_kernel<C, true><<<grid, C::NW * 64, 0, s>>>(a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, B, total, a.scl_e, a.scl_p, a.prof);
    else
        fused_block_early_kernel<C><<<grid, C::NW * 64, 0, s>>>(a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, B, total, a.scl_e, a.scl_p);
}

//                       CIN  HID COUT HIN S  RES   TH  TW  NW NG WN WP
using E2 = EarlyCfg<  16,  96,  24, 60, 2, false, 10, 10, 8, 1, 2, 4>;    // features.2   60 -> 30
using E3 = EarlyCfg<  24, 144,  24, 30, 1, true,  10, 10, 8, 1, 2, 4>;    // features.3   30
using E4 = EarlyCfg<  24, 144,  32, 30, 2, false,  5,  5, 8, 2, 2, 2>;    // features.4   30 -> 15

bool launch_fused_block_early(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.We3 || !a.Wp3 || !a.scl_e || !a.scl_p) return false;
    switch (feature) {
        case 2: launch_early<E2>(a, B, s); return true;
        case 3: launch_early<E3>(a, B, s); return true;
        case 4: launch_early<E4>(a, B, s); return true;
        default: return false;
    }
}

}  // namespace syn
