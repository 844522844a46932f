// Fused inverted-residual block with both 1x1 GEMMs on the fp16 matrix instructions at fp32-equivalent accuracy
// (the late MobileNetV2 blocks, CIN % 32 == 0).  Same dataflow as fused_block.hip:
//
//   stage 1  expand   E = ReLU6(BN(X . We^T))        LDS(fp16 x2) x regs(fp16 x2) -> fp32 acc -> LDS fp32
//   stage 2  dw 3x3   D = ReLU6(BN(dw(E)))            fp32 VALU, LDS -> LDS (split into fp16 x2 on the way out)
//   stage 3  project  acc += D . Wp[:, chunk]^T       LDS(fp16 x2) x regs(fp16 x2) -> fp32 acc in VGPRs
//
// Every fp32 operand x of a GEMM is carried as two fp16 pieces x = a + b (a = fp16(x), b = fp16(x - a), both toward zero: 11 + 11
// significant bits) and each 16x16x32 block product is rebuilt from three partial products a a + a b + b a (dropped: b b <= 2^-22
// relative).  fp16 x fp16 products are exact in fp32 and v_mfma_f32_16x16x32_f16 accumulates in fp32, so the result has
// fp32-class error at HALF the matrix instructions of the exact 3-way bf16 split (6 products) this kernel used before -- which
// matters because matrix and vector instructions of the waves of a SIMD do not overlap (tools/ubench/mfma_valu_kinds.hip).
// fp16's narrow exponent: the weights of a layer are scaled by a power of two S to max |w| in [2^13, 2^14) (exact; low pieces stay
// normal), the accumulators start at S x shift, ReLU6 clamps at 6 S and the inverse scale is folded into the depthwise filter
// (expand) or applied to the accumulator (project): scl = {S, 1/S, 6 S} per layer (synergy_abi.hip).
// Weights are split and lane-ordered offline ([n_tile][k32 chunk][piece][lane][4 dwords]); activations are split once,
// where they are written to LDS (input tile in stage 0, depthwise output in stage 2).
#include "syn_internal.h"

#include <cstdlib>

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ float relu6b(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }   // one v_med3 (fminf(fmaxf()) adds a canonicalising v_max)
__device__ __forceinline__ f32x4 relu6b(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = relu6b(v[i]);
    return r;
}
constexpr int cdivb(int a, int b) { return (a + b - 1) / b; }
constexpr int rupb(int a, int b) { return cdivb(a, b) * b; }
// two floats -> packed fp16 pieces a (high) and b (low), x = a + b to 22 bits
__device__ __forceinline__ void split2b(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ f32x4 mfmab(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the three partial products of one (channel tile, pixel tile, k32 chunk), smallest terms first
__device__ __forceinline__ f32x4 mac3(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x4 c) {
    c = mfmab(a[1], b[0], c);
    c = mfmab(a[0], b[1], c);
    c = mfmab(a[0], b[0], c);
    return c;
}
}  // namespace

template <int CIN_, int HID_, int COUT_, int HIN_, int S_, bool RES_, int NF_, int HC_, int NW_, int EPB_, int WN_, int WP_, int OCC_ = 1, bool PERSIST_ = false>
struct Bf3Cfg {   // whole-image tiles only (HIN <= 15): every late block
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_, HIN = HIN_, S = S_, NF = NF_, HC = HC_, NW = NW_,
                         EPB = EPB_, WN = WN_, WP = WP_;
    static constexpr bool RES = RES_, PERSIST = PERSIST_;
    static constexpr int SLOTS = 256 * OCC_;      // resident workgroups of the device (256 CUs)
    static constexpr int NT = NW * 64;
    static constexpr int WPE = OCC_ * NW / 4;     // waves per SIMD the register budget is sized for (OCC workgroups per CU)
    static constexpr int HOUT = S == 2 ? (HIN + 1) / 2 : HIN;
    static constexpr int TH = HOUT, TW = HOUT, IH = HIN, IW = HIN;
    static constexpr int PIN = NF * IH * IW, PINP = rupb(PIN, 16);
    static constexpr int POUT = NF * TH * TW, POUTP = rupb(POUT, 16);
    static constexpr int COUTP = rupb(COUT, 16);
    static constexpr int KE = CIN / 32, KP = HC / 32;                            // k32 chunks of expand / project
    static constexpr int XSD = CIN / 2 + 4, DSD = HC / 2 + 4, ES = HC + 4;       // LDS row strides (dwords / floats)
    static constexpr int XPL = PINP * XSD, DPL = POUTP * DSD;                    // dwords per bf16 plane
    static constexpr int NT_E = HC / 16, PT_IN = PINP / 16, PG = cdivb(PT_IN, EPB), JOBS = NT_E * PG;
    static constexpr int JPW = cdivb(JOBS, NW);
    static constexpr int NT_O = COUTP / 16, PT_O = POUTP / 16;
    static constexpr int AN = cdivb(NT_O, WN), AP = cdivb(PT_O, WP);
    static constexpr int X_ITEMS = PINP * (CIN / 4), X_IPT = cdivb(X_ITEMS, NT);
    static constexpr int C4N = HC / 4, COLS = NF * TW;
    static constexpr int RS_ = NT / (C4N * COLS);
    static constexpr int RS = RS_ < 1 ? 1 : (RS_ > TH ? TH : RS_);
    static constexpr int RPS = cdivb(TH, RS), DW_THREADS = C4N * COLS * cdivb(TH, RPS);
    static constexpr int WDR_THREADS = 11 * HC / 4;
    static constexpr int LDS_DWORDS = 2 * XPL + PINP * ES + 2 * DPL + 11 * HC;
    static_assert(CIN % 32 == 0 && HC % 32 == 0 && HID % HC == 0, "k32 chunking");
    static_assert(WN * WP == NW, "wave grid");
    static_assert(!RES || (S == 1 && CIN == COUT), "residual only on stride-1 same-width blocks");
    static_assert(LDS_DWORDS * 4 * OCC_ <= 160 * 1024, "LDS budget");
    static_assert(OCC_ * NW % 4 == 0, "whole waves per SIMD");
    static_assert(WDR_THREADS <= NT, "one depthwise-weight float4 per thread");
};

// PROF: as in fused_block.hip -- wave 0 accumulates s_memtime deltas per stage into prof[0..7].
#define SYNB_LAP(i) do { if (PROF) { tn = __builtin_amdgcn_s_memtime(); pt_[i] += tn - tk; tk = tn; } } while (0)

// NS > 1 (small batches only): blockIdx.y selects one of NS slices of the block's OUTPUT channel tiles.  Every slice repeats the
// expand and depthwise stages (idle compute at small B) but streams only its share of the project weights, and each output
// element is produced by exactly the same instruction sequence as with NS = 1 -- results do not depend on the batch size.
// PERSIST: the grid is the number of resident workgroups and a workgroup walks face groups blockIdx.x, + gridDim.x, ...; the input
// tile of the next group is fetched into registers while the current one computes (the staging -- 10-19 % of a workgroup's life at
// one group per workgroup -- shrinks to the split and the LDS writes), and the expand weights of chunk 0 arrive through the
// wrap-around of the per-chunk prefetch.
// the whole block as a device function (`smem`: the workgroup's C::LDS_DWORDS dwords), so that two blocks can share a launch (fused_pair_f16_kernel)
template <class C, bool PROF, int NS, bool PERSIST>
__device__ __forceinline__ void f16_block(unsigned *smem,
    const float *__restrict__ X, const unsigned *__restrict__ We3 /*[HID/16][KE][2][64][4]*/, const float *__restrict__ e_shift,
    const float *__restrict__ Wd, const float *__restrict__ d_shift, const unsigned *__restrict__ Wp3 /*[COUTP/16][HID/32][2][64][4]*/,
    const float *__restrict__ p_shift, float *__restrict__ Y, int B, const float *__restrict__ scl_e, const float *__restrict__ scl_p,
    unsigned long long *prof) {
    unsigned long long pt_[7] = {0, 0, 0, 0, 0, 0, 0}, tk = PROF ? __builtin_amdgcn_s_memtime() : 0ull, tn = 0;
    unsigned *Xb = smem;                                         // 2 planes [PINP][XSD]
    float *Es = reinterpret_cast<float *>(Xb + 2 * C::XPL);      // [PINP][ES] fp32 (scaled by Se)
    unsigned *Db = reinterpret_cast<unsigned *>(Es + C::PINP * C::ES);   // 2 planes [POUTP][DSD]
    float *Wds = reinterpret_cast<float *>(Db + 2 * C::DPL);     // [11][HC]: 9 taps / Se | (unused) | shift
    const float Se = scl_e[0], inv_se = scl_e[1], c6e = scl_e[2], Sp = scl_p[0], inv_sp = scl_p[1];
    constexpr int NT = C::NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int wn = wave % C::WN, wp = wave / C::WN;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    int f0 = blockIdx.x * C::NF;
    static_assert(!PERSIST || (NS == 1 && !PROF), "the persistent schedule is the plain large-batch one");
    static_assert(C::NT_O % NS == 0, "output channel tiles split evenly");
    constexpr int NTO_S = C::NT_O / NS, AN_S = cdivb(NTO_S, C::WN);            // channel tiles of this slice, per wave column
    const int nt_base = NS > 1 ? (int)blockIdx.y * NTO_S : 0;

    // ---- prefetch: expand weights of chunk 0 and its depthwise filter (registers) ----
    u32x4 a1[C::JPW][C::KE][2];
    f32x4 e1h[C::JPW];
    u32x4 a3[C::AN][C::KP][2];
    f32x4 wdr = z4;
    auto fetch_a1 = [&](int hc0) {
#pragma unroll
        for (int jj = 0; jj < C::JPW; ++jj) {
            const int job = wave + jj * C::NW;
            if (job < C::JOBS) {
                const unsigned *wa = We3 + ((size_t)(hc0 / 16 + job % C::NT_E) * C::KE) * 512 + lane * 4;
#pragma unroll
                for (int kc = 0; kc < C::KE; ++kc)
#pragma unroll
                    for (int p = 0; p < 2; ++p) a1[jj][kc][p] = *(const u32x4 *)(wa + (kc * 2 + p) * 256);
                e1h[jj] = *(const f32x4 *)&e_shift[hc0 + (job % C::NT_E) * 16 + 4 * g] * Se;
            }
        }
        if (tid < C::WDR_THREADS) {
            const int row = tid / (C::HC / 4), c4 = tid % (C::HC / 4);
            if (row < 9) wdr = *(const f32x4 *)&Wd[row * C::HID + hc0 + 4 * c4] * inv_se;
            else if (row == 10) wdr = *(const f32x4 *)&d_shift[hc0 + 4 * c4];
        }
    };
    auto fetch_a3 = [&](int hc0) {
#pragma unroll
        for (int i = 0; i < AN_S; ++i) {
            int nt = wn + i * C::WN;
            nt = nt_base + (nt < NTO_S ? nt : 0);
            const unsigned *wa = Wp3 + ((size_t)nt * (C::HID / 32) + hc0 / 32) * 512 + lane * 4;
#pragma unroll
            for (int kc = 0; kc < C::KP; ++kc)
#pragma unroll
                for (int p = 0; p < 2; ++p) a3[i][kc][p] = *(const u32x4 *)(wa + (kc * 2 + p) * 256);
        }
    };
    fetch_a1(0);

    // ---- stage 0: input tile -> three bf16 planes (exact split), zero beyond the batch / tile ----
    // All loads first, branch-free (indices clamped, the value zeroed afterwards): with a branch around each load the compiler
    // waited for every load before issuing the next one -- X_IPT serialised L2 round trips at the start of every workgroup.
    f32x4 xv[C::X_IPT];
    auto load_x = [&](int fb) {
#pragma unroll
        for (int ii = 0; ii < C::X_IPT; ++ii) {
            const int it = tid + ii * NT;
            const int c4 = it % (C::CIN / 4);
            int p = it / (C::CIN / 4);
            p = p < C::PIN ? p : C::PIN - 1;
            int f = fb + p / (C::IH * C::IW);
            f = f < B ? f : B - 1;
            xv[ii] = *(const f32x4 *)&X[((size_t)f * C::IH * C::IW + p % (C::IH * C::IW)) * C::CIN + 4 * c4];
        }
    };
    auto park_x = [&](int fb) {
#pragma unroll
        for (int ii = 0; ii < C::X_IPT; ++ii) asm volatile("" : "+v"(xv[ii]));     // (keeps the loads from being sunk to their uses)
#pragma unroll
        for (int ii = 0; ii < C::X_IPT; ++ii) {
            const int it = tid + ii * NT;
            if (it >= C::X_ITEMS) break;
            const int c4 = it % (C::CIN / 4), p = it / (C::CIN / 4);
            const bool real = p < C::PIN && fb + p / (C::IH * C::IW) < B;
            const f32x4 v = real ? xv[ii] : z4;
            unsigned a0, b0, a1_, b1;
            split2b(v[0], v[1], a0, b0);
            split2b(v[2], v[3], a1_, b1);
            *(u32x2 *)&Xb[0 * C::XPL + p * C::XSD + 2 * c4] = (u32x2){a0, a1_};
            *(u32x2 *)&Xb[1 * C::XPL + p * C::XSD + 2 * c4] = (u32x2){b0, b1};
        }
    };
    load_x(f0);
    if (C::POUTP > C::POUT)
        for (int it = tid; it < 2 * (C::POUTP - C::POUT) * C::DSD; it += NT) {
            const int p = it / ((C::POUTP - C::POUT) * C::DSD), r = it % ((C::POUTP - C::POUT) * C::DSD);
            Db[p * C::DPL + C::POUT * C::DSD + r] = 0u;
        }
    f32x4 psh[C::AN];
#pragma unroll
    for (int i = 0; i < AN_S; ++i) {
        const int n = (nt_base + wn + i * C::WN) * 16 + 4 * g;
        psh[i] = n < C::COUTP ? *(const f32x4 *)&p_shift[n] * Sp : z4;
    }
    const int gstep = (int)gridDim.x * C::NF;
    for (;;) {                                  // face groups of this workgroup (exactly one unless PERSIST)
    park_x(f0);
    const int f0n = f0 + gstep;
    const bool more = PERSIST && f0n < B;
    if (PERSIST) load_x(more ? f0n : f0);       // unconditional: in flight during all the chunks of this group
    f32x4 acc[C::AN][C::AP];
#pragma unroll
    for (int i = 0; i < AN_S; ++i)
#pragma unroll
        for (int j = 0; j < C::AP; ++j) acc[i][j] = psh[i];
    __syncthreads();
    SYNB_LAP(0);

    for (int hc0 = 0; hc0 < C::HID; hc0 += C::HC) {
        if (tid < C::WDR_THREADS) *(f32x4 *)&Wds[4 * tid] = wdr;
        // ---- stage 1: expand 1x1 (bf16 x3) + BN shift + ReLU6 -> Es (fp32) ----
#pragma unroll
        for (int jj = 0; jj < C::JPW; ++jj) {
            const int job = wave + jj * C::NW;
            if (job >= C::JOBS) break;
            const int nt = job % C::NT_E, pg = job / C::NT_E;
            f32x4 ea[C::EPB];
#pragma unroll
            for (int q = 0; q < C::EPB; ++q) ea[q] = e1h[jj];
            auto ldb = [&](int kc, u32x4(&b)[C::EPB][2]) {
#pragma unroll
                for (int q = 0; q < C::EPB; ++q) {
                    const int pt = pg * C::EPB + q;
                    const int row = ((pt < C::PT_IN ? pt : 0) * 16 + r16) * C::XSD + kc * 16 + 4 * g;
#pragma unroll
                    for (int p = 0; p < 2; ++p) b[q][p] = *(const u32x4 *)&Xb[p * C::XPL + row];
                }
            };
            u32x4 bq[2][C::EPB][2];                     // pixel operands of the current / next k32 chunk (ping-pong)
            ldb(0, bq[0]);
#pragma unroll
            for (int kc = 0; kc < C::KE; ++kc) {
                if (kc + 1 < C::KE) ldb(kc + 1, bq[(kc + 1) & 1]);
#pragma unroll
                for (int q = 0; q < C::EPB; ++q) ea[q] = mac3(a1[jj][kc], bq[kc & 1][q], ea[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < C::EPB; ++q) {
                const int pt = pg * C::EPB + q;
                if (pt < C::PT_IN) {
                    f32x4 e;
#pragma unroll
                    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_fmed3f(ea[q][i], 0.0f, c6e);
                    *(f32x4 *)&Es[(pt * 16 + r16) * C::ES + nt * 16 + 4 * g] = e;
                }
            }
        }
        fetch_a3(hc0);
        SYNB_LAP(1);
        __syncthreads();
        SYNB_LAP(2);
        // ---- stage 2: depthwise 3x3 + BN shift + ReLU6 (fp32 VALU), output split into bf16 x3 planes ----
        for (int t = tid; t < C::DW_THREADS; t += NT) {
            const int c4 = t % C::C4N, q = t / C::C4N;
            const int col = q % C::COLS, seg = q / C::COLS;
            const int fi = col / C::TW, oxl = col % C::TW;
            const int ixb = oxl * C::S - 1;
            f32x4 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x4 *)&Wds[k * C::HC + 4 * c4];
            if (ixb < 0) { w[0] = z4; w[3] = z4; w[6] = z4; }
            if (ixb + 2 >= C::HIN) { w[2] = z4; w[5] = z4; w[8] = z4; }
            const int lx1 = ixb + 1;
            const int lx0 = lx1 > 0 ? lx1 - 1 : 0, lx2 = lx1 + 1 < C::IW ? lx1 + 1 : C::IW - 1;
            const f32x4 sh = *(const f32x4 *)&Wds[10 * C::HC + 4 * c4];
            const float *ebase = Es + (size_t)fi * C::IH * C::IW * C::ES + 4 * c4;
            f32x4 rb[3][3];
            auto load_row = [&](int iy, f32x4(&dst)[3]) {
                const bool ok = (unsigned)iy < (unsigned)C::HIN;
                const int ly = iy < 0 ? 0 : (iy > C::IH - 1 ? C::IH - 1 : iy);
                const float *er = ebase + ly * C::IW * C::ES;
                dst[0] = *(const f32x4 *)(er + lx0 * C::ES);
                dst[1] = *(const f32x4 *)(er + lx1 * C::ES);
                dst[2] = *(const f32x4 *)(er + lx2 * C::ES);
                if (!ok) { dst[0] = z4; dst[1] = z4; dst[2] = z4; }
            };
#pragma unroll
            for (int r = 0; r < C::RPS; ++r) {
                const int oyl = seg * C::RPS + r;
                if (oyl >= C::TH) break;
                const int iyb = oyl * C::S - 1;
                if (r == 0) {
                    load_row(iyb, rb[0]); load_row(iyb + 1, rb[1]); load_row(iyb + 2, rb[2]);
                } else if (C::S == 1) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { rb[0][k] = rb[1][k]; rb[1][k] = rb[2][k]; }
                    load_row(iyb + 2, rb[2]);
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) rb[0][k] = rb[2][k];
                    load_row(iyb + 1, rb[1]); load_row(iyb + 2, rb[2]);
                }
                f32x4 a = sh;
                a += rb[0][0] * w[0]; a += rb[0][1] * w[1]; a += rb[0][2] * w[2];
                a += rb[1][0] * w[3]; a += rb[1][1] * w[4]; a += rb[1][2] * w[5];
                a += rb[2][0] * w[6]; a += rb[2][1] * w[7]; a += rb[2][2] * w[8];
                a = relu6b(a);
                const int po = (fi * C::TH + oyl) * C::TW + oxl;
                unsigned a0, b0, a1_, b1;
                split2b(a[0], a[1], a0, b0);
                split2b(a[2], a[3], a1_, b1);
                *(u32x2 *)&Db[0 * C::DPL + po * C::DSD + 2 * c4] = (u32x2){a0, a1_};
                *(u32x2 *)&Db[1 * C::DPL + po * C::DSD + 2 * c4] = (u32x2){b0, b1};
            }
        }
        fetch_a1(hc0 + C::HC < C::HID ? hc0 + C::HC : 0);
        SYNB_LAP(3);
        __syncthreads();
        SYNB_LAP(4);
        // ---- stage 3: project 1x1 (bf16 x3), K = this hidden chunk, accumulators stay in registers ----
        {
            auto ldb = [&](int kc, u32x4(&b)[C::AP][2]) {
#pragma unroll
                for (int j = 0; j < C::AP; ++j) {
                    const int pt = wp + j * C::WP;
                    const int row = ((pt < C::PT_O ? pt : 0) * 16 + r16) * C::DSD + kc * 16 + 4 * g;
#pragma unroll
                    for (int p = 0; p < 2; ++p) b[j][p] = *(const u32x4 *)&Db[p * C::DPL + row];
                }
            };
            u32x4 bq[2][C::AP][2];
            ldb(0, bq[0]);
#pragma unroll
            for (int kc = 0; kc < C::KP; ++kc) {
                if (kc + 1 < C::KP) ldb(kc + 1, bq[(kc + 1) & 1]);
#pragma unroll
                for (int i = 0; i < AN_S; ++i)
#pragma unroll
                    for (int j = 0; j < C::AP; ++j) acc[i][j] = mac3(a3[i][kc], bq[kc & 1][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        SYNB_LAP(5);
    }

    // the prefetched tile (and the wrapped-around weight fetch) are waited for BEFORE the output stores: vector memory retires in
    // order, a wait placed after them would also wait for their acknowledgements
    if (PERSIST) __builtin_amdgcn_s_waitcnt(0x0F70);
    // ---- epilogue: (+ residual rebuilt exactly from the three input planes) and NHWC store ----
#pragma unroll
    for (int i = 0; i < AN_S; ++i) {
        const int nt = wn + i * C::WN;
        const int n = (nt_base + nt) * 16 + 4 * g;
        if (nt >= NTO_S || n >= C::COUT) continue;
#pragma unroll
        for (int j = 0; j < C::AP; ++j) {
            const int pt = wp + j * C::WP;
            const int po = pt * 16 + r16;
            if (pt >= C::PT_O || po >= C::POUT) continue;
            const int f = f0 + po / (C::TH * C::TW);
            if (f >= B) continue;
            f32x4 v = acc[i][j] * inv_sp;
            const size_t at = ((size_t)f * C::TH * C::TW + po % (C::TH * C::TW)) * C::COUT + n;
            if (C::RES) v += *(const f32x4 *)&X[at];         // S == 1, CIN == COUT: same index (the two fp16 pieces in LDS are not the exact input)
            *(f32x4 *)&Y[at] = v;
        }
    }
    SYNB_LAP(6);
    if (!more) break;
    f0 = f0n;
    __syncthreads();                            // the planes of this group (residual reads) are done before the next tile is written
    }
    if (PROF && tid == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&prof[i], pt_[i]);
        atomicAdd(&prof[7], 1ull);
    }
}

template <class C, bool PROF = false, int NS = 1, bool PERSIST = false>
__global__ __launch_bounds__(C::NW * 64) __attribute__((amdgpu_waves_per_eu(C::WPE, C::WPE))) void fused_block_f16_kernel(
    const float *__restrict__ X, const unsigned *__restrict__ We3, const float *__restrict__ e_shift, const float *__restrict__ Wd,
    const float *__restrict__ d_shift, const unsigned *__restrict__ Wp3, const float *__restrict__ p_shift, float *__restrict__ Y, int B,
    const float *__restrict__ scl_e, const float *__restrict__ scl_p, unsigned long long *prof = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DWORDS];
    f16_block<C, PROF, NS, PERSIST>(smem, X, We3, e_shift, Wd, d_shift, Wp3, p_shift, Y, B, scl_e, scl_p, prof);
}

// Two consecutive blocks of one configuration in ONE launch (round 5, small batches: features.5 + 6 of a batch below 513 faces, where the row-marching
// pair of fused_block_rm.hip does not apply).  One workgroup per face (group): the second block reads what this workgroup's own waves stored,
// in the same layout and rows -- a workgroup barrier with workgroup-scope release / acquire instead of a kernel boundary.
struct F16StageArgs {
    const float *X; const unsigned *We3; const float *e_shift, *Wd, *d_shift; const unsigned *Wp3; const float *p_shift; float *Y; const float *scl_e, *scl_p;
};
template <class C>
__global__ __launch_bounds__(C::NW * 64) __attribute__((amdgpu_waves_per_eu(C::WPE, C::WPE))) void fused_pair_f16_kernel(F16StageArgs a, F16StageArgs b, int B) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DWORDS];
    f16_block<C, false, 1, false>(smem, a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, B, a.scl_e, a.scl_p, nullptr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    f16_block<C, false, 1, false>(smem, b.X, b.We3, b.e_shift, b.Wd, b.d_shift, b.Wp3, b.p_shift, b.Y, B, b.scl_e, b.scl_p, nullptr);
}

template <class C, int NS>
static void launch_f16_sliced(const FusedBlockArgs &a, int B, hipStream_t s) {
    const dim3 grid((B + C::NF - 1) / C::NF, NS);
    fused_block_f16_kernel<C, false, NS><<<grid, C::NW * 64, 0, s>>>(a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, B, a.scl_e, a.scl_p);
}

template <class C>
static void launch_f16(const FusedBlockArgs &a, int B, hipStream_t s) {
    const int grid = (B + C::NF - 1) / C::NF;
    if (a.prof)
        fused_block_f16_kernel<C, true><<<grid, C::NW * 64, 0, s>>>(a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, B, a.scl_e, a.scl_p, a.prof);
    else {
        if constexpr (C::PERSIST) {
            if (grid > C::SLOTS) {
                fused_block_f16_kernel<C, false, 1, true><<<C::SLOTS, C::NW * 64, 0, s>>>(a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, B, a.scl_e, a.scl_p);
                return;
            }
        }
        fused_block_f16_kernel<C><<<grid, C::NW * 64, 0, s>>>(a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, B, a.scl_e, a.scl_p);
    }
}

//                      CIN  HID COUT HIN S  RES   NF  HC NW EPB WN WP [OCC [PERSIST]]
// PERSIST where measured faster (features.5-7: -6...8 %; features.8-10 neutral; .11-.14 would spill: +44 registers)
using B5 = Bf3Cfg<   32, 192,  32, 15, 1, true,   1, 32, 8, 4, 2, 4, 1, true>;    // features.5,6
using B7 = Bf3Cfg<   32, 192,  64, 15, 2, false,  1, 32, 8, 4, 4, 2, 1, true>;    // features.7
using B8 = Bf3Cfg<   64, 384,  64,  8, 1, true,   1, 64, 4, 4, 4, 1, 2>;    // features.8-10
using B11 = Bf3Cfg<  64, 384,  96,  8, 1, false,  1, 64, 4, 4, 2, 2, 2>;    // features.11
using B12 = Bf3Cfg<  96, 576,  96,  8, 1, true,   1, 32, 4, 2, 2, 2, 2>;    // features.12,13
using B14 = Bf3Cfg<  96, 576, 160,  8, 2, false,  1, 64, 4, 4, 4, 1, 2>;    // features.14
using B15 = Bf3Cfg< 160, 960, 160,  4, 1, true,   4, 64, 4, 4, 2, 2>;    // features.15,16
using B17 = Bf3Cfg< 160, 960, 320,  4, 1, false,  4, 64, 4, 4, 4, 1>;    // features.17

// features.5 + 6 of a small batch in one launch (a[0..1]; one workgroup per face, not persistent: grids up to the resident 256 workgroups); false: one by one
bool launch_fused_pair_f16(const FusedBlockArgs &a, const FusedBlockArgs &b, int B, hipStream_t s) {
    static const bool on = test_knob("f16_pair56", 1) != 0;
    if (!on || a.prof || b.prof || !a.We3 || !a.Wp3 || !a.scl_e || !a.scl_p || !b.We3 || !b.Wp3 || !b.scl_e || !b.scl_p) return false;
    const int grid = (B + B5::NF - 1) / B5::NF;
    if (grid > B5::SLOTS) return false;                  // (larger batches take the persistent single launches, or the row-marching pair from 513 faces)
    fused_pair_f16_kernel<B5><<<grid, B5::NW * 64, 0, s>>>(F16StageArgs{a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Wp3, a.p_shift, a.Y, a.scl_e, a.scl_p},
                                                          F16StageArgs{b.X, b.We3, b.e_shift, b.Wd, b.d_shift, b.Wp3, b.p_shift, b.Y, b.scl_e, b.scl_p}, B);
    return true;
}

constexpr int kSliceMaxGrid = 48;      // workgroups (of 4 faces) below which the late blocks are sliced over output channels

bool launch_fused_block_f16(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.We3 || !a.Wp3 || !a.scl_e || !a.scl_p) return false;
    switch (feature) {
        case 5: case 6: launch_f16<B5>(a, B, s); return true;
        case 7: launch_f16<B7>(a, B, s); return true;
        case 8: case 9: case 10: launch_f16<B8>(a, B, s); return true;
        case 11: launch_f16<B11>(a, B, s); return true;
        case 12: case 13: launch_f16<B12>(a, B, s); return true;
        case 14: launch_f16<B14>(a, B, s); return true;
        // few faces: one workgroup would stream 1.8-2.8 MB of weights through a single CU; slice the output channels over 5
        case 15: case 16: if (!a.prof && (B + 3) / 4 <= kSliceMaxGrid) launch_f16_sliced<B15, 5>(a, B, s); else launch_f16<B15>(a, B, s); return true;
        case 17: if (!a.prof && (B + 3) / 4 <= kSliceMaxGrid) launch_f16_sliced<B17, 5>(a, B, s); else launch_f16<B17>(a, B, s); return true;
        default: return false;
    }
}

}  // namespace syn
