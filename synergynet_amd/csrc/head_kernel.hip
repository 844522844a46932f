// Fused network tail for gfx950: features.18 (1x1 conv 320->1280 + BN + ReLU6), adaptive_avg_pool2d(4x4 -> 1)
// and the three linear heads (12 + 40 + 10 outputs) in ONE kernel
// (reference mobilenetv2_backbone.py:140,177-188).  The 1280x4x4 activation (82 KB per face) never leaves the CU.
//
// A workgroup owns 4 faces = 64 pixels (4 MFMA pixel tiles) held in LDS; each of its 4 waves walks 20 of the 80
// output-channel tiles: per tile 20 k-chunks x 4 pixel tiles x 4 = 320 x v_mfma_f32_16x16x4_f32 with the weight
// fragments of the NEXT channel tile prefetched into registers meanwhile (weights in MFMA lane order, 1 KiB per
// fetch).  A lane owns 4 channels of one pixel, so the pool is a 16-lane butterfly (the 16 pixels of a face are
// the 16 lanes of a DPP row) and the pooled 1280-vector of each face lands in LDS for the 62 dot products.
#include <cstdlib>

#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int NF = 2, PX = NF * 16, K = 320, KCH = K / 16, N = 1280, NTL = N / 16, XS = K + 4;
__device__ __forceinline__ float r6h(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }
// sum over the 16 lanes of a DPP row, result in every lane of the row: 4 VALU adds, no LDS traffic
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ float wave64_sum(float v) {          // uniform result (via 4 v_readlane)
    const int b = __builtin_bit_cast(int, row16_sum(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}
}  // namespace

__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ X /*[B,16,320]*/, const float *__restrict__ Wpk,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   const float *__restrict__ Wfc /*[64,1280]*/, const float *__restrict__ bfc,
                                                   float *__restrict__ param, float *__restrict__ pool, int B) {
    __shared__ __attribute__((aligned(16))) float Xs[PX * XS];
    __shared__ __attribute__((aligned(16))) float Ps[NF * N];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int f0 = blockIdx.x * NF;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 an[KCH], hn;                             // prefetched: next channel tile's weights + BN shift (scale is folded in)
    auto fetch = [&](int nt) {
        const float *w = Wpk + (size_t)nt * KCH * 256 + lane * 4;
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) an[kc] = *(const f32x4 *)(w + kc * 256);
        hn = *(const f32x4 *)&shift[nt * 16 + 4 * g];
    };
    fetch(wave);
    for (int it = tid; it < PX * (K / 4); it += 256) {
        const int c4 = it % (K / 4), p = it / (K / 4);
        const int f = f0 + (p >> 4);
        f32x4 v = z4;
        if (f < B) v = *(const f32x4 *)&X[((size_t)f * 16 + (p & 15)) * K + 4 * c4];
        *(f32x4 *)&Xs[p * XS + 4 * c4] = v;
    }
    __syncthreads();

    for (int nt = wave; nt < NTL; nt += 4) {
        f32x4 a[KCH];
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) a[kc] = an[kc];
        const f32x4 sh = hn;
        if (nt + 4 < NTL) fetch(nt + 4);
        f32x4 acc[NF];                                // BN shift = accumulator start
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[j] = sh;
        // LDS operand reads run exactly one k-chunk ahead of the MFMAs; the scheduling barriers stop the compiler
        // from hoisting all 80 reads to the top (that spilled the weight fragments to scratch)
        f32x4 bc[NF], bn[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) bc[j] = *(const f32x4 *)&Xs[(j * 16 + r16) * XS + 4 * g];
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            if (kc + 1 < KCH) {
#pragma unroll
                for (int j = 0; j < NF; ++j) bn[j] = *(const f32x4 *)&Xs[(j * 16 + r16) * XS + (kc + 1) * 16 + 4 * g];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc][s], bc[j][s], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) bc[j] = bn[j];
            __builtin_amdgcn_sched_barrier(0);
        }
        // BN + ReLU6, then mean over the 16 pixels of each face = butterfly over the 16 lanes sharing g
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = r6h(acc[j][t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = row16_sum(v[t]);
            if (r16 == 0) *(f32x4 *)&Ps[j * N + nt * 16 + 4 * g] = v * 0.0625f;
        }
    }
    __syncthreads();
    // pooled feature out (optional), then the heads: wave w computes outputs w, w+4, ... for all NF faces, so
    // every row of Wfc is fetched once per workgroup; 64-lane dot products reduced with DPP + readlane
    if (pool) {
        for (int it = tid; it < NF * (N / 4); it += 256) {
            const int j = it / (N / 4), c4 = it % (N / 4);
            if (f0 + j < B) *(f32x4 *)&pool[(size_t)(f0 + j) * N + 4 * c4] = *(const f32x4 *)&Ps[j * N + 4 * c4];
        }
    }
    {
        f32x4 xv[NF][N / 256];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int i = 0; i < N / 256; ++i) xv[j][i] = *(const f32x4 *)&Ps[j * N + (i * 64 + lane) * 4];
#pragma unroll 2
        for (int o = wave; o < kParam; o += 4) {
            const float *wr = Wfc + (size_t)o * N;
            f32x4 wv[N / 256];
#pragma unroll
            for (int i = 0; i < N / 256; ++i) wv[i] = *(const f32x4 *)&wr[(i * 64 + lane) * 4];
            const float bo = bfc[o];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < N / 256; ++i)
                    a += wv[i][0] * xv[j][i][0] + wv[i][1] * xv[j][i][1] + wv[i][2] * xv[j][i][2] + wv[i][3] * xv[j][i][3];
                const float tot = wave64_sum(a);
                if (lane == 0 && f0 + j < B) param[(size_t)(f0 + j) * kParam + o] = tot + bo;
            }
        }
    }
}

void launch_head(const float *X, const float *Wpk, const float *scale, const float *shift, const float *Wfc,
                 const float *bfc, float *param, float *pool, int B, hipStream_t s) {
    head_kernel<<<(B + NF - 1) / NF, 256, 0, s>>>(X, Wpk, scale, shift, Wfc, bfc, param, pool, B);
}


// =====================================================================================
// Same tail, but the 320 -> 1280 GEMM runs on v_mfma_f32_16x16x32_f16 with fp32-equivalent accuracy: every fp32 operand x is
// carried as two fp16 pieces x = a + b (a = fp16(x), b = fp16(x - a), toward zero: 22 significant bits) and the product is rebuilt
// from three partial products a a + a b + b a (dropped: b b <= 2^-22).  fp16 x fp16 products are exact in fp32 and the MFMA
// accumulates in fp32.  The weights are scaled by a power of two S to the top of fp16's range (their low pieces stay normal), the
// accumulators start at S x shift and are rescaled before ReLU6.  Weights are split and lane-ordered offline; activations are
// split once when the input tile is staged into LDS.  (Four faces per workgroup -- half the weight traffic per face -- measured
// slower than two, 74 vs 65 us at B = 1024: one 4-wave workgroup per CU hides less than two.)
// =====================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int KC32 = K / 32;            // 10 k-chunks of 32
constexpr int XSD = K / 2 + 4;          // dwords per pixel row of one fp16 plane (164: 16-byte aligned, 4 mod 64 banks)
constexpr int PLANE = PX * XSD;         // dwords per plane
// two floats -> packed fp16 pieces a (high) and b (low)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
}  // namespace

// The 62 head rows of NFK faces whose pooled 1280-vectors are in `xv`: wave w of NWV computes outputs w + NWV * (rbase + r), r < NR;
// the next row's weights are fetched while the current one is reduced.  One function for the fused and the sliced tail,
// so a parameter is produced by the same instruction sequence at every batch size.
template <int NR, int NFK = NF, int NWV = 4>
__device__ __forceinline__ void fc_rows(const f32x4 (&xv)[NFK][N / 256], const float *__restrict__ Wfc, const float *__restrict__ bfc,
                                        float *__restrict__ param, int f0, int B, int rbase, int wave, int lane) {
    f32x4 wq[2][N / 256];
    auto ldw = [&](int o, f32x4(&w)[N / 256]) {
        const float *wr = Wfc + (size_t)(o < kParam ? o : kParam - 1) * N;
#pragma unroll
        for (int i = 0; i < N / 256; ++i) w[i] = *(const f32x4 *)&wr[(i * 64 + lane) * 4];
    };
    ldw(wave + NWV * rbase, wq[0]);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int o = wave + NWV * (rbase + r);
        if (r + 1 < NR) ldw(o + NWV, wq[(r + 1) & 1]);
        if (o >= kParam) continue;
        const f32x4(&wv)[N / 256] = wq[r & 1];
        const float bo = bfc[o];
#pragma unroll
        for (int j = 0; j < NFK; ++j) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < N / 256; ++i)
                a += wv[i][0] * xv[j][i][0] + wv[i][1] * xv[j][i][1] + wv[i][2] * xv[j][i][2] + wv[i][3] * xv[j][i][3];
            const float tot = wave64_sum(a);
            if (lane == 0 && f0 + j < B) param[(size_t)(f0 + j) * kParam + o] = tot + bo;
        }
    }
}

constexpr int kFcRounds = (kParam + 3) / 4;      // 16 rounds of 4 rows

// NS > 1 (few faces): blockIdx.y owns NTL / NS of the 80 output-channel tiles, the pooled slice goes to `pool` in HBM and
// head_fc_kernel finishes; NS == 1: the whole tail in this launch.
// NFK faces and NWV waves per workgroup: 2 x 4 (two workgroups per CU) for small and medium batches; 4 x 8 once the batch fills the
// chip with four faces per CU -- every workgroup streams the 1.6 MB of weight fragments from L2, so four faces per workgroup halve
// that traffic (0.84 -> 0.42 GB per launch at B = 1024), and eight waves keep two per SIMD (four faces on FOUR waves measured
// slower than two: 74 vs 65 us).  The per-output arithmetic is the same in every configuration.
// SIN (round 5, small batches): the input tile is features.17's output as the S hidden-slice partial sums of fused_block_lb4.hip's sliced schedule; the
// staging below adds them (slice 0 + 1 + ..., x 1 / (16 Sp), + BN shift: lb4_reduce_kernel's arithmetic and order, hence its bits) instead of a reduce
// launch in front of this kernel.
template <int NS, int NFK = NF, int NWV = 4, bool SIN = false>
__global__ __launch_bounds__(NWV * 64) void head_f16x2_kernel(const float *__restrict__ X /*[B,16,320]*/,
                                                          const unsigned *__restrict__ Wb3 /*[80][10][2][64][4] dwords, {S, 1/S}*/,
                                                          const float *__restrict__ shift, const float *__restrict__ Wfc,
                                                          const float *__restrict__ bfc, float *__restrict__ param,
                                                          float *__restrict__ pool, int B, HeadSliced hs = HeadSliced{nullptr, 0, nullptr, nullptr}) {
    constexpr int PXK = NFK * 16, PLANEK = PXK * XSD, NT = NWV * 64;
    __shared__ __attribute__((aligned(16))) unsigned Xb[2 * PLANEK];
    __shared__ __attribute__((aligned(16))) float Ps[NFK * N];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int gsw = g ^ (((r16 >> 2) ^ (r16 >> 3)) & 1);      // (the chunk swizzle of the staging below)
    const int f0 = blockIdx.x * NFK;
    const float S = __builtin_bit_cast(float, Wb3[80 * 10 * 2 * 256]), inv_s = __builtin_bit_cast(float, Wb3[80 * 10 * 2 * 256 + 1]);
    constexpr int NTS = NTL / NS;
    static_assert(NTL % NS == 0 && NTS % NWV == 0, "whole rounds of the waves per slice");
    const int nt0 = NS > 1 ? (int)blockIdx.y * NTS : 0, nt_end = nt0 + NTS;

    u32x4 ring[5][2];                                   // weight pieces of 5 k-chunks in flight
    // the 1.6 MB of weight fragments (and the heads' rows) into this XCD's L2 before the walk (syn_internal.h l2_touch; large batches: NS == 1):
    // measured 1 us SLOWER (50.9 against 49.8 us, gpurun_out/r5c3) -- the ring of five chunks in flight already covers the misses.  Off.
#ifndef SYN_HEAD_L2_TOUCH
#define SYN_HEAD_L2_TOUCH 0
#endif
#define SYN_L2_TOUCH SYN_HEAD_L2_TOUCH
    unsigned sink = 0;
    if (SYN_L2_TOUCH && NS == 1) {
        const unsigned gi = (blockIdx.x >> 3) * (unsigned)NT + tid, nth = ((gridDim.x + 7) >> 3) * (unsigned)NT;
        l2_touch(Wb3, 80u * 10u * 2u * 1024u, gi, nth, sink);
        l2_touch(Wfc, 64u * 1280u * 4u, gi, nth, sink);
    }
    auto lda = [&](int nt, int kc, u32x4(&dst)[2]) {
        const unsigned *w = Wb3 + ((size_t)(nt * KC32 + kc) * 2) * 256 + lane * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) dst[p] = *(const u32x4 *)(w + p * 256);
    };
#pragma unroll
    for (int kc = 0; kc < 5; ++kc) lda(nt0 + wave, kc, ring[kc]);
    // input tile -> three bf16 planes in LDS (the split happens exactly once per element).  All ten loads of a thread are
    // issued before the first is consumed, branch-free (face index clamped, value zeroed afterwards): as a plain loop with
    // `if (f < B)` around the load it compiled to ten serialised load -> s_waitcnt vmcnt(0) -> split round trips.
    {
        constexpr int XI = PXK * (K / 4) / NT;
        static_assert(PXK * (K / 4) % NT == 0, "whole rounds");
        f32x4 xv[XI];
        if constexpr (SIN) {
            const size_t total = (size_t)B * 16 * K;                // floats per slice
            size_t at[XI];
#pragma unroll
            for (int ii = 0; ii < XI; ++ii) {
                const int it = tid + ii * NT, c4 = it % (K / 4), p = it / (K / 4);
                int f = f0 + (p >> 4);
                f = f < B ? f : B - 1;
                at[ii] = ((size_t)f * 16 + (p & 15)) * K + 4 * c4;
                xv[ii] = *(const f32x4 *)&hs.part[at[ii]];
            }
            for (int sl = 1; sl < hs.S; sl += 2) {                  // two slices (2 XI loads) in flight per round
                const bool two = sl + 1 < hs.S;                     // (uniform)
                f32x4 t0[XI], t1[XI];
#pragma unroll
                for (int ii = 0; ii < XI; ++ii) t0[ii] = *(const f32x4 *)&hs.part[(size_t)sl * total + at[ii]];
#pragma unroll
                for (int ii = 0; ii < XI; ++ii) t1[ii] = *(const f32x4 *)&hs.part[(size_t)(two ? sl + 1 : sl) * total + at[ii]];
#pragma unroll
                for (int ii = 0; ii < XI; ++ii) { xv[ii] += t0[ii]; if (two) xv[ii] += t1[ii]; }
            }
            const float inv_p = *hs.inv_p;
#pragma unroll
            for (int ii = 0; ii < XI; ++ii) {
                const int c4 = (tid + ii * NT) % (K / 4);
                xv[ii] = xv[ii] * inv_p + *(const f32x4 *)&hs.p_shift[4 * c4];
            }
        } else {
#pragma unroll
        for (int ii = 0; ii < XI; ++ii) {
            const int it = tid + ii * NT, c4 = it % (K / 4), p = it / (K / 4);
            int f = f0 + (p >> 4);
            f = f < B ? f : B - 1;
            xv[ii] = *(const f32x4 *)&X[((size_t)f * 16 + (p & 15)) * K + 4 * c4];
        }
        }
#pragma unroll
        for (int ii = 0; ii < XI; ++ii) asm volatile("" : "+v"(xv[ii]));        // (keeps the loads from being sunk to their uses)
#pragma unroll
        for (int ii = 0; ii < XI; ++ii) {
            const int it = tid + ii * NT, c4 = it % (K / 4), p = it / (K / 4);
            const f32x4 v = f0 + (p >> 4) < B ? xv[ii] : (f32x4){0.f, 0.f, 0.f, 0.f};
            unsigned a0, b0, a1, b1;
            split2(v[0], v[1], a0, b0);
            split2(v[2], v[3], a1, b1);
            // 16-byte chunk q of a pixel row goes to slot q ^ m(row), m = 1 for rows 4 .. 11 of a 16-pixel tile: ds_read_b128 is serviced in
            // the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH, LDS), i.e. rows 4 .. 11 of a group read chunk
            // g ^ 1 of what the other rows read -- with every row at the same chunk the 36-bank row stride put seven pairs of them on the same
            // banks (SQ_LDS_BANK_CONFLICT 0.46 of this kernel's LDS cycles in profiles/r4); swapped, a group's sixteen reads hit distinct banks
            const int csw = ((((c4 >> 1) ^ ((((p & 15) >> 2) ^ ((p & 15) >> 3)) & 1)) << 1) | (c4 & 1)) * 2;
            *(u32x2 *)&Xb[0 * PLANEK + p * XSD + csw] = (u32x2){a0, a1};
            *(u32x2 *)&Xb[1 * PLANEK + p * XSD + csw] = (u32x2){b0, b1};
        }
    }
    __syncthreads();

    // the BN shift of a tile is requested one tile ahead: loaded at the top of its own tile it was consumed at once, and vector memory
    // retires in order -- the wait for it drained the five weight chunks in flight behind it at every tile
    f32x4 sh_n = *(const f32x4 *)&shift[(nt0 + wave) * 16 + 4 * g];
    auto tile = [&](int nt) __attribute__((always_inline)) {
        const f32x4 sh = sh_n * S;
        sh_n = *(const f32x4 *)&shift[(nt + NWV < nt_end ? nt + NWV : nt) * 16 + 4 * g];
        f32x4 acc[NFK];
#pragma unroll
        for (int j = 0; j < NFK; ++j) acc[j] = sh;
        u32x4 bq[2][NFK][2];                            // pixel operands of the current / next k-chunk (ping-pong, no copies)
        auto ldb = [&](int kc, u32x4(&b)[NFK][2]) {
#pragma unroll
            for (int j = 0; j < NFK; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) b[j][p] = *(const u32x4 *)&Xb[p * PLANEK + (j * 16 + r16) * XSD + kc * 16 + 4 * gsw];
        };
        ldb(0, bq[0]);
#pragma unroll
        for (int kc = 0; kc < KC32; ++kc) {
            if (kc + 1 < KC32) ldb(kc + 1, bq[(kc + 1) & 1]);
            const u32x4(&bc)[NFK][2] = bq[kc & 1];
            const u32x4 aa = ring[kc % 5][0], ab = ring[kc % 5][1];
            // three partial products, smallest first; NFK independent accumulators interleaved
#pragma unroll
            for (int j = 0; j < NFK; ++j) acc[j] = mfma_h(ab, bc[j][0], acc[j]);
#pragma unroll
            for (int j = 0; j < NFK; ++j) acc[j] = mfma_h(aa, bc[j][1], acc[j]);
#pragma unroll
            for (int j = 0; j < NFK; ++j) acc[j] = mfma_h(aa, bc[j][0], acc[j]);
            // refill this ring slot with the chunk 5 steps ahead (possibly of this wave's next channel tile)
            // (no branch: after the last tile its own chunks are fetched again, unused -- a conditional fetch makes the compiler's
            // wait-count bookkeeping fall back to vmcnt(0) at the join, i.e. every tile started by draining the ring)
            if (kc + 5 < KC32) lda(nt, kc + 5, ring[kc % 5]);
            else lda(nt + NWV < nt_end ? nt + NWV : nt, kc + 5 - KC32, ring[kc % 5]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NFK; ++j) {
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = row16_sum(r6h(acc[j][t] * inv_s));
            if (r16 == 0) *(f32x4 *)&Ps[j * N + nt * 16 + 4 * g] = v * 0.0625f;
        }
    };
    // first tile peeled: the loop is then entered with the loads in flight that every later tile finds (the compiler's wait counts at a
    // loop header are exact only when the entry and the back edge agree; otherwise: vmcnt(0) at the top of every tile)
    tile(nt0 + wave);
    for (int nt = nt0 + wave + NWV; nt < nt_end; nt += NWV) tile(nt);
    __syncthreads();
    if (NS > 1) {                                          // this slice of the pooled vectors -> HBM
        for (int it = tid; it < NFK * (NTS * 4); it += NT) {
            const int j = it / (NTS * 4), c4 = nt0 * 4 + it % (NTS * 4);
            if (f0 + j < B) *(f32x4 *)&pool[(size_t)(f0 + j) * N + 4 * c4] = *(const f32x4 *)&Ps[j * N + 4 * c4];
        }
        return;
    }
    if (pool) {
        for (int it = tid; it < NFK * (N / 4); it += NT) {
            const int j = it / (N / 4), c4 = it % (N / 4);
            if (f0 + j < B) *(f32x4 *)&pool[(size_t)(f0 + j) * N + 4 * c4] = *(const f32x4 *)&Ps[j * N + 4 * c4];
        }
    }
    f32x4 xv[NFK][N / 256];
#pragma unroll
    for (int j = 0; j < NFK; ++j)
#pragma unroll
        for (int i = 0; i < N / 256; ++i) xv[j][i] = *(const f32x4 *)&Ps[j * N + (i * 64 + lane) * 4];
    fc_rows<(kParam + NWV - 1) / NWV, NFK, NWV>(xv, Wfc, bfc, param, f0, B, 0, wave, lane);
    if (SYN_L2_TOUCH && NS == 1) l2_touch_done(sink);
}

// second half of the sliced tail: blockIdx.y owns 4 of the 16 rounds of head rows
constexpr int kFcSlices = 4;
__global__ __launch_bounds__(256) void head_fc_kernel(const float *__restrict__ pool, const float *__restrict__ Wfc,
                                                      const float *__restrict__ bfc, float *__restrict__ param, int B) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = blockIdx.x * NF;
    f32x4 xv[NF][N / 256];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = f0 + j < B ? f0 + j : B - 1;
#pragma unroll
        for (int i = 0; i < N / 256; ++i) xv[j][i] = *(const f32x4 *)&pool[(size_t)f * N + (i * 64 + lane) * 4];
    }
    fc_rows<kFcRounds / kFcSlices>(xv, Wfc, bfc, param, f0, B, blockIdx.y * (kFcRounds / kFcSlices), wave, lane);
}

// `scratch` ([B,1280] floats) receives the pooled vectors of the sliced schedule when the caller did not ask for them
void launch_head_f16x2(const float *X, const unsigned *Wb3, const float *shift, const float *Wfc, const float *bfc,
                        float *param, float *pool, float *scratch, int B, hipStream_t s, const HeadSliced *sin) {
    const int grid = (B + NF - 1) / NF;
    if (grid <= 96) {                                      // few faces: spread the 1.6 MB of weights over 5 workgroups per face pair
        float *pl = pool ? pool : scratch;
        if (sin) head_f16x2_kernel<5, NF, 4, true><<<dim3(grid, 5), 256, 0, s>>>(X, Wb3, shift, Wfc, bfc, param, pl, B, *sin);
        else head_f16x2_kernel<5><<<dim3(grid, 5), 256, 0, s>>>(X, Wb3, shift, Wfc, bfc, param, pl, B);
        head_fc_kernel<<<dim3(grid, kFcSlices), 256, 0, s>>>(pl, Wfc, bfc, param, B);
        return;
    }
    if (sin) {                                             // (the caller defers the reduce only below the wide tail's threshold)
        head_f16x2_kernel<1, NF, 4, true><<<grid, 256, 0, s>>>(X, Wb3, shift, Wfc, bfc, param, pool, B, *sin);
        return;
    }
    static const int wide_min = (int)test_knob("head_wide_min", 513);       // (B = 640 / 768 / 896: 60 / 62 / 61 -> 47 / 48 / 48 us; B = 512: the two-face workgroups, 49 us)
    if (B >= wide_min) {
        head_f16x2_kernel<1, 4, 8><<<(B + 3) / 4, 512, 0, s>>>(X, Wb3, shift, Wfc, bfc, param, pool, B);
        return;
    }
    head_f16x2_kernel<1><<<grid, 256, 0, s>>>(X, Wb3, shift, Wfc, bfc, param, pool, B);
}
}  // namespace syn
