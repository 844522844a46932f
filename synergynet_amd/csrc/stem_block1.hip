// Fused network head for gfx950: features.0 (3x3 s2 conv 3->32 + BN + ReLU6) and features.1
// (depthwise 3x3 + BN + ReLU6, then linear 1x1 32->16 + BN; t = 1 block without expand conv) in ONE
// kernel (reference mobilenetv2_backbone.py:129, :58-66).  The 60x60x32 stem output -- the largest
// activation of the network, 460 KB per face -- never leaves the CU: a workgroup reads the 25x25x3
// image patch behind a 10x10 output tile, builds the 12x12x32 stem tile in LDS with an im2col GEMM on
// the fp32 MFMA, runs the depthwise stage LDS->LDS (sliding 3-row window) and the 32->16 projection on
// the MFMA, and stores 10x10x16 (NHWC).  HBM traffic per face: 43 KB of uint8 crop in (+halo re-reads
// from L2), 230 KB out.  The uint8 variant also folds the HWC->CHW permute and (x-127.5)/128
// (synergy3DMM.py:189-192).
//
// Workgroups are PERSISTENT (filters / BN vectors are loaded into LDS and registers once) and the next
// tile's image patch is prefetched into registers while the current tile computes: uint8 rows as aligned
// dwords (25 px * 3 B = 75 B -> 20 dwords per patch row), fp32 NCHW planes as scalars.
#include <cstdlib>

#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int T = 10;                 // output tile edge (60 = 6 tiles)
constexpr int FT = T + 2;             // stem-output tile edge incl. the depthwise halo
constexpr int IT = 2 * FT + 1;        // image tile edge (stride-2 3x3 receptive field) = 25
constexpr int ITS = IT + 1;           // padded image row
constexpr int ES = 36;                // LDS row stride of the 32-channel tiles
constexpr int PIN = FT * FT;          // 144 stem pixels = 9 MFMA pixel tiles
constexpr int POUT = T * T, POUTP = 112;
constexpr int NTH = 256;
constexpr int ROWDW = 20;             // dwords per uint8 patch row (covers 3 lead bytes + 75 + tail)
constexpr int U8_IPT = (IT * ROWDW + NTH - 1) / NTH;      // 2 dword loads per thread per tile
constexpr int F32_IPT = (3 * IT * IT + NTH - 1) / NTH;    // 8 scalar loads per thread per tile
// LDS offset of im2col element k = ci*9 + ky*3 + kx inside the image planes (padded slots 27..31 read element 0)
constexpr int koff_of(int k) { return k < 27 ? (k / 9) * IT * ITS + ((k % 9) / 3) * ITS + k % 3 : 0; }
__device__ __forceinline__ float r6(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }
__device__ __forceinline__ f32x4 r6(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = r6(v[i]);
    return r;
}
}  // namespace

// BF: (uint8 crops only) the stem GEMM runs on the bf16 matrix pipe.  A normalised uint8 pixel (2x - 255) / 256 has 8
// significant bits, i.e. it IS a bf16 number, so only the filter needs the exact 3-way split (w0b3): 3 MFMAs of K = 32
// per (16 pixels x 16 channels) instead of 8 fp32-input MFMAs of K = 4, products exact, fp32 accumulation.
// ABL: stage ablation for profiling (SYN_ABLATE_STEM) is a separate instantiation -- runtime flags around the MFMA loops cost the
// production kernel scheduling freedom.
template <bool U8, bool BF, bool ABL = false>
__global__ __launch_bounds__(NTH) void stem_block1_kernel(
    const float *__restrict__ img, const uint8_t *__restrict__ img8, const float *__restrict__ w0 /*[27][32]*/,
    const unsigned *__restrict__ w0b3 /*[2][3][64][4]*/,
    const float *__restrict__ s0, const float *__restrict__ b0, const float *__restrict__ wd /*[9][32]*/,
    const float *__restrict__ sd, const float *__restrict__ bd, const float *__restrict__ wp /*Wpk[1][2][64][4]*/,
    const float *__restrict__ sp, const float *__restrict__ bp, float *__restrict__ Y, int total_tiles, int ablate_) {
    const int ablate = ABL ? ablate_ : 0;
    __shared__ __attribute__((aligned(16))) float im[3 * IT * ITS];
    __shared__ __attribute__((aligned(16))) float Es[PIN * ES];
    __shared__ __attribute__((aligned(16))) float Ds[POUTP * ES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    // ---- per-workgroup constants: stem / depthwise / project filters + BN in registers ----
    for (int i = tid; i < (POUTP - POUT) * ES; i += NTH) Ds[POUT * ES + i] = 0.f;
    // stem filter as MFMA "A" fragments: lane (channel r16 of tile nt, k-slot g): k = 16*kc + 4*g + q < 27
    f32x4 wa[2][2];
    u32x4 wb[2][3];                     // BF: filter pieces (h, m, l) of the two channel tiles
    int koff[2][4];                     // LDS offset of patch element k inside the image planes
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 4 * kc + q;
            const int k = BF ? 8 * g + j : 16 * kc + 4 * g + q;                // BF: lane group g holds k = 8g .. 8g+7
            koff[kc][q] = BF ? (g == 0 ? koff_of(j) : g == 1 ? koff_of(8 + j) : g == 2 ? koff_of(16 + j) : koff_of(24 + j))
                             : (g == 0 ? koff_of(16 * kc + q) : g == 1 ? koff_of(16 * kc + 4 + q)
                                : g == 2 ? koff_of(16 * kc + 8 + q) : koff_of(16 * kc + 12 + q));
            if (!BF)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) wa[nt][kc][q] = k < 27 ? w0[k * 32 + nt * 16 + r16] : 0.f;
        }
    if (BF)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) wb[nt][pc] = *(const u32x4 *)(w0b3 + ((nt * 3 + pc) * 64 + lane) * 4);
    // BN scales are folded into the filters by the host; accumulators start at the BN shift
    const f32x4 sh0 = *(const f32x4 *)&b0[4 * g], sh1 = *(const f32x4 *)&b0[16 + 4 * g];
    const f32x4 pa0 = *(const f32x4 *)(wp + lane * 4), pa1 = *(const f32x4 *)(wp + 256 + lane * 4);
    const f32x4 psh = *(const f32x4 *)&bp[4 * g];

    // depthwise filter + BN shift of this thread's channel quad: constant for the whole (persistent) kernel -> registers
    f32x4 dww[9], dwsh;
#pragma unroll
    for (int k = 0; k < 9; ++k) dww[k] = *(const f32x4 *)&wd[k * 32 + 4 * (tid & 7)];
    dwsh = *(const f32x4 *)&bd[4 * (tid & 7)];

    // ---- image patch prefetch (registers) ----
    unsigned xr8[U8_IPT];
    float xrf[U8 ? 1 : F32_IPT];
    auto tile_origin = [&](int tile, int &f, int &oy0, int &ox0) {
        ox0 = (tile % 6) * T;
        oy0 = ((tile / 6) % 6) * T;
        f = tile / 36;
    };
    auto load_patch = [&](int tile) {
        int f, oy0, ox0;
        tile_origin(tile, f, oy0, ox0);
        const int iy0 = 2 * (oy0 - 1) - 1, ix0 = 2 * (ox0 - 1) - 1;     // image coords of patch pixel (0,0)
        if (U8) {
            // patch row ly = bytes [ (iy*120 + ix0)*3 , +75 ); ix0*3 == 3 (mod 4) for every tile column, so the
            // aligned window starts 3 bytes earlier and is 20 dwords long; image rows are 360 B (a dword multiple),
            // so an aligned dword is either entirely inside its row or entirely outside (then it is masked later)
#pragma unroll
            for (int ii = 0; ii < U8_IPT; ++ii) {
                const int it = tid + ii * NTH;
                const int ly = it / ROWDW, d = it % ROWDW;
                const int iy = iy0 + ly;
                const int byte0 = ix0 * 3 - 3 + 4 * d;                   // byte offset inside the image row
                // Branch-free: rows / dwords outside the image are clamped into it (store_patch masks them, the value is
                // never used).  A conditional load would make the compiler guard the register's default value with
                // s_waitcnt vmcnt(0) -- which, vector memory being in order, also waits for the previous tile's stores.
                const int iyc = iy < 0 ? 0 : (iy >= kImg ? kImg - 1 : iy);
                const int bc = byte0 < 0 ? 0 : (byte0 + 3 < kImg * 3 ? byte0 : kImg * 3 - 4);
                xr8[ii] = *(const unsigned *)(img8 + ((size_t)(f * kImg + iyc) * kImg) * 3 + bc);
            }
        } else {
#pragma unroll
            for (int ii = 0; ii < F32_IPT; ++ii) {
                const int it = tid + ii * NTH;
                const int ci = it / (IT * IT), r = it % (IT * IT);
                const int iy = iy0 + r / IT, ix = ix0 + r % IT;
                float v = 0.f;
                if (it < 3 * IT * IT && iy >= 0 && iy < kImg && ix >= 0 && ix < kImg)
                    v = img[((size_t)(f * 3 + ci) * kImg + iy) * kImg + ix];
                xrf[ii] = v;
            }
        }
    };
    auto store_patch = [&](int tile) {      // registers -> normalised fp32 planes in LDS, zero outside the image
        int f, oy0, ox0;
        tile_origin(tile, f, oy0, ox0);
        const int iy0 = 2 * (oy0 - 1) - 1, ix0 = 2 * (ox0 - 1) - 1;
        if (U8) {
#pragma unroll
            for (int ii = 0; ii < U8_IPT; ++ii) {
                const int it = tid + ii * NTH;
                if (it >= IT * ROWDW) continue;
                const int ly = it / ROWDW, d = it % ROWDW;
                const bool row_ok = (unsigned)(iy0 + ly) < (unsigned)kImg;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int pb = 4 * d + b - 3;                        // byte index inside the 75-byte patch row
                    if (pb < 0 || pb >= IT * 3) continue;
                    const int lx = pb / 3, ci = pb % 3;
                    const bool ok = row_ok && (unsigned)(ix0 + lx) < (unsigned)kImg;
                    const float v = ((float)((xr8[ii] >> (8 * b)) & 0xffu) - 127.5f) * 0.0078125f;
                    im[ci * IT * ITS + ly * ITS + lx] = ok ? v : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int ii = 0; ii < F32_IPT; ++ii) {
                const int it = tid + ii * NTH;
                if (it >= 3 * IT * IT) continue;
                const int ci = it / (IT * IT), r = it % (IT * IT);
                im[ci * IT * ITS + (r / IT) * ITS + r % IT] = xrf[ii];
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < total_tiles) { load_patch(tile); if (!(ablate & 1)) store_patch(tile); }
    // de-phase the co-resident persistent workgroups of a CU (they start together and would otherwise hit the
    // same MFMA / VALU / LDS stage at the same time): workgroup "layer" k waits k * ~1/3 tile time once
    if (ablate & 16)
        for (int i = 0; i < (int)(blockIdx.x >> 8) * 3; ++i) __builtin_amdgcn_s_sleep(32);
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), visible to the compiler: the filter / BN constants are in; no wait
                                                     // for them is left inside the tile loop
    for (; tile < total_tiles; tile += gridDim.x) {
        int f, oy0, ox0;
        tile_origin(tile, f, oy0, ox0);
        const int fy0 = oy0 - 1, fx0 = ox0 - 1;      // stem-output coords of Es pixel (0,0)
        __syncthreads();                              // the patch planes `im` of this tile are complete
        // unconditional (the last iteration fetches its own tile again): with control flow around the prefetch the compiler
        // protects the reuse of its registers with s_waitcnt vmcnt(0) at the loop top, right behind the previous tile's stores
        const int next = tile + (int)gridDim.x < total_tiles ? tile + (int)gridDim.x : tile;
        load_patch(next);                             // in flight during the stem conv and the depthwise stage

        // ---- stem conv on the MFMA: im2col GEMM  E[144 px][32] = patch[px][27(+5 zero)] . W0^T ----
        // MFMA "B" = patches gathered straight from the LDS image planes: lane (pixel r16, k-slot g) reads
        // element k at the fixed per-lane offset koff[kc][q] plus the pixel's base 2*ly*ITS + 2*lx.
        if (!(ablate & 2))
        for (int pt = wave; pt < PIN / 16; pt += 4) {
            const int p = pt * 16 + r16;
            const int base = 2 * (p / FT) * ITS + 2 * (p % FT);
            f32x4 bv[2];
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bv[kc][q] = im[base + koff[kc][q]];          // padded k slots (27..31) read element 0: their filter taps are 0
                }
            f32x4 e0 = sh0, e1 = sh1;
            if (BF) {
                // the pixel values are exact bf16 numbers: pack their high halves, two k per dword (even k low)
                unsigned bw[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = bv[e >> 2][e & 3];      // (bit_cast applied to a vector-element lvalue reads element 0)
                    bw[e] = __builtin_bit_cast(unsigned, t);
                }
                u32x4 bb;
#pragma unroll
                for (int d = 0; d < 4; ++d) bb[d] = __builtin_amdgcn_perm(bw[2 * d + 1], bw[2 * d], 0x07060302u);
                const bf16x8 b8 = __builtin_bit_cast(bf16x8, bb);
#pragma unroll
                for (int pc = 2; pc >= 0; --pc) {          // smallest filter piece first
                    e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[0][pc]), b8, e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[1][pc]), b8, e1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][kc][q], bv[kc][q], e0, 0, 0, 0);
                        e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][kc][q], bv[kc][q], e1, 0, 0, 0);
                    }
            }
            // lane owns pixel p, channels 4g..4g+3 (tile 0) and 16+4g.. (tile 1)
            // stem pixels of the halo ring outside the 60x60 map are the depthwise stage's zero padding: clamp them to
            // [0, 0] instead of [0, 6] (the ceiling is a per-lane value, so the padding costs nothing downstream)
            const float hi = ((unsigned)(fy0 + p / FT) < 60u && (unsigned)(fx0 + p % FT) < 60u) ? 6.0f : 0.0f;
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = __builtin_amdgcn_fmed3f(e0[e], 0.0f, hi); v1[e] = __builtin_amdgcn_fmed3f(e1[e], 0.0f, hi); }
            *(f32x4 *)&Es[p * ES + 4 * g] = v0;
            *(f32x4 *)&Es[p * ES + 16 + 4 * g] = v1;
        }
        __syncthreads();
        // ---- depthwise 3x3 s1 on the stem tile: thread = (channel quad, output column, row half) ----
        // (out-of-map taps read the zeros the stem epilogue wrote)
        if (!(ablate & 4) && tid < 8 * T * 2) {
            const int c4 = tid & 7, q2 = tid >> 3;
            const int oxl = q2 % T, seg = q2 / T;
            const f32x4 (&w)[9] = dww;
            const f32x4 sh = dwsh;
            f32x4 rb[3][3];
            auto load_row = [&](int ly, f32x4(&dst)[3]) {               // ly = tile row (always inside the 12x12 tile)
                const float *er = Es + (ly * FT + oxl) * ES + 4 * c4;
                dst[0] = *(const f32x4 *)(er);
                dst[1] = *(const f32x4 *)(er + ES);
                dst[2] = *(const f32x4 *)(er + 2 * ES);
            };
#pragma unroll
            for (int r = 0; r < T / 2; ++r) {
                const int oyl = seg * (T / 2) + r;
                if (r == 0) { load_row(oyl, rb[0]); load_row(oyl + 1, rb[1]); load_row(oyl + 2, rb[2]); }
                else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { rb[0][k] = rb[1][k]; rb[1][k] = rb[2][k]; }
                    load_row(oyl + 2, rb[2]);
                }
                f32x4 a = sh;
                a += rb[0][0] * w[0]; a += rb[0][1] * w[1]; a += rb[0][2] * w[2];
                a += rb[1][0] * w[3]; a += rb[1][1] * w[4]; a += rb[1][2] * w[5];
                a += rb[2][0] * w[6]; a += rb[2][1] * w[7]; a += rb[2][2] * w[8];
                *(f32x4 *)&Ds[(oyl * T + oxl) * ES + 4 * c4] = r6(a);
            }
        }
        __syncthreads();
        // The next tile's patch goes to LDS HERE, before this tile's output stores are issued (`im` is free since the barrier
        // after the stem conv): vector memory retires in order, so consuming the prefetched registers after the stores -- at
        // the top of the next iteration -- meant waiting for every store's acknowledgement once per tile.
        store_patch(next);
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) the compiler can see: no load is pending once the stores start
        // ---- linear 1x1 32 -> 16 on the MFMA: 7 pixel tiles over 4 waves ----
        if (!(ablate & 8))
        for (int pt = wave; pt < POUTP / 16; pt += 4) {
            const f32x4 b0v = *(const f32x4 *)&Ds[(pt * 16 + r16) * ES + 4 * g];
            const f32x4 b1v = *(const f32x4 *)&Ds[(pt * 16 + r16) * ES + 16 + 4 * g];
            f32x4 acc = psh;
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa0[s], b0v[s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa1[s], b1v[s], acc, 0, 0, 0);
            const int po = pt * 16 + r16;
            if (po < POUT) {
                const int oy = oy0 + po / T, ox = ox0 + po % T;
                *(f32x4 *)&Y[((size_t)(f * 60 + oy) * 60 + ox) * 16 + 4 * g] = acc;
            }
        }
        // the next iteration barriers before `im` is read and before Es is rewritten; Ds is rewritten only after two more
        // barriers -> no extra barrier needed here
    }
}

void launch_stem_block1(const float *img, const uint8_t *img8, const float *w0, const unsigned *w0b3, const float *s0, const float *b0,
                        const float *wd, const float *sd, const float *bd, const float *wp, const float *sp,
                        const float *bp, float *Y, int B, hipStream_t s) {
    const int total = B * 36;
    // persistent: the register budget (~200 VGPRs, 4 waves per workgroup) admits two resident workgroups per CU; a third
    // layer would only start when the first two finish (measured 361 -> 322 us at B = 1024 going from 3 to 2)
    const int grid = total < 256 * 2 ? total : 256 * 2;
    static const int ablate = (int)test_knob("ablate_stem", 0);   // profiling only: skip stages
    if (img8 && w0b3 && ablate) stem_block1_kernel<true, true, true><<<grid, NTH, 0, s>>>(nullptr, img8, w0, w0b3, s0, b0, wd, sd, bd, wp, sp, bp, Y, total, ablate);
    else if (img8 && w0b3) stem_block1_kernel<true, true><<<grid, NTH, 0, s>>>(nullptr, img8, w0, w0b3, s0, b0, wd, sd, bd, wp, sp, bp, Y, total, 0);
    else if (img8)    stem_block1_kernel<true, false><<<grid, NTH, 0, s>>>(nullptr, img8, w0, nullptr, s0, b0, wd, sd, bd, wp, sp, bp, Y, total, ablate);
    else              stem_block1_kernel<false, false><<<grid, NTH, 0, s>>>(img, nullptr, w0, nullptr, s0, b0, wd, sd, bd, wp, sp, bp, Y, total, ablate);
}

}  // namespace syn
