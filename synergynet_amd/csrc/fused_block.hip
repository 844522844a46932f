// Fused MobileNetV2 inverted-residual block for gfx950 (reference mobilenetv2_backbone.py:45-74):
//
//   y = [x +] BN(pw_project( ReLU6(BN(dw3x3( ReLU6(BN(pw_expand(x))) ))) ))
//
// in ONE kernel.  The 6x-expanded activations (up to 1.38 MB per face at 60x60) never leave the CU:
// a workgroup owns a spatial tile of output pixels (of NF faces), keeps the input tile (with its 3x3
// halo) in LDS and walks the hidden channels HC at a time:
//
//   (BN scales are folded into the packed weights by the host; accumulators start at the BN shift, so each
//    BN+ReLU6 epilogue is a single clamp per element -- fp32 MFMA and VALU share the vector pipe on gfx950,
//    every VALU instruction saved is MFMA time gained)
//   stage 1  expand   E[pix_in ][HC] = ReLU6(BN(Xs[pix_in][CIN] . We^T))    fp32 MFMA 16x16x4 -> LDS
//   stage 2  dw 3x3   D[pix_out][HC] = ReLU6(BN(dw(E)))                       VALU, LDS -> LDS
//   stage 3  project  acc[pix_out][COUT] += D . Wp[:, chunk]^T                fp32 MFMA, accumulators in VGPRs
//
// and finishes with BN (+ residual, read back from the LDS input tile) and one NHWC store.  HBM
// traffic per block is input tile + output tile; everything else is LDS / L2-resident weights.
//
// MFMA operand convention (same as pointwise_kernel): MFMA "A" rows = 16 output channels (weights),
// MFMA "B" cols = 16 pixels (activations from LDS); a lane owns 4 consecutive channels of one pixel.
// Weights are pre-packed by the host in lane order  Wpk[n_tile][k_chunk][lane][4]  with
//   value = W[n = 16*n_tile + (lane&15)][k = 16*k_chunk + 4*(lane>>4) + s]
// so each weight fetch of a wave is one fully coalesced 1 KiB load (global) / conflict-free b128 (LDS).
//
// Latency plan (measured: the naive version spent 55-90 % of every MFMA stage waiting for weights):
//   WLDS = false (late blocks, weights 0.2-1.2 MB): the weights of the NEXT MFMA stage are fetched into
//       registers one stage ahead (project weights while the depthwise stage runs, the next chunk's expand
//       weights while the project stage runs), so no global load sits on an MFMA critical path.
//   WLDS = true  (early blocks, weights <= 42 KB): all weights live in LDS for the lifetime of a
//       PERSISTENT workgroup that loops over tiles, and the next tile's input is prefetched into registers
//       while the current tile computes (no other VMEM op is in flight, so it stays in flight).
#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu6_(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }   // one v_med3 (fminf(fmaxf()) adds a canonicalising v_max)
__device__ __forceinline__ f32x4 relu6_(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = relu6_(v[i]);
    return r;
}

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int rup(int a, int b) { return cdiv(a, b) * b; }

template <int CIN_, int HID_, int COUT_, int HIN_, int S_, bool RES_, int TH_, int TW_, int NF_, int HC_, int NW_,
          int EPB_, int WN_, int WP_, bool WLDS_, int DWRS_ = 0>
struct BlockCfg {
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_, HIN = HIN_, S = S_, TH = TH_, TW = TW_, NF = NF_,
                         HC = HC_, NW = NW_, EPB = EPB_, WN = WN_, WP = WP_;
    static constexpr bool RES = RES_, WLDS = WLDS_;
    static constexpr int NT = NW * 64;
    static constexpr int HOUT = S == 2 ? (HIN + 1) / 2 : HIN;
    static constexpr int TILES_Y = cdiv(HOUT, TH), TILES_X = cdiv(HOUT, TW);
    // a tile that covers the whole image needs no halo ring: everything outside is zero padding
    static constexpr bool WHOLE = (TILES_Y == 1 && TILES_X == 1);
    static constexpr int IH = WHOLE ? HIN : (TH - 1) * S + 3, IW = WHOLE ? HIN : (TW - 1) * S + 3;   // input tile
    static constexpr int PIN = NF * IH * IW, PINP = rup(PIN, 16);
    static constexpr int POUT = NF * TH * TW, POUTP = rup(POUT, 16);
    // expand K is walked in chunks of 16 (4 MFMA steps); a trailing half chunk of 8 (CIN = 24) takes 2 steps, with
    // lane group g holding k = 16*kc + 2g + {0,1} (the host packs the weights of that chunk the same way)
    static constexpr bool KHALF = (CIN % 16 == 8);
    static constexpr int CINP = rup(CIN, 8), COUTP = rup(COUT, 16);
    static constexpr int KCH = cdiv(CINP, 16), KC3 = HC / 16;                    // k-chunks of expand / project
    static constexpr int XS = CINP + 4, ES = HC + 4;                             // LDS row strides (floats)
    static constexpr int NT_E = HC / 16, PT_IN = PINP / 16, PG = cdiv(PT_IN, EPB), JOBS = NT_E * PG;
    static constexpr int JPW = cdiv(JOBS, NW);                                   // expand jobs per wave
    static constexpr int NT_O = COUTP / 16, PT_O = POUTP / 16;
    static constexpr int AN = cdiv(NT_O, WN), AP = cdiv(PT_O, WP);
    static constexpr int X_ITEMS = PINP * (CINP / 4), X_IPT = cdiv(X_ITEMS, NT); // input-tile float4 per thread
    // depthwise stage mapping: thread = (channel quad, output column of a face, row segment)
    static constexpr int C4N = HC / 4, COLS = NF * TW;
    static constexpr int RS_ = NT / (C4N * COLS);
    static constexpr int RS = DWRS_ > 0 ? DWRS_ : (RS_ < 1 ? 1 : (RS_ > TH ? TH : RS_));   // row segments per column (DWRS_ overrides)
    static constexpr int RPS = cdiv(TH, RS), DW_THREADS = C4N * COLS * cdiv(TH, RPS);
    // LDS carve (floats)
    static constexpr int XS_FLOATS = PINP * XS, ES_FLOATS = PINP * ES, DS_FLOATS = POUTP * ES;
    static constexpr int WE_FLOATS = (HID / 16) * KCH * 256, WP_FLOATS = NT_O * (HID / 16) * 256;
    static constexpr int WD_FLOATS = WLDS ? 11 * HID : 11 * HC;                  // 9 taps | scale | shift
    static constexpr int LDS_FLOATS = XS_FLOATS + ES_FLOATS + DS_FLOATS + WD_FLOATS + (WLDS ? WE_FLOATS + WP_FLOATS + HID : 0);
    static constexpr int WDR_THREADS = 11 * HC / 4;
    static_assert(HID % HC == 0 && HC % 16 == 0, "hidden chunking");
    static_assert(WN * WP == NW, "wave grid");
    static_assert(!RES || (S == 1 && CIN == COUT), "residual only on stride-1 same-width blocks");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static_assert(WLDS || WDR_THREADS <= NT, "one depthwise-weight float4 per thread");
};

// PROF: wave 0 accumulates s_memtime deltas (shader cycles) per stage into prof[0..7] (debug hook
// syn_debug_profile_block): 0 stage0, 1 expand, 2 barrier-after-expand, 3 depthwise, 4 barrier-after-dw,
// 5 project, 6 epilogue, 7 #tiles
#define SYN_TICK() (PROF ? __builtin_amdgcn_s_memtime() : 0ull)
#define SYN_LAP(i) do { if (PROF) { tn = SYN_TICK(); pt_[i] += tn - tk; tk = tn; } } while (0)

template <class C, bool PROF = false>
__global__ __launch_bounds__(C::NW * 64) void fused_block_kernel(
    const float *__restrict__ X, const float *__restrict__ We, const float *__restrict__ e_scale,
    const float *__restrict__ e_shift, const float *__restrict__ Wd, const float *__restrict__ d_scale,
    const float *__restrict__ d_shift, const float *__restrict__ Wp, const float *__restrict__ p_scale,
    const float *__restrict__ p_shift, float *__restrict__ Y, int B, int total_tiles,
    unsigned long long *prof = nullptr) {
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
    float *Xs = smem, *Es = Xs + C::XS_FLOATS, *Ds = Es + C::ES_FLOATS, *Wds = Ds + C::DS_FLOATS;
    float *Wle = Wds + C::WD_FLOATS, *Wlp = Wle + C::WE_FLOATS, *Ebn = Wlp + C::WP_FLOATS;   // only carved when WLDS
    constexpr int NT = C::NT;
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef SYN_NO_RFL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: branches on it stay scalar
#endif
    const int r16 = lane & 15, g = lane >> 4;
    const int wn = wave % C::WN, wp = wave / C::WN;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    unsigned long long pt_[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long tk = SYN_TICK(), tn;
    unsigned long long ntiles_done = 0;

    // input tile `tile` -> registers (halo included, zero outside the image / past CIN)
    f32x4 xr[C::X_IPT];
    auto load_x = [&](int tile) {
        const int tx = tile % C::TILES_X, ty = (tile / C::TILES_X) % C::TILES_Y;
        const int f0 = (tile / (C::TILES_X * C::TILES_Y)) * C::NF;
        const int iy0 = C::WHOLE ? 0 : ty * C::TH * C::S - 1, ix0 = C::WHOLE ? 0 : tx * C::TW * C::S - 1;
#pragma unroll
        for (int ii = 0; ii < C::X_IPT; ++ii) {
            const int it = tid + ii * NT;
            const int c4 = it % (C::CINP / 4), p = it / (C::CINP / 4);
            f32x4 v = z4;
            if (it < C::X_ITEMS && p < C::PIN && 4 * c4 < C::CIN) {
                const int i = p / (C::IH * C::IW), r = p % (C::IH * C::IW);
                const int iy = iy0 + r / C::IW, ix = ix0 + r % C::IW;
                const int f = f0 + i;
                if (f < B && iy >= 0 && iy < C::HIN && ix >= 0 && ix < C::HIN)
                    v = *(const f32x4 *)&X[((size_t)(f * C::HIN + iy) * C::HIN + ix) * C::CIN + 4 * c4];
            }
            xr[ii] = v;
        }
    };

    int tile = blockIdx.x;
    if (tile < total_tiles) load_x(tile);
    if (C::WLDS) {   // all weights of the block -> LDS, once per (persistent) workgroup
        for (int i = tid; i < C::WE_FLOATS / 4; i += NT) *(f32x4 *)&Wle[4 * i] = *(const f32x4 *)&We[4 * i];
        for (int i = tid; i < C::WP_FLOATS / 4; i += NT) *(f32x4 *)&Wlp[4 * i] = *(const f32x4 *)&Wp[4 * i];
        for (int i = tid; i < 11 * C::HID / 4; i += NT) {
            const int row = i / (C::HID / 4), c4 = i % (C::HID / 4);
            const float *src = row < 9 ? Wd + row * C::HID : (row == 9 ? d_scale : d_shift);
            *(f32x4 *)&Wds[row * C::HID + 4 * c4] = *(const f32x4 *)&src[4 * c4];
        }
        for (int i = tid; i < C::HID / 4; i += NT) {
            *(f32x4 *)&Ebn[4 * i] = *(const f32x4 *)&e_shift[4 * i];
        }
    }
    // project BN of the channels this lane owns: constant for the whole kernel
    f32x4 psh[C::AN];
#pragma unroll
    for (int i = 0; i < C::AN; ++i) {
        const int n = (wn + i * C::WN) * 16 + 4 * g;
        psh[i] = n < C::COUTP ? *(const f32x4 *)&p_shift[n] : z4;
    }
    if (C::POUTP > C::POUT)
        for (int it = tid; it < (C::POUTP - C::POUT) * C::ES; it += NT) Ds[C::POUT * C::ES + it] = 0.f;

    // register-prefetched weights (WLDS = false)
    f32x4 a1[C::JPW][C::KCH];      // expand weights of the coming chunk, per job of this wave
    f32x4 e1h[C::JPW];             // ... and the expand BN shift of the job's 4 channels
    f32x4 a3[C::AN][C::KC3];       // project weights of the current chunk
    f32x4 wdr = z4;                // this thread's float4 of the coming chunk's depthwise filter / BN
    auto fetch_a1 = [&](int hc0) {
#pragma unroll
        for (int jj = 0; jj < C::JPW; ++jj) {
            const int job = wave + jj * C::NW;
            if (job < C::JOBS) {
                const float *wa = We + ((size_t)(hc0 / 16 + job % C::NT_E) * C::KCH) * 256 + lane * 4;
#pragma unroll
                for (int kc = 0; kc < C::KCH; ++kc) a1[jj][kc] = *(const f32x4 *)(wa + kc * 256);
                const int ch = hc0 + (job % C::NT_E) * 16 + 4 * g;
                e1h[jj] = *(const f32x4 *)&e_shift[ch];
            }
        }
        if (tid < C::WDR_THREADS) {
            const int row = tid / (C::HC / 4), c4 = tid % (C::HC / 4);
            const float *src = row < 9 ? Wd + row * C::HID : (row == 9 ? d_scale : d_shift);
            wdr = *(const f32x4 *)&src[hc0 + 4 * c4];
        }
    };
    auto fetch_a3 = [&](int hc0) {
#pragma unroll
        for (int i = 0; i < C::AN; ++i) {
            int nt = wn + i * C::WN;
            nt = nt < C::NT_O ? nt : 0;
#pragma unroll
            for (int kc = 0; kc < C::KC3; ++kc)
                a3[i][kc] = *(const f32x4 *)(Wp + ((size_t)nt * (C::HID / 16) + hc0 / 16 + kc) * 256 + lane * 4);
        }
    };
    if (!C::WLDS) fetch_a1(0);

    for (; tile < total_tiles; tile += gridDim.x) {
        const int tx = tile % C::TILES_X, ty = (tile / C::TILES_X) % C::TILES_Y;
        const int f0 = (tile / (C::TILES_X * C::TILES_Y)) * C::NF;
        const int oy0 = ty * C::TH, ox0 = tx * C::TW;
        const int iy0 = C::WHOLE ? 0 : oy0 * C::S - 1, ix0 = C::WHOLE ? 0 : ox0 * C::S - 1;   // image coords of tile pixel (0,0)

        // ---- stage 0: input tile registers -> LDS; start fetching the next tile ----
#pragma unroll
        for (int ii = 0; ii < C::X_IPT; ++ii) {
            const int it = tid + ii * NT;
            if (it < C::X_ITEMS) *(f32x4 *)&Xs[(it / (C::CINP / 4)) * C::XS + 4 * (it % (C::CINP / 4))] = xr[ii];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) load_x(tile + gridDim.x);
        SYN_LAP(0);

        f32x4 acc[C::AN][C::AP];
#pragma unroll
        for (int i = 0; i < C::AN; ++i)
#pragma unroll
            for (int j = 0; j < C::AP; ++j) acc[i][j] = psh[i];        // BN shift = accumulator start
        // Tiled blocks: hidden pixels of the halo ring that lie outside the image must read as ZERO in the depthwise
        // stage (zero padding applies to the expanded map).  The expand epilogue writes them as med3(e, 0, 0) instead
        // of med3(e, 0, 6) -- the ceiling is a per-lane register, so the padding costs no instruction anywhere.
        float ehi[C::JPW][C::EPB];
        if (!C::WHOLE) {
#pragma unroll
            for (int jj = 0; jj < C::JPW; ++jj)
#pragma unroll
                for (int q = 0; q < C::EPB; ++q) {
                    const int job = wave + jj * C::NW;
                    const int p = ((job / C::NT_E) * C::EPB + q) * 16 + r16;
                    const int r = p % (C::IH * C::IW);
                    const int iy = iy0 + r / C::IW, ix = ix0 + r % C::IW;
                    ehi[jj][q] = ((unsigned)iy < (unsigned)C::HIN && (unsigned)ix < (unsigned)C::HIN) ? 6.0f : 0.0f;
                }
        }

        for (int hc0 = 0; hc0 < C::HID; hc0 += C::HC) {
            const float *wdc = C::WLDS ? Wds + hc0 : Wds;                 // depthwise filter of this chunk
            constexpr int WDS = C::WLDS ? C::HID : C::HC;                 // its row stride
            if (!C::WLDS && tid < C::WDR_THREADS) *(f32x4 *)&Wds[4 * tid] = wdr;   // rows are HC wide: [row][c4] == tid
            // ---- stage 1: expand 1x1 + BN + ReLU6 for every pixel of the input tile ----
#pragma unroll
            for (int jj = 0; jj < C::JPW; ++jj) {
                const int job = wave + jj * C::NW;
                if (job >= C::JOBS) break;
                const int nt = job % C::NT_E, pg = job / C::NT_E;
                const int ch = hc0 + nt * 16 + 4 * g;
                const f32x4 sh = C::WLDS ? *(const f32x4 *)&Ebn[ch] : e1h[jj];       // BN shift = accumulator start
                f32x4 ea[C::EPB];
#pragma unroll
                for (int q = 0; q < C::EPB; ++q) ea[q] = sh;
                // operand reads from LDS run exactly one k-chunk ahead of the MFMAs that use them (the scheduling
                // barrier keeps the compiler from re-serialising read -> wait -> MFMA per chunk)
                auto ldb = [&](int kc, f32x4(&b)[C::EPB]) {
#pragma unroll
                    for (int q = 0; q < C::EPB; ++q) {
                        const int pt = pg * C::EPB + q;
                        const float *xp = &Xs[((pt < C::PT_IN ? pt : 0) * 16 + r16) * C::XS + kc * 16];
                        if (C::KHALF && kc == C::KCH - 1) {
                            const float2 h2 = *(const float2 *)(xp + 2 * g);
                            b[q] = (f32x4){h2.x, h2.y, 0.f, 0.f};
                        } else {
                            b[q] = *(const f32x4 *)(xp + 4 * g);
                        }
                    }
                };
                auto lda = [&](int kc) -> f32x4 {
                    if (C::WLDS) return *(const f32x4 *)&Wle[((hc0 / 16 + nt) * C::KCH + kc) * 256 + lane * 4];
                    return a1[jj][kc];
                };
                f32x4 bc[C::EPB], bn[C::EPB], ac = lda(0), an = ac;
                ldb(0, bc);
#pragma unroll
                for (int kc = 0; kc < C::KCH; ++kc) {
                    if (kc + 1 < C::KCH) { ldb(kc + 1, bn); an = lda(kc + 1); }
                    // unconditional: a ragged last group multiplies a clamped (duplicate) pixel tile, never stored
#pragma unroll
                    for (int s = 0; s < ((C::KHALF && kc == C::KCH - 1) ? 2 : 4); ++s)
#pragma unroll
                        for (int q = 0; q < C::EPB; ++q)
                            ea[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[s], bc[q][s], ea[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < C::EPB; ++q) bc[q] = bn[q];
                    ac = an;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < C::EPB; ++q) {
                    const int pt = pg * C::EPB + q;
                    if (pt < C::PT_IN) {
                        f32x4 ev;
                        const float hi = C::WHOLE ? 6.0f : ehi[jj][q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) ev[e] = __builtin_amdgcn_fmed3f(ea[q][e], 0.0f, hi);
                        *(f32x4 *)&Es[(pt * 16 + r16) * C::ES + nt * 16 + 4 * g] = ev;
                    }
                }
            }
            if (!C::WLDS) fetch_a3(hc0);          // in flight during the depthwise stage
            SYN_LAP(1);
            __syncthreads();
            SYN_LAP(2);
            // ---- stage 2: depthwise 3x3 + BN + ReLU6, LDS -> LDS ----
            // thread = (channel quad, output column, row segment): walks down its column with a 3-row sliding
            // window (3 LDS reads per output at S=1).  Zero padding: column taps outside the image are folded
            // into the thread's filter copy (weight := 0, address clamped), rows outside the image load as zeros.
            for (int t = tid; t < C::DW_THREADS; t += NT) {
                const int c4 = t % C::C4N, q = t / C::C4N;
                const int col = q % C::COLS, seg = q / C::COLS;
                const int fi = col / C::TW, oxl = col % C::TW;
                const int ixb = (ox0 + oxl) * C::S - 1;                       // image x of tap kx = 0
                f32x4 w[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) w[k] = *(const f32x4 *)&wdc[k * WDS + 4 * c4];
                if (C::WHOLE) {     // no halo ring in LDS: out-of-image column taps get weight 0 (their address is clamped)
                    if (ixb < 0) { w[0] = z4; w[3] = z4; w[6] = z4; }
                    if (ixb + 2 >= C::HIN) { w[2] = z4; w[5] = z4; w[8] = z4; }
                }
                const int lx1 = ixb + 1 - ix0;
                const int lx0 = C::WHOLE ? (lx1 > 0 ? lx1 - 1 : 0) : lx1 - 1;
                const int lx2 = C::WHOLE ? (lx1 + 1 < C::IW ? lx1 + 1 : C::IW - 1) : lx1 + 1;
                const f32x4 sh = *(const f32x4 *)&wdc[10 * WDS + 4 * c4];
                const float *ebase = Es + (size_t)fi * C::IH * C::IW * C::ES + 4 * c4;
                f32x4 rb[3][3];
                auto load_row = [&](int iy, f32x4(&dst)[3]) {
                    int ly = iy - iy0;
                    if (C::WHOLE) ly = ly < 0 ? 0 : (ly > C::IH - 1 ? C::IH - 1 : ly);
                    const float *er = ebase + ly * C::IW * C::ES;
                    dst[0] = *(const f32x4 *)(er + lx0 * C::ES);
                    dst[1] = *(const f32x4 *)(er + lx1 * C::ES);
                    dst[2] = *(const f32x4 *)(er + lx2 * C::ES);
                    if (C::WHOLE && !((unsigned)iy < (unsigned)C::HIN)) { dst[0] = z4; dst[1] = z4; dst[2] = z4; }
                };
#pragma unroll
                for (int r = 0; r < C::RPS; ++r) {
                    const int oyl = seg * C::RPS + r;
                    if (oyl >= C::TH) break;
                    const int iyb = (oy0 + oyl) * C::S - 1;
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
                    const int po = (fi * C::TH + oyl) * C::TW + oxl;
                    *(f32x4 *)&Ds[po * C::ES + 4 * c4] = relu6_(a);
                }
            }
            if (!C::WLDS) fetch_a1(hc0 + C::HC < C::HID ? hc0 + C::HC : 0);   // next chunk (or next tile's chunk 0)
            SYN_LAP(3);
            __syncthreads();
            SYN_LAP(4);
            // ---- stage 3: project 1x1, K = this hidden chunk, accumulators stay in registers ----
            {
                auto ldab = [&](int kc, f32x4(&a)[C::AN], f32x4(&b)[C::AP]) {
#pragma unroll
                    for (int i = 0; i < C::AN; ++i) {
                        const int nt = wn + i * C::WN;
                        if (C::WLDS) a[i] = *(const f32x4 *)&Wlp[(((nt < C::NT_O ? nt : 0) * (C::HID / 16)) + hc0 / 16 + kc) * 256 + lane * 4];
                        else a[i] = a3[i][kc];
                    }
#pragma unroll
                    for (int j = 0; j < C::AP; ++j) {
                        const int pt = wp + j * C::WP;
                        b[j] = *(const f32x4 *)&Ds[((pt < C::PT_O ? pt : 0) * 16 + r16) * C::ES + kc * 16 + 4 * g];
                    }
                };
                f32x4 ac[C::AN], bc[C::AP], an[C::AN], bn[C::AP];
                ldab(0, ac, bc);
#pragma unroll
                for (int kc = 0; kc < C::KC3; ++kc) {
                    if (kc + 1 < C::KC3) ldab(kc + 1, an, bn);
                    // unconditional straight-line MFMAs: tiles past NT_O / PT_O use clamped operands and are never stored
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int i = 0; i < C::AN; ++i)
#pragma unroll
                            for (int j = 0; j < C::AP; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[i][s], bc[j][s], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < C::AN; ++i) ac[i] = an[i];
#pragma unroll
                    for (int j = 0; j < C::AP; ++j) bc[j] = bn[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            SYN_LAP(5);
            // no barrier here: the next stage 1 only writes Es / Wds (their readers finished before the barrier
            // above), and Ds is rewritten only after the next barrier, which every wave reaches after stage 3.
        }

        // ---- epilogue: BN (+ residual from the LDS input tile) and NHWC store ----
#pragma unroll
        for (int i = 0; i < C::AN; ++i) {
            const int nt = wn + i * C::WN;
            const int n = nt * 16 + 4 * g;
            if (nt >= C::NT_O || n >= C::COUT) continue;
#pragma unroll
            for (int j = 0; j < C::AP; ++j) {
                const int pt = wp + j * C::WP;
                const int po = pt * 16 + r16;
                if (pt >= C::PT_O || po >= C::POUT) continue;
                const int fi = po / (C::TH * C::TW), r = po % (C::TH * C::TW);
                const int oyl = r / C::TW, oxl = r % C::TW;
                const int f = f0 + fi, oy = oy0 + oyl, ox = ox0 + oxl;
                if (f >= B || oy >= C::HOUT || ox >= C::HOUT) continue;
                f32x4 v = acc[i][j];
                if (C::RES) v += *(const f32x4 *)&Xs[((fi * C::IH + (oy - iy0)) * C::IW + (ox - ix0)) * C::XS + n];   // x + conv(x)
                *(f32x4 *)&Y[((size_t)(f * C::HOUT + oy) * C::HOUT + ox) * C::COUT + n] = v;
            }
        }
        ++ntiles_done;
        if (tile + (int)gridDim.x < total_tiles) __syncthreads();   // Xs (residual) is rewritten by the next tile
        SYN_LAP(6);
    }
    if (PROF && tid == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&prof[i], pt_[i]);
        atomicAdd(&prof[7], ntiles_done);
    }
}

template <class C>
static void launch_cfg(const FusedBlockArgs &a, int B, hipStream_t s, int wgs_per_cu) {
    const int groups = (B + C::NF - 1) / C::NF;
    const int total = groups * C::TILES_Y * C::TILES_X;
    int grid = total;
    if (C::WLDS && grid > 256 * wgs_per_cu) grid = 256 * wgs_per_cu;     // persistent: one resident wave of workgroups
    if (a.prof)
        fused_block_kernel<C, true><<<grid, C::NW * 64, 0, s>>>(a.X, a.We, a.e_scale, a.e_shift, a.Wd, a.d_scale, a.d_shift,
                                                                 a.Wp, a.p_scale, a.p_shift, a.Y, B, total, a.prof);
    else
        fused_block_kernel<C, false><<<grid, C::NW * 64, 0, s>>>(a.X, a.We, a.e_scale, a.e_shift, a.Wd, a.d_scale, a.d_shift,
                                                                  a.Wp, a.p_scale, a.p_shift, a.Y, B, total);
}

//                      CIN  HID COUT HIN S  RES    TH  TW NF  HC NW EPB WN WP  WLDS
using Cfg2 = BlockCfg<  16,  96,  24, 60, 2, false, 10, 10, 1, 32, 8, 7, 2, 4, true>;       // features.2   60 -> 30
using Cfg3 = BlockCfg<  24, 144,  24, 30, 1, true,  10, 10, 1, 48, 8, 2, 2, 4, true>;       // features.3   30
using Cfg4 = BlockCfg<  24, 144,  32, 30, 2, false,  5,  5, 1, 144, 8, 2, 2, 4, true>;   // features.4   30 -> 15
using Cfg5 = BlockCfg<  32, 192,  32, 15, 1, true,  15, 15, 1, 32, 8, 4, 2, 4, false>;   // features.5,6 15
using Cfg7 = BlockCfg<  32, 192,  64, 15, 2, false,  8,  8, 1, 32, 4, 4, 4, 1, false>;   // features.7   15 -> 8
using Cfg8 = BlockCfg<  64, 384,  64,  8, 1, true,   8,  8, 1, 64, 4, 4, 4, 1, false>;   // features.8-10
using Cfg11 = BlockCfg< 64, 384,  96,  8, 1, false,  8,  8, 1, 64, 4, 4, 2, 2, false>;   // features.11
using Cfg12 = BlockCfg< 96, 576,  96,  8, 1, true,   8,  8, 1, 64, 4, 4, 2, 2, false>;   // features.12,13
using Cfg14 = BlockCfg< 96, 576, 160,  8, 2, false,  4,  4, 2, 64, 4, 4, 4, 1, false>;   // features.14  8 -> 4
using Cfg15 = BlockCfg<160, 960, 160,  4, 1, true,   4,  4, 4, 64, 4, 4, 2, 2, false>;   // features.15,16
using Cfg17 = BlockCfg<160, 960, 320,  4, 1, false,  4,  4, 4, 64, 4, 4, 4, 1, false>;   // features.17

bool launch_fused_block(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    switch (feature) {
        case 2: launch_cfg<Cfg2>(a, B, s, 1); return true;
        case 3: launch_cfg<Cfg3>(a, B, s, 1); return true;
        case 4: launch_cfg<Cfg4>(a, B, s, 1); return true;
        case 5: case 6: launch_cfg<Cfg5>(a, B, s, 1); return true;
        case 7: launch_cfg<Cfg7>(a, B, s, 1); return true;
        case 8: case 9: case 10: launch_cfg<Cfg8>(a, B, s, 1); return true;
        case 11: launch_cfg<Cfg11>(a, B, s, 1); return true;
        case 12: case 13: launch_cfg<Cfg12>(a, B, s, 1); return true;
        case 14: launch_cfg<Cfg14>(a, B, s, 1); return true;
        case 15: case 16: launch_cfg<Cfg15>(a, B, s, 1); return true;
        case 17: launch_cfg<Cfg17>(a, B, s, 1); return true;
        default: return false;
    }
}

}  // namespace syn
