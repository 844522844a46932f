// Register-resident fused inverted-residual block for the 8x8 MobileNetV2 blocks (features.8-13).
// Reference: backbone_nets/mobilenetv2_backbone.py:45-74 (InvertedResidual.forward), :33-42 (ConvBNReLU).
//
// The tiled kernel (fused_block_bf3.hip) walks the hidden width in chunks behind two workgroup barriers per chunk and moves
// every hidden activation through LDS twice; with two 4-wave workgroups per CU the matrix pipe is busy ~30 % of the time
// (profiles/r2/stage_profile_b1024.txt).  Here ONE WAVE carries a whole face through a hidden group of 32 channels without
// leaving registers, and a face is shared by two waves that split the hidden groups between them (even / odd):
//
//   * pixel layout: the 64 pixels of a face are four 16-column blocks of v_mfma_f32_16x16x32_bf16; block r, lane column
//     n = l & 15 is pixel (y = r + 4 (n >> 3), x = n & 7).  The vertical neighbours of a pixel are then the SAME LANE of
//     blocks r-1 / r+1 -- except row 3 <-> row 4, which is a shift by 8 lanes inside a 16-lane DPP row (row_shr:8 / row_shl:8,
//     whose zero fill is exactly the image border) -- and the horizontal ones are row_shr:1 / row_shl:1, with the filter
//     column zeroed on the lanes where that shift crosses x = 0 | 7;
//   * expand: D[tile t of 16 channels][block r] = shift + We . X on the exact 3-way bf16 split (6 products); the block input is
//     staged ONCE per face as pre-split B fragments in LDS, the weights stream from L2 straight into registers;
//   * depthwise 3x3 + BN shift + ReLU6 on the D registers (same tap order as the other kernels), split into bf16 pieces in
//     place: lane group g = l >> 4 holds channels 4g..4g+3 of both tiles = the 8 K slots of ONE k32 step of the project
//     GEMM (the host packs the project weights in that K order: slot e < 4 -> channel 4g + e, else 16 + 4g + e - 4);
//   * project: acc[out tile][block] += Wp[:, group] . D, accumulators in registers across all groups of the wave;
//   * the two waves of a face exchange half of their partial sums through LDS at the end (stream 0 + stream 1, fixed order),
//     add the BN shift and the residual and store NHWC.  Barriers: one after staging, two at the end.
#include "syn_internal.h"

#include <cstdio>
#include <cstdlib>

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ float relu6l(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f); }
// exact 3-way bf16 split of two floats, packed (x0 -> low half, x1 -> high half) per piece
__device__ __forceinline__ void split2l(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    const float s0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}
// ... into component d of the three piece vectors
__device__ __forceinline__ void split2v(float x0, float x1, u32x4 (&pc)[3], int d) {
    unsigned h, m, l;
    split2l(x0, x1, h, m, l);
    pc[0][d] = h; pc[1][d] = m; pc[2][d] = l;
}
__device__ __forceinline__ f32x4 mfmal(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six partial products of weight >= 2^-16, smallest terms first (same order as the other bf16x3 kernels)
__device__ __forceinline__ f32x4 mac6l(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {
    c = mfmal(a[2], b[0], c);
    c = mfmal(a[0], b[2], c);
    c = mfmal(a[1], b[1], c);
    c = mfmal(a[1], b[0], c);
    c = mfmal(a[0], b[1], c);
    c = mfmal(a[0], b[0], c);
    return c;
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
constexpr int kRowShr1 = 0x111, kRowShl1 = 0x101, kRowShr8 = 0x118, kRowShl8 = 0x108;   // lane n <- n-1 | n+1 | n-8 | n+8 of its 16-lane row, else 0
}  // namespace

template <int CIN_, int HID_, int COUT_, bool RES_, int FPW_ = 2>
struct LbCfg {
    static constexpr int CIN = CIN_, HID = HID_, COUT = COUT_, FPW = FPW_;
    static constexpr bool RES = RES_;
    static constexpr int KE = CIN / 32;                  // k32 steps of the expand GEMM
    static constexpr int NG = HID / 32;                  // hidden groups
    static constexpr int MT = COUT / 16;                 // output channel tiles
    static constexpr int NS = 2;                         // waves per face (hidden groups s, s + 2, ...)
    static constexpr int NW = FPW * NS, NT = NW * 64;
    static constexpr int XF_DW = KE * 4 * 3 * 256;       // block input of one face as fragments [KE][block 4][piece 3][lane 64][4 dwords]
    static constexpr int RED_DW = MT * 4 * 256;          // exchange buffer of one face: [stream 2][MT / 2][block 4][lane 64][4]
    static constexpr int LDS_DW = FPW * XF_DW;
    static_assert(CIN % 32 == 0 && HID % 64 == 0 && COUT % 32 == 0, "k32 steps, two streams, two halves of the output tiles");
    static_assert(RED_DW <= XF_DW, "the exchange buffer reuses the fragments of its face");
    static_assert(!RES || CIN == COUT, "residual only on same-width blocks");
    static_assert(2 * LDS_DW * 4 <= 160 * 1024, "two workgroups per CU");
};

// compiler fence between the phases of a hidden group: without it every load of a group is hoisted to the top of the loop body
// and unchained arithmetic floats across the scheduling barriers (~370 registers live)
#define SYNL_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <class C>
__global__ __launch_bounds__(C::NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_block_lb_kernel(const float *__restrict__ X, const unsigned *__restrict__ We3 /*[HID/16][KE][3][64][4]*/,
                           const float *__restrict__ e_shift, const float *__restrict__ Wd /*[9][HID] scaled*/,
                           const float *__restrict__ d_shift, const unsigned *__restrict__ Wlb /*[NG][MT][3][64][4]*/,
                           const float *__restrict__ p_shift, float *__restrict__ Y, int B) {
    __shared__ __attribute__((aligned(16))) unsigned smem[C::LDS_DW];
    constexpr int KE = C::KE, MT = C::MT, CIN = C::CIN, HID = C::HID, COUT = C::COUT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = wave >> 1, st = wave & 1;
    const int f = blockIdx.x * C::FPW + fl;
    const bool real = f < B;
    const int fc = real ? f : B - 1;
    const int n = lane & 15, g = lane >> 4;
    const unsigned l4 = lane * 4, g4 = g * 4;
    unsigned *Xf = smem + fl * C::XF_DW;
    const int pix0 = 32 * (n >> 3) + (n & 7);           // pixel index of (block 0, lane column n); block r adds 8 r

    // ---- stage: block input of this face -> pre-split B fragments (this wave: blocks 2 st, 2 st + 1) ----
    {
        f32x4 xv[KE][2][2];
#pragma unroll
        for (int kc = 0; kc < KE; ++kc)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float *src = X + ((size_t)fc * 64 + pix0 + 8 * (2 * st + rr)) * CIN + 32 * kc + 8 * g;
                xv[kc][rr][0] = *(const f32x4 *)src;
                xv[kc][rr][1] = *(const f32x4 *)(src + 4);
            }
#pragma unroll
        for (int kc = 0; kc < KE; ++kc)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                f32x4 a = xv[kc][rr][0], b = xv[kc][rr][1];
                if (!real) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }
                u32x4 pc[3];
                split2v(a[0], a[1], pc, 0);
                split2v(a[2], a[3], pc, 1);
                split2v(b[0], b[1], pc, 2);
                split2v(b[2], b[3], pc, 3);
#pragma unroll
                for (int p = 0; p < 3; ++p) *(u32x4 *)&Xf[((kc * 4 + 2 * st + rr) * 3 + p) * 256 + lane * 4] = pc[p];
            }
    }
    const float mL = (n & 7) != 0 ? 1.f : 0.f, mR = (n & 7) != 7 ? 1.f : 0.f;
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // Weight fragments are fetched one step ahead of their use and scheduling barriers keep the compiler from hoisting every
    // load of a group to its top (that spills): Ae = expand fragments of the current k32 step, Ap = project fragments of the
    // current output tile.
    u32x4 Ae[2][3];
    auto fetch_e = [&](int G, int kc) __attribute__((always_inline)) {
        const unsigned *we = We3 + (size_t)G * (2 * KE * 768);           // wave-uniform base + 32-bit lane offset
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) Ae[t][p] = *(const u32x4 *)((we + ((t * KE + kc) * 3 + p) * 256) + l4);
    };
    fetch_e(st, 0);
    for (int G = st; G < C::NG; G += C::NS) {
        // ---- expand 1x1 (bf16 x3) + BN shift + ReLU6: D[t][r], channels 32 G + 16 t + 4 g + i of pixel (r, n) ----
        f32x4 D[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 es = *(const f32x4 *)((e_shift + 32 * G + 16 * t) + g4);
#pragma unroll
            for (int r = 0; r < 4; ++r) D[t][r] = es;
        }
#pragma unroll
        for (int kc = 0; kc < KE; ++kc) {
            u32x4 A[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) A[t][p] = Ae[t][p];
            if (kc + 1 < KE) fetch_e(G, kc + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u32x4 Bx[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) Bx[p] = *(const u32x4 *)&Xf[((kc * 4 + r) * 3 + p) * 256 + lane * 4];
                D[0][r] = mac6l(A[0], Bx, D[0][r]);
                D[1][r] = mac6l(A[1], Bx, D[1][r]);
                if (r == 1) SYNL_FENCE();           // at most two blocks' fragments in flight
            }
            SYNL_FENCE();
        }
        // ---- depthwise 3x3 + BN shift + ReLU6, split in place into the B operand of the project step ----
        const unsigned *wp = Wlb + (size_t)G * (MT * 768);
        u32x4 Ap[3];
        u32x4 Bd[4][3];
        // two channels (one packed K dword) at a time: 18 filter registers live instead of 36
#pragma unroll
        for (int th = 0; th < 4; ++th) {
            const int t = th >> 1, hf = th & 1;
            if (th == 3) {                              // first project fragments: in flight behind the last depthwise pass
#pragma unroll
                for (int p = 0; p < 3; ++p) Ap[p] = *(const u32x4 *)((wp + p * 256) + l4);
            }
            const int c0 = 32 * G + 16 * t + 2 * hf;               // + 4 g per lane group
            f32x2 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *(const f32x2 *)((Wd + c0 + k * HID) + g4);
            const f32x2 dsh = *(const f32x2 *)((d_shift + c0) + g4);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { w[3 * dy] *= mL; w[3 * dy + 2] *= mR; }
            f32x2 E[4], O[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                E[r][0] = relu6l(D[t][r][2 * hf]);
                E[r][1] = relu6l(D[t][r][2 * hf + 1]);
                O[r] = dsh;
            }
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
                    O[r] += rt * w[3 * dy + 2];
                    asm volatile("" : "+v"(O[r]));
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                split2v(relu6l(O[r][0]), relu6l(O[r][1]), Bd[r], th);
                if (hf) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(Bd[r][p]));
                }
            }
            SYNL_FENCE();
        }
        // ---- project 1x1 (bf16 x3), K = this group ----
        if (G + C::NS < C::NG) fetch_e(G + C::NS, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            u32x4 A[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) A[p] = Ap[p];
            if (mt + 1 < MT) {
#pragma unroll
                for (int p = 0; p < 3; ++p) Ap[p] = *(const u32x4 *)((wp + ((mt + 1) * 3 + p) * 256) + l4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = mac6l(A, Bd[r], acc[mt][r]);
            SYNL_FENCE();
        }
    }

    // ---- exchange: wave `st` keeps the output tiles mt with (mt & 1) == st and hands the others to its partner ----
    __syncthreads();                                     // every wave is done reading the fragments
    float *Red = reinterpret_cast<float *>(Xf);
    int le = lane;
    asm volatile("" : "+v"(le));                         // (output addresses are computed here, not carried through the loop)
    const int ne = le & 15, ge = le >> 4, pixe = 32 * (ne >> 3) + (ne & 7);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if ((mt & 1) == st) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) *(f32x4 *)&Red[(((st * (MT / 2) + (mt >> 1)) * 4 + r) * 64 + lane) * 4] = acc[mt][r];
    }
    __syncthreads();
    if (!real) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if ((mt & 1) != st) continue;
        const int nch = 16 * mt + 4 * ge;
        const f32x4 psh = *(const f32x4 *)&p_shift[nch];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 o = *(const f32x4 *)&Red[((((1 - st) * (MT / 2) + (mt >> 1)) * 4 + r) * 64 + lane) * 4];
            f32x4 v = st == 0 ? acc[mt][r] + o : o + acc[mt][r];        // stream 0 + stream 1
            v += psh;
            const size_t at = ((size_t)f * 64 + pixe + 8 * r) * COUT + nch;
            if (C::RES) v += *(const f32x4 *)&X[at];
            *(f32x4 *)&Y[at] = v;
        }
    }
}

template <class C>
static void launch_lb(const FusedBlockArgs &a, int B, hipStream_t s) {
    const int grid = (B + C::FPW - 1) / C::FPW;
    fused_block_lb_kernel<C><<<grid, C::NT, 0, s>>>(a.X, a.We3, a.e_shift, a.Wd, a.d_shift, a.Alb_p, a.p_shift, a.Y, B);
}

//                    CIN  HID COUT  RES
using L8 = LbCfg<      64, 384,  64, true>;     // features.8-10
using L11 = LbCfg<     64, 384,  96, false>;    // features.11
using L12 = LbCfg<     96, 576,  96, true>;     // features.12, 13

static int lb_min_batch(int feature) {
    // below: too few workgroups to put two on every CU (the tiled kernel is faster); SYN_LB_MIN<f> overrides
    char name[32];
    snprintf(name, sizeof name, "SYN_LB_MIN%d", feature);
    if (const char *e = getenv(name)) return atoi(e);
    return 768;
}

bool launch_fused_block_lb(int feature, const FusedBlockArgs &a, int B, hipStream_t s) {
    if (!a.We3 || !a.Alb_p || a.prof) return false;
    if (B < lb_min_batch(feature)) return false;
    switch (feature) {
        case 8: case 9: case 10: launch_lb<L8>(a, B, s); return true;
        case 11: launch_lb<L11>(a, B, s); return true;
        case 12: case 13: launch_lb<L12>(a, B, s); return true;
        default: return false;
    }
}

}  // namespace syn
