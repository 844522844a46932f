// gfx950 kernels for the 3DMM parameter -> geometry stage of SynergyNet's inference path:
//   reconstruct_vertex_62  (reference synergy3DMM.py:116-149, batched torch)
//   param2vert / _predict_vertices (reference utils/inference.py:64-84,127-138, numpy per face)
//   predict_pose -> parse_pose -> P2sRt -> matrix2angle_corr (utils/inference.py:33-62,86-92,146-157)
//
//   S[b] = u + W_shp a_shp[b] + W_exp a_exp[b]          (3N-vector, xyz interleaved)
//   V[b] = P[b] * reshape(S[b], (3,N), 'F') + t[b];  V[b][1] = 121 - V[b][1];  ROI affine
//
// The contraction is a [B,52] x [52,3N] GEMM (K = 40 shape + 10 expression + the mean u with
// coefficient 1 + one zero pad) on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32, with
// rows = faces and columns = vertices so that a stored row segment is 32 consecutive vertices of
// one face (128 B).  The output (638,580 B per face) is the compulsory HBM traffic; the basis is
// read once per wave and kept in registers while the wave walks its share of the faces.
#include <cstdio>
#include <cstdlib>

#include "syn_internal.h"

namespace syn {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store to be
// acknowledged (vmcnt(0)), which would expose the latency of each face tile's 48 KB of output stores once per iteration.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kFaceRec = 64;   // floats per face in the prepared record: alpha[52] | M[9] | T[3]
constexpr int kStageStride = 132;  // floats per row of the store-transpose stage: 4 tiles x 32 vertices + pad (16-byte aligned rows)

// -------------------------------------------------------------------------------------
// Per-face prologue: de-whiten (param*std+mean, synergy3DMM.py:127), split into pose / alpha
// (parse_param_62, :30-37) and fold the pose matrix with the y flip (:139,:147) and the optional
// ROI affine (utils/inference.py:129-136) into one affine map  out = Mx * S + T.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void recon_prep_kernel(const float *__restrict__ param, const float *__restrict__ mean,
                                                        const float *__restrict__ stdv, const float *__restrict__ roi,
                                                        int transform, float *__restrict__ rec, int B) {
    const int b = blockIdx.x;
    const int l = threadIdx.x;
    __shared__ float p[64];
    if (l < kParam) p[l] = param[(size_t)b * kParam + l] * stdv[l] + mean[l];
    __syncthreads();
    float *r = rec + (size_t)b * kFaceRec;
    if (l < 50) r[l] = p[12 + l];
    else if (l == 50) r[l] = 1.0f;       // coefficient of the mean shape u
    else if (l == 51) r[l] = 0.0f;
    else if (l < 55) {
        const int c = l - 52;            // output row: 0 = x, 1 = y, 2 = z
        float sc = 1.0f, of = 0.0f;
        if (roi) {
            const float sx = roi[b * 5 + 0], sy = roi[b * 5 + 1], ex = roi[b * 5 + 2], ey = roi[b * 5 + 3];
            const float scx = (ex - sx) / 120.0f, scy = (ey - sy) / 120.0f;
            if (c == 0) { sc = scx; of = sx; }
            else if (c == 1) { sc = scy; of = sy; }
            else { sc = (scx + scy) * 0.5f; }
        }
        float m0 = p[4 * c + 0], m1 = p[4 * c + 1], m2 = p[4 * c + 2], t = p[4 * c + 3];
        if (transform && c == 1) { m0 = -m0; m1 = -m1; m2 = -m2; t = (float)(kImg + 1) - t; }
        r[52 + 3 * c + 0] = m0 * sc;
        r[52 + 3 * c + 1] = m1 * sc;
        r[52 + 3 * c + 2] = m2 * sc;
        r[61 + c] = t * sc + of;
    }
}

// -------------------------------------------------------------------------------------
// Main contraction + pose epilogue.
//
// v_mfma_f32_32x32x2_f32:  D[i][j] += sum_{k<2} A[i][k] B[k][j];  lane l supplies A[i = l&31][k = l>>5]
// and B[k = l>>5][j = l&31]; lane l owns D column j = l&31, rows i = (r&3) + 8*(r>>2) + 4*(l>>5), r < 16.
//   rows i  = 32 faces       (A operand = alpha)
//   cols j  = 32 vertices    (B operand = one coordinate plane of the basis)
// K = 52 is walked as 6 chunks of 8 + one chunk of 4: per chunk a lane fetches ONE float4 (float2 for
// the tail) holding logical k = 8t + 4h + s (h = l>>5) and feeds element s to MFMA step s, identically
// for both operands.  The basis is pre-packed by the host in exactly that per-lane order
//   Bp[tile][coord][chunk][lane][4]      (tile = 32 vertices; 6*1 KiB + 512 B per coord)
// so every basis load of a wave is one fully coalesced 1 KiB (512 B) line set.
// A wave owns vertex tile T and a contiguous range of 32-face tiles: 3 x 26 basis registers stay
// resident while alpha fragments stream in from the 256-byte per-face records.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void recon_kernel(const float *__restrict__ rec, const float *__restrict__ basis,
                                                    float *__restrict__ out, int B, int n_vert, int pitch, int n_tiles,
                                                    int n_split, int ftiles_per_split, int n_ftiles, int n_units) {
    __shared__ __attribute__((aligned(16))) float smt[4][32][12];
    __shared__ __attribute__((aligned(16))) float stage[96 * kStageStride];   // [face*3 + coord][4 tiles x 32 vertices]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware unit order: block b runs on XCD b % 8; give each XCD a contiguous range of (tile group, split) units so
    // that the row segments of neighbouring tile groups meet in the same L2 and leave it as full lines
    const int per_xcd = (n_units + 7) / 8;
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_units) return;
    const int tg = unit / n_split, split = unit - tg * n_split;
    int T = tg * 4 + wave;                        // the 4 waves of a workgroup own 4 consecutive vertex tiles
    const bool t_ok = T < n_tiles;
    T = t_ok ? T : n_tiles - 1;
    const int j = lane & 31, h = lane >> 5;

    // resident basis fragments for the three coordinate planes of this vertex tile
    f32x4 bw[3][6];
    f32x2 bt[3];
    const float *bp = basis + (size_t)T * 3 * (kBasisK * 32);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *bc = bp + c * (kBasisK * 32);
#pragma unroll
        for (int t = 0; t < 6; ++t) bw[c][t] = *(const f32x4 *)(bc + t * 256 + lane * 4);
        bt[c] = *(const f32x2 *)(bc + 6 * 256 + lane * 2);
    }
    const int ft0 = split * ftiles_per_split;
    int ft1 = ft0 + ftiles_per_split;
    ft1 = ft1 < n_ftiles ? ft1 : n_ftiles;
    float(*mt)[12] = smt[wave];
    const int v_base = tg * 128;                  // first vertex of the workgroup's 128-vertex run
    // the two co-resident workgroups of a CU start together; delaying every second one by about half an
    // iteration lets one's MFMA phase run in the shadow of the other's epilogue / store phase
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(56);

    for (int ft = ft0; ft < ft1; ++ft) {
        const int f0 = ft * 32;
        int fa = f0 + j;
        fa = fa < B ? fa : B - 1;
        const float *ra = rec + (size_t)fa * kFaceRec;
        f32x4 aw[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) aw[t] = *(const f32x4 *)(ra + 8 * t + 4 * h);
        const f32x2 at = *(const f32x2 *)(ra + 48 + 2 * h);
        // stage this face tile's 32 x (M[9],T[3]) into the wave's private LDS slice
        if (lane < 32) {
#pragma unroll
            for (int q = 0; q < 3; ++q) *(f32x4 *)&mt[lane][4 * q] = *(const f32x4 *)(ra + 52 + 4 * q);
        }
        f32x16 acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[t][s], bw[c][t][s], acc[c], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bt[c][s], acc[c], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // pose epilogue in registers -> this wave's 32-vertex column block of the workgroup stage
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const f32x4 q0 = *(const f32x4 *)&mt[i][0];
            const f32x4 q1 = *(const f32x4 *)&mt[i][4];
            const f32x4 q2 = *(const f32x4 *)&mt[i][8];
            const float sx = acc[0][r], sy = acc[1][r], sz = acc[2][r];
            float *st = stage + (i * 3) * kStageStride + wave * 32 + j;
            st[0] = q0[0] * sx + q0[1] * sy + q0[2] * sz + q2[1];
            st[kStageStride] = q0[3] * sx + q1[0] * sy + q1[1] * sz + q2[2];
            st[2 * kStageStride] = q1[2] * sx + q1[3] * sy + q2[0] * sz + q2[3];
        }
        lds_barrier();
        // cooperative store: 32 lanes x float4 = one 512-byte run of one (face, coord) row; 2 rows per instruction
        {
            const int seg = threadIdx.x & 31, rsub = threadIdx.x >> 5;
            const int vq = v_base + 4 * seg;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int row = k * 8 + rsub;                 // = face_in_tile * 3 + coord
                const int f = f0 + row / 3, c = row % 3;
                const f32x4 vv = *(const f32x4 *)&stage[row * kStageStride + 4 * seg];
                if (f < B) {
                    float *o = out + ((size_t)f * 3 + c) * pitch + vq;
                    if (vq + 3 < n_vert) *(f32x4 *)o = vv;       // 4-byte aligned 16-byte store (rows are n_vert floats)
                    else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (vq + t < n_vert) o[t] = vv[t];
                    }
                }
            }
        }
        lds_barrier();     // stage and the M/T slices are rewritten by the next face tile
    }
}

// rec: [B,64] scratch for the per-face records (part of the library workspace)
void launch_reconstruct(const float *param, const float *mean62, const float *std62, const float *basis, int n_vert,
                        int nvp, const float *roi, int transform, float *out, int pitch, int B, hipStream_t s, float *rec) {
    recon_prep_kernel<<<B, 64, 0, s>>>(param, mean62, std62, roi, transform, rec, B);
    const int n_tiles = nvp / 32;
    const int n_groups = (n_tiles + 3) / 4;                   // a workgroup = 4 consecutive vertex tiles
    const int n_ftiles = (B + 31) / 32;
    int n_split = (3072 + n_groups - 1) / n_groups;           // aim for >= 3072 workgroups: 6+ rounds of 512 resident ones,
                                                              // so the partially filled last round costs < 10 %
    n_split = n_split < 1 ? 1 : n_split;
    n_split = n_split > n_ftiles ? n_ftiles : n_split;
    const int per = (n_ftiles + n_split - 1) / n_split;
    n_split = (n_ftiles + per - 1) / per;
    const int n_units = n_groups * n_split;
    const int grid = ((n_units + 7) / 8) * 8;
    recon_kernel<<<grid, 256, 0, s>>>(rec, basis, out, B, n_vert, pitch, n_tiles, n_split, per, n_ftiles, n_units);
}

// =====================================================================================
// The same contraction on v_mfma_f32_32x32x16_f16 at fp32-equivalent accuracy.
//
// Both operands are carried as two fp16 pieces x = a + b (a = fp16(x), b = fp16(x - a), toward zero: 22 significant bits; the
// basis is split by the host, alpha by the prologue kernel) and each block product is rebuilt from three partial products
// a a + a b + b a (b b <= 2^-22 dropped; fp16 x fp16 is exact in fp32, fp32 accumulation).  K = 48 runs as 3 steps of 16 = 27 MFMAs
// per face tile and wave (the exact 3-way bf16 split used before: 54; the fp32-input MFMA: 78 of 64 cycles), plus ONE more MFMA
// whose K slots carry the partial products of the last two expression columns and of the mean shape.  fp16's exponent range: every
// basis COLUMN k (and the mean shape) is scaled by its own power of two 2^e_k (host: max |column| in [2^13, 2^14)), coefficient k by
// 2^-e_k (folded into its de-whitening constants: exact), each face's scaled coefficient vector by its own power of two Sa
// (prologue: max in [2^13, 2^14)), and 1 / Sa is folded into the face's pose matrix.  Nothing but the split itself rounds.
// =====================================================================================
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8r __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2r __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ void split2r(float x0, float x1, unsigned &a, unsigned &b) {
    // a = fp16 pair (toward zero); x - a in ONE v_fma_mix_f32 per value (fp16 source operand: no v_cvt_f32_f16)
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8r, a), __builtin_bit_cast(f16x8r, b), c, 0, 0, 0);
}
}  // namespace

// Prologue: one workgroup (64 lanes) per 32-face tile.  Lane (i = l&31, hh = l>>5) is face f0+i: it de-whitens the 24
// shape/expression coefficients k = 16*step + 8*hh + e it feeds to the MFMA, splits them and writes them in operand
// order (plus the fourth-step fragment, see below); lanes hh = 0 also write the face's 16-float record M[9] | T[3] | 0 x 4.
__global__ __launch_bounds__(64) void recon_prep_f16_kernel(const float *__restrict__ param, const float *__restrict__ mean,
                                                           const float *__restrict__ stdv, const float *__restrict__ roi,
                                                           int transform, unsigned *__restrict__ rec3, int B) {
    const int ft = blockIdx.x, l = threadIdx.x, i = l & 31, hh = l >> 5;
    const int b = ft * 32 + i;
    const bool ok = b < B;
    const float *pp = param + (size_t)(ok ? b : 0) * kParam;
    unsigned *rt = rec3 + (size_t)ft * kRecTileF16;
    // this lane's 24 coefficients + (both halves) columns 48, 49; the face's scale Sa = 2^e with max |alpha| Sa in [2^13, 2^14)
    float al[3][8];
    float amax = 0.f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 12 + 16 * ks + 8 * hh + e;                     // alpha_k = param[12 + k] (parse_param_62)
            al[ks][e] = ok ? pp[k] * stdv[k] + mean[k] : 0.f;
            amax = fmaxf(amax, fabsf(al[ks][e]));
        }
    const float a48 = ok ? pp[12 + 48] * stdv[12 + 48] + mean[12 + 48] : 0.f;
    const float a49 = ok ? pp[12 + 49] * stdv[12 + 49] + mean[12 + 49] : 0.f;
    amax = fmaxf(amax, fmaxf(fabsf(a48), fabsf(a49)));
    amax = fmaxf(amax, __shfl_xor(amax, 32));                            // the other half of the face's coefficients
    // `mean` / `stdv` are the COLUMN-SCALED de-whitening constants (synergy_abi.hip pack_basis): al[] = alpha_k 2^-e_k for the basis
    // column's own power of two 2^e_k, and cu = 2^-e_u is the coefficient of the scaled mean shape -- it takes part in the maximum
    const float cu = mean[62];
    amax = fmaxf(amax, cu);
    int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xff) - 126;     // amax = m 2^ex, m in [0.5, 1)
    ex = 14 - ex;
    // cu Sa is an operand of the fourth k-step: it must be a normal fp16 number (>= 2^-14; <= 2^14 holds because cu <= amax)
    const int ex_cu = (int)((__builtin_bit_cast(unsigned, cu) >> 23) & 0xff) - 127;   // cu = 2^ex_cu
    ex = ex < -14 - ex_cu ? -14 - ex_cu : ex;
    ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
    const float Sa = __builtin_bit_cast(float, (unsigned)(127 + ex) << 23);
    const float inv_ab = __builtin_bit_cast(float, (unsigned)(127 - ex) << 23);  // the column scales cancel in every product: only Sa is left
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        u32x4 pc[2];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned a, b2;
            split2r(al[ks][2 * d] * Sa, al[ks][2 * d + 1] * Sa, a, b2);
            pc[0][d] = a; pc[1][d] = b2;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) *(u32x4 *)&rt[((ks * 2 + p) * 64 + l) * 4] = pc[p];
    }
    // fourth k16 step: the two expression columns 48, 49 and the mean shape ride on ONE more MFMA -- eight of its K slots (lane half
    // 0; the other half is zero) carry the three split partial products of column 48, of column 49 and the two pieces of the mean
    // times Sa:
    //   basis side   [b48a b48a b48b | b49a b49a b49b | ua ub]
    //   alpha side   [a48a a48b a48a | a49a a49b a49a | cu Sa, cu Sa]
    {
        unsigned a, b2;
        split2r(a48 * Sa, a49 * Sa, a, b2);                              // low half = piece of a48, high half = piece of a49
        const unsigned a8 = a & 0xffffu, b8 = b2 & 0xffffu, a9 = a >> 16, b9 = b2 >> 16;
        const unsigned sa16 = ok ? (__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(cu * Sa, cu * Sa)) & 0xffffu) : 0u;   // a power of two: exact
        u32x4 fx = {0u, 0u, 0u, 0u};
        if (hh == 0) fx = (u32x4){a8 | (b8 << 16), a8 | (a9 << 16), b9 | (a9 << 16), sa16 | (sa16 << 16)};
        *(u32x4 *)&rt[(6 * 64 + l) * 4] = fx;
    }
    if (hh == 0) {
        float r[16];
        float p12[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) p12[q] = ok ? pp[q] * stdv[q] + mean[q] : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sc = 1.0f, of = 0.0f;
            if (roi && ok) {
                const float sx = roi[b * 5 + 0], sy = roi[b * 5 + 1], ex = roi[b * 5 + 2], ey = roi[b * 5 + 3];
                const float scx = (ex - sx) / 120.0f, scy = (ey - sy) / 120.0f;
                if (c == 0) { sc = scx; of = sx; }
                else if (c == 1) { sc = scy; of = sy; }
                else { sc = (scx + scy) * 0.5f; }
            }
            float m0 = p12[4 * c + 0], m1 = p12[4 * c + 1], m2 = p12[4 * c + 2], t = p12[4 * c + 3];
            if (transform && c == 1) { m0 = -m0; m1 = -m1; m2 = -m2; t = (float)(kImg + 1) - t; }
            r[3 * c + 0] = m0 * sc * inv_ab; r[3 * c + 1] = m1 * sc * inv_ab; r[3 * c + 2] = m2 * sc * inv_ab;     // the MFMA result is Sa x the shape
            r[9 + c] = t * sc + of;
        }
        r[12] = 0.f; r[13] = 0.f; r[14] = 0.f; r[15] = 0.f;
        float *rr = reinterpret_cast<float *>(rt + 7 * 256) + i * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4 *)&rr[4 * q] = (f32x4){r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
    }
}

// WPG waves per workgroup = WPG consecutive vertex tiles = one WPG*128-byte run per (face, coord) row and iteration.  Rows of
// the [B,3,53215] output are only 4-byte aligned, so every run shares its first and last cache line with a neighbouring
// workgroup; longer runs mean fewer such split lines (tools/ubench/store_pattern.hip: 512 B runs 3.5 TB/s, 1 KiB ~4).
//
// Vector memory operations retire IN ORDER on this ISA (one vmcnt for loads and stores): a load issued after the 12 stores of
// a face tile cannot be consumed before those stores are acknowledged (~6k cycles).  So the kernel never waits on memory
// right after its stores: the operands of the next face tile (alpha pieces + face records, 11 KB, the same for all WPG waves)
// are fetched cooperatively after the MFMAs, parked in registers during the epilogue and written to the other half of a
// double-buffered LDS tile before the stage barrier -- by then the previous tile's stores have had a whole MFMA + epilogue
// phase to drain -- and nothing in the kernel spills (a scratch reload is a vector load too: one `s_waitcnt vmcnt(0)` per
// store instruction cost 10k cycles per face tile before).
#define RLAP(i) do { if (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tn = __builtin_amdgcn_s_memtime(); pt_[i] += tn - tk; tk = tn; } } while (0)
// FAST: every face tile in [ft_lo, ft_hi) is whole (32 faces) and the padded tile columns fit in the pitch -> the stores are 12
// unconditional instructions in straight-line code, the only shape for which the compiler's wait-count bookkeeping stays exact
// (see the store phase).  The launcher runs the ragged last face tile / packed outputs through the guarded instantiation.
// PK (round 4, the reference's PACKED [B,3,n_vert] layout, synergy3DMM.py:131-147): rows are n_vert floats apart, so row r starts
// 4 r n_vert bytes into the tensor and a fixed 128-vertex run shares its first and last 128-byte line with the neighbouring
// workgroups' runs -- partial lines written at different times, which HBM takes as read-modify-writes (same bytes through L2 as the
// pitched layout by the counters, 3.0-3.5 TB/s against 4.4-5.4).  Here a workgroup COMPUTES WPG tiles, [W g, W g + 32 WPG) with
// W = 32 (WPG - 1), but STORES per row the W of them that start on a line boundary of THAT row: [W g + s_r, W g + W + s_r),
// s_r = -r n_vert mod 32 (W = whole lines, so every group shares the shift of its row and the windows tile the row; group 0 also
// writes the s_r vertices in front of its window).  The kernel is bound by its per-tile latency chain as much as by the stores, so the
// recomputed tile costs its share of iterations: WPG = 4 (+33 %) 0.166-0.180 ms, WPG = 8 (+14 %, eight waves, one workgroup per CU,
// 832 workgroups) 0.153 ms = 4.28 TB/s at B = 1024 -- within 5 % of the pitched layout (0.146).
template <int WPG, bool FAST, bool PROF = false, bool PK = false>
__global__ __launch_bounds__(WPG * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void recon_f16_kernel(const unsigned *__restrict__ rec3, const unsigned *__restrict__ basis3, float *__restrict__ out, int B,
                     int n_vert, int pitch, int n_tiles, int n_split, int ftiles_per_split, int ft_lo, int ft_hi, int n_units,
                     unsigned long long *prof = nullptr) {
    unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = PROF ? __builtin_amdgcn_s_memtime() : 0ull, tn = 0;
    constexpr int SS = WPG * 32 + 4;                                          // stage row stride (16-byte aligned rows)
    constexpr int NTH = WPG * 64, TQ = kRecTileF16 / 4, NPF = (TQ + NTH - 1) / NTH, TQP = NPF * NTH;
    __shared__ __attribute__((aligned(16))) unsigned optile[2][TQP * 4];
    __shared__ __attribute__((aligned(16))) float stage[96 * SS];            // [face*3 + coord][WPG tiles x 32 vertices]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per_xcd = (n_units + 7) / 8;                                    // XCD-aware unit order (see recon_kernel)
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_units) return;
    const int tg = unit / n_split, split = unit - tg * n_split;
    constexpr int RUN = WPG * 32;                                             // vertices per row and workgroup
    static_assert(!PK || !FAST, "the packed-row schedule has its own store path");
    constexpr int PKW = (WPG - 1) * 32;                                       // PK: vertices a workgroup stores per row (WPG - 1 whole lines)
    int T = tg * (PK ? WPG - 1 : WPG) + wave;
    T = T < n_tiles ? T : n_tiles - 1;
    const int j = lane & 31, h = lane >> 5;

    // resident basis fragments of this vertex tile: 3 coords x (3 k16 steps x 3 pieces + the fourth-step fragment)
    u32x4 bb[3][3][2], bx[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned *bc = basis3 + ((size_t)T * 3 + c) * kBasisF16;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) bb[c][ks][p] = *(const u32x4 *)(bc + ((ks * 2 + p) * 64 + lane) * 4);
        bx[c] = *(const u32x4 *)(bc + (6 * 64 + lane) * 4);
    }
    const int ft0 = ft_lo + split * ftiles_per_split;
    int ft1 = ft0 + ftiles_per_split;
    ft1 = ft1 < ft_hi ? ft1 : ft_hi;
    const int v_base = tg * (PK ? PKW : RUN);
    if (ft0 >= ft1) return;                          // (workgroup-uniform)

    u32x4 pf[NPF];                                   // this thread's quads of the next operand tile
    auto fetch = [&](int ft) {
        const unsigned *rt = rec3 + (size_t)ft * kRecTileF16;
#pragma unroll
        for (int i = 0; i < NPF; ++i) pf[i] = *(const u32x4 *)(rt + 4 * (i * NTH + (int)threadIdx.x));
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) *(u32x4 *)&optile[buf][4 * (i * NTH + (int)threadIdx.x)] = pf[i];
    };
    fetch(ft0);
    park(0);                                         // waits for everything issued so far: the loop starts with no load in flight
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), visible to the compiler: no wait on the basis loads is left inside the loop
    RLAP(0);
    lds_barrier();

    for (int ft = ft0; ft < ft1; ++ft) {
        const int f0 = ft * 32, buf = (ft - ft0) & 1;
        const unsigned *ot = optile[buf];
        // MFMA orientation: A = basis (rows = the tile's 32 vertices), B = alpha (columns = the 32 faces), so lane (j, h) ends
        // up with face f0 + j and vertices i = (r&3) + 8(r>>2) + 4h: the per-face pose record is a per-LANE constant
        // (12 registers, no LDS reads in the epilogue) and four consecutive registers are four consecutive vertices (one
        // ds_write_b128 per coordinate into the stage).
        f32x16 acc[3];
        {
            const u32x4 ax = *(const u32x4 *)(ot + (6 * 64 + lane) * 4);
            const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = mfma32(bx[c], ax, z16);     // mean + columns 48, 49
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            u32x4 aa[2];                                                     // alpha pieces of this k16 step (LDS; the SIMD's other wave covers the latency)
#pragma unroll
            for (int p = 0; p < 2; ++p) aa[p] = *(const u32x4 *)(ot + ((ks * 2 + p) * 64 + lane) * 4);
            // three partial products, smallest first; the three coordinate planes interleave as independent chains
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = mfma32(bb[c][ks][PB[q]], aa[PA[q]], acc[c]);
        }
        RLAP(1);
        fetch(ft + 1 < ft1 ? ft + 1 : ft);           // unconditional (the last tile is fetched twice): no control flow for the
                                                     // compiler's wait-count bookkeeping to be conservative about
        RLAP(2);
        {   // pose epilogue in registers -> this wave's 32-vertex column block of the workgroup stage
            const float *rec = reinterpret_cast<const float *>(ot + 7 * 256) + j * 16;
            const f32x4 q0 = *(const f32x4 *)&rec[0], q1 = *(const f32x4 *)&rec[4], q2 = *(const f32x4 *)&rec[8];
            float *st = stage + (j * 3) * SS + wave * 32 + 4 * h;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 ox, oy, oz;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sx = acc[0][4 * t + e], sy = acc[1][4 * t + e], sz = acc[2][4 * t + e];
                    ox[e] = q0[0] * sx + q0[1] * sy + q0[2] * sz + q2[1];
                    oy[e] = q0[3] * sx + q1[0] * sy + q1[1] * sz + q2[2];
                    oz[e] = q1[2] * sx + q1[3] * sy + q2[0] * sz + q2[3];
                }
                *(f32x4 *)&st[8 * t] = ox;
                *(f32x4 *)&st[SS + 8 * t] = oy;
                *(f32x4 *)&st[2 * SS + 8 * t] = oz;
            }
        }
        RLAP(3);
        lds_barrier();
        RLAP(4);
        {   // cooperative store: LPR lanes x float4 = one RUN*4-byte run of one (face, coord) row; WPG*64/LPR rows per
            // instruction.  Output row (f0 + r/3, r%3) is row 3*f0 + r of a [3B, pitch] matrix: one pointer, one constant
            // stride.  With pitch = n_vert (dense [B,3,53215]) rows are only 4-byte aligned and every run shares its first and
            // last 128-byte line with the neighbouring workgroups (3.3 TB/s of HBM writes in tools/ubench/store_pattern.hip);
            // with a pitch that is a multiple of 32 floats every run is four whole lines (5.2 TB/s) -- the host API
            // allocates dense outputs with such a pitch unless the caller brings a packed buffer.
            // Nothing of this addressing may stay live across the tile loop (the compiler once hoisted 12 offsets, spilled
            // them, and every scratch reload's s_waitcnt vmcnt(0) waited for all stores in flight): hence the opaque thread id.
            if constexpr (PK) {
                // 24 lanes x float4 = the 96-vertex window of one row, 10 rows per instruction (240 of the 256 threads), 10 instructions.
                // The stores are BUFFER stores on a per-tile resource and unconditional: a lane that must not write (row past the batch,
                // float4 not wholly inside the row, surplus thread) gets an out-of-range offset and the hardware drops it -- straight-line
                // code, so park()'s wait is a counted vmcnt and the stores stay in flight into the next tile (with branches around them
                // every tile waited for the acknowledgement of its stores: 0.180 ms).  What is left -- the vertices in front of the
                // first window (group 0) and a row's ragged last float4 (last group) -- follows park() under a workgroup-uniform branch.
                constexpr int LPRK = PKW / 4, RPIK = WPG * 64 / LPRK, NIT = (96 + RPIK - 1) / RPIK;     // lanes per row, rows per instruction, instructions
                int tid_ = threadIdx.x;
                asm volatile("" : "+v"(tid_));
                const int seg = tid_ % LPRK, rsub = tid_ / LPRK;
                const int nv32 = n_vert & 31;
                const long long r0 = 3ll * f0;
                long long tile_rows = 3ll * B - r0;
                tile_rows = tile_rows < 96 ? tile_rows : 96;
                const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)r0 * n_vert, 0, (int)(tile_rows * n_vert * 4), 0x00027000);
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int rr = RPIK * k + rsub;                                // row of the face tile: face f0 + rr / 3, coordinate rr % 3
                    const int sh = (32 - (int)(((r0 + rr) * nv32) & 31)) & 31;     // the row's line boundaries lie at vertices = sh mod 32
                    const int vq = v_base + sh + 4 * seg;
                    const float *sp = &stage[(rr < 96 ? rr : 95) * SS + sh + 4 * seg];     // (4-byte aligned: sh is any number)
                    const f32x4 vv = {sp[0], sp[1], sp[2], sp[3]};
                    const bool ok = rsub < RPIK && rr < 96 && vq + 3 < n_vert;     // (rows past the batch are past the resource)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv), rs_o, ok ? (unsigned)(rr * n_vert + vq) * 4u : 0x80000000u, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                park(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (tg == 0 || v_base + PKW + 31 + 3 >= n_vert) {                   // (workgroup-uniform) first / last group of the rows
#pragma unroll 1
                    for (int k = 0; k < NIT; ++k) {
                        const int rr = RPIK * k + rsub;
                        const long long r = r0 + rr;
                        if (rsub >= RPIK || rr >= 96 || r >= 3ll * B) continue;
                        const int sh = (32 - (int)((r * nv32) & 31)) & 31;
                        const int vq = v_base + sh + 4 * seg;
                        if (vq < n_vert && vq + 3 >= n_vert) {                      // the ragged last float4 of the row
                            for (int t = 0; t < 4; ++t) if (vq + t < n_vert) out[(size_t)r * n_vert + vq + t] = stage[rr * SS + sh + 4 * seg + t];
                        }
                        if (tg == 0 && 4 * seg < sh) {                              // the vertices in front of the first window of the row
                            for (int t = 0; t < 4; ++t) if (4 * seg + t < sh) out[(size_t)r * n_vert + 4 * seg + t] = stage[rr * SS + 4 * seg + t];
                        }
                    }
                }
            } else {
            constexpr int LPR = RUN / 4, RPI = WPG * 64 / LPR;
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));
            const int seg = tid_ % LPR, rsub = tid_ / LPR;
            const int vq = v_base + 4 * seg;
            const int rows_live = 3 * (B - f0);                       // rows of this face tile that exist (ragged last tile)
            float *o = out + ((size_t)3 * f0 + rsub) * pitch + vq;
            const size_t ostep = (size_t)RPI * pitch;
            const float *sp = &stage[rsub * SS + 4 * seg];
            if (FAST) {
                // 12 unconditional stores: the compiler knows exactly how many stores follow the fetch, so park's wait for
                // the fetched operands is s_waitcnt vmcnt(12) -- these stores stay in flight across the barrier and into the
                // next tile's MFMAs (with any branch around a store the wait degrades to vmcnt(0))
#pragma unroll
                for (int k = 0; k < 96 / RPI; ++k, o += ostep) *(f32x4 *)o = *(const f32x4 *)(sp + k * RPI * SS);      // (non-temporal stores: 0.136 vs 0.137 ms alone, 1.036 vs 1.026 ms in the step -- no)
            } else {
                const bool whole = vq + 3 < n_vert;
#pragma unroll
                for (int k = 0; k < 96 / RPI; ++k, o += ostep) {
                    const f32x4 vv = *(const f32x4 *)(sp + k * RPI * SS);
                    if (k * RPI + rsub < rows_live) {
                        if (whole) *(f32x4 *)o = vv;
                        else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) if (vq + t < n_vert) o[t] = vv[t];
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // park() must stay BELOW the stores (its wait counts them)
            park(buf ^ 1);
            }
        }
        RLAP(5);
        lds_barrier();     // the stage is rewritten by the next face tile
        RLAP(6);
    }
    if (PROF && threadIdx.x == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&prof[i], pt_[i]);
        atomicAdd(&prof[7], 1ull);
    }
}

void launch_reconstruct_f16(const float *param, const float *mean62, const float *std62, const unsigned *basis3, int n_vert,
                           int nvp, const float *roi, int transform, float *out, int pitch, int pad_writable, int B, hipStream_t s, float *rec3f,
                           hipEvent_t *marks /* nullable: [3] recorded before the prologue, after it, after the contraction */) {
    unsigned *rec3 = reinterpret_cast<unsigned *>(rec3f);
    const int n_ftiles = (B + 31) / 32;
    if (marks) (void)hipEventRecord(marks[0], s);
    recon_prep_f16_kernel<<<n_ftiles, 64, 0, s>>>(param, mean62, std62, roi, transform, rec3, B);
    if (marks) (void)hipEventRecord(marks[1], s);
    const int n_tiles = nvp / 32;
    constexpr int WPG = 4;                                    // 8 (1 KiB runs, one workgroup per CU) measured slower
    // the reference's packed rows (pitch == n_vert, dense mesh, a line-aligned tensor): the whole-line schedule PK of the kernel
    // (the straight-line FAST path counts groups of WPG x 32 = 128 vertices, PK counts groups of 224 | 96 -- two counts, kept apart: a
    // caller with pitch == n_vert AND pad_writable used to reach the FAST kernel with PK's group count and leave vertices >= 128 x that unwritten)
    const int fast_groups = (n_tiles + WPG - 1) / WPG;
    const bool fast_ok = pad_writable && pitch >= fast_groups * WPG * 32;
    const bool pk = !fast_ok && pitch == n_vert && n_vert >= 4096 && (reinterpret_cast<uintptr_t>(out) & 127) == 0 && !test_knob("recon_no_pk", 0);
    static const int pk_wpg = (int)test_knob("recon_pk_wpg", 8);       // tiles a PK workgroup computes: 8 (stores 7 lines per row) | 4 (stores 3)
    const int pkw = (pk_wpg == 8 ? 7 : 3) * 32;
    const int n_groups = pk ? (n_vert + pkw - 1) / pkw : fast_groups;       // a workgroup = WPG consecutive vertex tiles (PK: a stride of WPG - 1)
    static const int wg_target = (int)test_knob("recon_wgs", 1664);   // (3072: -0.8 % in the two-stream pipeline at B = 1024)
    static const int prof3 = (int)test_knob("recon_prof", 0);              // profiling only
    // face tiles [lo, hi) in one launch of >= wg_target workgroups: vertex groups x splits of the face-tile range
    auto run = [&](int lo, int hi, bool fast) {
        const int nft = hi - lo;
        if (nft <= 0) return;
        // (eight-wave PK workgroups, one per CU: half the target -- 832: 0.153 ms, 1664: 0.167, 512: 0.156 at B = 1024)
        const int target = (pk && pk_wpg == 8 && !test_knob_set("recon_wgs")) ? wg_target / 2 : wg_target;
        int n_split = (target + n_groups - 1) / n_groups;
        n_split = n_split < 1 ? 1 : n_split;
        n_split = n_split > nft ? nft : n_split;
        const int per = (nft + n_split - 1) / n_split;
        n_split = (nft + per - 1) / per;
        const int n_units = n_groups * n_split;
        const int grid = ((n_units + 7) / 8) * 8;
        if (prof3 && fast && n_vert > 1000) {
            unsigned long long *d = nullptr, hst[8];
            (void)hipMalloc((void **)&d, sizeof(hst));
            (void)hipMemsetAsync(d, 0, sizeof(hst), s);
            recon_f16_kernel<WPG, true, true><<<grid, WPG * 64, 0, s>>>(rec3, basis3, out, B, n_vert, pitch, n_tiles, n_split, per, lo, hi, n_units, d);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost);
            (void)hipFree(d);
            const char *nm[7] = {"basis+first tile", "mfma", "fetch issue", "epilogue", "barrier1", "store+park", "barrier2"};
            for (int i = 0; i < 7; ++i) fprintf(stderr, "recon prof %-18s %10.0f ticks/wg\n", nm[i], (double)hst[i] / (double)hst[7]);
            fprintf(stderr, "recon prof workgroups %llu\n", hst[7]);
        } else if (fast)
            recon_f16_kernel<WPG, true><<<grid, WPG * 64, 0, s>>>(rec3, basis3, out, B, n_vert, pitch, n_tiles, n_split, per, lo, hi, n_units);
        else if (pk && pk_wpg == 8)
            recon_f16_kernel<8, false, false, true><<<grid, 8 * 64, 0, s>>>(rec3, basis3, out, B, n_vert, pitch, n_tiles, n_split, per, lo, hi, n_units);
        else if (pk)
            recon_f16_kernel<WPG, false, false, true><<<grid, WPG * 64, 0, s>>>(rec3, basis3, out, B, n_vert, pitch, n_tiles, n_split, per, lo, hi, n_units);
        else
            recon_f16_kernel<WPG, false><<<grid, WPG * 64, 0, s>>>(rec3, basis3, out, B, n_vert, pitch, n_tiles, n_split, per, lo, hi, n_units);
    };
    if (fast_ok) {                        // pitched output with room for whole 128-vertex runs: whole face tiles on the straight-line
                                          // store path (columns [n_vert, pitch) receive padding values), the ragged last one guarded
        run(0, B / 32, true);
        run(B / 32, n_ftiles, false);
    } else
        run(0, n_ftiles, false);
    if (marks) (void)hipEventRecord(marks[2], s);
}

// -------------------------------------------------------------------------------------
// predict_pose: one lane per face.  fp32 for the de-whitening / normalisation / cross product
// (numpy float32 in the reference), double for asin/atan2/cos (python math on the float32 values).
// -------------------------------------------------------------------------------------
// the pose arithmetic of one face (shared by pose_kernel and lmk_pose_kernel so that both produce the same bits): p = the face's 12
// de-whitened pose parameters
__device__ __forceinline__ void pose_of_face(const float (&p)[12], const float *__restrict__ roi5 /*nullable*/, double *__restrict__ angles3 /*nullable*/,
                                             float *__restrict__ t3d3, float *__restrict__ pmat12 /*nullable*/) {
    const float n1 = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);      // P2sRt (:33-43)
    const float n2 = sqrtf(p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    const float r1[3] = {p[0] / n1, p[1] / n1, p[2] / n1};
    const float r2[3] = {p[4] / n2, p[5] / n2, p[6] / n2};
    const float r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    // matrix2angle_corr (:45-62): R = [r1; r2; r3]
    const double PI = 3.14159265358979323846;
    double x, y, z;
    if (r3[0] != 1.0f && r3[0] != -1.0f) {
        x = asin((double)r3[0]);
        const double cx = cos(x);
        y = atan2((double)r2[2] / cx, (double)r3[2] / cx);
        z = atan2((double)r1[1] / cx, (double)r1[0] / cx);
    } else {                                                                // gimbal lock
        z = 0.0;
        if (r3[0] == -1.0f) { x = PI / 2; y = z + atan2((double)r1[1], (double)r1[2]); }
        else                { x = -PI / 2; y = -z + atan2(-(double)r1[1], -(double)r1[2]); }
    }
    if (pmat12) {           // parse_pose's P = [R | t3d] "without scale" (:90), built BEFORE predict_pose's ROI affine touches t3d
        float *m = pmat12;
        m[0] = r1[0]; m[1] = r1[1]; m[2] = r1[2]; m[3] = p[3];
        m[4] = r2[0]; m[5] = r2[1]; m[6] = r2[2]; m[7] = p[7];
        m[8] = r3[0]; m[9] = r3[1]; m[10] = r3[2]; m[11] = p[11];
    }
    if (!angles3) return;
    angles3[0] = x * 180.0 / PI;
    angles3[1] = y * 180.0 / PI;
    angles3[2] = z * 180.0 / PI;
    float tx = p[3], ty = p[7], tz = p[11];
    if (roi5) {                                                             // predict_pose (:149-154)
        const float sx = roi5[0], sy = roi5[1], ex = roi5[2], ey = roi5[3];
        tx = tx * ((ex - sx) / 120.0f) + sx;
        ty = ty * ((ey - sy) / 120.0f) + sy;
    }
    t3d3[0] = tx;
    t3d3[1] = ty;
    t3d3[2] = tz;
}

__global__ __launch_bounds__(64) void pose_kernel(const float *__restrict__ param, const float *__restrict__ mean,
                                                  const float *__restrict__ stdv, const float *__restrict__ roi,
                                                  double *__restrict__ angles, float *__restrict__ t3d,
                                                  float *__restrict__ pmat /*nullable [B,3,4]*/, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float p[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = param[(size_t)b * kParam + i] * stdv[i] + mean[i];
    pose_of_face(p, roi ? roi + (size_t)b * 5 : nullptr, angles ? angles + (size_t)b * 3 : nullptr, t3d + (size_t)b * 3, pmat ? pmat + (size_t)b * 12 : nullptr);
}

void launch_pose(const float *param, const float *mean62, const float *std62, const float *roi, double *angles,
                 float *t3d, float *pmat, int B, hipStream_t s) {
    pose_kernel<<<(B + 63) / 64, 64, 0, s>>>(param, mean62, std62, roi, angles, t3d, pmat, B);
}

// -------------------------------------------------------------------------------------
// Landmarks + pose of a batch in ONE launch (round 5): what get_all_outputs computes per face besides the mesh
// (synergy3DMM.py:194-201: predict_sparseVert + predict_pose).  As separate calls it is three dependent launches of 4-8 us each
// (prologue, contraction, pose) for 10.6 k multiply-adds per face -- 19 of the 300 us of a BASELINE configs[1] step.  One workgroup per
// face: de-whitening, the [204 x 52] landmark contraction as plain fp32 fused multiply-adds over k = 0 .. 51 in order (the exact-fp32
// landmark tiles of the handle, layout of recon_kernel above: no operand split, nothing to range-check), the pose / flip / ROI affine
// of recon_prep_kernel, and pose_of_face on lane 0.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lmk_pose_kernel(const float *__restrict__ param, const float *__restrict__ mean, const float *__restrict__ stdv,
                                                       const float *__restrict__ basis /*fp32 landmark tiles*/, int n_lmk, int nlp,
                                                       const float *__restrict__ roi, int transform, float *__restrict__ lmk /*[B,3,n_lmk]*/,
                                                       double *__restrict__ angles, float *__restrict__ t3d, int B) {
    const int b = blockIdx.x, l = threadIdx.x;
    __shared__ float p[64];
    if (l < kParam) p[l] = param[(size_t)b * kParam + l] * stdv[l] + mean[l];
    __syncthreads();                                 // the only barrier: a landmark's thread computes all three coordinates and its own copy of
                                                     // the affine map, so the pose -- double-precision asin / atan2 on ONE lane, the longest chain
                                                     // of the kernel -- runs BESIDE the contraction on the last thread instead of in front of a barrier
    if (l == 255 && angles) {
        float pp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) pp[i] = p[i];
        pose_of_face(pp, roi ? roi + (size_t)b * 5 : nullptr, angles + (size_t)b * 3, t3d + (size_t)b * 3, nullptr);
    }
    // out = Mx * S + T of recon_prep_kernel (same expressions), per thread
    float aff[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sc = 1.0f, of = 0.0f;
        if (roi) {
            const float sx = roi[b * 5 + 0], sy = roi[b * 5 + 1], ex = roi[b * 5 + 2], ey = roi[b * 5 + 3];
            const float scx = (ex - sx) / 120.0f, scy = (ey - sy) / 120.0f;
            if (c == 0) { sc = scx; of = sx; }
            else if (c == 1) { sc = scy; of = sy; }
            else { sc = (scx + scy) * 0.5f; }
        }
        float m0 = p[4 * c + 0], m1 = p[4 * c + 1], m2 = p[4 * c + 2], t = p[4 * c + 3];
        if (transform && c == 1) { m0 = -m0; m1 = -m1; m2 = -m2; t = (float)(kImg + 1) - t; }
        aff[3 * c + 0] = m0 * sc;
        aff[3 * c + 1] = m1 * sc;
        aff[3 * c + 2] = m2 * sc;
        aff[9 + c] = t * sc + of;
    }
    // S[c] = sum_k basis[v][c][k] alpha[k]: element k = 8 t + 4 h + s of (tile T, coord c, column j) sits at
    // T * 3 * 52 * 32 + c * 52 * 32 + t * 256 + (32 h + j) * 4 + s; the tail k = 48 + 2 h + s at ... + 6 * 256 + (32 h + j) * 2 + s (k = 50: the mean
    // shape, coefficient 1; k = 51: zero pad)
    for (int v = l; v < n_lmk; v += 256) {
        const int T = v >> 5, j = v & 31;
        float S[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *bc = basis + ((size_t)T * 3 + c) * (kBasisK * 32);
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 w = *(const f32x4 *)(bc + t * 256 + (32 * h + j) * 4);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_fmaf(w[s], p[12 + 8 * t + 4 * h + s], acc);
                }
            const f32x2 w0 = *(const f32x2 *)(bc + 6 * 256 + j * 2), w1 = *(const f32x2 *)(bc + 6 * 256 + (32 + j) * 2);
            acc = __builtin_fmaf(w0[0], p[12 + 48], acc);
            acc = __builtin_fmaf(w0[1], p[12 + 49], acc);
            acc += w1[0];                            // the mean shape
            S[c] = acc;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            lmk[((size_t)b * 3 + c) * n_lmk + v] = aff[3 * c + 0] * S[0] + aff[3 * c + 1] * S[1] + aff[3 * c + 2] * S[2] + aff[9 + c];
    }
}

void launch_lmk_pose(const float *param, const float *mean62, const float *std62, const float *basis_lmk, int n_lmk, int nlp, const float *roi,
                     int transform, float *lmk, double *angles, float *t3d, int B, hipStream_t s) {
    lmk_pose_kernel<<<B, 256, 0, s>>>(param, mean62, std62, basis_lmk, n_lmk, nlp, roi, transform, lmk, angles, t3d, B);
}

}  // namespace syn
