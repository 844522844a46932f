// gfx950 kernels for the FaceBoxes face detector: the step that produces the boxes the hot path crops.
//   FaceBoxesNet.forward     FaceBoxes/models/faceboxes.py:116-150 (CRelu :48-61, Inception :21-46, BasicConv2d :8-18, heads :104-114)
//   PriorBox.forward         FaceBoxes/utils/prior_box.py:22-48
//   decode                   FaceBoxes/utils/box_utils.py:177-196
//   FaceBoxes.__call__       FaceBoxes/FaceBoxes.py:60-143 (frame down-scaling, mean subtraction, thresholds, top-k, NMS)
//   cpu_nms                  FaceBoxes/utils/nms/cpu_nms.pyx:17-68
// The detector runs once per frame (~1.3 GFLOP at 720x1080), not once per face, so its convolutions are plain fp32 VALU
// direct convolutions over NHWC activations -- no MFMA tiling: the whole network is ~30 short launches.  Channel
// concatenations (CReLU, Inception, the multibox heads) are free: every kernel writes into a channel window of a wider
// NHWC buffer.  Everything after the network (priors, decoding, threshold, sort, NMS) stays on the device as well.
#include "syn_internal.h"

namespace syn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- uint8 BGR frame -> fp32 NHWC minus the channel means (FaceBoxes.py:83,90), optionally through cv2.resize's bilinear ----
// scale == 1: plain conversion.  Otherwise OpenCV's fixed-point INTER_LINEAR for uint8 (11-bit coefficients, see the
// oracle's resize_linear_u8; cv2 itself is absent, so that path is restated, not pinned).
__global__ __launch_bounds__(256) void det_preproc_kernel(const unsigned char *__restrict__ frame, int H, int W, float *__restrict__ out,
                                                          int Ho, int Wo) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int oy = p / Wo, ox = p % Wo;
    const float mean[3] = {104.f, 117.f, 123.f};
    if (Ho == H && Wo == W) {
        for (int c = 0; c < 3; ++c) out[(size_t)p * 3 + c] = (float)frame[(size_t)p * 3 + c] - mean[c];
        return;
    }
    auto taps = [](int d, int n_dst, int n_src, int &s0, int &s1, int &c0, int &c1) {
        const double scale = (double)n_src / n_dst;
        const float fd = (float)(((double)d + 0.5) * scale - 0.5);      // fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx
        long s = (long)floorf(fd);
        float f = fd - (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
        c1 = (int)rintf(f * 2048.0f);
        c0 = (int)rintf((1.0f - f) * 2048.0f);
        s0 = (int)s; s1 = min((int)s + 1, n_src - 1);
    };
    int x0, x1, cx0, cx1, y0, y1, cy0, cy1;
    taps(ox, Wo, W, x0, x1, cx0, cx1);
    taps(oy, Ho, H, y0, y1, cy0, cy1);
    for (int c = 0; c < 3; ++c) {
        const int a = frame[((size_t)y0 * W + x0) * 3 + c] * cx0 + frame[((size_t)y0 * W + x1) * 3 + c] * cx1;
        const int b = frame[((size_t)y1 * W + x0) * 3 + c] * cx0 + frame[((size_t)y1 * W + x1) * 3 + c] * cx1;
        int v = (((cy0 * (a >> 4)) >> 16) + ((cy1 * (b >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        out[(size_t)p * 3 + c] = (float)v - mean[c];
    }
}

// ---- direct convolution, NHWC, fp32 VALU.  thread = (output pixel, 4 consecutive output channels) ----
// in:  [Hi][Wi][cs_in] channels [ci0, ci0+Cin);  W: [K][K][Cin][Cp] (Cp = Cout rounded up to 4, BN scale folded in), shift [Cp]
// out: [Ho][Wo][cs_out] channels [co0, co0+Cout) (+ [co0+Cout, co0+2*Cout) for CReLU);  act: 0 none, 1 ReLU, 2 CReLU
__global__ __launch_bounds__(256) void det_conv_kernel(const float *__restrict__ in, const float *__restrict__ Wt, const float *__restrict__ shift,
                                                       float *__restrict__ out, int Hi, int Wi, int cs_in, int ci0, int Cin, int Ho, int Wo,
                                                       int cs_out, int co0, int Cout, int Cp, int K, int stride, int pad, int act) {
    const int ncog = Cp >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)Ho * Wo * ncog) return;
    const int cog = (int)(t % ncog), p = (int)(t / ncog);
    const int oy = p / Wo, ox = p % Wo;
    f32x4 acc = *(const f32x4 *)&shift[4 * cog];
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * stride - pad + ky;
        if ((unsigned)iy >= (unsigned)Hi) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ox * stride - pad + kx;
            if ((unsigned)ix >= (unsigned)Wi) continue;
            const float *xp = in + ((size_t)iy * Wi + ix) * cs_in + ci0;
            const float *wp = Wt + ((size_t)(ky * K + kx) * Cin) * Cp + 4 * cog;
            if ((Cin & 3) == 0 && ((cs_in | ci0) & 3) == 0) {
#pragma unroll 4                                  // 20 independent loads in flight per trip instead of 5 (the kernel is latency bound)
                for (int ci = 0; ci < Cin; ci += 4) {
                    const f32x4 x = *(const f32x4 *)(xp + ci);
                    acc += x[0] * *(const f32x4 *)(wp + (size_t)ci * Cp);
                    acc += x[1] * *(const f32x4 *)(wp + (size_t)(ci + 1) * Cp);
                    acc += x[2] * *(const f32x4 *)(wp + (size_t)(ci + 2) * Cp);
                    acc += x[3] * *(const f32x4 *)(wp + (size_t)(ci + 3) * Cp);
                }
            } else {
                for (int ci = 0; ci < Cin; ++ci) acc += xp[ci] * *(const f32x4 *)(wp + (size_t)ci * Cp);
            }
        }
    }
    float *o = out + (size_t)p * cs_out + co0;
    for (int q = 0; q < 4; ++q) {
        const int c = 4 * cog + q;
        if (c >= Cout) break;
        const float v = acc[q];
        if (act == 2) { o[c] = fmaxf(v, 0.f); o[c + Cout] = fmaxf(-v, 0.f); }
        else o[c] = act == 1 ? fmaxf(v, 0.f) : v;
    }
}

// ---- 3x3 pooling, NHWC, C % 4 == 0: max (stride 2, pad 1, -inf padding) or average (stride 1, pad 1, divisor 9) ----
__global__ __launch_bounds__(256) void det_pool_kernel(const float *__restrict__ in, float *__restrict__ out, int Hi, int Wi, int C, int Ho,
                                                       int Wo, int stride, int is_max) {
    const int c4n = C >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)Ho * Wo * c4n) return;
    const int c4 = (int)(t % c4n), p = (int)(t / c4n);
    const int oy = p / Wo, ox = p % Wo;
    f32x4 a = is_max ? (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY} : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - 1 + ky;
        if ((unsigned)iy >= (unsigned)Hi) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride - 1 + kx;
            if ((unsigned)ix >= (unsigned)Wi) continue;
            const f32x4 v = *(const f32x4 *)&in[((size_t)iy * Wi + ix) * C + 4 * c4];
            if (is_max) for (int q = 0; q < 4; ++q) a[q] = fmaxf(a[q], v[q]);
            else a += v;
        }
    }
    if (!is_max) a = a / 9.0f;                     // F.avg_pool2d default count_include_pad=True (faceboxes.py:34)
    *(f32x4 *)&out[(size_t)p * C + 4 * c4] = a;
}

// ---- priors + softmax + decode + score threshold (prior_box.py:22-48, box_utils.py:189-196, FaceBoxes.py:98-112) ----
// thread = prior.  Candidates (score > thr) are appended to cand[] = {x1, y1, x2, y2, score, prior index} via an atomic counter.
__global__ __launch_bounds__(256) void det_decode_kernel(const float *__restrict__ loc, const float *__restrict__ conf, int P, int Hn, int Wn,
                                                         int H4, int W4, int H5, int W5, int H6, int W6, float scale, float thr,
                                                         float *__restrict__ cand, int *__restrict__ n_cand, int max_cand,
                                                         float *__restrict__ boxes_out /*nullable [P,4]*/, float *__restrict__ scores_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    // which source / cell / anchor (anchors per cell: 21 = 16 (32 px, 4x4 dense) + 4 (64 px, 2x2) + 1 (128 px); 1; 1)
    double cx, cy, sk;
    const int n0 = H4 * W4 * 21, n1 = H5 * W5;
    if (i < n0) {
        const int cell = i / 21, a = i % 21, ci = cell / W4, cj = cell % W4;
        if (a < 16) { sk = 32; cx = (cj + 0.25 * (a & 3)) * 32.0 / Wn; cy = (ci + 0.25 * (a >> 2)) * 32.0 / Hn; }
        else if (a < 20) { sk = 64; cx = (cj + 0.5 * ((a - 16) & 1)) * 32.0 / Wn; cy = (ci + 0.5 * ((a - 16) >> 1)) * 32.0 / Hn; }
        else { sk = 128; cx = (cj + 0.5) * 32.0 / Wn; cy = (ci + 0.5) * 32.0 / Hn; }
    } else if (i < n0 + n1) {
        const int cell = i - n0; sk = 256; cx = (cell % W5 + 0.5) * 64.0 / Wn; cy = (cell / W5 + 0.5) * 64.0 / Hn;
    } else {
        const int cell = i - n0 - n1; sk = 512; cx = (cell % W6 + 0.5) * 128.0 / Wn; cy = (cell / W6 + 0.5) * 128.0 / Hn;
    }
    const float pcx = (float)cx, pcy = (float)cy, pw = (float)(sk / Wn), ph = (float)(sk / Hn);
    const float l0 = loc[4 * i], l1 = loc[4 * i + 1], l2 = loc[4 * i + 2], l3 = loc[4 * i + 3];
    float bx = pcx + (l0 * 0.1f) * pw, by = pcy + (l1 * 0.1f) * ph;
    float bw = pw * expf(l2 * 0.2f), bh = ph * expf(l3 * 0.2f);
    bx -= bw / 2; by -= bh / 2;
    bw += bx; bh += by;
    // boxes * scale_bbox / scale / resize   (FaceBoxes.py:101-104): scale_bbox = (Wn, Hn, Wn, Hn)
    const float x1 = (bx * (float)Wn) / scale, y1 = (by * (float)Hn) / scale, x2 = (bw * (float)Wn) / scale, y2 = (bh * (float)Hn) / scale;
    const float c0 = conf[2 * i], c1 = conf[2 * i + 1], m = fmaxf(c0, c1);
    const float e0 = expf(c0 - m), e1 = expf(c1 - m);
    const float score = e1 / (e0 + e1);
    if (boxes_out) { boxes_out[4 * i] = x1; boxes_out[4 * i + 1] = y1; boxes_out[4 * i + 2] = x2; boxes_out[4 * i + 3] = y2; scores_out[i] = score; }
    if (score > thr) {
        const int k = atomicAdd(n_cand, 1);
        if (k < max_cand) {
            float *c = cand + (size_t)k * 6;
            c[0] = x1; c[1] = y1; c[2] = x2; c[3] = y2; c[4] = score; c[5] = __builtin_bit_cast(float, i);
        }
    }
}

// ---- sort by score (descending; equal scores: lower prior index first) + greedy NMS, one workgroup (FaceBoxes.py:114-127) ----
// keys: [score bits : 32 | ~prior index : 32] sorted descending with an in-LDS bitonic network over kSortN slots.  The reference
// sorts ALL candidates and keeps the first top_k (FaceBoxes.py:115); when more candidates than the network holds pass the
// threshold, the top_k largest keys are first selected exactly (8-bit radix select on the unique 64-bit keys, most significant
// digit first) and only those enter the network -- the same set in the same order as sorting everything.
constexpr int kSortN = 8192;
__global__ __launch_bounds__(1024) void det_nms_kernel(const float *__restrict__ cand, const int *__restrict__ n_cand, int max_cand, int top_k,
                                                       float nms_thr, int keep_top_k, float *__restrict__ dets /*[keep_top_k,5]*/,
                                                       int *__restrict__ n_out) {
    __shared__ unsigned long long key[kSortN];
    __shared__ unsigned slot[kSortN];                  // candidate slot of each key
    __shared__ unsigned char dead[kSortN];
    __shared__ int n_keep, n_sel;
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel_prefix;
    __shared__ int sel_want;
    int n_all = *n_cand;
    n_all = n_all < max_cand ? n_all : max_cand;
    const int tid = threadIdx.x;
    auto key_of = [&](int i) {
        const unsigned sb = __builtin_bit_cast(unsigned, cand[(size_t)i * 6 + 4]);       // scores are positive: bit order = value order
        const unsigned idx = __builtin_bit_cast(unsigned, cand[(size_t)i * 6 + 5]);
        return ((unsigned long long)sb << 32) | (unsigned long long)(~idx);
    };
    int n = n_all;
    if (n_all > kSortN) {
        // threshold key T = the K-th largest key, K = min(top_k, kSortN) (host guarantees top_k <= kSortN)
        const int K = top_k < kSortN ? top_k : kSortN;
        if (tid == 0) { sel_prefix = 0ull; sel_want = K; }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned long long pre = sel_prefix;
            const unsigned long long mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
            for (int i = tid; i < n_all; i += 1024) {
                const unsigned long long k = key_of(i);
                if ((k & mask) == pre) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int want = sel_want, d = 255;
                for (; d > 0; --d) {                   // digits from the top: skip whole bins while they fit
                    if ((int)hist[d] >= want) break;
                    want -= (int)hist[d];
                }
                sel_prefix = pre | ((unsigned long long)d << shift);
                sel_want = want;
            }
            __syncthreads();
        }
        const unsigned long long T = sel_prefix;       // exactly K keys are >= T (keys are unique)
        if (tid == 0) n_sel = 0;
        for (int i = tid; i < kSortN; i += 1024) { key[i] = 0ull; slot[i] = 0; dead[i] = 0; }
        __syncthreads();
        for (int i = tid; i < n_all; i += 1024) {
            const unsigned long long k = key_of(i);
            if (k >= T) {
                const int q = atomicAdd(&n_sel, 1);
                if (q < kSortN) { key[q] = k; slot[q] = (unsigned)i; }
            }
        }
        __syncthreads();
        n = n_sel < kSortN ? n_sel : kSortN;
    }
    int sn = 1024;                                     // sort network size: next power of two >= n (>= one element per thread)
    while (sn < n) sn <<= 1;
    if (n_all <= kSortN) {
        for (int i = tid; i < sn; i += 1024) {
            key[i] = i < n ? key_of(i) : 0ull;
            slot[i] = (unsigned)i; dead[i] = 0;
        }
    }
    if (tid == 0) n_keep = 0;
    __syncthreads();
    for (int k2 = 2; k2 <= sn; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < sn; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k2) == 0;
                    const unsigned long long a = key[i], b = key[l];
                    if (desc ? (a < b) : (a > b)) {
                        key[i] = b; key[l] = a;
                        const unsigned s = slot[i]; slot[i] = slot[l]; slot[l] = s;
                    }
                }
            }
            __syncthreads();
        }
    n = n < top_k ? n : top_k;                         // keep top-K before NMS (FaceBoxes.py:115)
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;                         // uniform: LDS value read by every thread after the barrier below
        const float *ci = cand + (size_t)slot[i] * 6;
        const float ix1 = ci[0], iy1 = ci[1], ix2 = ci[2], iy2 = ci[3];
        const float iarea = (ix2 - ix1 + 1) * (iy2 - iy1 + 1);
        int kk = 0;
        if (tid == 0) {
            kk = n_keep;
            if (kk < keep_top_k) { for (int q = 0; q < 5; ++q) dets[(size_t)kk * 5 + q] = ci[q]; }
            n_keep = kk + 1;
        }
        for (int j = i + 1 + tid; j < n; j += 1024) {
            if (dead[j]) continue;
            const float *cj = cand + (size_t)slot[j] * 6;
            const float xx1 = fmaxf(ix1, cj[0]), yy1 = fmaxf(iy1, cj[1]), xx2 = fminf(ix2, cj[2]), yy2 = fminf(iy2, cj[3]);
            const float w = fmaxf(0.0f, xx2 - xx1 + 1), h = fmaxf(0.0f, yy2 - yy1 + 1);
            const float inter = w * h;
            const float jarea = (cj[2] - cj[0] + 1) * (cj[3] - cj[1] + 1);
            const float ovr = inter / (iarea + jarea - inter);
            if (ovr >= nms_thr) dead[j] = 1;           // cpu_nms.pyx:64-65
        }
        __syncthreads();
    }
    if (tid == 0) *n_out = n_keep < keep_top_k ? n_keep : keep_top_k;
}

void launch_det_preproc(const unsigned char *frame, int H, int W, float *out, int Ho, int Wo, hipStream_t s) {
    det_preproc_kernel<<<(Ho * Wo + 255) / 256, 256, 0, s>>>(frame, H, W, out, Ho, Wo);
}
void launch_det_conv(const float *in, const float *Wt, const float *shift, float *out, int Hi, int Wi, int cs_in, int ci0, int Cin, int Ho,
                     int Wo, int cs_out, int co0, int Cout, int K, int stride, int pad, int act, hipStream_t s) {
    const int Cp = (Cout + 3) & ~3;
    const long total = (long)Ho * Wo * (Cp >> 2);
    det_conv_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(in, Wt, shift, out, Hi, Wi, cs_in, ci0, Cin, Ho, Wo, cs_out, co0, Cout, Cp, K,
                                                                    stride, pad, act);
}
void launch_det_pool(const float *in, float *out, int Hi, int Wi, int C, int Ho, int Wo, int stride, int is_max, hipStream_t s) {
    const long total = (long)Ho * Wo * (C >> 2);
    det_pool_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(in, out, Hi, Wi, C, Ho, Wo, stride, is_max);
}
void launch_det_decode(const float *loc, const float *conf, int P, int Hn, int Wn, int H4, int W4, int H5, int W5, int H6, int W6,
                       float scale, float thr, float *cand, int *n_cand, int max_cand, float *boxes_out, float *scores_out,
                       hipStream_t s) {
    (void)hipMemsetAsync(n_cand, 0, sizeof(int), s);
    det_decode_kernel<<<(P + 255) / 256, 256, 0, s>>>(loc, conf, P, Hn, Wn, H4, W4, H5, W5, H6, W6, scale, thr, cand, n_cand, max_cand,
                                                      boxes_out, scores_out);
}
void launch_det_nms(const float *cand, const int *n_cand, int max_cand, int top_k, float nms_thr, int keep_top_k, float *dets, int *n_out,
                    hipStream_t s) {
    det_nms_kernel<<<1, 1024, 0, s>>>(cand, n_cand, max_cand, top_k, nms_thr, keep_top_k, dets, n_out);
}
int det_sort_capacity() { return kSortN; }

}  // namespace syn
