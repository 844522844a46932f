// AFLW2000-3D landmark error on the device (the data-gated evaluator): calc_nme of the reference's
// benchmark_aflw2000.py:107-139 -- per sample: map the fitted 68 landmarks from the 120x120 crop back to the image with the crop
// box, mean Euclidean distance to the ground truth, normalised by sqrt(area of the ground-truth bounding box).
// The reference mixes float32 arrays with python-float scalars (numpy promotes to float64 for sqrt / mean of the distances of a
// float32 array? no: np.sqrt / np.sum / np.mean keep float32; `sqrt((maxx-minx)*(maxy-miny))` is python math on float32 scalars
// -> float64); the kernel follows that: float32 for the per-point distances and their mean (pairwise summation as numpy does
// for 68 contiguous elements), float64 for the box diagonal, result rounded to float32.
#pragma clang fp contract(off)

#include "syn_internal.h"

namespace syn {

// one lane per sample (68 points: a few hundred flops; 2000 samples in AFLW2000-3D)
__global__ __launch_bounds__(64) void nme_kernel(const float *__restrict__ fit /*[N,2,68] crop coords*/, const float *__restrict__ gt /*[N,3,68]*/,
                                                 const float *__restrict__ roi /*[N,4]*/, float *__restrict__ nme, int N) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= N) return;
    const float sx = roi[4 * i], sy = roi[4 * i + 1], ex = roi[4 * i + 2], ey = roi[4 * i + 3];
    const float scale_x = (ex - sx) / 120, scale_y = (ey - sy) / 120;        // float32 / python int -> float32
    const float *f = fit + (size_t)i * 2 * 68, *g = gt + (size_t)i * 3 * 68;
    float minx = g[0], maxx = g[0], miny = g[68], maxy = g[68];
    float d[68];
    for (int k = 0; k < 68; ++k) {
        minx = fminf(minx, g[k]); maxx = fmaxf(maxx, g[k]); miny = fminf(miny, g[68 + k]); maxy = fmaxf(maxy, g[68 + k]);
        const float dx = (f[k] * scale_x + sx) - g[k], dy = (f[68 + k] * scale_y + sy) - g[68 + k];
        d[k] = sqrtf(dx * dx + dy * dy);                                      // np.sqrt(np.sum(np.power(dis, 2), 0)): 2 terms
    }
    // np.mean over 68 contiguous float32: pairwise summation -- blocks of 8 accumulators for the first 64, then the tail
    float r[8];
    for (int q = 0; q < 8; ++q) r[q] = d[q];
    for (int k = 8; k < 64; k += 8)
        for (int q = 0; q < 8; ++q) r[q] += d[k + q];
    float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (int k = 64; k < 68; ++k) s += d[k];
    const float mean = s / 68.0f;
    const double llength = sqrt((double)((maxx - minx) * (maxy - miny)));   // float32 product, python math.sqrt
    nme[i] = (float)((double)mean / llength);
}

void launch_nme(const float *fit, const float *gt, const float *roi, float *nme, int N, hipStream_t s) {
    nme_kernel<<<(N + 63) / 64, 64, 0, s>>>(fit, gt, roi, nme, N);
}

}  // namespace syn
