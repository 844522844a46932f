// On-device pre-processing of get_all_outputs (reference synergy3DMM.py:177-192): square crop of each detection
// with zero padding outside the frame (crop_img, utils/inference.py:95-125) followed by the 8-tap Lanczos resize to
// 120x120 (cv2.resize INTER_LANCZOS4, synergy3DMM.py:188) -- OpenCV's 8-bit fixed-point path restated: 11-bit
// coefficients per destination column / row (computed on the host exactly as synergynet_amd/inference.py does and
// passed in as tables), replicated borders at the CROP edges, one rounding shift by 22 at the end.
// Output: uint8 [B,120,120,3] HWC crops, the input format of syn_backbone_forward_u8.  Integer arithmetic only:
// bit-identical to the host restatement.  One thread per output pixel (3 channels), 64 source taps each.
#include "syn_internal.h"

namespace syn {

__global__ __launch_bounds__(256) void crop_resize_kernel(const uint8_t *__restrict__ frame, int H, int W,
                                                          const int *__restrict__ box /*[B,4] sx,sy,ex,ey (rounded)*/,
                                                          const int *__restrict__ xofs /*[B,120] first tap, crop coords*/,
                                                          const short *__restrict__ xcoef /*[B,120,8]*/,
                                                          const int *__restrict__ yofs, const short *__restrict__ ycoef,
                                                          uint8_t *__restrict__ out, int B,
                                                          const long long *__restrict__ foff /* nullable: faces of SEVERAL frames (syn_crop_resize_frames) */,
                                                          const int *__restrict__ fdim, const int *__restrict__ fidx) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * kImg * kImg) return;
    const int b = idx / (kImg * kImg), r = idx % (kImg * kImg);
    if (foff) {                                            // (kernel-uniform) frame of face b: byte offset from `frame`, its height and width
        const int f = fidx[b];
        frame += foff[f];
        H = fdim[2 * f];
        W = fdim[2 * f + 1];
    }
    const int oy = r / kImg, ox = r % kImg;
    const int sx = box[4 * b + 0], sy = box[4 * b + 1], ex = box[4 * b + 2], ey = box[4 * b + 3];
    const int cw = ex - sx, ch = ey - sy;                 // crop size
    const int x0 = xofs[b * kImg + ox], y0 = yofs[b * kImg + oy];
    const short *cx = xcoef + ((size_t)b * kImg + ox) * 8, *cy = ycoef + ((size_t)b * kImg + oy) * 8;
    long long acc[3] = {0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 8; ++ky) {
        int yc = y0 + ky;
        yc = yc < 0 ? 0 : (yc > ch - 1 ? ch - 1 : yc);     // replicate at the crop border
        const int fy = sy + yc;                            // frame row; outside the frame the crop is zero
        long long hor[3] = {0, 0, 0};
        if (fy >= 0 && fy < H) {
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) {
                int xc = x0 + kx;
                xc = xc < 0 ? 0 : (xc > cw - 1 ? cw - 1 : xc);
                const int fx = sx + xc;
                if (fx >= 0 && fx < W) {
                    const uint8_t *p = frame + ((size_t)fy * W + fx) * 3;
                    const int c = cx[kx];
                    hor[0] += c * (int)p[0]; hor[1] += c * (int)p[1]; hor[2] += c * (int)p[2];
                }
            }
        }
        const int c = cy[ky];
        acc[0] += c * hor[0]; acc[1] += c * hor[1]; acc[2] += c * hor[2];
    }
    uint8_t *o = out + (size_t)idx * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        long long v = (acc[k] + (1ll << 21)) >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o[k] = (uint8_t)v;
    }
}

void launch_crop_resize(const uint8_t *frame, int H, int W, const int *box, const int *xofs, const short *xcoef,
                        const int *yofs, const short *ycoef, uint8_t *out, int B, hipStream_t s, const long long *foff, const int *fdim,
                        const int *fidx) {
    const int total = B * kImg * kImg;
    crop_resize_kernel<<<(total + 255) / 256, 256, 0, s>>>(frame, H, W, box, xofs, xcoef, yofs, ycoef, out, B, foff, fdim, fidx);
}

}  // namespace syn
