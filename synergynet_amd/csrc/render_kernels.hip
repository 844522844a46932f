// gfx950 kernels for the mesh consumers that follow the reconstruction in the reference's demo path:
//   Sim3DR.get_normal   (Sim3DR/lib/rasterize_kernel.cpp:158-215)  per-vertex normals
//   RenderPipeline      (Sim3DR/lighting.py:37-71)                   ambient + diffuse + specular vertex colours
//   Sim3DR.rasterize    (Sim3DR/lib/rasterize_kernel.cpp:219-287)  z-buffer rasteriser, barycentric colours
//   cv2.addWeighted     (utils/render.py:45)                         alpha overlay
// The reference walks the 105 840 triangles one at a time on the host.  Here every stage is data parallel and still
// reproduces the sequential result BIT FOR BIT (tests/test_gpu_render.py):
//   * arithmetic is plain IEEE single precision in the reference's operation order -- FMA contraction is switched off for
//     this file, divisions and square roots are correctly rounded;
//   * vertex normals are summed per vertex over a CSR list of its incident triangles in ascending triangle order (built
//     once per topology by the host), which is the order the sequential loop adds them in;
//   * the z-buffer is a 64-bit atomicMax on  [face index | order-preserving depth bits | ~triangle index] : the sequential
//     rule "overwrite when strictly deeper" ends at the deepest triangle, earliest index among equals, and a later face
//     overwrites an earlier one wherever it covers (each face starts from a fresh depth buffer); a second pass re-derives
//     the barycentric weights of the winning triangle and writes the colour (alpha = 1, the binding's default).
// Exception: numpy evaluates (v2v*reflection)**5 with glibc powf; the kernel uses an exactly rounded double product, which
// differs from powf in the last bit for a small fraction of inputs (light agrees to 1e-6, pixels to one grey level).
#pragma clang fp contract(off)

#include "syn_internal.h"

namespace syn {

namespace {
// `planar`: 0 -> [nver,3] interleaved (the reference's layout); >= nver -> planar rows `planar` floats apart ([3,pitch][:, :nver],
// what syn_reconstruct_pitched writes; pitch == nver is the packed [3,nver]).  The launchers map the ABI's planar = 1 to nver.
__device__ __forceinline__ float vtx(const float *v, int planar, int nver, int i, int c) {
    return planar ? v[(size_t)c * planar + i] : v[(size_t)i * 3 + c];
}
__device__ __forceinline__ size_t face_stride(int planar, int nver) { return planar ? (size_t)3 * planar : (size_t)3 * nver;
}
}  // namespace

// ---- triangle cross products (rasterize_kernel.cpp:166-185), one thread per (face, triangle) ----
__global__ __launch_bounds__(256) void tri_normal_kernel(const float *__restrict__ vertices, const int *__restrict__ tri,
                                                         float *__restrict__ tri_normal, int nver, int ntri, int planar) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= ntri) return;
    const float *v = vertices + (size_t)f * face_stride(planar, nver);
    const int p0 = tri[3 * i], p1 = tri[3 * i + 1], p2 = tri[3 * i + 2];
    const float v1x = vtx(v, planar, nver, p1, 0) - vtx(v, planar, nver, p0, 0);
    const float v1y = vtx(v, planar, nver, p1, 1) - vtx(v, planar, nver, p0, 1);
    const float v1z = vtx(v, planar, nver, p1, 2) - vtx(v, planar, nver, p0, 2);
    const float v2x = vtx(v, planar, nver, p2, 0) - vtx(v, planar, nver, p0, 0);
    const float v2y = vtx(v, planar, nver, p2, 1) - vtx(v, planar, nver, p0, 1);
    const float v2z = vtx(v, planar, nver, p2, 2) - vtx(v, planar, nver, p0, 2);
    float *o = tri_normal + ((size_t)f * ntri + i) * 3;
    o[0] = v1y * v2z - v1z * v2y;
    o[1] = v1z * v2x - v1x * v2z;
    o[2] = v1x * v2y - v1y * v2x;
}

// ---- vertex normals (rasterize_kernel.cpp:187-212): ordered sum over incident triangles, then normalise ----
// normal out is [F, nver, 3] (the reference's layout).
__global__ __launch_bounds__(256) void ver_normal_kernel(const float *__restrict__ tri_normal, const int *__restrict__ adj_off,
                                                         const int *__restrict__ adj_tri, float *__restrict__ normal, int nver,
                                                         int ntri) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= nver) return;
    float x = 0.f, y = 0.f, z = 0.f;
    const float *tn = tri_normal + (size_t)f * ntri * 3;
    for (int a = adj_off[i]; a < adj_off[i + 1]; ++a) {
        const int t = adj_tri[a];
        x += tn[3 * t]; y += tn[3 * t + 1]; z += tn[3 * t + 2];
    }
    const float det = sqrtf(x * x + y * y + z * z);
    float *o = normal + ((size_t)f * nver + i) * 3;
    o[0] = x / det; o[1] = y / det; o[2] = z / det;
}

// ---- per-face, per-axis min / max of the vertices (norm_vertices, lighting.py:9-14) as order-preserving integer keys ----
// grid (kMinMaxBlocks, F): block-level reduction, then six atomics per block (a few dozen per face)
constexpr int kMinMaxBlocks = 8;
__global__ __launch_bounds__(1024) void minmax_kernel(const float *__restrict__ vertices, unsigned *__restrict__ mm, int nver, int planar) {
    __shared__ unsigned red[16][6];
    const int f = blockIdx.y;
    const float *v = vertices + (size_t)f * face_stride(planar, nver);
    auto key = [](float a) { const unsigned u = __builtin_bit_cast(unsigned, a); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    unsigned k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < nver; i += kMinMaxBlocks * 1024)
        for (int c = 0; c < 3; ++c) {
            const unsigned kk = key(vtx(v, planar, nver, i, c));
            k[c] = min(k[c], kk); k[3 + c] = max(k[3 + c], kk);
        }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            k[c] = min(k[c], (unsigned)__shfl_xor((int)k[c], s));
            k[3 + c] = max(k[3 + c], (unsigned)__shfl_xor((int)k[3 + c], s));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 6; ++c) red[wave][c] = k[c];
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned r = red[0][threadIdx.x];
        for (int w2 = 1; w2 < 16; ++w2) r = threadIdx.x < 3 ? min(r, red[w2][threadIdx.x]) : max(r, red[w2][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&mm[f * 6 + threadIdx.x], r); else atomicMax(&mm[f * 6 + threadIdx.x], r);
    }
}

// ---- Phong vertex colours (lighting.py:37-71 with norm_vertices :9-14), one thread per (face, vertex) ----
// cfg: [0] intensity_ambient [1..3] color_ambient [4] intensity_directional [5..7] color_directional
//      [8] intensity_specular [9] specular_exp(int) [10..12] light_pos [13..15] view_pos
__global__ __launch_bounds__(256) void lighting_kernel(const float *__restrict__ vertices, const float *__restrict__ normal,
                                                       const unsigned *__restrict__ mm, const float *__restrict__ cfg,
                                                       float *__restrict__ light, int nver, int planar) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= nver) return;
    auto unkey = [](unsigned k) { const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __builtin_bit_cast(float, u); };
    const unsigned *m = mm + f * 6;
    const float *v = vertices + (size_t)f * face_stride(planar, nver);
    // norm_vertices: v -= min(0); v /= v.max(); v *= 2; v -= v.max(0) / 2   (all monotone: the extrema map to the extrema)
    float mn[3], mx[3], vn[3];
    for (int c = 0; c < 3; ++c) { mn[c] = unkey(m[c]); mx[c] = unkey(m[3 + c]) - mn[c]; }
    const float s = fmaxf(mx[0], fmaxf(mx[1], mx[2]));
    for (int c = 0; c < 3; ++c) {
        const float hi = (mx[c] / s) * 2.0f;
        vn[c] = ((vtx(v, planar, nver, i, c) - mn[c]) / s) * 2.0f - hi / 2.0f;
    }
    const float *n = normal + ((size_t)f * nver + i) * 3;
    float l[3] = {0.f, 0.f, 0.f};
    if (cfg[0] > 0)
        for (int c = 0; c < 3; ++c) l[c] += cfg[0] * cfg[1 + c];
    if (cfg[4] > 0) {
        float d[3], dn = 0.f;
        for (int c = 0; c < 3; ++c) { d[c] = cfg[10 + c] - vn[c]; dn += d[c] * d[c]; }
        dn = sqrtf(dn);
        float cs = 0.f;
        for (int c = 0; c < 3; ++c) { d[c] = d[c] / dn; cs += n[c] * d[c]; }
        const float cc = fminf(fmaxf(cs, 0.0f), 1.0f);          // np.clip propagates NaN exactly like fmin(fmax()) does not:
        const float ccl = (cs != cs) ? cs : cc;                  // keep NaN (vertices without triangles), as numpy does
        for (int c = 0; c < 3; ++c) l[c] += cfg[4] * (cfg[5 + c] * ccl);
        if (cfg[8] > 0) {
            float w[3], wn = 0.f;
            for (int c = 0; c < 3; ++c) { w[c] = cfg[13 + c] - vn[c]; wn += w[c] * w[c]; }
            wn = sqrtf(wn);
            const int e = (int)cfg[9];
            float spe = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float r = (2.0f * cs) * n[c] - d[c];
                const double b = (double)((w[c] / wn) * r);
                double p = 1.0;
                for (int q = 0; q < e; ++q) p *= b;
                spe += (float)p;
            }
            float sc = (spe != spe) ? spe : fminf(fmaxf(spe, 0.0f), 1.0f);
            if (!(cs != 0.0f)) sc = 0.0f;                         // np.where(cos != 0, clip(spe), 0): NaN != 0 is True
            const float sc2 = (sc != sc) ? sc : fminf(fmaxf(sc, 0.0f), 1.0f);
            for (int c = 0; c < 3; ++c) l[c] += (cfg[8] * cfg[5 + c]) * sc2;
        }
    }
    float *o = light + ((size_t)f * nver + i) * 3;
    for (int c = 0; c < 3; ++c) o[c] = (l[c] != l[c]) ? l[c] : fminf(fmaxf(l[c], 0.0f), 1.0f);
}

namespace {
struct Bary { float w0, w1, w2; bool in; };
// is_point_in_tri + get_point_weight (rasterize_kernel.cpp:26-82)
__device__ __forceinline__ Bary bary(float px, float py, float p0x, float p0y, float p1x, float p1y, float p2x, float p2y) {
    const float v0x = p2x - p0x, v0y = p2y - p0y, v1x = p1x - p0x, v1y = p1y - p0y, v2x = px - p0x, v2y = py - p0y;
    const float dot00 = v0x * v0x + v0y * v0y, dot01 = v0x * v1x + v0y * v1y, dot02 = v0x * v2x + v0y * v2y;
    const float dot11 = v1x * v1x + v1y * v1y, dot12 = v1x * v2x + v1y * v2y;
    const float den = dot00 * dot11 - dot01 * dot01;
    const float inv = den == 0 ? 0.0f : 1 / den;
    const float u = (dot11 * dot02 - dot01 * dot12) * inv, v = (dot00 * dot12 - dot01 * dot02) * inv;
    Bary b;
    b.w0 = 1 - u - v; b.w1 = v; b.w2 = u;
    b.in = (u >= 0) && (v >= 0) && (u + v < 1);
    return b;
}
__device__ __forceinline__ unsigned depth_key(float d) {
    const unsigned u = __builtin_bit_cast(unsigned, d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
}  // namespace

// ---- z-buffer pass 1 (rasterize_kernel.cpp:229-262): one thread per (face, triangle) walks its bounding box ----
__global__ __launch_bounds__(256) void raster_depth_kernel(const float *__restrict__ vertices, const int *__restrict__ tri,
                                                           unsigned long long *__restrict__ zkey, int nver, int ntri, int h,
                                                           int w, int planar) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= ntri) return;
    const float *v = vertices + (size_t)f * face_stride(planar, nver);
    const int t0 = tri[3 * i], t1 = tri[3 * i + 1], t2 = tri[3 * i + 2];
    const float p0x = vtx(v, planar, nver, t0, 0), p0y = vtx(v, planar, nver, t0, 1), d0 = vtx(v, planar, nver, t0, 2);
    const float p1x = vtx(v, planar, nver, t1, 0), p1y = vtx(v, planar, nver, t1, 1), d1 = vtx(v, planar, nver, t1, 2);
    const float p2x = vtx(v, planar, nver, t2, 0), p2y = vtx(v, planar, nver, t2, 1), d2 = vtx(v, planar, nver, t2, 2);
    const int x_min = max((int)floorf(fminf(p0x, fminf(p1x, p2x))), 0), x_max = min((int)ceilf(fmaxf(p0x, fmaxf(p1x, p2x))), w - 1);
    const int y_min = max((int)floorf(fminf(p0y, fminf(p1y, p2y))), 0), y_max = min((int)ceilf(fmaxf(p0y, fmaxf(p1y, p2y))), h - 1);
    if (x_max < x_min || y_max < y_min) return;
    const unsigned long long hi = ((unsigned long long)(f + 1) << 56), lo = (unsigned long long)(0xffffffu - (unsigned)i);
    for (int y = y_min; y <= y_max; ++y)
        for (int x = x_min; x <= x_max; ++x) {
            const Bary b = bary((float)x, (float)y, p0x, p0y, p1x, p1y, p2x, p2y);
            if (!b.in) continue;
            const float depth = b.w0 * d0 + b.w1 * d1 + b.w2 * d2;
            if (!(depth > -1e8f)) continue;                       // the fresh depth buffer holds -1e8 (Sim3DR.py:23)
            atomicMax(&zkey[(size_t)y * w + x], hi | ((unsigned long long)depth_key(depth) << 24) | lo);
        }
}

// ---- pass 2 (rasterize_kernel.cpp:264-280, alpha = 1): one thread per pixel shades the winning triangle ----
__global__ __launch_bounds__(256) void raster_shade_kernel(const float *__restrict__ vertices, const int *__restrict__ tri,
                                                           const float *__restrict__ colors, const unsigned long long *__restrict__ zkey,
                                                           unsigned char *__restrict__ image, int nver, int h, int w, int c,
                                                           int planar, int reverse) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= h * w) return;
    const unsigned long long k = zkey[px];
    if (k == 0ull) return;
    const int f = (int)(k >> 56) - 1;
    int i = (int)(0xffffffu - (unsigned)(k & 0xffffffull));
    // hipcc 7.2 folds ((k & 0xffffff) ^ 0xffffff) * 12 into a 24-bit multiply pattern and then widens it again WITHOUT the mask
    // (memory fault on tri[3*i]); an opaque copy keeps the masked value
    asm volatile("" : "+v"(i));
    const int y = px / w, x = px % w;
    const float *v = vertices + (size_t)f * face_stride(planar, nver);
    const float *col = colors + (size_t)f * nver * c;
    const int t0 = tri[3 * i], t1 = tri[3 * i + 1], t2 = tri[3 * i + 2];
    const Bary b = bary((float)x, (float)y, vtx(v, planar, nver, t0, 0), vtx(v, planar, nver, t0, 1), vtx(v, planar, nver, t1, 0),
                        vtx(v, planar, nver, t1, 1), vtx(v, planar, nver, t2, 0), vtx(v, planar, nver, t2, 1));
    unsigned char *o = image + ((size_t)(reverse ? (h - 1 - y) : y) * w + x) * c;
    for (int q = 0; q < c; ++q) {
        const float pc = b.w0 * col[c * t0 + q] + b.w1 * col[c * t1 + q] + b.w2 * col[c * t2 + q];
        o[q] = (unsigned char)(int)(255 * pc);      // (1-alpha)*image + alpha*255*p_color with alpha = 1: 0*image + 255*p_color exactly
    }
}

// ---- cv2.addWeighted for uint8 images (utils/render.py:45): saturate(round-half-even(a*alpha + b*beta)) ----
__global__ __launch_bounds__(256) void add_weighted_kernel(const unsigned char *__restrict__ a, float alpha,
                                                           const unsigned char *__restrict__ b, float beta,
                                                           unsigned char *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = (float)a[i] * alpha + (float)b[i] * beta;
    out[i] = (unsigned char)fminf(fmaxf(rintf(v), 0.0f), 255.0f);
}

void launch_mesh_normals(const float *vertices, const int *tri, const int *adj_off, const int *adj_tri, float *tri_normal,
                         float *normal, unsigned *mm, int F, int nver, int ntri, int planar, hipStream_t s) {
    // per-face extrema start at +inf / -inf in key space
    (void)hipMemsetAsync(mm, 0, sizeof(unsigned) * 6 * F, s);
    (void)hipMemset2DAsync(mm, 6 * sizeof(unsigned), 0xff, 3 * sizeof(unsigned), F, s);
    tri_normal_kernel<<<dim3((ntri + 255) / 256, F), 256, 0, s>>>(vertices, tri, tri_normal, nver, ntri, planar);
    ver_normal_kernel<<<dim3((nver + 255) / 256, F), 256, 0, s>>>(tri_normal, adj_off, adj_tri, normal, nver, ntri);
    minmax_kernel<<<dim3(kMinMaxBlocks, F), 1024, 0, s>>>(vertices, mm, nver, planar);
}

void launch_mesh_lighting(const float *vertices, const float *normal, const unsigned *mm, const float *cfg, float *light, int F,
                          int nver, int planar, hipStream_t s) {
    lighting_kernel<<<dim3((nver + 255) / 256, F), 256, 0, s>>>(vertices, normal, mm, cfg, light, nver, planar);
}

void launch_rasterize(const float *vertices, const int *tri, const float *colors, unsigned long long *zkey, unsigned char *image,
                      int F, int nver, int ntri, int h, int w, int c, int planar, int reverse, hipStream_t s) {
    (void)hipMemsetAsync(zkey, 0, sizeof(unsigned long long) * (size_t)h * w, s);
    raster_depth_kernel<<<dim3((ntri + 255) / 256, F), 256, 0, s>>>(vertices, tri, zkey, nver, ntri, h, w, planar);
    raster_shade_kernel<<<(h * w + 255) / 256, 256, 0, s>>>(vertices, tri, colors, zkey, image, nver, h, w, c, planar, reverse);
}

void launch_add_weighted(const unsigned char *a, float alpha, const unsigned char *b, float beta, unsigned char *out, size_t n,
                         hipStream_t s) {
    add_weighted_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a, alpha, b, beta, out, n);
}

}  // namespace syn
