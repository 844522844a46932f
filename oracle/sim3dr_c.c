/* TEST INFRASTRUCTURE -- CPU restatement of the reference's mesh consumers (Sim3DR), plain C.
 *
 * Only tests/, __graft_entry__.smoke() and tools that time a CPU baseline may use this file; the product path
 * (synergynet_amd/csrc/render_kernels.hip) never links or calls it.
 *
 * Restates, operation for operation (single-precision IEEE arithmetic, no FMA contraction -- build with
 * -ffp-contract=off), the two functions of /root/reference/Sim3DR/lib/rasterize_kernel.cpp that the reference's Python
 * side calls (Sim3DR/Sim3DR.py:8-29 through lib/rasterize.pyx:66-74, 96-110):
 *   _get_normal   rasterize_kernel.cpp:158-215   per-vertex normals = normalised sum of incident triangle cross products
 *   _rasterize    rasterize_kernel.cpp:219-287   z-buffer rasteriser with barycentric colour interpolation
 * with their helpers is_point_in_tri (:26-52) and get_point_weight (:54-82).
 * Pinned against the real reference (oracle/_ref/libsim3dr_ref.so, built by oracle/Makefile from the reference's own
 * source where it lies) by tests/test_render_cpu.py, and against tests/golden/render_golden.npz produced by it.
 */
#include <math.h>
#include <stdlib.h>

/* rasterize_kernel.cpp:158-215 (note: the det <= 0 guard is commented out there: isolated vertices give 0/0 = NaN) */
void sim_get_normal(float *ver_normal, const float *vertices, const int *triangles, int nver, int ntri) {
    float *tri_normal = (float *)malloc(sizeof(float) * 3 * (size_t)(ntri > 0 ? ntri : 1));
    for (int i = 0; i < ntri; i++) {
        const int p0 = triangles[3 * i], p1 = triangles[3 * i + 1], p2 = triangles[3 * i + 2];
        const float v1x = vertices[3 * p1] - vertices[3 * p0];
        const float v1y = vertices[3 * p1 + 1] - vertices[3 * p0 + 1];
        const float v1z = vertices[3 * p1 + 2] - vertices[3 * p0 + 2];
        const float v2x = vertices[3 * p2] - vertices[3 * p0];
        const float v2y = vertices[3 * p2 + 1] - vertices[3 * p0 + 1];
        const float v2z = vertices[3 * p2 + 2] - vertices[3 * p0 + 2];
        tri_normal[3 * i] = v1y * v2z - v1z * v2y;
        tri_normal[3 * i + 1] = v1z * v2x - v1x * v2z;
        tri_normal[3 * i + 2] = v1x * v2y - v1y * v2x;
    }
    for (int i = 0; i < ntri; i++) {          /* accumulation in triangle order: the float sums depend on it */
        const int p0 = triangles[3 * i], p1 = triangles[3 * i + 1], p2 = triangles[3 * i + 2];
        for (int j = 0; j < 3; j++) {
            ver_normal[3 * p0 + j] += tri_normal[3 * i + j];
            ver_normal[3 * p1 + j] += tri_normal[3 * i + j];
            ver_normal[3 * p2 + j] += tri_normal[3 * i + j];
        }
    }
    for (int i = 0; i < nver; ++i) {
        const float nx = ver_normal[3 * i], ny = ver_normal[3 * i + 1], nz = ver_normal[3 * i + 2];
        const float det = sqrtf(nx * nx + ny * ny + nz * nz);
        ver_normal[3 * i] = nx / det;
        ver_normal[3 * i + 1] = ny / det;
        ver_normal[3 * i + 2] = nz / det;
    }
    free(tri_normal);
}

typedef struct { float x, y; } pt2;

/* rasterize_kernel.cpp:26-52 and :54-82 share the arithmetic: returns inside flag, writes the weights */
static int bary(pt2 p, pt2 p0, pt2 p1, pt2 p2, float *weight) {
    const float v0x = p2.x - p0.x, v0y = p2.y - p0.y;
    const float v1x = p1.x - p0.x, v1y = p1.y - p0.y;
    const float v2x = p.x - p0.x, v2y = p.y - p0.y;
    const float dot00 = v0x * v0x + v0y * v0y;
    const float dot01 = v0x * v1x + v0y * v1y;
    const float dot02 = v0x * v2x + v0y * v2y;
    const float dot11 = v1x * v1x + v1y * v1y;
    const float dot12 = v1x * v2x + v1y * v2y;
    float inverDeno;
    if (dot00 * dot11 - dot01 * dot01 == 0) inverDeno = 0;
    else inverDeno = 1 / (dot00 * dot11 - dot01 * dot01);
    const float u = (dot11 * dot02 - dot01 * dot12) * inverDeno;
    const float v = (dot00 * dot12 - dot01 * dot02) * inverDeno;
    weight[0] = 1 - u - v;
    weight[1] = v;
    weight[2] = u;
    return (u >= 0) && (v >= 0) && (u + v < 1);
}

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* rasterize_kernel.cpp:219-287 */
void sim_rasterize(unsigned char *image, const float *vertices, const int *triangles, const float *colors,
                   float *depth_buffer, int ntri, int h, int w, int c, float alpha, int reverse) {
    for (int i = 0; i < ntri; i++) {
        const int t0 = triangles[3 * i], t1 = triangles[3 * i + 1], t2 = triangles[3 * i + 2];
        pt2 p0 = {vertices[3 * t0], vertices[3 * t0 + 1]}, p1 = {vertices[3 * t1], vertices[3 * t1 + 1]},
            p2 = {vertices[3 * t2], vertices[3 * t2 + 1]};
        const float d0 = vertices[3 * t0 + 2], d1 = vertices[3 * t1 + 2], d2 = vertices[3 * t2 + 2];
        const int x_min = imax((int)floorf(fminf(p0.x, fminf(p1.x, p2.x))), 0);
        const int x_max = imin((int)ceilf(fmaxf(p0.x, fmaxf(p1.x, p2.x))), w - 1);
        const int y_min = imax((int)floorf(fminf(p0.y, fminf(p1.y, p2.y))), 0);
        const int y_max = imin((int)ceilf(fmaxf(p0.y, fmaxf(p1.y, p2.y))), h - 1);
        if (x_max < x_min || y_max < y_min) continue;
        for (int y = y_min; y <= y_max; y++)
            for (int x = x_min; x <= x_max; x++) {
                pt2 p = {(float)x, (float)y};
                float weight[3];
                if (!bary(p, p0, p1, p2, weight)) continue;
                const float p_depth = weight[0] * d0 + weight[1] * d1 + weight[2] * d2;
                if (p_depth > depth_buffer[y * w + x]) {
                    for (int k = 0; k < c; k++) {
                        const float p_color = weight[0] * colors[c * t0 + k] + weight[1] * colors[c * t1 + k] + weight[2] * colors[c * t2 + k];
                        unsigned char *px = &image[(reverse ? (h - 1 - y) : y) * w * c + x * c + k];
                        *px = (unsigned char)((1 - alpha) * *px + alpha * 255 * p_color);
                    }
                    depth_buffer[y * w + x] = p_depth;
                }
            }
    }
}
