// TEST INFRASTRUCTURE -- C-linkage entry points around the REAL reference rasteriser, compiled together with
// /root/reference/Sim3DR/lib/rasterize_kernel.cpp where it lies (oracle/Makefile -> oracle/_ref/libsim3dr_ref.so).
// Nothing of the reference is copied: this file only forwards to the functions its own Cython binding calls
// (Sim3DR/lib/rasterize.pyx:66-74 get_normal, :96-110 rasterize).
#include "rasterize.h"

extern "C" void ref_get_normal(float *ver_normal, float *vertices, int *triangles, int nver, int ntri) {
    _get_normal(ver_normal, vertices, triangles, nver, ntri);
}
extern "C" void ref_rasterize(unsigned char *image, float *vertices, int *triangles, float *colors, float *depth_buffer,
                              int ntri, int h, int w, int c, float alpha, int reverse) {
    _rasterize(image, vertices, triangles, colors, depth_buffer, ntri, h, w, c, alpha, reverse != 0);
}
