// Microbenchmark + correctness check: the two-piece fp16 split of a pair of fp32 values,
//   A (current):  a = v_cvt_pkrtz(x0, x1);  r0 = v_fma_mix_f32(a.lo, -1, x0);  r1 = v_fma_mix_f32(a.hi, -1, x1);  b = v_cvt_pkrtz(r0, r1)   (4 VALU)
//   B (candidate): a = v_cvt_pkrtz(x0, x1);  b.lo = v_fma_mixlo_f16(a.lo, -1, x0);  b.hi = v_fma_mixhi_f16(a.hi, -1, x1)                    (3 VALU)
// B's low piece is rounded to nearest (the f16 rounding mode) instead of toward zero: |x - (a + b)| can only shrink.
// build: hipcc --offload-arch=gfx950 -O3 -o split_mixlo tools/ubench/split_mixlo.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ void splitA(float x0, float x1, unsigned &a, unsigned &b) {
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
    b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ void splitB(float x0, float x1, unsigned &a, unsigned &b) {
    a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(b) : "v"(a), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(b) : "v"(a), "v"(x1));
}
template <int V> __global__ void check(const float *x, unsigned *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a, b;
    if (V == 0) splitA(x[2 * i], x[2 * i + 1], a, b); else splitB(x[2 * i], x[2 * i + 1], a, b);
    out[2 * i] = a; out[2 * i + 1] = b;
}
template <int V> __global__ void bench(float *io, int iters) {
    float x0 = io[threadIdx.x], x1 = io[threadIdx.x + 256];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            unsigned a, b;
            if (V == 0) splitA(x0, x1, a, b); else splitB(x0, x1, a, b);
            acc ^= a + b;
            x0 += 1.0f; x1 -= 0.5f;
        }
    }
    io[threadIdx.x + blockIdx.x * 0] = (float)acc;
}
static float h2f(unsigned h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    const float v = e ? ldexpf(1.0f + m / 1024.0f, e - 15) : ldexpf((float)m, -24);
    return (h & 0x8000u) ? -v : v;
}
int main() {
    const int n = 1 << 20;
    std::vector<float> hx(n);
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) {                       // magnitudes 2^-20 .. 2^15, both signs, plus exact f16 values and zeros
        s = s * 1664525u + 1013904223u;
        const float m = 1.0f + (s >> 9) / 8388608.0f;
        s = s * 1664525u + 1013904223u;
        const int e = (int)(s >> 26) % 36 - 20;
        hx[i] = ((s >> 5) & 1 ? -1.f : 1.f) * ldexpf(m, e);
        if (i % 97 == 0) hx[i] = 0.f;
        if (i % 101 == 0) hx[i] = h2f((s >> 8) & 0x7bff);
    }
    float *dx; unsigned *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> o(n);
    for (int v = 0; v < 2; ++v) {
        if (v == 0) check<0><<<n / 2 / 256, 256>>>(dx, dout, n); else check<1><<<n / 2 / 256, 256>>>(dx, dout, n);
        hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        double worst = 0; int bad_a = 0;
        for (int i = 0; i < n / 2; ++i)
            for (int k = 0; k < 2; ++k) {
                const float x = hx[2 * i + k];
                const unsigned a = (o[2 * i] >> (16 * k)) & 0xffff, b = (o[2 * i + 1] >> (16 * k)) & 0xffff;
                const double rec = (double)h2f(a) + (double)h2f(b);
                if (x != 0.f && fabs(x) >= ldexp(1.0, -3)) { const double r = fabs(rec - x) / fabs(x); if (r > worst) worst = r; }
                if (fabsf(h2f(a)) > fabsf(x)) ++bad_a;
            }
        printf("variant %c: worst relative |x - (a+b)| / |x| for |x| >= 2^-3: 2^%.2f   high pieces beyond |x|: %d\n", "AB"[v], log2(worst), bad_a);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v == 0) bench<0><<<2048, 256>>>(dx, 2000); else bench<1><<<2048, 256>>>(dx, 2000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("variant %c: %.3f ms for 2048 x 256 threads x 32000 splits\n", "AB"[v], ms);
    }
    return 0;
}
