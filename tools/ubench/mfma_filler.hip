// Microbenchmark: how many VALU instructions of the SAME wave hide in the shadow of one v_mfma_f32_16x16x32_bf16 /
// v_mfma_f32_16x16x4_f32?  One or two waves per SIMD run  { MFMA ; NF x v_fma_f32 }  in a loop; prints cycles per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -o ub2 tools/ubench/mfma_filler.hip   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND, int NF, int OP>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f32x16 c0 = {0}, c1 = {0};
    bf16x8 bx, by;
    for (int e = 0; e < 8; ++e) { bx[e] = (__bf16)(x + e); by[e] = (__bf16)(y + e); }
    float v[8];
    unsigned w[8];
    for (int e = 0; e < 8; ++e) { v[e] = x + e; w[e] = threadIdx.x * 7 + e; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 &acc = u == 0 ? a0 : u == 1 ? a1 : u == 2 ? a2 : a3;
            if (KIND == 2) { if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, bx, c0, 0, 0, 0); }
            else if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, acc, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (OP == 0) v[f % 8] = __builtin_fmaf(v[f % 8], 1.0001f, 0.5f);
                else if (OP == 1) w[f % 8] = (w[f % 8] & 0xffff0f0fu) + 3u;        // integer: v_and + v_add
                else v[f % 8] = __builtin_amdgcn_fmed3f(v[f % 8], 0.25f, 6.0f) ;
            }
        }
    }
    float s = a0[0] + a1[1] + a2[2] + a3[3] + c0[5] + c1[9];
    for (int e = 0; e < 8; ++e) s += v[e] + (float)w[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KIND, int NF, int OP> void run1(const char *name, int threads) {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND, NF, OP><<<256, threads>>>(d, 100); hipDeviceSynchronize();
    const int iters = 4000;
    hipEventRecord(a); k<KIND, NF, OP><<<256, threads>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mfma_per_simd = (double)iters * 4 * (threads / 256);
    printf("%s op=%d waves/SIMD=%d fillers/MFMA=%d : %.1f ns per MFMA slot (%.1f cycles @2.4GHz)\n", name, OP, threads / 256, NF, ms * 1e6 / mfma_per_simd,
           ms * 1e6 / mfma_per_simd * 2.4);
    hipFree(d);
}
template <int KIND, int OP> void sweep(const char *name) {
    run1<KIND, 0, OP>(name, 256); run1<KIND, 2, OP>(name, 256); run1<KIND, 4, OP>(name, 256); run1<KIND, 6, OP>(name, 256);
    run1<KIND, 8, OP>(name, 256); run1<KIND, 12, OP>(name, 256);
    run1<KIND, 4, OP>(name, 512); run1<KIND, 8, OP>(name, 512);
}
int main() {
    sweep<2, 0>("bf16_32x32x16 + v_fma");
    sweep<2, 1>("bf16_32x32x16 + v_and/v_add");
    run1<2, 16, 0>("bf16_32x32x16 + v_fma", 256); run1<2, 24, 0>("bf16_32x32x16 + v_fma", 256);
    return 0;
}
