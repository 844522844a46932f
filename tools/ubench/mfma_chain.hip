// Microbenchmark: v_mfma_f32_16x16x32_bf16 issue rate vs number of independent accumulator chains (1, 2, 3, 4, 6, 8),
// one or two waves per SIMD.   build: hipcc --offload-arch=gfx950 -O3 -o mc tools/ubench/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NC>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    float x = threadIdx.x * 1e-3f;
    bf16x8 bx, by;
    for (int e = 0; e < 8; ++e) { bx[e] = (__bf16)(x + e); by[e] = (__bf16)(1.0f + e); }
    f32x4 a[NC];
    for (int c = 0; c < NC; ++c) a[c] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 24 / NC; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) a[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < NC; ++c) s += a[c][c & 3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NC> void run(int threads) {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NC><<<256, threads>>>(d, 100); hipDeviceSynchronize();
    const int iters = 4000;
    hipEventRecord(a); k<NC><<<256, threads>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)iters * 24 * (threads / 256);
    printf("chains=%d waves/SIMD=%d : %.2f ns per MFMA per SIMD\n", NC, threads / 256, ms * 1e6 / n);
    hipFree(d);
}
int main() {
    run<1>(256); run<2>(256); run<3>(256); run<4>(256); run<6>(256); run<8>(256);
    run<1>(512); run<2>(512); run<3>(512); run<4>(512);
    return 0;
}
