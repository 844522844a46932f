// Microbenchmark (round 4): SIMD cycles per wave64 instruction by kind and by waves per SIMD -- what the row-marching kernels'
// vector work really costs.  One workgroup per CU (256 x W*4 waves), every wave issues N instructions of ONE kind on eight
// independent register chains; ns per instruction per SIMD = time / (N x W).  Kinds: 0 v_fma_f32, 1 v_pk_fma_f32 (2 FMAs per lane),
// 2 v_mov_b32_dpp wave_shr:1, 3 v_med3_f32, 4 v_cvt_pkrtz_f16_f32, 5 v_fma_mix_f32, 6 v_perm_b32, 7 ds_read_b128 (all lanes of a
// half one address), 8 v_mfma_f32_32x32x16_f16, 9 v_fmac_f32 with an SGPR operand, 10 v_pk_mul_f32, 11 v_add_f32_dpp (DPP folded into the op)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_issue tools/ubench/valu_issue.hip     run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K>
__global__ __launch_bounds__(1024) void bench(float *out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed + threadIdx.x + e;
    f32x2 p[8];
    for (int e = 0; e < 8; ++e) p[e] = (f32x2){seed + e, seed - e};
    f32x16 acc = {};
    f16x8 a = {}, b = {};
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + e); b[e] = (_Float16)(seed - e); }
    f32x4 l = {0, 0, 0, 0};
    const float *lp = lds + (threadIdx.x >> 5) * 4;
    float sg = seed * 1.5f;
    sg = __builtin_amdgcn_readfirstlane(sg);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(v[(e + 1) & 7]), "v"(v[(e + 2) & 7]));
                if (K == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[e]) : "v"(p[(e + 1) & 7]), "v"(p[(e + 2) & 7]));
                if (K == 2) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v[e]) : "v"(v[(e + 1) & 7]));
                if (K == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(v[(e + 1) & 7]), "v"(v[(e + 2) & 7]));
                if (K == 4) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[(e + 1) & 7]), "v"(v[(e + 2) & 7]));
                if (K == 5) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v[e]) : "v"(v[(e + 1) & 7]), "v"(v[(e + 2) & 7]));
                if (K == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(v[e]) : "v"(v[(e + 1) & 7]), "v"(v[(e + 2) & 7]), "v"(v[(e + 3) & 7]));
                if (K == 7) { f32x4 t = *(const volatile f32x4 *)(lp + 8 * e + 64 * u); l += t; }
                if (K == 8) { if ((e & 3) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0); }
                if (K == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[e]) : "s"(sg), "v"(v[(e + 1) & 7]));
                if (K == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[e]) : "v"(p[(e + 1) & 7]));
                if (K == 11) asm volatile("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v[e]) : "v"(v[(e + 1) & 7]), "v"(v[(e + 2) & 7]));
            }
    }
    float r = 0;
    for (int e = 0; e < 8; ++e) r += v[e] + p[e][0] + p[e][1];
    for (int e = 0; e < 16; ++e) r += acc[e];
    r += l[0] + l[1] + l[2] + l[3];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int K>
static void run(const char *name, float *out, int per_iter) {
    printf("%-28s", name);
    for (int W = 1; W <= 4; ++W) {
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        bench<K><<<256, W * 256, 0, 0>>>(out, 10, 1.0f);
        hipEventRecord(e0, 0);
        bench<K><<<256, W * 256, 0, 0>>>(out, iters, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  W=%d %6.2f ns", W, ms * 1e6 / ((double)iters * per_iter * W));
    }
    printf("   (per instruction per SIMD)\n");
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    run<0>("v_fma_f32", out, 64);
    run<1>("v_pk_fma_f32", out, 64);
    run<2>("v_mov_b32_dpp wave_shr", out, 64);
    run<3>("v_med3_f32", out, 64);
    run<4>("v_cvt_pkrtz_f16_f32", out, 64);
    run<5>("v_fma_mix_f32", out, 64);
    run<6>("v_perm_b32", out, 64);
    run<7>("ds_read_b128 (bcast) + 4 add", out, 64);
    run<8>("v_mfma_f32_32x32x16_f16", out, 16);
    run<9>("v_fmac_f32 (SGPR operand)", out, 64);
    run<10>("v_pk_mul_f32", out, 64);
    run<11>("v_add_f32_dpp wave_shr", out, 64);
    return 0;
}
