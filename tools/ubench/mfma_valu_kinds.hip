// Microbenchmark: which VALU instruction kinds of one wave overlap with v_mfma_f32_16x16x32_bf16 of the OTHER wave of its SIMD?
// block = 512 threads = 8 waves = 2 per SIMD (waves w and w + 4).  modes as mfma_valu_overlap.hip:
//   2: waves 0-3 MFMA, 4-7 VALU    3: 4 MFMA waves alone    4: 4 VALU waves alone    5: waves 0-3 MFMA then VALU, 4-7 VALU then MFMA (phased)
// VK 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_mov_dpp + v_fma, 3 = bf16-split mix (v_and, v_sub, v_perm), 4 = ds_read_b128 + v_fma
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubk tools/ubench/mfma_valu_kinds.hip   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float dppl(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true)); }

template <int VK>
__device__ __forceinline__ float valu_work(float x, float y, int iters, const float *lds) {
    if (VK == 0) {
        float v[8];
        for (int e = 0; e < 8; ++e) v[e] = x + e;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
        return v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
    } else if (VK == 1) {
        f32x2 v[4], m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
        for (int e = 0; e < 4; ++e) v[e] = (f32x2){x + e, y + e};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v[e] * m + c; asm volatile("" : "+v"(v[e])); }      // 64 v_pk_fma = 128 FMAs
        return v[0][0] + v[1][1] + v[2][0] + v[3][1];
    } else if (VK == 2) {
        float v[8];
        for (int e = 0; e < 8; ++e) v[e] = x + e;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(dppl(v[e]), 1.0001f, 0.5f);      // 64 dpp + 64 fma
        return v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
    } else if (VK == 3) {
        float v[8];
        unsigned acc = 0;
        for (int e = 0; e < 8; ++e) v[e] = x + e;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {                                                     // 8 ops per pair step: 32 x 4 = 128
                    const unsigned u0 = __builtin_bit_cast(unsigned, v[e]), u1 = __builtin_bit_cast(unsigned, v[e + 1]);
                    const float r0 = v[e] - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = v[e + 1] - __builtin_bit_cast(float, u1 & 0xffff0000u);
                    acc ^= __builtin_amdgcn_perm(u1, u0, 0x07060302u);
                    v[e] = r0 + 1.5f; v[e + 1] = r1 + 2.5f;
                }
        return v[0] + v[1] + __builtin_bit_cast(float, acc);
    } else {
        f32x4 s = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const f32x4 t = *(const f32x4 *)&lds[((threadIdx.x & 63) * 4 + u * 256 + i * 4) & 8191];
                s += t * 1.0001f;                                                                    // 32 ds_read_b128 + 128 fma
            }
        return s[0] + s[1] + s[2] + s[3];
    }
}

__device__ __forceinline__ float mfma_work(float x, float y, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 bx, by;
    for (int e = 0; e < 8; ++e) { bx[e] = (__bf16)(x + e); by[e] = (__bf16)(y + e); }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(by, bx, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, bx, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(by, by, a3, 0, 0, 0);
        }
    return a0[0] + a1[1] + a2[2] + a3[3];
}

template <int VK>
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 1e-4f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float r = 0;
    if (mode == 5) {            // phased: each wave alternates 40 MFMA-iterations and the equivalent VALU block, the partner in antiphase
        for (int rep = 0; rep < iters / 40; ++rep) {
            if (wave < 4) { r += mfma_work(x + rep, y, 40); r += valu_work<VK>(x + rep, y, 40, lds); }
            else { r += valu_work<VK>(x + rep, y, 40, lds); r += mfma_work(x + rep, y, 40); }
        }
    } else {
        if ((mode == 2 || mode == 3) && wave < 4) r = mfma_work(x, y, iters);
        if ((mode == 2 || mode == 4) && wave >= 4) r = valu_work<VK>(x, y, iters, lds);
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int VK> void run(const char *name) {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    static const char *what[] = {"", "", "4 MFMA + 4 VALU waves", "4 MFMA waves alone", "4 VALU waves alone", "8 waves, antiphase MFMA / VALU blocks"};
    for (int mode = 2; mode < 6; ++mode) {
        k<VK><<<256, 512>>>(d, 200, mode); hipDeviceSynchronize();
        hipEventRecord(a); k<VK><<<256, 512>>>(d, 2000, mode); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-14s mode %d (%-38s): %.3f ms\n", name, mode, what[mode], ms);
    }
}
int main() {
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("dpp + fma"); run<3>("split mix"); run<4>("ds_read + fma");
    return 0;
}
