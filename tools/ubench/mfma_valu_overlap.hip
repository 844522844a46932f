// Microbenchmark: do MFMAs and plain VALU FMAs of two co-resident waves of one SIMD overlap?
// block = 512 threads = 8 waves = 2 per SIMD.
//   mode 0: all 8 waves MFMA        mode 1: all 8 waves VALU       mode 2: waves 0-3 MFMA, 4-7 VALU (one of each per SIMD)
//   mode 3: waves 0-3 MFMA, rest idle   mode 4: waves 4-7 VALU, rest idle
// Separate pipes: t(2) ~ max(t3, t4); shared pipe: t(2) ~ t3 + t4.
// KIND 0 = 16x16x4 f32, 1 = 32x32x2 f32, 2 = 16x16x32 bf16, 3 = 16x16x16 bf16 (_1k)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench/mfma_valu_overlap.hip   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_valu = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    if (do_mfma) {
        if (KIND == 1) {
            f32x16 a0 = {0}, a1 = {0};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); }
            }
            out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[3];
        } else {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            bf16x8 bx, by;
            s16x4 sx, sy;
            for (int e = 0; e < 8; ++e) { bx[e] = (__bf16)(x + e); by[e] = (__bf16)(y + e); }
            for (int e = 0; e < 4; ++e) { sx[e] = (short)(threadIdx.x + e); sy[e] = (short)(threadIdx.x * 3 + e); }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (KIND == 0) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
                    } else if (KIND == 2) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(by, bx, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, bx, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(by, by, a3, 0, 0, 0);
                    } else {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sx, sy, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sy, sx, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sx, sx, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sy, sy, a3, 0, 0, 0);
                    }
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        }
    }
    if (do_valu) {
        float v0 = x, v1 = y, v2 = x + 1, v3 = y + 1, v4 = x + 2, v5 = y + 2, v6 = x + 3, v7 = y + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f); v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
                v4 = __builtin_fmaf(v4, 1.0001f, 0.5f); v5 = __builtin_fmaf(v5, 1.0001f, 0.5f); v6 = __builtin_fmaf(v6, 1.0001f, 0.5f); v7 = __builtin_fmaf(v7, 1.0001f, 0.5f);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
}
template <int KIND> void run(const char *name) {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    static const char *what[] = {"8 waves MFMA", "8 waves VALU", "4 MFMA + 4 VALU waves", "4 MFMA waves alone", "4 VALU waves alone"};
    for (int mode = 0; mode < 5; ++mode) {
        k<KIND><<<256, 512>>>(d, 200, mode); hipDeviceSynchronize();
        hipEventRecord(a); k<KIND><<<256, 512>>>(d, 2000, mode); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per wave: 2000 iters x 32 MFMAs (16 for 32x32) or 2000 x 128 FMAs
        printf("%-18s mode %d (%-22s): %.3f ms\n", name, mode, what[mode], ms);
    }
}
int main() {
    run<0>("mfma_f32_16x16x4"); run<1>("mfma_f32_32x32x2"); run<2>("mfma_16x16x32_bf16"); run<3>("mfma_16x16x16_bf16");
    return 0;
}
