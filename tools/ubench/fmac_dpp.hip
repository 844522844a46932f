// GPU box: does folding the lane shift into the multiply-add (v_fmac_f32_dpp, inline asm: the compiler never forms it) pay?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fmac_dpp tools/ubench/fmac_dpp.hip && /tmp/fmac_dpp
// One "group" = one shifted source feeding three accumulators (the three filter rows of a depthwise column tap).
//   mode 0: v_mov_b32_dpp + 3 v_fmac_f32          (what the kernels do)
//   mode 1: 3 v_fmac_f32_dpp
//   mode 2: s_nop 1 + 3 v_fmac_f32_dpp            (the DPP read-after-VALU-write hazard covered by hand)
//   mode 3: 3 v_fmac_f32                          (no shift at all: the floor)
// s_memtime ticks (100 MHz) per group and wave; the check value proves the shifted product is the same.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(768) void k(float *o, const float *a, int iters, unsigned long long *cyc) {
    float e[8], acc[8][3];
    for (int i = 0; i < 8; ++i) { e[i] = a[threadIdx.x + 64 * i]; acc[i][0] = acc[i][1] = acc[i][2] = 0.f; }
    const float w0 = a[1], w1 = a[2], w2 = a[3];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {
                const float s = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, e[i]), 0x111, 0xf, 0xf, true));
                acc[i][0] = __builtin_fmaf(s, w0, acc[i][0]);
                acc[i][1] = __builtin_fmaf(s, w1, acc[i][1]);
                acc[i][2] = __builtin_fmaf(s, w2, acc[i][2]);
            } else if (MODE == 1 || MODE == 2) {
                if (MODE == 2) asm volatile("s_nop 1");
                asm volatile("v_fmac_f32_dpp %0, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %1, %3, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %2, %3, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]) : "v"(e[i]), "v"(w0), "v"(w1), "v"(w2));
            } else {
                acc[i][0] = __builtin_fmaf(e[i], w0, acc[i][0]);
                acc[i][1] = __builtin_fmaf(e[i], w1, acc[i][1]);
                acc[i][2] = __builtin_fmaf(e[i], w2, acc[i][2]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(e[i]));      // (keeps the shifts inside the loop without an extra instruction)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + 2.f * acc[i][1] + 3.f * acc[i][2];
    o[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *o, *a; unsigned long long *c;
    hipMalloc(&o, 256 * 768 * 4); hipMalloc(&a, 4096 * 4); hipMalloc(&c, 256 * 8);
    std::vector<float> ha(4096);
    for (int i = 0; i < 4096; ++i) ha[i] = (float)((i * 37) % 101) * 0.01f;
    hipMemcpy(a, ha.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    const char *names[4] = {"mov_dpp + 3 fmac", "3 fmac_dpp", "s_nop 1 + 3 fmac_dpp", "3 fmac (no shift)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps = 1; wps <= 3; ++wps) {
            const int threads = 256 * wps;
            auto launch = [&] {
                switch (mode) {
                    case 0: k<0><<<256, threads>>>(o, a, iters, c); break;
                    case 1: k<1><<<256, threads>>>(o, a, iters, c); break;
                    case 2: k<2><<<256, threads>>>(o, a, iters, c); break;
                    default: k<3><<<256, threads>>>(o, a, iters, c); break;
                }
            };
            launch(); launch();
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256);
            std::vector<float> ho(768);
            hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(ho.data(), o, 768 * 4, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            double chk = 0; for (int i = 0; i < 64; ++i) chk += ho[i];
            printf("%-22s %d waves/SIMD: %.3f ticks per group per wave, %.3f per SIMD   (check %.4e)\n", names[mode], wps, s / 256 / iters / 8,
                   s / 256 / iters / 8 / wps, chk);
        }
    return 0;
}
