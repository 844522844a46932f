// GPU box: cost of the cross-lane shifts the row-marching early blocks use for the horizontal depthwise taps.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_shift tools/ubench/dpp_shift.hip && /tmp/dpp_shift
// Per variant: cycles per instruction-pair (shift + dependent v_fmac) per wave, with 1 / 2 / 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(768) void k(float *o, const float *a, int iters, unsigned long long *cyc) {
    float e[8], acc[8];
    for (int i = 0; i < 8; ++i) { e[i] = a[threadIdx.x + 64 * i]; acc[i] = 0.f; }
    const float w = a[1];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s;
            if (MODE == 0) s = e[i];                                                                                                  // plain fmac
            else if (MODE == 1) s = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, e[i]), 0x111, 0xf, 0xf, true));   // row_shr:1
            else if (MODE == 2) s = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, e[i]), 0x138, 0xf, 0xf, true));   // wave_shr:1
            else if (MODE == 3) s = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, e[i]), 0x130, 0xf, 0xf, true));   // wave_shl:1
            else s = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((int)threadIdx.x - 1) * 4, __builtin_bit_cast(int, e[i])));        // ds_bpermute
            acc[i] = __builtin_fmaf(s, w, acc[i]);
            e[i] += 1.0f;            // new value every round (keeps the shift from being hoisted)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *o, *a; unsigned long long *c;
    hipMalloc(&o, 256 * 768 * 4); hipMalloc(&a, 4096 * 4); hipMalloc(&c, 256 * 8);
    hipMemset(a, 0, 4096 * 4);
    const int iters = 2000;
    const char *names[5] = {"plain fmac+add", "row_shr:1 mov + fmac+add", "wave_shr:1 mov + fmac+add", "wave_shl:1 mov + fmac+add", "ds_bpermute + fmac+add"};
    for (int mode = 0; mode < 5; ++mode)
        for (int wps = 1; wps <= 3; ++wps) {
            const int threads = 256 * wps;
            auto launch = [&] {
                switch (mode) {
                    case 0: k<0><<<256, threads>>>(o, a, iters, c); break;
                    case 1: k<1><<<256, threads>>>(o, a, iters, c); break;
                    case 2: k<2><<<256, threads>>>(o, a, iters, c); break;
                    case 3: k<3><<<256, threads>>>(o, a, iters, c); break;
                    default: k<4><<<256, threads>>>(o, a, iters, c); break;
                }
            };
            launch(); launch();
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("%-28s %d waves/SIMD: %.2f cycles per (element step) per wave, %.2f per SIMD\n", names[mode], wps, s / 256 / iters / 8,
                   s / 256 / iters / 8 / wps);
        }
    return 0;
}
