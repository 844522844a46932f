// Microbenchmark (round 6): HBM write throughput of a convolution epilogue's store pattern.  out[M][N] fp32-sized elements (N = 1024, M = 32768: layer 3's
// conv3 at B = 512); a workgroup of 8 waves (4 x 2) owns 128 pixels x 128 channels, workgroups mapped to XCDs as conv_lt_kernel does.
//   pattern 0: the pair-format epilogue -- per instruction lane (r16, g) writes 16 bytes: 4 lanes = one 64-byte HALF line (high pieces), the low pieces of the
//              same 32 channels one instruction later
//   pattern 1: the same bytes through an LDS transpose -- per instruction 32 lanes write one pixel's whole 512-byte run (4 lines)
//   pattern 2: fp32 epilogue (16-byte pieces at a 32-byte stride, the partner tile fills the gaps)
// build: hipcc --offload-arch=gfx950 -O3 -o tile_store tools/ubench/tile_store.hip ; run: ./tile_store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512) void k(float *out, int M, int N, int n_tiles, int m_tiles) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt_idx = q % n_tiles, mt_idx = (q / n_tiles) * 8 + xcd;
    if (mt_idx >= m_tiles) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int m0 = mt_idx * 128, n0 = nt_idx * 128;
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    if (PAT == 1) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {           // 64 instructions per workgroup, 8 per wave: 2 pixels x 512 B each
            const int pix = (wave * 8 + k8) * 2 + (lane >> 5);
            *(f32x4 *)(out + (size_t)(m0 + pix) * N + n0 + (lane & 31) * 4) = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + wm * 32 + j * 16 + r16;
                const int col = n0 + wn * 64 + 32 * (i >> 1) + (PAT == 0 ? 4 * g + 16 * (i & 1) : 8 * g + 4 * (i & 1));
                *(f32x4 *)(out + (size_t)m * N + col) = v;
            }
    }
}
int main() {
    const int M = 32768, N = 1024, n_tiles = N / 128, m_tiles = M / 128, grid = ((m_tiles + 7) / 8) * n_tiles * 8;
    float *out; hipMalloc(&out, (size_t)M * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 3; ++pat)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) {
                if (pat == 0) k<0><<<grid, 512>>>(out, M, N, n_tiles, m_tiles);
                else if (pat == 1) k<1><<<grid, 512>>>(out, M, N, n_tiles, m_tiles);
                else k<2><<<grid, 512>>>(out, M, N, n_tiles, m_tiles);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("pattern %d: %.1f us  %.2f TB/s\n", pat, ms * 1e3, (double)M * N * 4 / ms / 1e9);
        }
    return 0;
}
