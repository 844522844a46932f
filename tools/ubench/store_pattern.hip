// Microbenchmark: HBM write throughput of the reconstruction kernel's output pattern.
// out[B*3][n_vert] fp32; a workgroup (256 threads) writes, per iteration, ROWS=96 row segments of RUN bytes each
// (rows = 32 faces x 3 coords of one face tile, segment = the workgroup's vertex range), then moves to the next face tile.
// Variants: RUN = 512 B / 1 KiB / 2 KiB; unit order as in recon_kernel (XCD-aware) or plain.
// build: hipcc --offload-arch=gfx950 -O3 -o sp tools/ubench/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int RUNF /*floats per run*/>
__global__ __launch_bounds__(256) void k(float *out, int B, int n_vert, int n_groups, int n_split, int per, int n_ftiles, int n_units, int xcd) {
    int unit;
    if (xcd >= 2) {      // split-major, each XCD a contiguous run of vertex groups: whole rows are written at the same time
        const int gpx = (n_groups + 7) / 8;             // vertex groups per XCD
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int split = idx / gpx, tgx = x * gpx + idx % gpx;
        if (split >= n_split || tgx >= n_groups) return;
        unit = tgx * n_split + split;
    } else if (xcd) {
        const int per_xcd = (n_units + 7) / 8;
        unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_units) return;
    } else { unit = blockIdx.x; if (unit >= n_units) return; }
    const int tg = unit / n_split, split = unit - tg * n_split;
    const int ft0 = split * per, ft1 = min(ft0 + per, n_ftiles);
    constexpr int LPR = RUNF / 4;                 // lanes per run
    constexpr int RPI = 256 / LPR;                // rows per store instruction
    const int seg = threadIdx.x % LPR, rsub = threadIdx.x / LPR;
    const int vq = tg * RUNF + 4 * seg;
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int ft = ft0; ft < ft1; ++ft) {
        for (int kk = 0; kk < 96 / RPI; ++kk) {
            const int row = kk * RPI + rsub;
            const int f = ft * 32 + row / 3, c = row % 3;
            if (f >= B) continue;
            if (xcd < 3) { if (vq + 3 < n_vert) *(f32x4 *)(out + ((size_t)f * 3 + c) * n_vert + vq) = v; }
            else {
                // 16-byte aligned body: the run [v0, v0 + RUNF) of row r starts `lead` floats before the next aligned address
                const size_t rowoff = ((size_t)f * 3 + c) * n_vert;
                const int v0 = tg * RUNF;
                const int lead = (int)((4 - ((rowoff + v0) & 3)) & 3);
                float *o = out + rowoff + v0;
                const int vv = lead + 4 * seg;                 // float4 of this lane inside the run
                if (vv + 3 < RUNF) { if (v0 + vv + 3 < n_vert) *(f32x4 *)(o + vv) = v; }
                else {                                         // last lane: tail floats, then the head floats
                    for (int t = vv; t < RUNF; ++t) if (v0 + t < n_vert) o[t] = 1.f;
                    for (int t = 0; t < lead; ++t) o[t] = 2.f;
                }
            }
        }
        __syncthreads();
    }
}
// Line-aligned variant: a workgroup owns 96 vertices per row, shifted per row so that every run is three whole 128-byte
// lines of the flat [3B * n_vert] array (row r starts at flat offset r * n_vert; aligned columns are v = -r*n_vert mod 32 + 32m).
__global__ __launch_bounds__(256) void k_aligned(float *out, int B, int n_vert, int n_groups, int n_split, int per, int n_ftiles, int n_units) {
    const int per_xcd = (n_units + 7) / 8;
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_units) return;
    const int tg = unit / n_split, split = unit - tg * n_split;
    const int ft0 = split * per, ft1 = min(ft0 + per, n_ftiles);
    const int seg = threadIdx.x % 32, rsub = threadIdx.x / 32;
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int ft = ft0; ft < ft1; ++ft) {
        for (int kk = 0; kk < 12; ++kk) {
            const int row = kk * 8 + rsub;
            const long r = (long)ft * 96 + row;
            if (r >= 3L * B || seg >= 24) continue;
            const size_t rowoff = (size_t)r * n_vert;
            const int s = (int)((32 - (rowoff & 31)) & 31);
            const int vq = tg * 96 + s + 4 * seg;
            if (vq + 3 < n_vert) *(f32x4 *)(out + rowoff + vq) = v;
        }
        __syncthreads();
    }
}
// Peeled variant: the same 128-vertex ownership as k<128>, but per row the run is written as its line-aligned interior (three whole
// 128-byte lines: lanes 0-23, float4 each, 8 lanes per line) plus the head and tail floats around it (32 in total: lanes 24-31, four
// scalar stores each).  Same bytes, same partial lines at the ends -- only the packaging of the store instructions differs.
__global__ __launch_bounds__(256) void k_peel(float *out, int B, int n_vert, int n_groups, int n_split, int per, int n_ftiles, int n_units) {
    const int per_xcd = (n_units + 7) / 8;
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_units) return;
    const int tg = unit / n_split, split = unit - tg * n_split;
    const int ft0 = split * per, ft1 = min(ft0 + per, n_ftiles);
    const int seg = threadIdx.x % 32, rsub = threadIdx.x / 32;
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int ft = ft0; ft < ft1; ++ft) {
        for (int kk = 0; kk < 12; ++kk) {
            const int row = kk * 8 + rsub;
            const long r = (long)ft * 96 + row;
            if (r >= 3L * B) continue;
            const size_t rowoff = (size_t)r * n_vert;
            const int v0 = tg * 128;
            const int lead = (int)((32 - ((rowoff + v0) & 31)) & 31);     // floats before the first line boundary inside the run
            float *o = out + rowoff + v0;
            if (seg < 24) {
                const int vv = lead + 4 * seg;
                if (v0 + vv + 3 < n_vert) *(f32x4 *)(o + vv) = v;
            } else {
                // the 32 floats around the body: [0, lead) and [lead + 96, 128)
                for (int t = 0; t < 4; ++t) {
                    int i = 4 * (seg - 24) + t;
                    i = i < lead ? i : i + 96;
                    if (v0 + i < n_vert) o[i] = 1.f;
                }
            }
        }
        __syncthreads();
    }
}
void run_peel(float *d, int B, int nv, int target_wgs) {
    const int n_groups = (nv + 127) / 128, n_ftiles = (B + 31) / 32;
    int n_split = (target_wgs + n_groups - 1) / n_groups; n_split = n_split < 1 ? 1 : (n_split > n_ftiles ? n_ftiles : n_split);
    const int per = (n_ftiles + n_split - 1) / n_split; n_split = (n_ftiles + per - 1) / per;
    const int n_units = n_groups * n_split, grid = ((n_units + 7) / 8) * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_peel<<<grid, 256>>>(d, B, nv, n_groups, n_split, per, n_ftiles, n_units); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) k_peel<<<grid, 256>>>(d, B, nv, n_groups, n_split, per, n_ftiles, n_units);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("peeled 512 B runs (3 whole lines + head/tail)  wgs=%5d (split %2d, %d ftiles each) : %.1f us  %.2f TB/s\n", n_units, n_split, per, ms * 1e3,
           (double)B * 3 * nv * 4 / ms / 1e9);
}
void run_aligned(float *d, int B, int nv, int target_wgs) {
    const int n_groups = (nv + 95) / 96, n_ftiles = (B + 31) / 32;
    int n_split = (target_wgs + n_groups - 1) / n_groups; n_split = n_split < 1 ? 1 : (n_split > n_ftiles ? n_ftiles : n_split);
    const int per = (n_ftiles + n_split - 1) / n_split; n_split = (n_ftiles + per - 1) / per;
    const int n_units = n_groups * n_split, grid = ((n_units + 7) / 8) * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_aligned<<<grid, 256>>>(d, B, nv, n_groups, n_split, per, n_ftiles, n_units); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) k_aligned<<<grid, 256>>>(d, B, nv, n_groups, n_split, per, n_ftiles, n_units);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("line-aligned 384 B runs  wgs=%5d (split %2d, %d ftiles each) : %.1f us  %.2f TB/s\n", n_units, n_split, per, ms * 1e3,
           (double)B * 3 * nv * 4 / ms / 1e9);
}
template <int RUNF> void run(float *d, int B, int nv, int target_wgs, int xcd) {
    const int n_groups = (nv + RUNF - 1) / RUNF, n_ftiles = (B + 31) / 32;
    int n_split = (target_wgs + n_groups - 1) / n_groups; n_split = n_split < 1 ? 1 : (n_split > n_ftiles ? n_ftiles : n_split);
    const int per = (n_ftiles + n_split - 1) / n_split; n_split = (n_ftiles + per - 1) / per;
    const int n_units = n_groups * n_split, grid = xcd >= 2 ? ((n_groups + 7) / 8) * 8 * n_split : ((n_units + 7) / 8) * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<RUNF><<<grid, 256>>>(d, B, nv, n_groups, n_split, per, n_ftiles, n_units, xcd); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) k<RUNF><<<grid, 256>>>(d, B, nv, n_groups, n_split, per, n_ftiles, n_units, xcd);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("run=%4d B  wgs=%5d (split %2d, %d ftiles each)  xcd_order=%d : %.1f us  %.2f TB/s\n", RUNF * 4, n_units, n_split, per, xcd, ms * 1e3,
           (double)B * 3 * nv * 4 / ms / 1e9);
}
int main(int argc, char **argv) {
    const int B = 1024, nv = argc > 1 ? atoi(argv[1]) : 53215;
    printf("n_vert = %d\n", nv);
    float *d; hipMalloc(&d, (size_t)B * 3 * nv * 4 + 4096);
    for (int xcd = 0; xcd < 4; ++xcd) {
        run<128>(d, B, nv, 3072, xcd); run<256>(d, B, nv, 3072, xcd); run<512>(d, B, nv, 3072, xcd); run<1024>(d, B, nv, 3072, xcd);
    }
    run_aligned(d, B, nv, 3072); run_aligned(d, B, nv, 4400); run_aligned(d, B, nv, 1600);
    run_peel(d, B, nv, 3072); run_peel(d, B, nv, 1664); run<128>(d, B, nv, 1664, 1);
    run<128>(d, B, nv, 512, 2); run<128>(d, B, nv, 1600, 2); run<128>(d, B, nv, 100000, 2); run<256>(d, B, nv, 100000, 2);
    return 0;
}
