// Microbenchmark: what HBM rate does a march of row pieces reach, by piece length?
// Workgroups of 128 lanes walk down a strip of rows reading 10 planes per row (as the producers of
// k_flow_iter_pc read R0/R1) and writing 8 bytes per pixel; PLANAR: ten 512-byte pieces per row, one per
// plane; INTERLEAVED: the strip's ten planes of a row are adjacent (one 5 KB piece per row).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/march_bench tools/micro/march_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int PF>
__global__ void __launch_bounds__(128) k_march(const float *__restrict__ R, float2 *__restrict__ out, int W, int H, int seg)
{
    extern __shared__ char pad[];
    const int strip = blockIdx.x, sg = blockIdx.y, pair = blockIdx.z, lane = threadIdx.x;
    const size_t N = (size_t)W * H;
    const float *base = R + (size_t)pair * 10 * N;
    const int y0 = sg * seg, y1 = min(H, y0 + seg);
    float v[PF][10];
    auto issue = [&](int y, float *d) {
        if (y >= y1) y = y1 - 1;
#pragma unroll
        for (int c = 0; c < 10; c++) {
            size_t o;
            if (MODE == 0) o = (size_t)c * N + (size_t)y * W + strip * 128 + lane;                 // planar
            else if (MODE == 1) o = (((size_t)strip * H + y) * 10 + c) * 128 + lane;               // strip-major, planes adjacent per row
            else o = ((size_t)y * 10 + c) * W + strip * 128 + lane;                               // row-major, planes adjacent per frame row
            d[c] = base[o];
        }
    };
#pragma unroll
    for (int p = 0; p < PF; p++)
        issue(y0 + p, v[p]);
    for (int y = y0; y < y1; y += PF) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            float acc = 0;
#pragma unroll
            for (int c = 0; c < 10; c++)
                acc += v[p][c];
            issue(y + p + PF, v[p]);
            if (y + p < y1)
                out[(size_t)pair * N + (size_t)(y + p) * W + strip * 128 + lane] = make_float2(acc, acc);
        }
    }
    if (pad[0] == 77 && lane == 1000) out[0].x = 1;
}

template <int MODE, int PF>
static int run(const char *name, const float *R, float2 *out, int W, int H, int pairs, int seg, int lds)
{
    dim3 grid(W / 128, (H + seg - 1) / seg, pairs);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k_march<MODE, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 2; i++)
        hipLaunchKernelGGL((k_march<MODE, PF>), grid, dim3(128), lds, 0, R, out, W, H, seg);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((k_march<MODE, PF>), grid, dim3(128), lds, 0, R, out, W, H, seg);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double bytes = (double)pairs * W * H * 48.0;
    printf("%-28s lds=%6d  %.3f ms  %.2f TB/s\n", name, lds, ms, bytes / ms / 1e9);
    return 0;
}

int main(int argc, char **argv)
{
    const int W = 3840, H = 2160, pairs = 16, seg = 270;
    const size_t N = (size_t)W * H;
    float *R;
    float2 *out;
    CK(hipMalloc(&R, pairs * 10 * N * 4));
    CK(hipMalloc(&out, pairs * N * 8));
    CK(hipMemset(R, 0, pairs * 10 * N * 4));
    for (int lds : {51200, 25600, 12800, 0}) {
        if (run<0, 2>("planar PF2", R, out, W, H, pairs, seg, lds)) return 1;
        if (run<1, 2>("strip-interleaved PF2", R, out, W, H, pairs, seg, lds)) return 1;
        if (run<2, 2>("row-interleaved PF2", R, out, W, H, pairs, seg, lds)) return 1;
        if (run<0, 4>("planar PF4", R, out, W, H, pairs, seg, lds)) return 1;
        if (run<1, 4>("strip-interleaved PF4", R, out, W, H, pairs, seg, lds)) return 1;
        if (run<2, 4>("row-interleaved PF4", R, out, W, H, pairs, seg, lds)) return 1;
    }
    return 0;
}
