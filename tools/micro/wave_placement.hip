// Which SIMD does wave i of a 256-thread workgroup land on?  Same shape as k_flow_iter_pc (256 threads, 53.8 KB of
// LDS: three workgroups per CU), every wave records HW_REG_HW_ID and spins a little so the chip fills up.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/wave_placement.hip -o tools/micro/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_place(unsigned *out, int spin)
{
    __shared__ float pad[53760 / 4];
    const int wave = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID, bits 0..3? (size-1 << 11 | offset << 6 | id)
    hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++)
        a = a * 1.0001f + 0.5f;
    pad[threadIdx.x] = a;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + wave] = hw;
    if (pad[(threadIdx.x + 1) & 255] == 12345.f)
        out[0] = 0;
}

int main()
{
    const int blocks = 2240;
    unsigned *out;
    CHECK(hipMalloc(&out, blocks * 4 * sizeof(unsigned)));
    k_place<<<blocks, 256>>>(out, 200000);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    int hist[4][4] = {};
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < 4; w++)
            hist[w][(h[b * 4 + w] >> 4) & 3]++;
    for (int w = 0; w < 4; w++)
        printf("wave %d: SIMD 0..3 = %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 12; b++)
        printf("block %d: hw_id %08x %08x %08x %08x  simd %u %u %u %u  cu %u se %u\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2],
               h[b * 4 + 3], (h[b * 4] >> 4) & 3, (h[b * 4 + 1] >> 4) & 3, (h[b * 4 + 2] >> 4) & 3, (h[b * 4 + 3] >> 4) & 3,
               (h[b * 4] >> 8) & 15, (h[b * 4] >> 13) & 7);
    return 0;
}
