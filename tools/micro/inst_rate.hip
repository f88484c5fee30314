// Issue rate of a few vector instructions on one SIMD: one wave per SIMD (256 CUs x 4), 8 independent chains,
// N iterations of 8 instructions each; cycles per wave instruction = time * clock / (N * 8).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/inst_rate.hip -o tools/micro/inst_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k_rate(float *out, int n, float seed)
{
    float f[8];
    double d[8];
    int sc[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    unsigned long long mask = 0x5555555555555555ull + (unsigned long long)n;
    for (int i = 0; i < 8; i++) {
        f[i] = seed + i + threadIdx.x;
        d[i] = seed * 3 + i;
    }
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
            if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (MODE == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (MODE == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (MODE == 5) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (MODE == 7) asm volatile("v_floor_f32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 8) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (MODE == 10) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 11) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(f[(i + 1) & 7]) : "vcc");
            if (MODE == 14) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[i]));
            // selects: the mask in an SGPR pair set once (no VCC traffic), and the compare + select pair a ternary compiles to
            if (MODE == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "s"(mask));
            if (MODE == 16) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(f[(i + 1) & 7]) : "vcc");
            if (MODE == 17) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 19) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 20) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (MODE == 21) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 22) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            if (MODE == 23) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (MODE == 24) asm volatile("v_cmp_gt_f32_e64 %1, %0, %0" : "+v"(f[i]), "=s"(mask));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++)
        s += f[i] + (float)d[i] + sc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
int run(const char *name, float *out, int waves_per_simd)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int n = 20000, blocks = 256 * 4 * waves_per_simd;
    k_rate<MODE><<<blocks, 64>>>(out, 100, 1.f);
    CHECK(hipEventRecord(e0));
    k_rate<MODE><<<blocks, 64>>>(out, n, 1.f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // cycles per wave instruction per SIMD at 2.4 GHz (the clock is not read: compare rows with each other)
    printf("%-20s waves/SIMD %d: %.3f ms  %.2f cycles per wave instruction at 2.4 GHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / ((double)n * 8 * waves_per_simd));
    return 0;
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 256 * 4 * 4 * 64 * sizeof(float)));
    for (int w = 1; w <= 4; w++) {
        run<0>("v_add_f32", out, w);
        run<1>("v_add_f64", out, w);
        run<9>("v_mul_f64", out, w);
        run<4>("v_fma_f64", out, w);
        run<2>("v_cvt_f64_f32", out, w);
        run<3>("v_cvt_f32_f64", out, w);
        run<5>("v_rcp_f64", out, w);
        run<6>("v_pk_fma_f32", out, w);
        run<7>("v_floor_f32", out, w);
        run<8>("v_cvt_i32_f32", out, w);
        run<10>("v_lshlrev_b32", out, w);
        run<11>("v_mad_u32_u24", out, w);
        run<12>("v_add_f32 (2 regs)", out, w);
        run<13>("v_cndmask_b32", out, w);
        run<14>("s_add_u32", out, w);
        run<15>("v_cndmask (sgpr mask)", out, w);
        run<16>("v_cmp + v_cndmask vcc (x2)", out, w);
        run<17>("v_bfi_b32", out, w);
        run<18>("v_max_f32", out, w);
        run<19>("v_med3_i32", out, w);
        run<20>("v_pk_mul_f32", out, w);
        run<21>("v_mul_hi_u32", out, w);
        run<22>("v_mul_lo_u32", out, w);
        run<23>("v_lshl_add_u64", out, w);
        run<24>("v_cmp_gt_f32 -> sgpr", out, w);
    }
    return 0;
}
