// A2 of the Farneback path, the polynomial expansion, and the forms fused with A1 for the full-resolution and the half-size
// level (optflowgf.cpp FarnebackPrepareGaussian / FarnebackPolyExp, cv.py:479-490).  A1 by itself:
// fb_level_image.hip.
#include "fb_common.h"

namespace {

// ---------------------------------------------------------------------------------
// A2: FarnebackPolyExp.  Tile 64x16 outputs; LDS holds the image tile with an
// n-pixel halo, then the three vertical-pass planes; the horizontal pass runs in
// double.  Clamped loads reproduce OpenCV's row clamping (vertical) and its
// replication of the edge triple (horizontal).
// ---------------------------------------------------------------------------------
#ifndef TF_PX_TH
#define TF_PX_TH 16
#endif
constexpr int PX_TW = 64, PX_TH = TF_PX_TH;

__global__ void k_polyexp(const float *__restrict__ img, float *__restrict__ R, int Wk, int Hk, PolyConst pc)
{
    extern __shared__ float s_mem[];
    const int n = pc.n;
    const int LW = PX_TW + 2 * n;     // columns incl. halo
    const int LH = PX_TH + 2 * n;     // rows incl. halo
    float *sI = s_mem;                // [LH][LW]
    float *sT = s_mem + LH * LW;      // [3][PX_TH][LW]
    const int pi = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    const float *src = img + (size_t)pi * Nk;
    const int x0 = blockIdx.x * PX_TW, y0 = blockIdx.y * PX_TH;
    for (int idx = threadIdx.x; idx < LH * LW; idx += blockDim.x) {
        int ry = idx / LW, cx = idx % LW;
        int y = clampi(y0 - n + ry, 0, Hk - 1), x = clampi(x0 - n + cx, 0, Wk - 1);
        sI[idx] = src[(size_t)y * Wk + x];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < PX_TH * LW; idx += blockDim.x) {
        int ty = idx / LW, cx = idx % LW;
        int y = y0 + ty;
        // rows y-k / y+k are clamped to the image: compute their LDS row from the clamped index
        const float c = sI[(ty + n) * LW + cx];
        float t0 = c * pc.g[0], t1 = 0.f, t2 = 0.f;
        for (int k = 1; k <= n; k++) {
            int ya = max(y - k, 0), yb = min(y + k, Hk - 1);
            // LDS row of image row yy is (yy - (y0 - n)); clamped loads make out-of-image halo
            // rows equal to the edge row, so indexing by ty works when y itself is in range
            float a = sI[(ya - y0 + n) * LW + cx];
            float b = sI[(yb - y0 + n) * LW + cx];
            float p = a + b;
            t0 = t0 + pc.g[k] * p;
            t1 = t1 + pc.xg[k] * (b - a);
            t2 = t2 + pc.xxg[k] * p;
        }
        if (y >= Hk) {
            t0 = t1 = t2 = 0.f;
        }
        sT[(0 * PX_TH + ty) * LW + cx] = t0;
        sT[(1 * PX_TH + ty) * LW + cx] = t1;
        sT[(2 * PX_TH + ty) * LW + cx] = t2;
    }
    __syncthreads();
    float *dst = R + (size_t)pi * 5 * Nk;
    for (int idx = threadIdx.x; idx < PX_TH * PX_TW; idx += blockDim.x) {
        int ty = idx / PX_TW, cx = idx % PX_TW;
        int x = x0 + cx, y = y0 + ty;
        if (x >= Wk || y >= Hk)
            continue;
        const float *T0 = sT + (0 * PX_TH + ty) * LW + cx + n;
        const float *T1 = sT + (1 * PX_TH + ty) * LW + cx + n;
        const float *T2 = sT + (2 * PX_TH + ty) * LW + cx + n;
        float g0 = pc.g[0];
        double b1 = T0[0] * g0, b2 = 0, b3 = T1[0] * g0, b4 = 0, b5 = T2[0] * g0, b6 = 0;
        for (int k = 1; k <= n; k++) {
            double tg = T0[k] + T0[-k];
            g0 = pc.g[k];
            b1 = __builtin_fma(tg, (double)g0, b1); // exact products of float values: fused == separate
            b4 = __builtin_fma(tg, (double)pc.xxg[k], b4);
            b2 += (T0[k] - T0[-k]) * pc.xg[k];
            b3 += (T1[k] + T1[-k]) * g0;
            b6 += (T1[k] - T1[-k]) * pc.xg[k];
            b5 += (T2[k] + T2[-k]) * g0;
        }
        size_t o = (size_t)y * Wk + x;
        const float v[5] = {(float)(b3 * pc.ig11), (float)(b2 * pc.ig11), (float)(b1 * pc.ig03 + b5 * pc.ig33),
                            (float)(b1 * pc.ig03 + b4 * pc.ig33), (float)(b6 * pc.ig55)};
        r_store_px(dst, Nk, o, v);
    }
}

// A2, register-blocked form for a compile-time poly_n.  Same arithmetic, statement for
// statement; what changes is how often LDS is read: the vertical pass slides a window of
// 4+2N rows down a column in registers (4 outputs per item), the horizontal pass computes two
// adjacent outputs from one window of 2+2N triples.  ~25 DS operations per pixel instead of ~62.
typedef float float2w __attribute__((ext_vector_type(2), aligned(4)));

// FarnebackPolyExp's vertical pass for four consecutive rows of two adjacent columns: v[j] holds
// rows y0-N+j of the column pair; results go to the three planes at rows 0..3 (row stride LW).
template <int N>
__device__ __forceinline__ void polyexp_vertical4(const f32x2 (&v)[4 + 2 * N], const PolyConst &pc, float *T0, float *T1,
                                                  float *T2, int LW)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x2 c = v[q + N];
        f32x2 t0 = c * pc.g[0], t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
        for (int k = 1; k <= N; k++) {
            const f32x2 a = v[q + N - k], b = v[q + N + k]; // rows y-k and y+k (clamped when staged)
            const f32x2 p = a + b;
            t0 = t0 + pc.g[k] * p;
            t1 = t1 + pc.xg[k] * (b - a);
            t2 = t2 + pc.xxg[k] * p;
        }
        *reinterpret_cast<f32x2 *>(T0 + q * LW) = t0;
        *reinterpret_cast<f32x2 *>(T1 + q * LW) = t1;
        *reinterpret_cast<f32x2 *>(T2 + q * LW) = t2;
    }
}

template <int N>
__global__ void __launch_bounds__(256)
k_polyexp_t(const float *__restrict__ img, float *__restrict__ R, int Wk, int Hk, PolyConst pc)
{
    constexpr int TW = PX_TW, TH = PX_TH, LW = TW + 2 * N, LH = TH + 2 * N;
    __shared__ float sI[LH * LW];
    __shared__ float sT[3][TH][LW];
    const int pi = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    const float *src = img + (size_t)pi * Nk;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    for (int idx = threadIdx.x; idx < LH * LW; idx += 256) {
        int ry = idx / LW, cx = idx - ry * LW; // LW is a compile-time constant
        int y = clampi(y0 - N + ry, 0, Hk - 1), x = clampi(x0 - N + cx, 0, Wk - 1);
        sI[idx] = src[(size_t)y * Wk + x];
    }
    __syncthreads();
    // vertical pass (float): item = (pair of columns, group of 4 rows); the two columns ride in
    // the halves of packed fp32 operations
    static_assert(LW % 2 == 0, "column pairs");
    for (int idx = threadIdx.x; idx < (TH / 4) * (LW / 2); idx += 256) {
        const int g = idx / (LW / 2), cx = 2 * (idx - g * (LW / 2));
        f32x2 v[4 + 2 * N];
#pragma unroll
        for (int j = 0; j < 4 + 2 * N; j++)
            v[j] = *reinterpret_cast<const f32x2 *>(&sI[(4 * g + j) * LW + cx]);
        polyexp_vertical4<N>(v, pc, &sT[0][4 * g][cx], &sT[1][4 * g][cx], &sT[2][4 * g][cx], LW);
    }
    __syncthreads();
    // horizontal pass (double): item = (row, pair of columns)
    float *dst = R + (size_t)pi * 5 * Nk;
    for (int idx = threadIdx.x; idx < TH * (TW / 2); idx += 256) {
        const int ty = idx / (TW / 2), cp = idx - ty * (TW / 2);
        const int x = x0 + 2 * cp, y = y0 + ty;
        if (x >= Wk || y >= Hk)
            continue;
        float w0[2 + 2 * N], w1[2 + 2 * N], w2[2 + 2 * N]; // triples at columns x-N .. x+1+N
#pragma unroll
        for (int j = 0; j < 2 + 2 * N; j++) {
            w0[j] = sT[0][ty][2 * cp + j];
            w1[j] = sT[1][ty][2 * cp + j];
            w2[j] = sT[2][ty][2 * cp + j];
        }
        float out[2][5];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float *T0 = w0 + q + N, *T1 = w1 + q + N, *T2 = w2 + q + N;
            float g0 = pc.g[0];
            double b1 = T0[0] * g0, b2 = 0, b3 = T1[0] * g0, b4 = 0, b5 = T2[0] * g0, b6 = 0;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                double tg = T0[k] + T0[-k];
                g0 = pc.g[k];
                b1 = __builtin_fma(tg, (double)g0, b1); // exact products of float values: fused == separate
                b4 = __builtin_fma(tg, (double)pc.xxg[k], b4);
                b2 += (T0[k] - T0[-k]) * pc.xg[k];
                b3 += (T1[k] + T1[-k]) * g0;
                b6 += (T1[k] - T1[-k]) * pc.xg[k];
                b5 += (T2[k] + T2[-k]) * g0;
            }
            out[q][0] = (float)(b3 * pc.ig11);
            out[q][1] = (float)(b2 * pc.ig11);
            out[q][2] = (float)(b1 * pc.ig03 + b5 * pc.ig33);
            out[q][3] = (float)(b1 * pc.ig03 + b4 * pc.ig33);
            out[q][4] = (float)(b6 * pc.ig55);
        }
        const size_t o = (size_t)y * Wk + x;
        if (x + 1 < Wk)
            r_store_px2(dst, Nk, o, out[0], out[1]);
        else
            r_store_px(dst, Nk, o, out[0]);
    }
}

// The two passes of FarnebackPolyExp over a blurred level tile held in LDS (sI, indexed by real level
// coordinates relative to (xr0, yr0); virtual coordinates outside the level clamp, as OpenCV
// replicates edge rows/columns of the level image): shared by the fused level-0 and level-1 kernels.
// W, H: the LEVEL's size.  Call with sI complete and the workgroup synchronised.
template <int N, int TW, int TH>
__device__ __forceinline__ void tile_expansion(const float *sI, float (*sT)[TH][TW + 2 * N], int x0, int y0, int xr0,
                                               int yr0, int W, int H, const PolyConst &pc, float *dst, size_t Nk)
{
    constexpr int LW = TW + 2 * N;
    // polynomial expansion, vertical pass: virtual row y0-N+j reads real row clamp(...) - yr0
    // (an interior tile reads rows/columns 4g+j / cx directly; a border tile clamps them)
    const bool interior = x0 - N >= 0 && x0 + TW - 1 + N <= W - 1 && y0 - N >= 0 && y0 + TH - 1 + N <= H - 1;
    for (int idx = threadIdx.x; idx < (TH / 4) * (LW / 2); idx += 256) {
        const int g = idx / (LW / 2), cx = 2 * (idx - g * (LW / 2));
        f32x2 v[4 + 2 * N];
        if (interior) {
#pragma unroll
            for (int j = 0; j < 4 + 2 * N; j++)
                v[j] = *reinterpret_cast<const f32x2 *>(&sI[(4 * g + j) * LW + cx]);
        } else {
            const int xa = clampi(x0 - N + cx, 0, W - 1) - xr0, xb = clampi(x0 - N + cx + 1, 0, W - 1) - xr0;
#pragma unroll
            for (int j = 0; j < 4 + 2 * N; j++) {
                const float *row = sI + (clampi(y0 - N + 4 * g + j, 0, H - 1) - yr0) * LW;
                v[j] = f32x2{row[xa], row[xb]};
            }
        }
        polyexp_vertical4<N>(v, pc, &sT[0][4 * g][cx], &sT[1][4 * g][cx], &sT[2][4 * g][cx], LW);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TH * (TW / 2); idx += 256) {
        const int ty = idx / (TW / 2), cp = idx - ty * (TW / 2);
        const int x = x0 + 2 * cp, y = y0 + ty;
        if (x >= W || y >= H)
            continue;
        float w0[2 + 2 * N], w1[2 + 2 * N], w2[2 + 2 * N];
#pragma unroll
        for (int j = 0; j < 2 + 2 * N; j++) {
            w0[j] = sT[0][ty][2 * cp + j];
            w1[j] = sT[1][ty][2 * cp + j];
            w2[j] = sT[2][ty][2 * cp + j];
        }
        float out[2][5];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float *T0 = w0 + q + N, *T1 = w1 + q + N, *T2 = w2 + q + N;
            float g0 = pc.g[0];
            double b1 = T0[0] * g0, b2 = 0, b3 = T1[0] * g0, b4 = 0, b5 = T2[0] * g0, b6 = 0;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                double tg = T0[k] + T0[-k];
                g0 = pc.g[k];
                b1 = __builtin_fma(tg, (double)g0, b1); // exact products of float values: fused == separate
                b4 = __builtin_fma(tg, (double)pc.xxg[k], b4);
                b2 += (T0[k] - T0[-k]) * pc.xg[k];
                b3 += (T1[k] + T1[-k]) * g0;
                b6 += (T1[k] - T1[-k]) * pc.xg[k];
                b5 += (T2[k] + T2[-k]) * g0;
            }
            out[q][0] = (float)(b3 * pc.ig11);
            out[q][1] = (float)(b2 * pc.ig11);
            out[q][2] = (float)(b1 * pc.ig03 + b5 * pc.ig33);
            out[q][3] = (float)(b1 * pc.ig03 + b4 * pc.ig33);
            out[q][4] = (float)(b6 * pc.ig55);
        }
        const size_t o = (size_t)y * W + x;
        if (x + 1 < W)
            r_store_px2(dst, Nk, o, out[0], out[1]);
        else
            r_store_px(dst, Nk, o, out[0]);
    }
}

// rows of a tile of the fused expansion kernels (4K x 33 frames, round 3: level 0 with 12 / 16 / 20 rows 2.43 / 2.10 / 2.18 ms,
// level 1 with 16 / 20 / 24 / 32 rows 1.31 (before its 2 x 2 blocks) / 0.87 / 0.95 / 1.05 ms; 4K x 129 frames, round 5: level 0
// with 12 / 16 / 20 / 24 rows 9.55 / 8.79 - 8.88 / 8.81 / 8.76 ms, level 1 with 16 / 20 / 24 / 28 / 32 rows 4.20 / 3.96 - 4.02 /
// 3.78 - 3.80 / 4.37 / 4.00 ms)
#ifndef TF_EXP_TH0
#define TF_EXP_TH0 16
#endif
#ifndef TF_EXP_TH1
#define TF_EXP_TH1 24
#endif
// A1+A2 fused for the full-resolution level (resize is a copy, the blur has 3 taps): the level
// image never leaves the CU.  Stages the u8 region by REAL image coordinates (REFLECT_101 ring of
// one pixel), blurs it into LDS exactly as k_level_image does (row pass, then column pass), then
// runs k_polyexp_t's two passes, reading the blurred tile with CLAMPED coordinates (OpenCV's
// polynomial expansion replicates edge rows/columns of the already blurred image).
template <int N>
__global__ void __launch_bounds__(256)
k_level0_polyexp_t(const uint8_t *__restrict__ frames, const int2 *__restrict__ pairs, float *__restrict__ R, int W,
                   int H, float kc, float k1, PolyConst pc)
{
    constexpr int TW = 64, TH = TF_EXP_TH0, LW = TW + 2 * N, LH = TH + 2 * N; // blurred tile (virtual extent)
    constexpr int SW = ((LW + 2 + 3 + 3) + 3) & ~3, SH = LH + 2; // staged bytes: one more pixel all round, dword slack
    // LDS: the staged bytes and the row-pass values are dead once the blurred tile exists, so the
    // three planes of the expansion's vertical pass reuse their space
    constexpr int RS = (LW + 3) & ~3; // row stride of the row-pass values: whole groups of four columns
    static_assert(SW >= RS + 4 + 4 && (SH * SW) % 16 == 0, "a group's three dwords stay inside its staged row; sRow 16-byte aligned");
    constexpr int BYTES_A = SH * SW + SH * RS * 4, BYTES_T = 3 * TH * LW * 4;
    constexpr int BYTES_U = ((BYTES_A > BYTES_T ? BYTES_A : BYTES_T) + 15) & ~15;
    __shared__ __attribute__((aligned(16))) uint8_t s_u[BYTES_U];
    __shared__ float sI[LH * LW]; // blurred level image, indexed by real coordinate offsets
    uint8_t *sS = s_u;                                          // [SH][SW] staged bytes
    float *sRow = reinterpret_cast<float *>(s_u + SH * SW);     // [SH][RS] row-pass values, rows yr0-1 .. yr1+1
    float(*sT)[TH][LW] = reinterpret_cast<float(*)[TH][LW]>(s_u); // [3][TH][LW], after the blur
    const int pi = blockIdx.z;
    const int2 pr = pairs[pi >> 1];
    const uint8_t *src = frames + (size_t)((pi & 1) ? pr.y : pr.x) * W * H;
    const size_t Nk = (size_t)W * H;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // real image region the tile's (clamped) reads touch
    const int xr0 = max(x0 - N, 0), xr1 = min(x0 + TW - 1 + N, W - 1);
    const int yr0 = max(y0 - N, 0), yr1 = min(y0 + TH - 1 + N, H - 1);
    const int ny = yr1 - yr0 + 1; // (columns past xr1 are computed from whatever was staged and never read)
    // stage bytes for columns xr0-1 .. xr1+1, rows yr0-1 .. yr1+1 (reflected outside the image);
    // the staged columns start at a multiple of 4 so interior tiles copy dwords
    const int xs = (xr0 - 1) & ~3, off = xr0 - 1 - xs; // column xr0-1 sits at byte `off` of a staged row
    const int ncols = xr1 + 1 - xs + 1;
    const bool dwords = (W & 3) == 0 && xs >= 0 && xs + ((ncols + 3) & ~3) <= W;
    constexpr int U = 8;
    if (dwords) {
        const int nq = (ncols + 3) >> 2;
        for (int c = lane; c < nq; c += 64) {
            for (int j0 = wave; j0 < ny + 2; j0 += 4 * U) {
                uint32_t v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < ny + 2)
                        v[u] = *reinterpret_cast<const uint32_t *>(src + (size_t)reflect101_near(yr0 - 1 + ry, H) * W + xs + 4 * c);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < ny + 2)
                        *reinterpret_cast<uint32_t *>(sS + ry * SW + 4 * c) = v[u];
                }
            }
        }
    } else {
        for (int c = lane; c < ncols; c += 64) {
            const int x = reflect101_near(xs + c, W);
            for (int j0 = wave; j0 < ny + 2; j0 += 4 * U) {
                uint8_t v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < ny + 2)
                        v[u] = src[(size_t)reflect101_near(yr0 - 1 + ry, H) * W + x];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < ny + 2)
                        sS[ry * SW + c] = v[u];
                }
            }
        }
    }
    __syncthreads();
    // row pass (tap order of k_level_image, ksz == 3), four adjacent pixels per item: their six bytes come out of
    // three aligned dwords (the staged row starts on a dword; `off` is the same for the whole tile)
    for (int idx = threadIdx.x; idx < (ny + 2) * (RS / 4); idx += 256) {
        const int ry = idx / (RS / 4), cx = 4 * (idx - ry * (RS / 4));
        const uint32_t *q = reinterpret_cast<const uint32_t *>(sS + ry * SW + cx); // bytes off + cx .. are pixels cx-1 ..
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, off), hi = __builtin_amdgcn_alignbyte(d2, d1, off);
        const float b0 = (float)(lo & 0xff), b1 = (float)((lo >> 8) & 0xff), b2 = (float)((lo >> 16) & 0xff),
                    b3 = (float)(lo >> 24), b4 = (float)(hi & 0xff), b5 = (float)((hi >> 8) & 0xff);
        float *o = sRow + ry * RS + cx;
        *reinterpret_cast<f32x2 *>(o) = f32x2{b1 * kc + (b0 + b2) * k1, b2 * kc + (b1 + b3) * k1};
        *reinterpret_cast<f32x2 *>(o + 2) = f32x2{b3 * kc + (b2 + b4) * k1, b4 * kc + (b3 + b5) * k1};
    }
    __syncthreads();
    // column pass -> blurred image at real coordinates (xr0 + cx, yr0 + ry), four columns per item
    for (int idx = threadIdx.x; idx < ny * (RS / 4); idx += 256) {
        const int ry = idx / (RS / 4), cx = 4 * (idx - ry * (RS / 4));
        const float *c = sRow + (ry + 1) * RS + cx;
        const f32x4 m = *reinterpret_cast<const f32x4 *>(c), dn = *reinterpret_cast<const f32x4 *>(c + RS),
                    up = *reinterpret_cast<const f32x4 *>(c - RS);
        f32x4 v = kc * m;
        v += k1 * (dn + up);
        float *o = sI + ry * LW + cx;
        *reinterpret_cast<f32x2 *>(o) = f32x2{v.x, v.y};
        if (cx + 2 < LW)
            *reinterpret_cast<f32x2 *>(o + 2) = f32x2{v.z, v.w};
    }
    __syncthreads();
    tile_expansion<N, TW, TH>(sI, sT, x0, y0, xr0, yr0, W, H, pc, R + (size_t)pi * 5 * Nk, Nk);
}

// A1+A2 fused for a level that is exactly half the frame in both directions with the 3-tap blur
// (level 1 of a pyr_scale = 0.5 pyramid over even frame sizes).  resize.cpp's coordinates are then
// (2X, 2Y) with both fractions exactly 0.5, so a level pixel is the lerp of the blurred frame at a 2x2
// block, each of those a 3x3 separable blur: computed per level pixel from its 4x4 bytes with
// k_level_image's statements (row pass, column pass centre-then-pair, horizontal lerp, vertical
// lerp); the level image never leaves the CU.  Then the expansion passes shared with level 0.
template <int N>
__global__ void __launch_bounds__(256)
k_level1_polyexp_t(const uint8_t *__restrict__ frames, const int2 *__restrict__ pairs, float *__restrict__ R, int W,
                   int H, float kc, float k1, PolyConst pc)
{
    constexpr int TW = 64, TH = TF_EXP_TH1, LW = TW + 2 * N, LH = TH + 2 * N;
    constexpr int SW = ((2 * LW + 2 + 3 + 3) + 3) & ~3, SH = 2 * LH + 2; // staged bytes: 2 per level pixel + 1 all round
    constexpr int BYTES_A = SH * SW, BYTES_T = 3 * TH * LW * 4;
    constexpr int BYTES_U = ((BYTES_A > BYTES_T ? BYTES_A : BYTES_T) + 15) & ~15;
    __shared__ __attribute__((aligned(16))) uint8_t s_u[BYTES_U];
    __shared__ float sI[LH * LW];
    uint8_t *sS = s_u;
    float(*sT)[TH][LW] = reinterpret_cast<float(*)[TH][LW]>(s_u);
    const int Wk = W >> 1, Hk = H >> 1;
    const int pi = blockIdx.z;
    const int2 pr = pairs[pi >> 1];
    const uint8_t *src = frames + (size_t)((pi & 1) ? pr.y : pr.x) * W * H;
    const size_t Nk = (size_t)Wk * Hk;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xr0 = max(x0 - N, 0), xr1 = min(x0 + TW - 1 + N, Wk - 1);
    const int yr0 = max(y0 - N, 0), yr1 = min(y0 + TH - 1 + N, Hk - 1);
    const int ny = yr1 - yr0 + 1;
    // frame bytes: columns 2*xr0-1 .. 2*xr1+2, rows 2*yr0-1 .. 2*yr1+2 (REFLECT_101 outside the frame)
    const int cfirst = 2 * xr0 - 1, rfirst = 2 * yr0 - 1, nrows = 2 * ny + 2;
    const int xs = cfirst >= 0 ? (cfirst & ~3) : cfirst, off = cfirst - xs;
    const int ncols = 2 * xr1 + 2 - xs + 1;
    const bool dwords = (W & 3) == 0 && xs >= 0 && (xs & 3) == 0 && xs + ((ncols + 3) & ~3) <= W;
    constexpr int U = 8;
    if (dwords) {
        const int nq = (ncols + 3) >> 2;
        for (int c = lane; c < nq; c += 64) {
            for (int j0 = wave; j0 < nrows; j0 += 4 * U) {
                uint32_t v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        v[u] = *reinterpret_cast<const uint32_t *>(src + (size_t)reflect101_near(rfirst + ry, H) * W + xs + 4 * c);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        *reinterpret_cast<uint32_t *>(sS + ry * SW + 4 * c) = v[u];
                }
            }
        }
    } else {
        for (int c = lane; c < ncols; c += 64) {
            const int x = reflect101_near(xs + c, W);
            for (int j0 = wave; j0 < nrows; j0 += 4 * U) {
                uint8_t v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        v[u] = src[(size_t)reflect101_near(rfirst + ry, H) * W + x];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        sS[ry * SW + c] = v[u];
                }
            }
        }
    }
    __syncthreads();
    // Level pixel (xr0 + cx, yr0 + ry): frame rows 2Y-1 .. 2Y+2 are staged rows 2*ry .. 2*ry+3, frame columns
    // 2X-1 .. 2X+2 staged bytes off + 2*cx .. +3.  An item is a 2 x 2 block of level pixels: six staged rows, six
    // bytes of each out of three aligned dwords (`off` is the tile's), 24 row-pass values instead of 32.  Rows and
    // columns past the tile's real region are computed from whatever was staged and never read.
    static_assert(LW % 2 == 0 && SW >= 2 * LW + 8, "2 x 2 blocks; a block's three dwords stay inside its staged row");
    for (int idx = threadIdx.x; idx < ((ny + 1) >> 1) * (LW / 2); idx += 256) {
        const int by2 = idx / (LW / 2), ry = 2 * by2, cx = 2 * (idx - by2 * (LW / 2));
        float rp[6][4];
#pragma unroll
        for (int dy = 0; dy < 6; dy++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(sS + (2 * ry + dy) * SW + 2 * cx);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, off), hi = __builtin_amdgcn_alignbyte(d2, d1, off);
            const float b0 = (float)(lo & 0xff), b1 = (float)((lo >> 8) & 0xff), b2 = (float)((lo >> 16) & 0xff),
                        b3 = (float)(lo >> 24), b4 = (float)(hi & 0xff), b5 = (float)((hi >> 8) & 0xff);
            rp[dy][0] = b1 * kc + (b0 + b2) * k1; // row pass at frame columns 2X, 2X+1 (this pixel), 2X+2, 2X+3 (the next)
            rp[dy][1] = b2 * kc + (b1 + b3) * k1;
            rp[dy][2] = b3 * kc + (b2 + b4) * k1;
            rp[dy][3] = b4 * kc + (b3 + b5) * k1;
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            float out[2];
#pragma unroll
            for (int b = 0; b < 2; b++) {
                float v00 = kc * rp[2 * a + 1][2 * b], v01 = kc * rp[2 * a + 1][2 * b + 1];
                float v10 = kc * rp[2 * a + 2][2 * b], v11 = kc * rp[2 * a + 2][2 * b + 1];
                v00 += k1 * (rp[2 * a + 2][2 * b] + rp[2 * a][2 * b]); // column pass at frame row 2Y: centre, then (row+1 + row-1)
                v01 += k1 * (rp[2 * a + 2][2 * b + 1] + rp[2 * a][2 * b + 1]);
                v10 += k1 * (rp[2 * a + 3][2 * b] + rp[2 * a + 1][2 * b]); //                          2Y+1
                v11 += k1 * (rp[2 * a + 3][2 * b + 1] + rp[2 * a + 1][2 * b + 1]);
                const float h0 = v00 * (1.f - 0.5f) + v01 * 0.5f, h1 = v10 * (1.f - 0.5f) + v11 * 0.5f;
                out[b] = h0 * (1.f - 0.5f) + h1 * 0.5f;
            }
            if (ry + a < ny)
                *reinterpret_cast<f32x2 *>(sI + (ry + a) * LW + cx) = f32x2{out[0], out[1]};
        }
    }
    __syncthreads();
    tile_expansion<N, TW, TH>(sI, sT, x0, y0, xr0, yr0, Wk, Hk, pc, R + (size_t)pi * 5 * Nk, Nk);
}

} // namespace

namespace tf {
namespace fb {

int fb_polyexp(tf_fb *fb, int w, int h, int n_images, int k)
{
    const int n = fb->pc.n;
    dim3 grid(cdiv(w, PX_TW), cdiv(h, PX_TH), n_images);
    if (n == 5)
        return launch(lvl_name("fb_polyexp", k), k_polyexp_t<5>, grid, dim3(256), 0, (const float *)fb->imgk(k),
                      fb->Rk_out(k), w, h, fb->pc);
    if (n == 7)
        return launch(lvl_name("fb_polyexp", k), k_polyexp_t<7>, grid, dim3(256), 0, (const float *)fb->imgk(k),
                      fb->Rk_out(k), w, h, fb->pc);
    size_t smem = ((size_t)(PX_TH + 2 * n) * (PX_TW + 2 * n) + 3 * (size_t)PX_TH * (PX_TW + 2 * n)) * sizeof(float);
    return launch(lvl_name("fb_polyexp_generic", k), k_polyexp, grid, dim3(256), smem,
                  (const float *)fb->imgk(k), fb->Rk_out(k), w, h, fb->pc);
}

// A1+A2 fusion applies to a level that is a copy-sized resize of the frame with the 3-tap blur
// (level 0 of every pyramid) and a poly_n the blocked expansion is instantiated for.
bool fb_can_fuse_level(tf_fb *fb, int k)
{
    static const bool off = tune("TF_FB_NO_A1A2", 0) != 0;
    const Level &L = *fb->lv[k];
    return !off && L.W == fb->W && L.H == fb->H && L.ksz == 3 && (fb->pc.n == 5 || fb->pc.n == 7);
}

// ... and to a level that is exactly half the frame (k_level1_polyexp_t)
bool fb_can_fuse_half_level(tf_fb *fb, int k)
{
    static const bool off = tune("TF_FB_NO_A1A2", 0) != 0;
    const Level &L = *fb->lv[k];
    return !off && 2 * L.W == fb->W && 2 * L.H == fb->H && L.ksz == 3 && (fb->pc.n == 5 || fb->pc.n == 7);
}

int fb_level1_polyexp(tf_fb *fb, int k, int n_images)
{
    Level &L = *fb->lv[k];
    dim3 grid(cdiv(L.W, 64), cdiv(L.H, TF_EXP_TH1), n_images);
    const float kc = L.kern_host[1], k1 = L.kern_host[2];
    if (fb->pc.n == 5)
        return launch(lvl_name("fb_level_polyexp", k), k_level1_polyexp_t<5>, grid, dim3(256), 0,
                      (const uint8_t *)fb->frames.as<uint8_t>(), fb->image_list(), fb->Rk_out(k), fb->W,
                      fb->H, kc, k1, fb->pc);
    return launch(lvl_name("fb_level_polyexp", k), k_level1_polyexp_t<7>, grid, dim3(256), 0,
                  (const uint8_t *)fb->frames.as<uint8_t>(), fb->image_list(), fb->Rk_out(k), fb->W, fb->H,
                  kc, k1, fb->pc);
}

int fb_level0_polyexp(tf_fb *fb, int k, int n_images)
{
    Level &L = *fb->lv[k];
    dim3 grid(cdiv(L.W, 64), cdiv(L.H, TF_EXP_TH0), n_images);
    const float kc = L.kern_host[1], k1 = L.kern_host[2];
    if (fb->pc.n == 5)
        return launch(lvl_name("fb_level_polyexp", k), k_level0_polyexp_t<5>, grid, dim3(256), 0,
                      (const uint8_t *)fb->frames.as<uint8_t>(), fb->image_list(), fb->Rk_out(k),
                      L.W, L.H, kc, k1, fb->pc);
    return launch(lvl_name("fb_level_polyexp", k), k_level0_polyexp_t<7>, grid, dim3(256), 0,
                  (const uint8_t *)fb->frames.as<uint8_t>(), fb->image_list(), fb->Rk_out(k), L.W,
                  L.H, kc, k1, fb->pc);
}

} // namespace fb
} // namespace tf
