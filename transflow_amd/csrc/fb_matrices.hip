// A3 and A4 of the Farneback path as separate kernels (small levels; every level with option fb_fused = 0, the Gaussian
// window, windows the one-kernel iteration is not instantiated for): update-matrices with A5 fused, the marching box blur +
// 2x2 solve with OpenCV's column sums carried across row segments, FarnebackUpdateFlow_GaussianBlur, and the INTER_AREA
// shrink of a caller's initial flow (optflowgf.cpp; cv.py:479-490).
#include "fb_common.h"

namespace {

#ifndef BLUR_PREFETCH
#define BLUR_PREFETCH 3 // rows of M kept in flight per wave in the blur march (2: 1219 us, 3: 1177, 4: 1225 at 4K x16)
#endif

// ---------------------------------------------------------------------------------
// A3: one pixel of FarnebackUpdateMatrices.  R0/R1 in the channel-pair layout; out[5] = M.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void update_matrix_px(const float *__restrict__ R0, const float *__restrict__ R1, size_t Nk,
                                                 int Wk, int Hk, int x, int y, float dx, float dy, float out[5])
{
    const size_t o = (size_t)y * Wk + x;
    float r0[5];
    r_load_px(R0, Nk, o, r0);
    float fx = x + dx, fy = y + dy;
    int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    float r2, r3, r4, r5, r6;
    fx -= x1;
    fy -= y1;
    if ((unsigned)x1 < (unsigned)(Wk - 1) && (unsigned)y1 < (unsigned)(Hk - 1)) {
        float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        size_t q = (size_t)y1 * Wk + x1;
        float2u t[5], b[5];
        r_load_taps(R1, Nk, q, Wk, t, b);
        r2 = a00 * t[0].x + a01 * t[0].y + a10 * b[0].x + a11 * b[0].y;
        r3 = a00 * t[1].x + a01 * t[1].y + a10 * b[1].x + a11 * b[1].y;
        r4 = a00 * t[2].x + a01 * t[2].y + a10 * b[2].x + a11 * b[2].y;
        r5 = a00 * t[3].x + a01 * t[3].y + a10 * b[3].x + a11 * b[3].y;
        r6 = a00 * t[4].x + a01 * t[4].y + a10 * b[4].x + a11 * b[4].y;
        r4 = (r0[2] + r4) * 0.5f;
        r5 = (r0[3] + r5) * 0.5f;
        r6 = (r0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = r0[2];
        r5 = r0[3];
        r6 = r0[4] * 0.5f;
    }
    r2 = (r0[0] - r2) * 0.5f;
    r3 = (r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    {
        float scale = border_scale(x, y, Wk, Hk);
        r2 *= scale;
        r3 *= scale;
        r4 *= scale;
        r5 *= scale;
        r6 *= scale;
    }
    out[0] = r4 * r4 + r6 * r6;
    out[1] = (r4 + r5) * r6;
    out[2] = r5 * r5 + r6 * r6;
    out[3] = r4 * r2 + r6 * r3;
    out[4] = r6 * r2 + r5 * r3;
}
struct GatherRegs {
    float2 r0[5];   // R0 at the two pixels, per channel (x: first pixel, y: second)
    float2 t[2][5]; // R1 pair (x1, x1+1) on row y1, per pixel and channel
    float2 b[2][5]; // R1 pair on row y1+1
    float dx[2], dy[2], fx[2], fy[2];
    bool inb[2];
};

// loads for the matrices of pixels (xa, y) and (xb, y); flow already known
__device__ __forceinline__ void gather_issue(GatherRegs &g, const float *__restrict__ R0, const float *__restrict__ R1,
                                             size_t Nk, int Wk, int Hk, int xa, int xb, int y, float2 fa, float2 fb)
{
    const int xs[2] = {xa, xb};
    const float2 fl[2] = {fa, fb};
    const size_t oa = (size_t)y * Wk + xa, ob = (size_t)y * Wk + xb;
    {
        float va[5], vb[5];
        r_load_px(R0, Nk, oa, va);
        r_load_px(R0, Nk, ob, vb);
#pragma unroll
        for (int c = 0; c < 5; c++)
            g.r0[c] = make_float2(va[c], vb[c]);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        float dx = fl[j].x, dy = fl[j].y;
        float fx = xs[j] + dx, fy = y + dy;
        int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
        fx -= x1;
        fy -= y1;
        g.dx[j] = dx;
        g.dy[j] = dy;
        g.fx[j] = fx;
        g.fy[j] = fy;
        g.inb[j] = (unsigned)x1 < (unsigned)(Wk - 1) && (unsigned)y1 < (unsigned)(Hk - 1);
        // out-of-frame taps load from a clamped (valid) address and are discarded: no branch
        // around the loads, so they all stay in flight together
        int x1c = clampi(x1, 0, Wk - 2), y1c = clampi(y1, 0, Hk - 2);
        float2u tv[5], bv[5];
        r_load_taps(R1, Nk, (size_t)y1c * Wk + x1c, Wk, tv, bv);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            g.t[j][c] = make_float2(tv[c].x, tv[c].y);
            g.b[j][c] = make_float2(bv[c].x, bv[c].y);
        }
    }
}

// the arithmetic of update_matrix_px on the gathered values; m[j][5] for the two pixels
__device__ __forceinline__ void gather_finish(const GatherRegs &g, int Wk, int Hk, int xa, int xb, int y,
                                              float m[2][5])
{
    const int xs[2] = {xa, xb};
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int x = xs[j];
        const float R00 = j ? g.r0[0].y : g.r0[0].x, R01 = j ? g.r0[1].y : g.r0[1].x, R02 = j ? g.r0[2].y : g.r0[2].x,
                    R03 = j ? g.r0[3].y : g.r0[3].x, R04 = j ? g.r0[4].y : g.r0[4].x;
        const float fx = g.fx[j], fy = g.fy[j], dx = g.dx[j], dy = g.dy[j];
        float r2, r3, r4, r5, r6;
        if (g.inb[j]) {
            float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
            r2 = a00 * g.t[j][0].x + a01 * g.t[j][0].y + a10 * g.b[j][0].x + a11 * g.b[j][0].y;
            r3 = a00 * g.t[j][1].x + a01 * g.t[j][1].y + a10 * g.b[j][1].x + a11 * g.b[j][1].y;
            r4 = a00 * g.t[j][2].x + a01 * g.t[j][2].y + a10 * g.b[j][2].x + a11 * g.b[j][2].y;
            r5 = a00 * g.t[j][3].x + a01 * g.t[j][3].y + a10 * g.b[j][3].x + a11 * g.b[j][3].y;
            r6 = a00 * g.t[j][4].x + a01 * g.t[j][4].y + a10 * g.b[j][4].x + a11 * g.b[j][4].y;
            r4 = (R02 + r4) * 0.5f;
            r5 = (R03 + r5) * 0.5f;
            r6 = (R04 + r6) * 0.25f;
        } else {
            r2 = r3 = 0.f;
            r4 = R02;
            r5 = R03;
            r6 = R04 * 0.5f;
        }
        r2 = (R00 - r2) * 0.5f;
        r3 = (R01 - r3) * 0.5f;
        r2 += r4 * dy + r6 * dx;
        r3 += r6 * dy + r5 * dx;
        {
            float scale = border_scale(x, y, Wk, Hk);
            r2 *= scale;
            r3 *= scale;
            r4 *= scale;
            r5 *= scale;
            r6 *= scale;
        }
        m[j][0] = r4 * r4 + r6 * r6;
        m[j][1] = (r4 + r5) * r6;
        m[j][2] = r5 * r5 + r6 * r6;
        m[j][3] = r4 * r2 + r6 * r3;
        m[j][4] = r6 * r2 + r5 * r3;
    }
}

constexpr int UM_TW = 128, UM_TH = 4; // tile: 128 columns x 4 rows, 256 threads, two pixels each

__global__ void __launch_bounds__(256)
k_update_matrices(const float *__restrict__ R, float *__restrict__ M, int Wk, int Hk, FlowInit fi)
{
    // 256 threads = 64 columns x 4 rows: the two R1 rows a bilinear tap straddles are shared by the
    // block's neighbouring output rows.  Measured R1 traffic model: 20 B x (rows+1)/rows x
    // (lines+1)/lines of a flow-shifted 256-byte row segment (profiles/README.md); 128x8 tiles were
    // measured and bring nothing more
    // XCD-aware tile order: vertically adjacent tiles (which share R1 rows) meet in one L2
    unsigned bx, by;
    xcd_tile(bx, by);
    // a thread owns two pixels of one row, 64 columns apart (both halves of a 128-column tile row
    // are coalesced), and gathers for both with every load in flight at once
    const int xa = bx * UM_TW + (threadIdx.x & 63), xb = xa + 64;
    const int y = by * UM_TH + (threadIdx.x >> 6);
    if (xa >= Wk || y >= Hk)
        return;
    const bool has_b = xb < Wk;
    const int xbc = has_b ? xb : xa;
    const int pair = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    float2 fa = make_float2(0.f, 0.f), fb = fa;
    if (fi.mode == 1) {
        const float2 *c = fi.src + (size_t)pair * fi.Wc * fi.Hc;
        const int sy = fi.yofs[y];
        const float fy = fi.yfrac[y], b0 = 1.f - fy;
        const int sy0 = clampi(sy, 0, fi.Hc - 1), sy1 = clampi(sy + 1, 0, fi.Hc - 1);
        const int xs[2] = {xa, xbc};
        float2 res[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int sx = fi.xofs[xs[j]];
            const float fx = fi.xfrac[xs[j]];
            const int sx1 = min(sx + 1, fi.Wc - 1);
            float2 a = c[(size_t)sy0 * fi.Wc + sx], b = c[(size_t)sy0 * fi.Wc + sx1];
            float2 d = c[(size_t)sy1 * fi.Wc + sx], e = c[(size_t)sy1 * fi.Wc + sx1];
            float2 h0, h1;
            if (sx >= fi.Wc - 1) { // resize.cpp: columns past xmax copy S[sx]
                h0 = a;
                h1 = d;
            } else {
                float a0 = 1.f - fx;
                h0 = make_float2(a.x * a0 + b.x * fx, a.y * a0 + b.y * fx);
                h1 = make_float2(d.x * a0 + e.x * fx, d.y * a0 + e.y * fx);
            }
            res[j] = make_float2((h0.x * b0 + h1.x * fy) * fi.mul, (h0.y * b0 + h1.y * fy) * fi.mul);
        }
        fa = res[0];
        fb = res[1];
    } else if (fi.mode == 2) {
        const float2 *f = fi.src + (size_t)pair * Nk + (size_t)y * Wk;
        fa = f[xa];
        fb = f[xbc];
    }
    const int2 im = pair_images(fi, pair);
    const float *R0 = R + (size_t)im.x * 5 * Nk, *R1 = R + (size_t)im.y * 5 * Nk;
    float m[2][5];
    if (Wk >= 2 && Hk >= 2) {
        GatherRegs g;
        gather_issue(g, R0, R1, Nk, Wk, Hk, xa, xbc, y, fa, fb);
        gather_finish(g, Wk, Hk, xa, xbc, y, m);
    } else { // degenerate one-pixel-wide levels: no in-frame bilinear cell exists
        update_matrix_px(R0, R1, Nk, Wk, Hk, xa, y, fa.x, fa.y, m[0]);
        update_matrix_px(R0, R1, Nk, Wk, Hk, xbc, y, fb.x, fb.y, m[1]);
    }
    float *Mo = M + (size_t)pair * 5 * Nk + (size_t)y * Wk;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        Mo[c * Nk + xa] = m[0][c];
        if (has_b)
            Mo[c * Nk + xb] = m[1][c];
    }
}
// ---------------------------------------------------------------------------------
// A4, any window width: box blur of M over (2m+1)^2 with replicated borders, 2x2 solve.
// One block marches a strip of 256 columns (256-2m outputs + halo) down `seg`
// rows, holding the vertical window sums of its column in double registers.
// ---------------------------------------------------------------------------------
constexpr int BS_THREADS = 256;

__global__ void __launch_bounds__(BS_THREADS)
k_blur_solve(const float *__restrict__ Min, float2 *__restrict__ flow_out, int Wk, int Hk, int m, double scale, int seg,
             const double *__restrict__ carry)
{
    __shared__ double s_v[2][5][BS_THREADS];
    const int tid = threadIdx.x;
    const int out_cols = BS_THREADS - 2 * m;
    const int col = blockIdx.x * out_cols - m + tid;
    const int colc = clampi(col, 0, Wk - 1);
    const int pair = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    const float *Mi = Min + (size_t)pair * 5 * Nk + colc;
    const int r0 = blockIdx.y * seg, r1 = min(r0 + seg, Hk);
    // OpenCV's vertical chain (ColumnCarry's note): primed at row 0, or continued from the segment above
    double vs[5];
    if (r0 == 0) {
#pragma unroll
        for (int c = 0; c < 5; c++)
            vs[c] = (double)(Mi[c * Nk] * (float)(m + 2)); // vsum[x] = srow0[x] * (m + 2): a float product
        for (int j = 1; j < m; j++) {
            size_t ro = (size_t)min(j, Hk - 1) * Wk;
#pragma unroll
            for (int c = 0; c < 5; c++)
                vs[c] += (double)Mi[c * Nk + ro];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 5; c++)
            vs[c] = carry[((((size_t)blockIdx.y * gridDim.z + pair) * 5 + c) * Wk) + colc];
    }
    const bool is_out = tid >= m && tid < BS_THREADS - m && col < Wk;
    int buf = 0;
    for (int y = r0; y < r1; y++) {
        {
            size_t ra = (size_t)min(y + m, Hk - 1) * Wk, rb = (size_t)max(y - m - 1, 0) * Wk;
#pragma unroll
            for (int c = 0; c < 5; c++)
                vs[c] += (double)(Mi[c * Nk + ra] - Mi[c * Nk + rb]); // FarnebackUpdateFlow_Blur: vsum += srow1[x] - srow0[x], a float difference
        }
#pragma unroll
        for (int c = 0; c < 5; c++)
            s_v[buf][c][tid] = vs[c];
        __syncthreads();
        if (is_out) {
            double g[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                double a = 0;
                for (int j = -m; j <= m; j++)
                    a += s_v[buf][c][tid + j];
                g[c] = a * scale;
            }
            double idet = 1. / (g[0] * g[2] - g[1] * g[1] + 1e-3);
            float fx = (float)((g[0] * g[4] - g[1] * g[3]) * idet);
            float fy = (float)((g[2] * g[3] - g[1] * g[4]) * idet);
            flow_out[(size_t)pair * Nk + (size_t)y * Wk + col] = make_float2(fx, fy);
        }
        buf ^= 1;
    }
}
// ---------------------------------------------------------------------------------
// A4, fast path: one WAVE marches a strip of 128 columns (two per lane) down `seg`
// rows; no block barrier, so waves run decoupled and hide each other's latency.
// Per row: the lane updates the vertical window sums of its two columns (double
// registers, float2 loads, next row prefetched), publishes them to the wave's LDS
// row, and reads back the 2M+2 neighbours it needs for its two outputs with
// 16-byte LDS reads.  HALO = M rounded up to even keeps column pairs aligned.
// ---------------------------------------------------------------------------------
struct dpair {
    double x, y;
};

template <int M, bool VEC>
__device__ __forceinline__ void blur_solve_wave_body(const float *__restrict__ Min, float2 *__restrict__ flow_out,
                                                     int Wk, int Hk, double scale, int seg, const double *__restrict__ carry,
                                                     double (*s_e)[64], double (*s_o)[64], double (*s_p)[64])
{
    constexpr int HALO = (M + 1) & ~1;
    constexpr int OUTC = 128 - 2 * HALO;
    const int lane = threadIdx.x;
    unsigned bx, by;
    xcd_tile(bx, by);
    const int c0 = bx * OUTC - HALO + 2 * lane;
    const int pair = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    const float *Mi = Min + (size_t)pair * 5 * Nk;
    const int r0 = by * seg, r1 = min(r0 + seg, Hk);
    // VEC: the whole strip lies inside the image (wave-uniform), so every lane loads its two
    // columns with one 8-byte load; otherwise two clamped scalar loads (replicated border)
    const int ca = clampi(c0, 0, Wk - 1), cb = clampi(c0 + 1, 0, Wk - 1);

    auto load_row = [&](int row, float2 out[5]) {
        const size_t ro = (size_t)row * Wk;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const float *p = Mi + c * Nk + ro;
            if (VEC) {
                float2u v = *reinterpret_cast<const float2u *>(p + c0);
                out[c] = make_float2(v.x, v.y);
            } else {
                out[c] = make_float2(p[ca], p[cb]);
            }
        }
    };

    // OpenCV's vertical chain (ColumnCarry's note): the segment at the top primes it -- vsum = row 0 * (m + 2), a float
    // product, plus rows 1 .. m - 1 --, the others continue from the chain's value after the row above them
    double vs[5][2];
    if (r0 == 0) {
        float2 v[5];
        load_row(0, v);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            vs[c][0] = (double)(v[c].x * (float)(M + 2));
            vs[c][1] = (double)(v[c].y * (float)(M + 2));
        }
        for (int j = 1; j < M; j++) {
            load_row(min(j, Hk - 1), v);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                vs[c][0] += (double)v[c].x;
                vs[c][1] += (double)v[c].y;
            }
        }
    } else {
        const double *C = carry + (((size_t)by * gridDim.z + pair) * 5) * Wk;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            vs[c][0] = C[(size_t)c * Wk + ca];
            vs[c][1] = C[(size_t)c * Wk + cb];
        }
    }
    // rows entering / leaving the window, prefetched PD steps ahead (slot t % PD serves step
    // r0+t); always loaded from clamped row indices, so no branch surrounds a load
    constexpr int PD = BLUR_PREFETCH;
    float2 pin[PD][5], pout[PD][5];
#pragma unroll
    for (int t = 0; t < PD; t++) {
        load_row(min(r0 + t + M, Hk - 1), pin[t]);
        load_row(clampi(r0 + t - M - 1, 0, Hk - 1), pout[t]);
    }
    const bool is_out = lane >= HALO / 2 && lane < 64 - HALO / 2 && c0 < Wk;
    for (int yb = r0; yb < r1; yb += PD) {
#pragma unroll
      for (int h = 0; h < PD; h++) {
        const int y = yb + h;
        if (y >= r1)
            break;
        {
            float2(&in)[5] = pin[h];
            float2(&out)[5] = pout[h];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                // OpenCV's increment: vsum[x] += srow1[x] - srow0[x] -- the difference in float, accumulated in double
                vs[c][0] += (double)(in[c].x - out[c].x);
                vs[c][1] += (double)(in[c].y - out[c].y);
            }
            load_row(min(y + PD + M, Hk - 1), in);                // step y+PD: entering row
            load_row(clampi(y + PD - M - 1, 0, Hk - 1), out);     //            leaving row
        }
#pragma unroll
        for (int c = 0; c < 5; c++) {
            // the lane publishes its two column sums and their pair sum, each in its own LDS row
            // (8-byte accesses at an 8-byte lane stride are bank-conflict free)
            s_e[c][lane] = vs[c][0];
            s_o[c][lane] = vs[c][1];
            s_p[c][lane] = vs[c][0] + vs[c][1];
        }
        lds_wave_sync(); // single-wave workgroup: orders the LDS writes before the reads, leaves the loads in flight
        if (is_out) {
            double g0[5], g1[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                // windows of the lane's columns 2l and 2l+1 as whole neighbour pairs plus one single
                // column at each end: M+2 (odd M) LDS reads and adds instead of 2M+2
                if (M & 1) {
                    constexpr int h = (M - 1) / 2, k = (M + 1) / 2;
                    double common = s_p[c][lane - h];
#pragma unroll
                    for (int j = -h + 1; j <= h; j++)
                        common += s_p[c][lane + j];
                    g0[c] = (s_o[c][lane - k] + common) * scale; // columns 2l-M .. 2l+M
                    g1[c] = (common + s_e[c][lane + k]) * scale; // columns 2l+1-M .. 2l+1+M
                } else {
                    constexpr int h = M / 2;
                    double mid = s_p[c][lane - h + 1];
#pragma unroll
                    for (int j = -h + 2; j <= h - 1; j++)
                        mid += s_p[c][lane + j];
                    g0[c] = (s_p[c][lane - h] + mid + s_e[c][lane + h]) * scale;
                    g1[c] = (s_o[c][lane - h] + mid + s_p[c][lane + h]) * scale;
                }
            }
            double idet0 = 1. / (g0[0] * g0[2] - g0[1] * g0[1] + 1e-3);
            double idet1 = 1. / (g1[0] * g1[2] - g1[1] * g1[1] + 1e-3);
            float2 f0 = make_float2((float)((g0[0] * g0[4] - g0[1] * g0[3]) * idet0),
                                    (float)((g0[2] * g0[3] - g0[1] * g0[4]) * idet0));
            float2 f1 = make_float2((float)((g1[0] * g1[4] - g1[1] * g1[3]) * idet1),
                                    (float)((g1[2] * g1[3] - g1[1] * g1[4]) * idet1));
            float2 *o = flow_out + (size_t)pair * Nk + (size_t)y * Wk + c0;
            o[0] = f0;
            if (c0 + 1 < Wk)
                o[1] = f1;
        }
        lds_wave_sync(); // the next row's writes must not overtake this row's reads
      }
    }
}

template <int M>
__global__ void __launch_bounds__(64, 3)
k_blur_solve_wave(const float *__restrict__ Min, float2 *__restrict__ flow_out, int Wk, int Hk, double scale, int seg,
                  const double *__restrict__ carry)
{
    constexpr int HALO = (M + 1) & ~1;
    constexpr int OUTC = 128 - 2 * HALO;
    __shared__ double s_e[5][64], s_o[5][64], s_p[5][64];
    unsigned bx, by;
    xcd_tile(bx, by);
    const int first = (int)bx * OUTC - HALO;
    if (first >= 0 && first + 127 < Wk)
        blur_solve_wave_body<M, true>(Min, flow_out, Wk, Hk, scale, seg, carry, s_e, s_o, s_p);
    else
        blur_solve_wave_body<M, false>(Min, flow_out, Wk, Hk, scale, seg, carry, s_e, s_o, s_p);
}
// The same from M in memory (the two-kernel iteration): one thread per column and channel of a segment.
__global__ void __launch_bounds__(64)
k_blur_carry(const float *__restrict__ Min, double *__restrict__ S, int Wk, int Hk, int m, int seg)
{
    const int x = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y % 5, by = blockIdx.y / 5, pair = blockIdx.z;
    if (x >= Wk)
        return;
    const size_t Nk = (size_t)Wk * Hk;
    const float *P = Min + ((size_t)pair * 5 + c) * Nk + x;
    const int r0 = by * seg, r1 = min(r0 + seg, Hk);
    double vs = 0.0;
    if (r0 == 0) {
        vs = (double)(P[0] * (float)(m + 2)); // vsum[x] = srow0[x] * (m + 2): a float product
        for (int y = 1; y < m; y++)
            vs += (double)P[(size_t)min(y, Hk - 1) * Wk];
    }
#pragma unroll 8
    for (int y = r0; y < r1; y++) {
        const float in = P[(size_t)min(y + m, Hk - 1) * Wk], out = P[(size_t)max(y - m - 1, 0) * Wk];
        vs += (double)(in - out); // vsum[x] += srow1[x] - srow0[x]
    }
    S[(((size_t)by * gridDim.z + pair) * 5 + c) * Wk + x] = vs;
}

// S[s][i] (what segment s adds to the chain; S[0]: the chain after segment 0) -> the chain's value in front of segment s.
// Sixteen segments' values are loaded together (the additions are a serial chain, the loads need not be: with up to 64
// segments a load per addition made this the longest kernel of a small level).
__global__ void k_carry_scan(double *__restrict__ S, size_t n, int segs)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double acc = S[i];
    for (int s0 = 1; s0 < segs; s0 += 16) {
        double t[16];
#pragma unroll
        for (int j = 0; j < 16; j++)
            t[j] = s0 + j < segs ? S[(size_t)(s0 + j) * n + i] : 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (s0 + j < segs) {
                S[(size_t)(s0 + j) * n + i] = acc;
                acc += t[j];
            }
        }
    }
}
// ---------------------------------------------------------------------------------
// A4 with OPTFLOW_FARNEBACK_GAUSSIAN: FarnebackUpdateFlow_GaussianBlur.  The 2x2 systems are smoothed
// by a separable Gaussian of winsize / 2 taps a side (sigma = 0.3 * winsize / 2) in FLOAT -- vertical
// pass then horizontal pass, centre tap first, then pairs outwards, replicated borders -- and solved
// with +1e-3 in double.  One plane of M at a time through an LDS tile (64 x 16 outputs, halo m); the
// statements are the scalar loops of optflowgf.cpp, so the result is bit-identical to the oracle's.
// ---------------------------------------------------------------------------------
#define GS_TW 64
#define GS_TH 16
__global__ void __launch_bounds__(256)
k_gauss_solve(const float *__restrict__ Min, float2 *__restrict__ flow_out, int Wk, int Hk, int m, const float *__restrict__ taps)
{
    extern __shared__ float gs_lds[];
    const int LW = GS_TW + 2 * m, LH = GS_TH + 2 * m;
    float *sM = gs_lds;               // [LH][LW] one plane of M with its halo (clamped coordinates)
    float *sV = gs_lds + LH * LW;     // [GS_TH][LW] vertical pass
    float *sK = sV + GS_TH * LW;      // [m + 1] taps
    const int pair = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    const int x0 = blockIdx.x * GS_TW, y0 = blockIdx.y * GS_TH;
    for (int i = threadIdx.x; i <= m; i += 256)
        sK[i] = taps[i];
    float h[4][5]; // the thread's four outputs (rows ty, ty + 4, ty + 8, ty + 12 of column tx), five planes
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = 0; c < 5; c++) {
        const float *src = Min + ((size_t)pair * 5 + c) * Nk;
        __syncthreads(); // the previous plane's passes are done with sM / sV (and sK is written)
        for (int idx = threadIdx.x; idx < LH * LW; idx += 256) {
            const int ry = idx / LW, cx = idx - ry * LW;
            sM[idx] = src[(size_t)clampi(y0 - m + ry, 0, Hk - 1) * Wk + clampi(x0 - m + cx, 0, Wk - 1)];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < GS_TH * LW; idx += 256) {
            const int ry = idx / LW, cx = idx - ry * LW;
            const float *col = sM + (ry + m) * LW + cx;
            float s0 = col[0] * sK[0];
            for (int i = 1; i <= m; i++)
                s0 += (col[i * LW] + col[-i * LW]) * sK[i];
            sV[idx] = s0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float *row = sV + (ty + 4 * q) * LW + tx + m;
            float sum = row[0] * sK[0];
            for (int i = 1; i <= m; i++)
                sum += sK[i] * (row[-i] + row[i]);
            h[q][c] = sum;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = x0 + tx, y = y0 + ty + 4 * q;
        if (x >= Wk || y >= Hk)
            continue;
        const double g11 = h[q][0], g12 = h[q][1], g22 = h[q][2], h1 = h[q][3], h2 = h[q][4];
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        flow_out[(size_t)pair * Nk + (size_t)y * Wk + x] =
            make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
    }
}
// ---------------------------------------------------------------------------------
// OPTFLOW_USE_INITIAL_FLOW: the caller's full-resolution flow shrunk to the coarsest scale with
// resize(INTER_AREA) and multiplied by that scale (optflowgf.cpp: `resize(flow0, flow, size, 0, 0,
// INTER_AREA); flow *= scale`).  Integer factors: the sum of the block, four at a time, times 1/area;
// otherwise computeResizeAreaTab's weights -- per source row buf = sum_k S * alpha_k, then sum = beta0 *
// buf and sum += beta * buf over the rows of the cell, all float.  One thread per output pixel.
// ---------------------------------------------------------------------------------
__global__ void k_flow_area_init(const float2 *__restrict__ init, float2 *__restrict__ out, int W, int H, int Wc, int Hc,
                                 AreaTabs t, float mul)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y, pair = blockIdx.z;
    if (dx >= Wc)
        return;
    const float2 *src = init + (size_t)pair * W * H;
    float2 r;
    if (t.ix > 0) {
        const float2 *S = src + (size_t)dy * t.iy * W + (size_t)dx * t.ix;
        const int area = t.ix * t.iy;
        const float scale = 1.f / area;
        auto at = [&](int k) { return S[(k / t.ix) * W + (k % t.ix)]; };
        float sx = 0.f, sy = 0.f;
        int k = 0;
        for (; k <= area - 4; k += 4) {
            const float2 a = at(k), b = at(k + 1), c = at(k + 2), d = at(k + 3);
            sx += a.x + b.x + c.x + d.x;
            sy += a.y + b.y + c.y + d.y;
        }
        for (; k < area; k++) {
            const float2 a = at(k);
            sx += a.x;
            sy += a.y;
        }
        r = make_float2(sx * scale, sy * scale);
    } else {
        float sumx = 0.f, sumy = 0.f;
        for (int j = t.ystart[dy]; j < t.ystart[dy + 1]; j++) {
            const float2 *S = src + (size_t)t.ysi[j] * W;
            float bx = 0.f, by = 0.f;
            for (int k = t.xstart[dx]; k < t.xstart[dx + 1]; k++) {
                const float2 v = S[t.xsi[k]];
                bx = bx + v.x * t.xalpha[k];
                by = by + v.y * t.xalpha[k];
            }
            const float beta = t.yalpha[j];
            if (j == t.ystart[dy]) {
                sumx = beta * bx;
                sumy = beta * by;
            } else {
                sumx += beta * bx;
                sumy += beta * by;
            }
        }
        r = make_float2(sumx, sumy);
    }
    out[(size_t)pair * Wc * Hc + (size_t)dy * Wc + dx] = make_float2(r.x * mul, r.y * mul);
}

} // namespace

namespace tf {
namespace fb {

// M is sized by the largest level that ever takes the two-kernel path (levels the one-kernel iteration serves never
// store it: at 4K x 128 pairs that is levels 0 - 4, and a full-resolution M would be 21 GB nobody touches)
int fb_m_room(tf_fb *fb, int w, int h, int n_pairs)
{
    const size_t need = (size_t)n_pairs * 5 * w * h * sizeof(float);
    if (fb->M.bytes < need)
        TF_TRY(fb->M.alloc(need)); // (hipFree waits for whatever still reads the old one)
    return TF_OK;
}

int fb_update_matrices(tf_fb *fb, int w, int h, int n_pairs, const FlowInit &fi, int k)
{
    TF_TRY(fb_m_room(fb, w, h, n_pairs));
    dim3 grid(cdiv(w, UM_TW), cdiv(h, UM_TH), n_pairs);
    FlowInit f = fi;
    f.rmap = fb->rmap_dev;
    return launch(lvl_name("fb_update_matrices", k), k_update_matrices, grid, dim3(256), 0, (const float *)fb->Rk(k),
                  fb->M.as<float>(), w, h, f);
}

// room for the carries of a launch (and, for hand-offs inside it, its flags); the fault word
int fb_carry_room(tf_fb *fb, size_t carry_doubles, size_t flags)
{
    if (fb->col_carry.bytes < carry_doubles * sizeof(double))
        TF_TRY(fb->col_carry.alloc(carry_doubles * sizeof(double)));
    const size_t words = ((16 + flags + 3) & ~(size_t)3);
    if (flags && fb->chain_words.bytes < words * 4) {
        TF_TRY(fb->chain_words.alloc(words * 4)); // (hipFree waits for whatever still uses the old one)
        TF_HIP(hipMemsetAsync(fb->chain_words.p, 0, words * 4, stream()));
        fb->chain_epoch = 0;
    }
    if (!fb->chain_fault) {
        TF_HIP(hipHostMalloc((void **)&fb->chain_fault, 64, hipHostMallocDefault));
        *fb->chain_fault = 0;
    }
    return TF_OK;
}

// a wait inside an earlier launch gave up: its flow is wrong
int fb_check_fault(tf_fb *fb, const char *where)
{
    if (fb->chain_fault && *(volatile unsigned *)fb->chain_fault) {
        *fb->chain_fault = 0;
        return set_error(TF_ERR_HIP, "%s: a segment of k_flow_iter_pc waited more than 5 s for the column sums of the segment "
                                     "above it; the flow of that call is invalid", where);
    }
    return TF_OK;
}

int fb_carry_scan(tf_fb *fb, int w, int n_pairs, int segs, int k)
{
    const size_t n = (size_t)n_pairs * 5 * w;
    return launch(lvl_name("fb_carry_scan", k), k_carry_scan, dim3(cdiv(n, 256)), dim3(256), 0, fb->col_carry.as<double>(), n, segs);
}

template <int M>
static int launch_blur_solve_wave(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, double scale, int k)
{
    constexpr int HALO = (M + 1) & ~1;
    constexpr int OUTC = 128 - 2 * HALO;
    const unsigned strips = cdiv(w, OUTC);
    // One-wave workgroups, 12 resident per CU.  A lone wave takes ~0.9 us per row (load -> LDS -> solve -> store is one
    // dependent chain), so even a small level is cut into segments: tall enough to repay the march's start, many enough
    // for a few waves per resident slot (measured at 4K x 16: 4096 / 8192 / 12288 / 16384 waves -> 5.03 / 4.92 / 4.88 /
    // 4.86 ms for all levels).  The segments' carries come from k_blur_carry + k_carry_scan (M is in memory: the pre-pass
    // reads the entering and the leaving row of every step; ~17 us for the two launches at a small level), never from
    // inside the launch; a level of <= 40 rows is marched whole (35 us against 31 us at 60 x 34).
    static const long waves_wanted = tune("TF_BLUR_WAVES", 12288);
    March mc;
    mc.mode = 0;
    {
        const long segs_wanted = std::max(1l, waves_wanted / std::max(1l, (long)strips * n_pairs));
        int seg = (int)std::min<long>(256, std::max<long>(16, (h + segs_wanted - 1) / segs_wanted));
        if (h <= 40)
            seg = h;
        if (option(OPT_FB_SEGS) > 0)
            seg = std::max(1, (int)((h + option(OPT_FB_SEGS) - 1) / option(OPT_FB_SEGS)));
        mc.seg = seg;
        mc.segs = (int)cdiv(h, seg);
    }
    const double *carry = nullptr;
    if (mc.segs > 1) {
        TF_TRY(fb_carry_room(fb, (size_t)mc.segs * n_pairs * 5 * w, 0));
        TF_TRY(launch(lvl_name("fb_blur_carry", k), k_blur_carry, dim3(cdiv(w, 64), 5 * mc.segs, n_pairs), dim3(64), 0,
                      (const float *)fb->M.as<float>(), fb->col_carry.as<double>(), w, h, M, mc.seg));
        TF_TRY(fb_carry_scan(fb, w, n_pairs, mc.segs, k));
        carry = fb->col_carry.as<double>();
    }
    dim3 grid(strips, mc.segs, n_pairs);
    return launch(lvl_name("fb_blur_solve", k), k_blur_solve_wave<M>, grid, dim3(64), 0, (const float *)fb->M.as<float>(),
                  flow_out, w, h, scale, mc.seg, carry);
}

int fb_blur_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k)
{
    const int m = fb->prm.winsize / 2;
    const double scale = 1. / ((double)fb->prm.winsize * fb->prm.winsize);
    if (fb_exact(fb)) // OpenCV's own running sums, in its order (fb_exact.hip)
        return fb_exact_from_matrices(fb, w, h, n_pairs, flow_out, k);
    switch (m) {
    case 2: return launch_blur_solve_wave<2>(fb, w, h, n_pairs, flow_out, scale, k);
    case 3: return launch_blur_solve_wave<3>(fb, w, h, n_pairs, flow_out, scale, k);
    case 4: return launch_blur_solve_wave<4>(fb, w, h, n_pairs, flow_out, scale, k);
    case 5: return launch_blur_solve_wave<5>(fb, w, h, n_pairs, flow_out, scale, k);
    case 6: return launch_blur_solve_wave<6>(fb, w, h, n_pairs, flow_out, scale, k);
    case 7: return launch_blur_solve_wave<7>(fb, w, h, n_pairs, flow_out, scale, k);
    case 8: return launch_blur_solve_wave<8>(fb, w, h, n_pairs, flow_out, scale, k);
    case 10: return launch_blur_solve_wave<10>(fb, w, h, n_pairs, flow_out, scale, k);
    case 12: return launch_blur_solve_wave<12>(fb, w, h, n_pairs, flow_out, scale, k);
    default: break;
    }
    if (m == 0)
        return fb_w1_solve(fb, w, h, n_pairs, flow_out);
    // any other window: the generic block-per-strip kernel
    const int out_cols = BS_THREADS - 2 * m;
    const unsigned strips = cdiv(w, out_cols);
    long segs_wanted = std::max(1l, 1024 / std::max(1l, (long)strips * n_pairs));
    if (option(OPT_FB_SEGS) > 0)
        segs_wanted = option(OPT_FB_SEGS);
    const int seg = (int)std::min<long>(h, std::max<long>(16, (h + segs_wanted - 1) / segs_wanted));
    const int segs = (int)cdiv(h, seg);
    const double *carry = nullptr;
    if (segs > 1) {
        TF_TRY(fb_carry_room(fb, (size_t)segs * n_pairs * 5 * w, 0));
        TF_TRY(launch(lvl_name("fb_blur_carry", k), k_blur_carry, dim3(cdiv(w, 64), 5 * segs, n_pairs), dim3(64), 0,
                      (const float *)fb->M.as<float>(), fb->col_carry.as<double>(), w, h, m, seg));
        TF_TRY(fb_carry_scan(fb, w, n_pairs, segs, k));
        carry = fb->col_carry.as<double>();
    }
    dim3 grid(strips, segs, n_pairs);
    return launch("fb_blur_solve_generic", k_blur_solve, grid, dim3(BS_THREADS), 0, (const float *)fb->M.as<float>(),
                  flow_out, w, h, m, scale, seg, carry);
}

int fb_gauss_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k)
{
    const int m = fb->prm.winsize / 2;
    const size_t smem = ((size_t)(GS_TH + 2 * m) * (GS_TW + 2 * m) + (size_t)GS_TH * (GS_TW + 2 * m) + m + 1) * sizeof(float);
    dim3 grid(cdiv(w, GS_TW), cdiv(h, GS_TH), n_pairs);
    return launch(lvl_name("fb_gauss_solve", k), k_gauss_solve, grid, dim3(256), smem, (const float *)fb->M.as<float>(), flow_out,
                  w, h, m, (const float *)fb->gauss_taps.as<float>());
}

// computeResizeAreaTab (imgproc/resize.cpp) for one axis: entries grouped by destination index.
static void area_axis(int ssize, int dsize, std::vector<int> &si, std::vector<int> &start, std::vector<float> &alpha)
{
    const double scale = (double)ssize / dsize;
    start.assign(dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        start[dx] = (int)si.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) {
            si.push_back(sx1 - 1);
            alpha.push_back((float)((sx1 - fsx1) / cell));
        }
        for (int sx = sx1; sx < sx2; sx++) {
            si.push_back(sx);
            alpha.push_back((float)(1.0 / cell));
        }
        if (fsx2 - sx2 > 1e-3) {
            si.push_back(sx2);
            alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell));
        }
    }
    start[dsize] = (int)si.size();
}

int fb_setup_flags(tf_fb *fb)
{
    if (fb->gaussian()) { // FarnebackUpdateFlow_GaussianBlur's taps: exp in double -> float, normalised by the double sum
        const int m = fb->prm.winsize / 2;
        std::vector<float> k(m + 1);
        const double sigma = m * 0.3;
        double sum = 1;
        k[0] = 1.f;
        for (int i = 1; i <= m; i++) {
            k[i] = (float)std::exp(-i * i / (2 * sigma * sigma));
            sum += k[i] * 2;
        }
        sum = 1. / sum;
        for (int i = 0; i <= m; i++)
            k[i] = (float)(k[i] * sum);
        TF_TRY(fb->gauss_taps.alloc(k.size() * 4));
        TF_HIP(hipMemcpy(fb->gauss_taps.p, k.data(), k.size() * 4, hipMemcpyHostToDevice));
    }
    if (fb->use_initial()) {
        const size_t N0 = (size_t)fb->W * fb->H;
        TF_TRY(fb->init_flow.alloc((size_t)fb->max_pairs * N0 * 8));
        TF_HIP(hipMemset(fb->init_flow.p, 0, (size_t)fb->max_pairs * N0 * 8));
        const Level &C = *fb->lv[fb->K];
        const double sx = (double)fb->W / C.W, sy = (double)fb->H / C.H;
        const int ix = (int)std::lrint(sx), iy = (int)std::lrint(sy);
        memset(&fb->area, 0, sizeof(fb->area));
        if (std::fabs(sx - ix) < DBL_EPSILON && std::fabs(sy - iy) < DBL_EPSILON) {
            fb->area.ix = ix;
            fb->area.iy = iy;
        } else {
            std::vector<int> xsi, xst, ysi, yst;
            std::vector<float> xa, ya;
            area_axis(fb->W, C.W, xsi, xst, xa);
            area_axis(fb->H, C.H, ysi, yst, ya);
            std::vector<int> ints;
            ints.insert(ints.end(), xsi.begin(), xsi.end());
            ints.insert(ints.end(), xst.begin(), xst.end());
            ints.insert(ints.end(), ysi.begin(), ysi.end());
            ints.insert(ints.end(), yst.begin(), yst.end());
            std::vector<float> fl(xa);
            fl.insert(fl.end(), ya.begin(), ya.end());
            TF_TRY(fb->area_i.alloc(ints.size() * 4));
            TF_TRY(fb->area_f.alloc(fl.size() * 4));
            TF_HIP(hipMemcpy(fb->area_i.p, ints.data(), ints.size() * 4, hipMemcpyHostToDevice));
            TF_HIP(hipMemcpy(fb->area_f.p, fl.data(), fl.size() * 4, hipMemcpyHostToDevice));
            const int *bi = fb->area_i.as<int>();
            const float *bf = fb->area_f.as<float>();
            fb->area.xsi = bi;
            fb->area.xstart = bi + xsi.size();
            fb->area.ysi = bi + xsi.size() + xst.size();
            fb->area.ystart = bi + xsi.size() + xst.size() + ysi.size();
            fb->area.xalpha = bf;
            fb->area.yalpha = bf + xa.size();
        }
    }
    return TF_OK;
}

// The coarsest scale's flow from the pairs' initial flows: resize(INTER_AREA) * scale  (flag 4).
int fb_initial_flow(tf_fb *fb, int n_pairs, float2 *out)
{
    const Level &C = *fb->lv[fb->K];
    double scale = 1;
    for (int i = 0; i < fb->K; i++)
        scale *= fb->prm.pyr_scale;
    dim3 block(64), grid(cdiv(C.W, 64), C.H, n_pairs);
    return launch("fb_initial_flow", k_flow_area_init, grid, block, 0, (const float2 *)fb->init_flow.as<float2>(), out, fb->W,
                  fb->H, C.W, C.H, fb->area, (float)scale);
}

} // namespace fb
} // namespace tf
