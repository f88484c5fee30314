/*
 * CPU oracle for the Farnebäck half of the hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under transflow_amd/ may link, load or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / reported CPU baseline.
 *
 * What it restates: the arithmetic behind the single call the reference makes
 * on this path, cv2.calcOpticalFlowFarneback(prev, next, flow, pyr_scale,
 * levels, winsize, iterations, poly_n, poly_sigma, flags)
 * (reference: transflow/flow/sources/cv.py:479-490; defaults cv.py:273-281).
 * That arithmetic lives in the third-party dependency `opencv-python`
 * (requirements.txt:3, setup.py:28 -- UNPINNED, no version specifier), which is
 * not vendored under /root/reference and is not installed in the build image.
 * This file restates the published algorithm of OpenCV 4.x's CPU path,
 * modules/video/src/optflowgf.cpp (FarnebackOpticalFlowImpl::calc,
 * FarnebackPrepareGaussian, FarnebackPolyExp, FarnebackUpdateMatrices,
 * FarnebackUpdateFlow_Blur) plus the imgproc pieces it calls (GaussianBlur's
 * separable float filter with BORDER_REFLECT_101, getGaussianKernel, and
 * resize INTER_LINEAR), following SURVEY.md Appendix A.
 *
 * PARITY UNPINNED: the reference's own tests hold no flow values for this
 * path (tests/test_flow_source.py:22-33 check type/shape/dtype only) and cv2
 * cannot be run here, so nothing pins these numbers to OpenCV's.  What pins
 * the restatement is analytic behaviour (tests/test_oracle_farneback.py):
 * zero flow on identical frames, recovery of known translations, exact
 * polynomial-expansion coefficients of a quadratic image, and a cv2-gated
 * comparison that runs wherever `import cv2` works.
 *
 * Float/double discipline follows OpenCV statement by statement (which
 * products are float, which accumulators are double); build with
 * -ffp-contract=off so no FMA contraction changes the rounding.
 *
 * Supported flags (cv.py:281, 489 pass CvFlowConfig.fb_flags straight through): 0 (transflow's default),
 * OPTFLOW_USE_INITIAL_FLOW (4: the caller's flow, which cv.py:478 fills with the previous output, is
 * shrunk with resize(INTER_AREA) to the coarsest scale and multiplied by that scale) and
 * OPTFLOW_FARNEBACK_GAUSSIAN (256: FarnebackUpdateFlow_GaussianBlur, a separable float Gaussian of the
 * 2x2 systems instead of the box window), alone or together.  Both are restated from the same upstream
 * file and are as unpinned as the rest; of FarnebackUpdateFlow_GaussianBlur the SCALAR loops are
 * restated (upstream's SIMD body uses v_muladd, which fuses on FMA builds: a 1-ulp build dependence).
 *
 * [VERIFY] list -- details of the upstream code that this restatement assumes and that only a live
 * cv2 can confirm (the cv2-gated tests in tests/test_oracle_farneback.py run wherever cv2 imports):
 *   1. scales run k = levels..0 (levels + 1 scales), stopping early below 32 px      
 *   2. poly_n is the RADIUS of the expansion's taps (2n + 1 of them)                
 *   3. Gaussian taps are computed and normalised in double, then cast to float     
 *   4. cv::resize(INTER_LINEAR) switches to its INTER_AREA fast path when both scale factors are
 *      exactly 2 -- level 1 of a 0.5 pyramid over even frame sizes.  That path averages the 2x2
 *      block; its vector body computes ((a + b) + (c + d)) * 0.25f, which is bit-identical to the
 *      bilinear statement used here (both fractions are exactly 0.5), but its scalar tail computes
 *      (((a + b) + c) + d) * 0.25f, which can differ by 1 ulp in the last few columns of a row.
 *      Not restated: which columns fall to the tail depends on the build's SIMD width.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FB_EXPORT __attribute__((visibility("default")))

static inline int cv_round(double v) { return (int)lrint(v); } /* half to even */
static inline int cv_floorf(float v) { return (int)floorf(v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---------------------------------------------------------------------- */
/* imgproc: getGaussianKernel(n, sigma, CV_32F)                            */
/* sigma<=0 with n in {1,3,5,7}: fixed tables; else exp() in double,       */
/* normalised in double, cast to float.                       */
/* ---------------------------------------------------------------------- */
FB_EXPORT void fbref_gaussian_kernel(int n, double sigma, float *out)
{
    static const float t1[] = {1.f};
    static const float t3[] = {0.25f, 0.5f, 0.25f};
    static const float t5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float t7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    if (sigma <= 0 && (n == 1 || n == 3 || n == 5 || n == 7)) {
        const float *t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : t7;
        memcpy(out, t, sizeof(float) * (size_t)n);
        return;
    }
    double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2x = -0.5 / (sx * sx);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        v[i] = exp(scale2x * x * x);
        sum += v[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++)
        out[i] = (float)(v[i] * sum);
    free(v);
}

static inline int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    }
    return p;
}

/* ---------------------------------------------------------------------- */
/* A1a: convertTo(CV_32F) + GaussianBlur(ksz x ksz, sigma), REFLECT_101.    */
/* Row pass then column pass, float accumulation.  Tap order as OpenCV's   */
/* scalar filters: row ksz<=5 symmetric-paired, row ksz>5 left-to-right;   */
/* column centre first, then pairs outwards.                               */
/* ---------------------------------------------------------------------- */
FB_EXPORT void fbref_gaussian_blur_u8(const uint8_t *src, int W, int H, int ksz, double sigma, float *dst)
{
    float *k = (float *)malloc(sizeof(float) * (size_t)ksz);
    fbref_gaussian_kernel(ksz, sigma, k);
    int r = ksz / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)W * H);
    int *xi = (int *)malloc(sizeof(int) * (size_t)(W + 2 * r));
    for (int x = -r; x < W + r; x++)
        xi[x + r] = reflect101(x, W);
    for (int y = 0; y < H; y++) {
        const uint8_t *s = src + (size_t)y * W;
        float *t = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            const int *xx = xi + x + r; /* xx[d] = source column of tap d */
            float acc;
            if (ksz == 1) {
                acc = (float)s[x] * k[0];
            } else if (ksz == 3) {
                acc = (float)s[xx[0]] * k[1] + ((float)s[xx[-1]] + (float)s[xx[1]]) * k[2];
            } else if (ksz == 5) {
                acc = (float)s[xx[0]] * k[2] + ((float)s[xx[-1]] + (float)s[xx[1]]) * k[3] +
                      ((float)s[xx[-2]] + (float)s[xx[2]]) * k[4];
            } else {
                acc = k[0] * (float)s[xx[-r]];
                for (int i = 1; i < ksz; i++)
                    acc += k[i] * (float)s[xx[i - r]];
            }
            t[x] = acc;
        }
    }
    for (int y = 0; y < H; y++) {
        float *d = dst + (size_t)y * W;
        const float *c = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float acc = k[r] * c[x];
            for (int i = 1; i <= r; i++) {
                const float *a = tmp + (size_t)reflect101(y + i, H) * W;
                const float *b = tmp + (size_t)reflect101(y - i, H) * W;
                acc += k[r + i] * (a[x] + b[x]);
            }
            d[x] = acc;
        }
    }
    free(xi);
    free(tmp);
    free(k);
}

/* ---------------------------------------------------------------------- */
/* imgproc: resize(..., INTER_LINEAR) for float images with cn channels.   */
/* Same size: copy.  Coefficients: fx = (float)((dx+0.5)*scale - 0.5) with */
/* scale = 1/(dst/src) in double; edge handling as resize.cpp (sx<0 ->     */
/* (0, fx=0); sx>=W-1 -> (W-1, fx=0)); rows clamped, beta kept.            */
/* Horizontal lerp first, then vertical, float.                            */
/* ---------------------------------------------------------------------- */
typedef struct {
    int *ofs;
    float *a0, *a1;
} lerp_tab;

static void make_lerp(int src, int dst, lerp_tab *t, int zero_at_edges)
{
    t->ofs = (int *)malloc(sizeof(int) * (size_t)dst);
    t->a0 = (float *)malloc(sizeof(float) * (size_t)dst);
    t->a1 = (float *)malloc(sizeof(float) * (size_t)dst);
    double inv_scale = (double)dst / src;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floorf(f);
        f -= s;
        if (zero_at_edges) {
            if (s < 0) {
                f = 0;
                s = 0;
            }
            if (s >= src - 1) {
                f = 0;
                s = src - 1;
            }
        }
        t->ofs[d] = s;
        t->a0[d] = 1.f - f;
        t->a1[d] = f;
    }
}

static void free_lerp(lerp_tab *t)
{
    free(t->ofs);
    free(t->a0);
    free(t->a1);
}

FB_EXPORT void fbref_resize_linear(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    if (sw == dw && sh == dh) {
        memcpy(dst, src, sizeof(float) * (size_t)sw * sh * cn);
        return;
    }
#ifdef FBREF_AREA_TAIL
    /* Sensitivity variant ([VERIFY] 4): at exactly half size cv::resize(INTER_LINEAR) takes INTER_AREA's fast
     * path; this build gives EVERY column the order of its scalar tail, (((a + b) + c) + d) * 0.25f -- the
     * widest a real build's mix of vector body and tail can be from the default build's bilinear statement. */
    if (cn == 1 && sw == 2 * dw && sh == 2 * dh) {
        for (int dy = 0; dy < dh; dy++) {
            const float *s0 = src + (size_t)(2 * dy) * sw, *s1 = s0 + sw;
            for (int dx = 0; dx < dw; dx++)
                dst[(size_t)dy * dw + dx] = (((s0[2 * dx] + s0[2 * dx + 1]) + s1[2 * dx]) + s1[2 * dx + 1]) * 0.25f;
        }
        return;
    }
#endif
    lerp_tab tx, ty;
    make_lerp(sw, dw, &tx, 1);
    make_lerp(sh, dh, &ty, 0);
    float *row0 = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    float *row1 = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = ty.ofs[dy], sy1 = sy0 + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);
        sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
        const float *s0 = src + (size_t)sy0 * sw * cn;
        const float *s1 = src + (size_t)sy1 * sw * cn;
        for (int dx = 0; dx < dw; dx++) {
            int sx = tx.ofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx; /* weight is 0 there */
            for (int c = 0; c < cn; c++) {
                if (sx >= sw - 1) { /* dx >= xmax: D = S[sx]*1 */
                    row0[dx * cn + c] = s0[sx * cn + c];
                    row1[dx * cn + c] = s1[sx * cn + c];
                } else {
                    row0[dx * cn + c] = s0[sx * cn + c] * tx.a0[dx] + s0[sx1 * cn + c] * tx.a1[dx];
                    row1[dx * cn + c] = s1[sx * cn + c] * tx.a0[dx] + s1[sx1 * cn + c] * tx.a1[dx];
                }
            }
        }
        float b0 = ty.a0[dy], b1 = ty.a1[dy];
        float *d = dst + (size_t)dy * dw * cn;
        for (int i = 0; i < dw * cn; i++)
            d[i] = row0[i] * b0 + row1[i] * b1;
    }
    free(row0);
    free(row1);
    free_lerp(&tx);
    free_lerp(&ty);
}

/* ---------------------------------------------------------------------- */
/* A2: FarnebackPrepareGaussian + FarnebackPolyExp             */
/* ---------------------------------------------------------------------- */
static void cholesky_inverse6(const double G[36], double inv[36])
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j <= i; j++) {
            double s = G[i * 6 + j];
            for (int k = 0; k < j; k++)
                s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = (i == j) ? sqrt(s) : s / L[j * 6 + j];
        }
    }
    for (int c = 0; c < 6; c++) {
        double y[6], x[6];
        for (int i = 0; i < 6; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++)
                s -= L[i * 6 + k] * y[k];
            y[i] = s / L[i * 6 + i];
        }
        for (int i = 5; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 6; k++)
                s -= L[k * 6 + i] * x[k];
            x[i] = s / L[i * 6 + i];
        }
        for (int i = 0; i < 6; i++)
            inv[i * 6 + c] = x[i];
    }
}

/* g, xg, xxg point at the centre of arrays of 2n+1 floats; ig = {ig11, ig03, ig33, ig55} */
FB_EXPORT void fbref_prepare_gaussian(int n, double sigma, float *g, float *xg, float *xxg, double ig[4])
{
    if (sigma < FLT_EPSILON)
        sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[36];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++) {
        for (int x = -n; x <= n; x++) {
            float gg = g[y] * g[x]; /* float products, double accumulation */
            G[0] += gg;
            G[1 * 6 + 1] += gg * x * x;
            G[3 * 6 + 3] += gg * x * x * x * x;
            G[5 * 6 + 5] += gg * x * x * y * y;
        }
    }
    G[2 * 6 + 2] = G[0 * 6 + 3] = G[0 * 6 + 4] = G[3 * 6 + 0] = G[4 * 6 + 0] = G[1 * 6 + 1];
    G[4 * 6 + 4] = G[3 * 6 + 3];
    G[3 * 6 + 4] = G[4 * 6 + 3] = G[5 * 6 + 5];
    double inv[36];
    cholesky_inverse6(G, inv);
    ig[0] = inv[1 * 6 + 1];
    ig[1] = inv[0 * 6 + 3];
    ig[2] = inv[3 * 6 + 3];
    ig[3] = inv[5 * 6 + 5];
}

/* dst: [H][W][5] interleaved, as OpenCV's CV_32FC5 */
FB_EXPORT void fbref_polyexp(const float *src, int W, int H, int n, double sigma, float *dst)
{
    float *kbuf = (float *)malloc(sizeof(float) * (size_t)(n * 6 + 3));
    float *g = kbuf + n, *xg = g + n * 2 + 1, *xxg = xg + n * 2 + 1;
    double ig[4];
    fbref_prepare_gaussian(n, sigma, g, xg, xxg, ig);
    double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
    float *rowbuf = (float *)malloc(sizeof(float) * (size_t)(W + n * 2) * 3);
    float *row = rowbuf + n * 3;
    for (int y = 0; y < H; y++) {
        float g0 = g[0], g1, g2;
        const float *srow0 = src + (size_t)y * W, *srow1;
        float *drow = dst + (size_t)y * W * 5;
        for (int x = 0; x < W; x++) { /* vertical part, float */
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k];
            g1 = xg[k];
            g2 = xxg[k];
            srow0 = src + (size_t)imax(y - k, 0) * W;
            srow1 = src + (size_t)imin(y + k, H - 1) * W;
            for (int x = 0; x < W; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0;
                row[x * 3 + 1] = t1;
                row[x * 3 + 2] = t2;
            }
        }
        for (int x = 0; x < n * 3; x++) { /* replicate the edge triples */
            row[-1 - x] = row[2 - x];
            row[W * 3 + x] = row[W * 3 + x - 3];
        }
#ifdef FBREF_POLY_F32
        /* Sensitivity variant: the horizontal part accumulated in FLOAT with fused multiply-adds -- the
         * arithmetic of the HIP library's default (non-exact) mode, restated here only to measure on the CPU
         * how far such a build moves the flow (tests/test_oracle_farneback.py: the envelope).  Never the
         * parity target: that is the default build below. */
        for (int x = 0; x < W; x++) {
            g0 = g[0];
            float b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                float tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 = fmaf(tg, g0, b1);
                b4 = fmaf(tg, xxg[k], b4);
                b2 = fmaf(row[(x + k) * 3] - row[(x - k) * 3], xg[k], b2);
                b3 = fmaf(row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1], g0, b3);
                b6 = fmaf(row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1], xg[k], b6);
                b5 = fmaf(row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2], g0, b5);
            }
#if FBREF_POLY_F32 >= 2 /* ... and the combination with the inverse's entries in float as well */
            const float f11 = (float)ig11, f03 = (float)ig03, f33 = (float)ig33, f55 = (float)ig55;
            const float b103 = b1 * f03;
            drow[x * 5 + 1] = b2 * f11;
            drow[x * 5] = b3 * f11;
            drow[x * 5 + 3] = fmaf(b4, f33, b103);
            drow[x * 5 + 2] = fmaf(b5, f33, b103);
            drow[x * 5 + 4] = b6 * f55;
#else
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
#endif
        }
        continue;
#endif
        for (int x = 0; x < W; x++) { /* horizontal part, double accumulators */
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(rowbuf);
    free(kbuf);
}

/* ---------------------------------------------------------------------- */
/* A3: FarnebackUpdateMatrices on rows [y0, y1)               */
/* ---------------------------------------------------------------------- */
FB_EXPORT void fbref_update_matrices(const float *R0, const float *R1all, const float *flowall, float *Mall, int W, int H,
                                     int y0, int y1)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    size_t step1 = (size_t)W * 5;
    for (int y = y0; y < y1; y++) {
        const float *flow = flowall + (size_t)y * W * 2;
        const float *R0r = R0 + (size_t)y * W * 5;
        float *M = Mall + (size_t)y * W * 5;
        for (int x = 0; x < W; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floorf(fx), yy1 = cv_floorf(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1;
            fy -= yy1;
            if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)yy1 < (unsigned)(H - 1)) {
                const float *ptr = R1all + (size_t)yy1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0r[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0r[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0r[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0r[x * 5 + 2];
                r5 = R0r[x * 5 + 3];
                r6 = R0r[x * 5 + 4] * 0.5f;
            }
            r2 = (R0r[x * 5] - r2) * 0.5f;
            r3 = (R0r[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(W - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(H - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) * (x >= W - BORDER ? border[W - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) * (y >= H - BORDER ? border[H - y - 1] : 1.f);
                r2 *= scale;
                r3 *= scale;
                r4 *= scale;
                r5 *= scale;
                r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

/* ---------------------------------------------------------------------- */
/* A4: FarnebackUpdateFlow_Blur: running box sums in      */
/* double, 2x2 solve with +1e-3, interleaved stripe updates of M.          */
/* ---------------------------------------------------------------------- */
FB_EXPORT void fbref_update_flow_blur(const float *R0, const float *R1, float *flowall, float *M, int W, int H,
                                      int block_size, int update_matrices)
{
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = imax((1 << 10) / W, block_size);
    double scale = 1. / (block_size * block_size);
    double *vbuf = (double *)malloc(sizeof(double) * (size_t)(W + m * 2 + 2) * 5);
    double *vsum = vbuf + (m + 1) * 5;
    const float *srow0 = M;
    for (int x = 0; x < W * 5; x++)
        vsum[x] = srow0[x] * (m + 2); /* float product */
    for (int y = 1; y < m; y++) {
        srow0 = M + (size_t)imin(y, H - 1) * W * 5;
        for (int x = 0; x < W * 5; x++)
            vsum[x] += srow0[x];
    }
    for (int y = 0; y < H; y++) {
        double g11, g12, g22, h1, h2;
        float *flow = flowall + (size_t)y * W * 2;
        srow0 = M + (size_t)imax(y - m - 1, 0) * W * 5;
        const float *srow1 = M + (size_t)imin(y + m, H - 1) * W * 5;
        for (int x = 0; x < W * 5; x++)
            vsum[x] += srow1[x] - srow0[x]; /* float difference */
        for (int x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[W * 5 + x] = vsum[W * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2);
        g12 = vsum[1] * (m + 2);
        g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2);
        h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; x++) {
            g11 += vsum[x * 5];
            g12 += vsum[x * 5 + 1];
            g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3];
            h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < W; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale;
            double h1_ = h1 * scale, h2_ = h2 * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
        y1 = y == H - 1 ? H : y - block_size;
        if (update_matrices && (y1 == H || y1 >= y0 + min_update_stripe)) {
            fbref_update_matrices(R0, R1, flowall, M, W, H, y0, y1);
            y0 = y1;
        }
    }
    free(vbuf);
}

/* ---------------------------------------------------------------------- */
/* imgproc: resize(..., INTER_AREA) shrinking a float image with cn        */
/* channels (what OPTFLOW_USE_INITIAL_FLOW does to the caller's flow).     */
/* Integer factors: resizeAreaFast_ -- every destination value is the sum  */
/* of its sy x sx block, four at a time (CV_ENABLE_UNROLLED), times        */
/* 1/area, float.  Otherwise: resizeArea_ with computeResizeAreaTab's      */
/* fractional cell coverage (tables in double, weights cast to float),     */
/* rows accumulated in float: buf = sum_k S*alpha_k, sum = beta0*buf, then  */
/* sum += beta*buf.                                                        */
/* ---------------------------------------------------------------------- */
typedef struct {
    int si, di;
    float alpha;
} area_tab;

static int make_area_tab(int ssize, int dsize, double scale, area_tab *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = imin(sx2, ssize - 1);
        sx1 = imin(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx;
            tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cell);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx;
            tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cell);
        }
        if (fsx2 - sx2 > 1e-3) {
            double w = fsx2 - sx2;
            w = w < 1. ? w : 1.;
            w = w < cell ? w : cell;
            tab[k].di = dx;
            tab[k].si = sx2;
            tab[k++].alpha = (float)(w / cell);
        }
    }
    return k;
}

/* Exported so that the tests and the GPU build share one statement of the tables. */
FB_EXPORT int fbref_area_tab(int ssize, int dsize, int *si, int *di, float *alpha)
{
    area_tab *t = (area_tab *)malloc(sizeof(area_tab) * (size_t)(ssize * 2 + 2));
    int n = make_area_tab(ssize, dsize, (double)ssize / dsize, t);
    for (int i = 0; i < n; i++) {
        si[i] = t[i].si;
        di[i] = t[i].di;
        alpha[i] = t[i].alpha;
    }
    free(t);
    return n;
}

FB_EXPORT void fbref_resize_area(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int ix = (int)lrint(scale_x), iy = (int)lrint(scale_y);
    if (fabs(scale_x - ix) < DBL_EPSILON && fabs(scale_y - iy) < DBL_EPSILON) {
        int area = ix * iy;
        float scale = 1.f / area;
        int *ofs = (int *)malloc(sizeof(int) * (size_t)area);
        for (int sy = 0, k = 0; sy < iy; sy++)
            for (int sx = 0; sx < ix; sx++)
                ofs[k++] = (sy * sw + sx) * cn;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    const float *S = src + ((size_t)dy * iy * sw + (size_t)dx * ix) * cn + c;
                    float sum = 0;
                    int k = 0;
                    for (; k <= area - 4; k += 4)
                        sum += S[ofs[k]] + S[ofs[k + 1]] + S[ofs[k + 2]] + S[ofs[k + 3]];
                    for (; k < area; k++)
                        sum += S[ofs[k]];
                    dst[((size_t)dy * dw + dx) * cn + c] = sum * scale;
                }
        free(ofs);
        return;
    }
    area_tab *xt = (area_tab *)malloc(sizeof(area_tab) * (size_t)(sw * 2 + 2));
    area_tab *yt = (area_tab *)malloc(sizeof(area_tab) * (size_t)(sh * 2 + 2));
    int nx = make_area_tab(sw, dw, scale_x, xt), ny = make_area_tab(sh, dh, scale_y, yt);
    float *buf = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    float *sum = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    int prev_dy = yt[0].di;
    for (int i = 0; i < dw * cn; i++)
        sum[i] = 0;
    for (int j = 0; j < ny; j++) {
        float beta = yt[j].alpha;
        int dy = yt[j].di;
        const float *S = src + (size_t)yt[j].si * sw * cn;
        for (int i = 0; i < dw * cn; i++)
            buf[i] = 0;
        for (int k = 0; k < nx; k++)
            for (int c = 0; c < cn; c++)
                buf[xt[k].di * cn + c] = buf[xt[k].di * cn + c] + S[xt[k].si * cn + c] * xt[k].alpha;
        if (dy != prev_dy) {
            float *D = dst + (size_t)prev_dy * dw * cn;
            for (int i = 0; i < dw * cn; i++) {
                D[i] = sum[i];
                sum[i] = beta * buf[i];
            }
            prev_dy = dy;
        } else {
            for (int i = 0; i < dw * cn; i++)
                sum[i] += beta * buf[i];
        }
    }
    {
        float *D = dst + (size_t)prev_dy * dw * cn;
        for (int i = 0; i < dw * cn; i++)
            D[i] = sum[i];
    }
    free(buf);
    free(sum);
    free(xt);
    free(yt);
}

/* ---------------------------------------------------------------------- */
/* A4': FarnebackUpdateFlow_GaussianBlur (OPTFLOW_FARNEBACK_GAUSSIAN):      */
/* kernel[i] = exp(-i^2 / (2 sigma^2)), sigma = m * 0.3, m = winsize / 2,   */
/* taps as float, normalised by their double sum; vertical then horizontal */
/* pass in FLOAT, centre first then pairs outwards, replicate border;      */
/* 2x2 solve with +1e-3 in double; the same interleaved stripe updates of M.*/
/* ---------------------------------------------------------------------- */
FB_EXPORT void fbref_gaussian_window(int m, float *kernel /* m + 1 taps */)
{
    double sigma = m * 0.3, s = 1;
    kernel[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++)
        kernel[i] = (float)(kernel[i] * s);
}

FB_EXPORT void fbref_update_flow_gaussian(const float *R0, const float *R1, float *flowall, float *M, int W, int H,
                                          int block_size, int update_matrices)
{
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = imax((1 << 10) / W, block_size);
    float *kernel = (float *)malloc(sizeof(float) * (size_t)(m + 1));
    fbref_gaussian_window(m, kernel);
    float *vbuf = (float *)malloc(sizeof(float) * (size_t)(W + m * 2 + 2) * 5);
    float *vsum = vbuf + (m + 1) * 5;
    float *hsum = (float *)malloc(sizeof(float) * (size_t)W * 5);
    for (int y = 0; y < H; y++) {
        float *flow = flowall + (size_t)y * W * 2;
        for (int x = 0; x < W * 5; x++) {
            float s0 = M[(size_t)y * W * 5 + x] * kernel[0];
            for (int i = 1; i <= m; i++)
                s0 += (M[(size_t)imin(y + i, H - 1) * W * 5 + x] + M[(size_t)imax(y - i, 0) * W * 5 + x]) * kernel[i];
            vsum[x] = s0;
        }
        for (int x = 0; x < m * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[W * 5 + x] = vsum[W * 5 + x - 5];
        }
        for (int x = 0; x < W * 5; x++) {
            float sum = vsum[x] * kernel[0];
            for (int i = 1; i <= m; i++)
                sum += kernel[i] * (vsum[x - i * 5] + vsum[x + i * 5]);
            hsum[x] = sum;
        }
        for (int x = 0; x < W; x++) {
            double g11 = hsum[x * 5], g12 = hsum[x * 5 + 1], g22 = hsum[x * 5 + 2], h1 = hsum[x * 5 + 3], h2 = hsum[x * 5 + 4];
            double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            flow[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
        }
        y1 = y == H - 1 ? H : y - block_size;
        if (update_matrices && (y1 == H || y1 >= y0 + min_update_stripe)) {
            fbref_update_matrices(R0, R1, flowall, M, W, H, y0, y1);
            y0 = y1;
        }
    }
    free(hsum);
    free(vbuf);
    free(kernel);
}

/* ---------------------------------------------------------------------- */
/* A.1 driver: level schedule shared with the tests                        */
/* ---------------------------------------------------------------------- */
/* Returns the number K of usable coarse scales (scales are k = K..0). */
FB_EXPORT int fbref_num_levels(int W, int H, double pyr_scale, int levels)
{
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (W * scale < min_size || H * scale < min_size)
            break;
    }
    return k;
}

/* out[0..3] = Wk, Hk, ksz ; sigma returned */
FB_EXPORT double fbref_level_geometry(int W, int H, double pyr_scale, int k, int *Wk, int *Hk, int *ksz)
{
    double scale = 1;
    for (int i = 0; i < k; i++)
        scale *= pyr_scale;
    double sigma = (1. / scale - 1) * 0.5;
    int smooth_sz = cv_round(sigma * 5) | 1;
    smooth_sz = imax(smooth_sz, 3);
    *Wk = cv_round(W * scale);
    *Hk = cv_round(H * scale);
    *ksz = smooth_sz;
    return sigma;
}

/* A1: one image -> pyramid level image (blur at full res, then resize) */
FB_EXPORT void fbref_level_image(const uint8_t *img, int W, int H, double pyr_scale, int k, float *out)
{
    int Wk, Hk, ksz;
    double sigma = fbref_level_geometry(W, H, pyr_scale, k, &Wk, &Hk, &ksz);
    float *blur = (float *)malloc(sizeof(float) * (size_t)W * H);
    fbref_gaussian_blur_u8(img, W, H, ksz, sigma, blur);
    fbref_resize_linear(blur, W, H, 1, out, Wk, Hk);
    free(blur);
}

/*
 * The whole call.  flow: [H][W][2] float32 (x = dx, y = dy); read first when flags has
 * OPTFLOW_USE_INITIAL_FLOW (4), output only otherwise.  OPTFLOW_FARNEBACK_GAUSSIAN (256) selects the
 * Gaussian window.  Returns 0, or -1 on bad args.
 */
FB_EXPORT int fbref_calc(const uint8_t *prev, const uint8_t *next, int W, int H, float *flow0, double pyr_scale,
                         int levels, int winsize, int iterations, int poly_n, double poly_sigma, int flags)
{
    if ((flags & ~(4 | 256)) != 0 || !(pyr_scale < 1) || W <= 0 || H <= 0 || winsize < 1 || poly_n < 1)
        return -1;
    const uint8_t *img[2] = {prev, next};
    int K = fbref_num_levels(W, H, pyr_scale, levels);
    float *prevFlow = NULL;
    int pW = 0, pH = 0;
    for (int k = K; k >= 0; k--) {
        int Wk, Hk, ksz;
        fbref_level_geometry(W, H, pyr_scale, k, &Wk, &Hk, &ksz);
        size_t nk = (size_t)Wk * Hk;
        float *flow = k > 0 ? (float *)malloc(sizeof(float) * nk * 2) : flow0;
        if (!prevFlow) {
            if (flags & 4) { /* resize(flow0, flow, (Wk, Hk), INTER_AREA); flow *= scale */
                double scale = 1;
                for (int i = 0; i < k; i++)
                    scale *= pyr_scale;
                float *tmp = k > 0 ? flow : (float *)malloc(sizeof(float) * nk * 2);
                fbref_resize_area(flow0, W, H, 2, tmp, Wk, Hk);
                for (size_t i = 0; i < nk * 2; i++)
                    flow[i] = tmp[i] * (float)scale;
                if (k == 0)
                    free(tmp);
            } else {
                memset(flow, 0, sizeof(float) * nk * 2);
            }
        } else {
            fbref_resize_linear(prevFlow, pW, pH, 2, flow, Wk, Hk);
            float mul = (float)(1. / pyr_scale);
            for (size_t i = 0; i < nk * 2; i++)
                flow[i] *= mul;
        }
        float *R[2], *I = (float *)malloc(sizeof(float) * nk), *M = (float *)malloc(sizeof(float) * nk * 5);
        for (int i = 0; i < 2; i++) {
            R[i] = (float *)malloc(sizeof(float) * nk * 5);
            fbref_level_image(img[i], W, H, pyr_scale, k, I);
            fbref_polyexp(I, Wk, Hk, poly_n, poly_sigma, R[i]);
        }
        fbref_update_matrices(R[0], R[1], flow, M, Wk, Hk, 0, Hk);
        for (int i = 0; i < iterations; i++) {
            if (flags & 256)
                fbref_update_flow_gaussian(R[0], R[1], flow, M, Wk, Hk, winsize, i < iterations - 1);
            else
                fbref_update_flow_blur(R[0], R[1], flow, M, Wk, Hk, winsize, i < iterations - 1);
        }
        free(R[0]);
        free(R[1]);
        free(I);
        free(M);
        free(prevFlow);
        prevFlow = flow;
        pW = Wk;
        pH = Hk;
    }
    /* prevFlow == flow0 here; not ours to free */
    return 0;
}

/*
 * The same call over n independent frame pairs, one pair per thread (OpenMP): with flags == 0 nothing
 * crosses pairs (cv.py:478-490), so every pair's result is bit-identical to fbref_calc's.  This is the
 * "all cores" leg of bench.py's cpu_baseline -- a timing aid, not a second statement of the algorithm.
 * Returns 0, or -1 if any pair's arguments were refused.
 */
FB_EXPORT int fbref_calc_batch(const uint8_t *const *prev, const uint8_t *const *next, int n, int threads, int W, int H,
                               float *const *flow, double pyr_scale, int levels, int winsize, int iterations, int poly_n,
                               double poly_sigma, int flags)
{
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(| : bad)
    for (int i = 0; i < n; i++)
        bad |= fbref_calc(prev[i], next[i], W, H, flow[i], pyr_scale, levels, winsize, iterations, poly_n, poly_sigma,
                          flags) != 0;
    return bad ? -1 : 0;
}
