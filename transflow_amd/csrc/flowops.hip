// Flow-array steps either side of the hot path, on gfx950:
//   flow merging            transflow/pipeline.py:149-158, transflow/utils.py:359-381
//   integer upscaling       transflow/utils.py:417-418
//   convolution kernel      transflow/flow/sources/source.py:344-348 (scipy.signal.convolve2d)
//   post_process in the convolution's type (float64 unless the kernel is float32), :349-362
//   flow visualisation      transflow/output/render.py:9-48
//   BGR -> grey             transflow/flow/sources/cv.py:461-466 (cv2.cvtColor) + nearest resize
// All element-wise or small-stencil, HBM-bound; arithmetic in the type and order numpy / scipy use.
#include <cstdlib>
#include <cstring>

#include "common.h"

using namespace tf;

namespace {

constexpr int BLOCK = 256;

struct MergeArgs {
    const float *in[TF_MAX_MERGE];
    int n;
};

// Pipeline.FLOW_MERGING_FUNCTIONS: float32, operands combined left to right
__global__ void k_flow_merge(MergeArgs a, float *__restrict__ out, size_t count, int kind)
{
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= count)
        return;
    const float f0 = a.in[0][t];
    float r = f0;
    switch (kind) {
    case TF_MERGE_FIRST:
        break;
    case TF_MERGE_SUM:
    case TF_MERGE_AVERAGE:
        for (int i = 1; i < a.n; i++)
            r = r + a.in[i][t];
        if (kind == TF_MERGE_AVERAGE)
            r = r / (float)a.n;
        break;
    case TF_MERGE_DIFFERENCE: { // flows[0] - sum(flows[1:])
        float s = 0.f;
        if (a.n > 1) {
            s = a.in[1][t];
            for (int i = 2; i < a.n; i++)
                s = s + a.in[i][t];
        }
        r = f0 - s;
        break;
    }
    case TF_MERGE_PRODUCT:
        for (int i = 1; i < a.n; i++)
            r = r * a.in[i][t];
        break;
    case TF_MERGE_MASKBIN: // utils.py:367-373: |x| > 0.2 (float32) -> 1, else 0
        for (int i = 1; i < a.n; i++)
            r = r * (fabsf(a.in[i][t]) > 0.2f ? 1.f : 0.f);
        break;
    case TF_MERGE_MASKLIN:
        for (int i = 1; i < a.n; i++)
            r = r * fabsf(a.in[i][t]);
        break;
    case TF_MERGE_ABSMAX: { // utils.py:376-381: numpy.argmax keeps the first maximum; a NaN is a maximum
        const float f1 = a.in[1][t];
        const float a0 = fabsf(f0), a1 = fabsf(f1);
        if (a1 > a0 || (a1 != a1 && a0 == a0))
            r = f1;
        break;
    }
    }
    out[t] = r;
}

// utils.upscale_array: (x * wf, y * hf) repeated hf x wf times
__global__ void k_flow_upscale(const float2 *__restrict__ in, float2 *__restrict__ out, int W, int H, int wf, int hf)
{
    const int Wo = W * wf;
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (size_t)Wo * H * hf)
        return;
    const int xo = (int)(t % Wo), yo = (int)(t / Wo);
    float2 v = in[(size_t)(yo / hf) * W + xo / wf];
    out[t] = make_float2(v.x * (float)wf, v.y * (float)hf);
}

template <typename T> struct Pair;
template <> struct Pair<float> {
    typedef float2 type;
};
template <> struct Pair<double> {
    typedef double2 type;
};

// scipy.signal.convolve2d(mode="same", boundary="fill", fillvalue=0) on both channels: kernel rows j
// then columns k, each product and each sum rounded in T (scipy/signal/_firfilter.c)
template <typename T>
__global__ void k_flow_convolve(const float2 *__restrict__ flow, const T *__restrict__ kern, int kh, int kw,
                                typename Pair<T>::type *__restrict__ out, int W, int H)
{
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (size_t)W * H)
        return;
    const int n = (int)(t % W), m = (int)(t / W);
    const int oy = (kh - 1) >> 1, ox = (kw - 1) >> 1;
    T sx = 0, sy = 0;
    for (int j = 0; j < kh; j++) {
        const int y = m + oy - j;
        const bool row_in = y >= 0 && y < H;
        for (int k = 0; k < kw; k++) {
            const int x = n + ox - k;
            const T hv = kern[j * kw + k];
            T vx = 0, vy = 0;
            if (row_in && x >= 0 && x < W) {
                float2 f = flow[(size_t)y * W + x];
                vx = (T)f.x;
                vy = (T)f.y;
            }
            sx = sx + hv * vx; // the fill value takes part: 0 * h
            sy = sy + hv * vy;
        }
    }
    typename Pair<T>::type r;
    r.x = sx;
    r.y = sy;
    out[t] = r;
}

// The same sums from an LDS tile: a block of 64x4 threads produces 64x16 outputs (four rows per
// thread); the tile holds the inputs those need, zero outside the image (the fill value), and the
// kernel values sit behind it.  Same j-then-k order per output, so the results are bit-identical to
// k_flow_convolve; used when tile + kernel fit 64 KB.
constexpr int CV_TW = 64, CV_TH = 16;

template <typename T>
__global__ void __launch_bounds__(256)
k_flow_convolve_tiled(const float2 *__restrict__ flow, const T *__restrict__ kern, int kh, int kw,
                      typename Pair<T>::type *__restrict__ out, int W, int H)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_cv[];
    const int LW = CV_TW + kw - 1, LH = CV_TH + kh - 1;
    float2 *tile = reinterpret_cast<float2 *>(s_cv);
    T *sk = reinterpret_cast<T *>(s_cv + (((size_t)LW * LH * sizeof(float2) + 15) & ~(size_t)15));
    const int x0 = blockIdx.x * CV_TW, y0 = blockIdx.y * CV_TH;
    const int oy = (kh - 1) >> 1, ox = (kw - 1) >> 1;
    // tile (ry, rx) <-> image (y0 + oy - (kh-1) + ry, x0 + ox - (kw-1) + rx)
    const int ty0 = y0 + oy - (kh - 1), tx0 = x0 + ox - (kw - 1);
    for (int idx = threadIdx.x; idx < LW * LH; idx += 256) {
        const int ry = idx / LW, rx = idx - ry * LW;
        const int y = ty0 + ry, x = tx0 + rx;
        tile[idx] = (y >= 0 && y < H && x >= 0 && x < W) ? flow[(size_t)y * W + x] : make_float2(0.f, 0.f);
    }
    for (int idx = threadIdx.x; idx < kh * kw; idx += 256)
        sk[idx] = kern[idx];
    __syncthreads();
    const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
    T sx[4] = {0, 0, 0, 0}, sy[4] = {0, 0, 0, 0};
    for (int j = 0; j < kh; j++) {
        for (int k = 0; k < kw; k++) {
            const T hv = sk[j * kw + k];
            const float2 *p = tile + (tq + (kh - 1) - j) * LW + tx + (kw - 1) - k;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float2 f = p[4 * q * LW];
                sx[q] = sx[q] + hv * (T)f.x;
                sy[q] = sy[q] + hv * (T)f.y;
            }
        }
    }
    const int x = x0 + tx;
    if (x < W) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int y = y0 + tq + 4 * q;
            if (y < H) {
                typename Pair<T>::type r;
                r.x = sx[q];
                r.y = sy[q];
                out[(size_t)y * W + x] = r;
            }
        }
    }
}

template <typename T> __device__ __forceinline__ T clip_nan_t(T v, T lo, T hi)
{
    return v != v ? v : (v < lo ? lo : (v > hi ? hi : v));
}

template <typename T2> __device__ __forceinline__ T2 clip_frame_t(T2 f, int i, int j, int W, int H)
{
    typedef decltype(f.x) T;
    f.x = clip_nan_t<T>(f.x, (T)(-j), (T)(W - 1 - j));
    f.y = clip_nan_t<T>(f.y, (T)(-i), (T)(H - 1 - i));
    return f;
}

template <typename T2> __global__ void k_pp_clip_t(T2 *flow, int W, int H)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= W * H)
        return;
    flow[t] = clip_frame_t(flow[t], t / W, t % W, W, H);
}

// source.py:350-358 (see k_pp_fwd_scatter in farneback.hip)
template <typename T2>
__global__ void k_pp_fwd_scatter_t(const T2 *__restrict__ flow, int *__restrict__ winner, int W, int H)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    const int N = W * H;
    if (t >= N)
        return;
    T2 f = clip_frame_t(flow[t], t / W, t % W, W, H);
    int ix = (int)rint((double)f.x), iy = (int)rint((double)f.y);
    int d = iy * W + ix;
    if (d == 0)
        return;
    int target = min(max(t + d, 0), N - 1);
    atomicMax(&winner[target], t);
}

template <typename T2>
__global__ void k_pp_fwd_resolve_t(T2 *__restrict__ flow, const int *__restrict__ winner, int W, int H)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= W * H)
        return;
    typedef decltype(flow[0].x) T;
    int w = winner[t];
    int src = w >= 0 ? w : t;
    int i = t / W, j = t % W;
    T2 f;
    f.x = (T)(src % W - j);
    f.y = (T)(src / W - i);
    flow[t] = clip_frame_t(f, i, j, W, H);
}

struct Colors {
    float c[4][3];
};

__device__ __forceinline__ float clip01(float v) { return clip_nan(v, 0.f, 1.f); }
__device__ __forceinline__ uint8_t to_u8(float v) { return (uint8_t)(int)clip_nan(v, 0.f, 255.f); }

// Four pixels per thread: their 12 output bytes leave as three dwords.
__device__ __forceinline__ void store_rgb4(uint8_t *__restrict__ rgb, size_t t4, size_t N, const uint8_t (&px)[12])
{
    if (t4 + 4 <= N) {
        uint32_t *o = reinterpret_cast<uint32_t *>(rgb + t4 * 3); // t4 is a multiple of 4: 12-byte groups stay 4-aligned
#pragma unroll
        for (int d = 0; d < 3; d++)
            o[d] = px[4 * d] | (px[4 * d + 1] << 8) | (px[4 * d + 2] << 16) | ((uint32_t)px[4 * d + 3] << 24);
    } else {
        for (size_t i = 0; t4 + i < N; i++)
            for (int c = 0; c < 3; c++)
                rgb[(t4 + i) * 3 + c] = px[3 * i + c];
    }
}

// output/render.py:9-27
__global__ void k_render1d(const float *__restrict__ arr, uint8_t *__restrict__ rgb, size_t N, float scale, Colors col,
                           int binary)
{
    const size_t t4 = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * 4;
    if (t4 >= N)
        return;
    uint8_t px[12];
    float a4[4];
    if (t4 + 4 <= N) {
        const float4 v = *reinterpret_cast<const float4 *>(arr + t4);
        a4[0] = v.x, a4[1] = v.y, a4[2] = v.z, a4[3] = v.w;
    } else {
        for (int i = 0; i < 4; i++)
            a4[i] = arr[min(t4 + i, N - 1)];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float sa = scale * a4[i];
        float ka, kb;
        if (binary) {
            kb = clip01(rintf(sa));
            ka = 1.f - kb;
        } else {
            ka = clip01(1.f - sa);
            kb = clip01(sa);
        }
#pragma unroll
        for (int c = 0; c < 3; c++)
            px[3 * i + c] = to_u8(ka * col.c[0][c] + kb * col.c[1][c]);
    }
    store_rgb4(rgb, t4, N, px);
}

// output/render.py:30-48
__global__ void k_render2d(const float2 *__restrict__ flow, uint8_t *__restrict__ rgb, size_t N, float scale, Colors col)
{
    const size_t t4 = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * 4;
    if (t4 >= N)
        return;
    uint8_t px[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float2 f = flow[min(t4 + i, N - 1)];
        const float sx = scale * f.x, sy = scale * f.y;
        const float ky = clip01(1.f + sx), kb = clip01(1.f - sx), km = clip01(1.f + sy), kg = clip01(1.f - sy);
#pragma unroll
        for (int c = 0; c < 3; c++)
            px[3 * i + c] = to_u8(0.5f * (((ky * col.c[0][c] + kb * col.c[1][c]) + km * col.c[2][c]) + kg * col.c[3][c]));
    }
    store_rgb4(rgb, t4, N, px);
}

// cv2.cvtColor(frame, COLOR_BGR2GRAY) for 8-bit images (cv.py:463, 466): OpenCV 4.x fixed point with
// 15 fractional bits, Y = (B*3735 + G*19235 + R*9798 + 16384) >> 15  (imgproc color_rgb: the BT.601
// weights 0.114, 0.587, 0.299 times 2^15, rounded; they sum to 32768).  RECALLED, cv2 is not
// installable here: parity unpinned.
// Fused with the nearest-neighbour resize of cv.py:461-462 / :464-465 (cv2.resize INTER_NEAREST:
// source index = min(floor(dst * src / dst_size), src - 1)); equal sizes make it a plain conversion.
__global__ void k_bgr_to_grey(const uint8_t *__restrict__ bgr, int Ws, int Hs, uint8_t *__restrict__ grey, int W, int H,
                              double fx, double fy)
{
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (size_t)W * H)
        return;
    const int x = (int)(t % W), y = (int)(t / W);
    const int sx = min((int)floor(x * fx), Ws - 1), sy = min((int)floor(y * fy), Hs - 1);
    const uint8_t *p = bgr + ((size_t)sy * Ws + sx) * 3;
    grey[t] = (uint8_t)((p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + 16384) >> 15);
}

// equal sizes: four pixels per thread, 12 bytes in as three dwords, one dword out
__global__ void k_bgr_to_grey4(const uint32_t *__restrict__ bgr, uint32_t *__restrict__ grey, size_t n4)
{
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n4)
        return;
    const uint32_t a = bgr[3 * t], b = bgr[3 * t + 1], c = bgr[3 * t + 2];
    const uint8_t p[12] = {(uint8_t)a, (uint8_t)(a >> 8), (uint8_t)(a >> 16), (uint8_t)(a >> 24),
                           (uint8_t)b, (uint8_t)(b >> 8), (uint8_t)(b >> 16), (uint8_t)(b >> 24),
                           (uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24)};
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        o |= (uint32_t)((p[3 * i] * 3735 + p[3 * i + 1] * 19235 + p[3 * i + 2] * 9798 + 16384) >> 15) << (8 * i);
    grey[t] = o;
}


// ---------------------------------------------------------------------------------
// The `polar` flow filter (transflow/flow/filters.py:75-87): radius and angle of every vector,
// two user expressions of (t, r, a) compiled to postfix programs by transflow_amd/exprs.py (the
// parts that are not arrays are evaluated on the host and arrive as constants), then
// flow = (R cos A, R sin A).  A step computes in float32 unless its `wide` flag says float64, which
// is how numpy types the same expression; values ride on a stack of doubles (every float is one).
// ---------------------------------------------------------------------------------
enum PolarOp { P_PUSH_R, P_PUSH_A, P_PUSH_CONST, P_ADD, P_SUB, P_MUL, P_DIV, P_POW, P_MOD, P_FLOORDIV, P_NEG, P_SIN, P_COS,
               P_TAN, P_ASIN, P_ACOS, P_ATAN, P_ATAN2, P_SQRT, P_ABS, P_EXP, P_LOG, P_LOG2, P_LOG10, P_MIN, P_MAX,
               P_FLOOR, P_CEIL, P_RINT, P_SIGN, P_SQUARE, P_HYPOT, P_LT, P_LE, P_GT, P_GE, P_EQ, P_NE, P_WHERE, P_CLIP,
               P_RECIP, P_NOT, P_COUNT };

struct PolarProg {
    int n;
    unsigned char op[TF_MAX_POLAR_STEPS];
    unsigned char wide[TF_MAX_POLAR_STEPS];
    double imm[TF_MAX_POLAR_STEPS];
};

template <typename T> __device__ __forceinline__ T pmod(T a, T b)
{ // numpy.mod: the result takes the divisor's sign
    T r = sizeof(T) == 4 ? (T)fmodf((float)a, (float)b) : (T)fmod((double)a, (double)b);
    if (r != 0 && ((r < 0) != (b < 0)))
        r += b;
    return r;
}

template <typename T> __device__ __forceinline__ T pnanmin(T a, T b) { return (a != a || b != b) ? (a != a ? a : b) : (a < b ? a : b); }
template <typename T> __device__ __forceinline__ T pnanmax(T a, T b) { return (a != a || b != b) ? (a != a ? a : b) : (a > b ? a : b); }

template <typename T> __device__ __forceinline__ double polar_unary(int op, T x)
{
    const bool f = sizeof(T) == 4;
    switch (op) {
    case P_NEG: return -x;
    case P_SIN: return f ? sinf(x) : sin((double)x);
    case P_COS: return f ? cosf(x) : cos((double)x);
    case P_TAN: return f ? tanf(x) : tan((double)x);
    case P_ASIN: return f ? asinf(x) : asin((double)x);
    case P_ACOS: return f ? acosf(x) : acos((double)x);
    case P_ATAN: return f ? atanf(x) : atan((double)x);
    case P_SQRT: return f ? sqrtf(x) : sqrt((double)x);
    case P_ABS: return f ? fabsf(x) : fabs((double)x);
    case P_EXP: return f ? expf(x) : exp((double)x);
    case P_LOG: return f ? logf(x) : log((double)x);
    case P_LOG2: return f ? log2f(x) : log2((double)x);
    case P_LOG10: return f ? log10f(x) : log10((double)x);
    case P_FLOOR: return f ? floorf(x) : floor((double)x);
    case P_CEIL: return f ? ceilf(x) : ceil((double)x);
    case P_RINT: return f ? rintf(x) : rint((double)x);
    case P_SIGN: return x != x ? x : (T)((x > 0) - (x < 0));
    case P_SQUARE: return x * x;
    case P_RECIP: return (T)1 / x;
    default: return x;
    }
}

template <typename T> __device__ __forceinline__ double polar_binary(int op, T a, T b)
{
    const bool f = sizeof(T) == 4;
    switch (op) {
    case P_ADD: return a + b;
    case P_SUB: return a - b;
    case P_MUL: return a * b;
    case P_DIV: return a / b;
    case P_POW: return f ? powf(a, b) : pow((double)a, (double)b);
    case P_MOD: return pmod<T>(a, b);
    case P_FLOORDIV: return f ? floorf(a / b) : floor((double)a / (double)b);
    case P_ATAN2: return f ? atan2f(a, b) : atan2((double)a, (double)b);
    case P_MIN: return pnanmin<T>(a, b);
    case P_MAX: return pnanmax<T>(a, b);
    case P_HYPOT: return f ? hypotf(a, b) : hypot((double)a, (double)b);
    case P_LT: return a < b;
    case P_LE: return a <= b;
    case P_GT: return a > b;
    case P_GE: return a >= b;
    case P_EQ: return a == b;
    case P_NE: return a != b;
    default: return a;
    }
}

__device__ double polar_eval(const PolarProg &p, float r, float a)
{
    double st[TF_MAX_POLAR_STACK];
    int sp = 0;
    for (int i = 0; i < p.n; i++) {
        const int op = p.op[i];
        const bool wide = p.wide[i] != 0;
        if (op == P_PUSH_R) {
            st[sp++] = r;
        } else if (op == P_PUSH_A) {
            st[sp++] = a;
        } else if (op == P_PUSH_CONST) {
            st[sp++] = p.imm[i];
        } else if (op == P_WHERE) {
            const double y = st[--sp], x = st[--sp], c = st[--sp];
            const double v = c != 0 ? x : y;
            st[sp++] = wide ? v : (double)(float)v;
        } else if (op == P_CLIP) {
            const double hi = st[--sp], lo = st[--sp], x = st[--sp];
            st[sp++] = wide ? pnanmin<double>(pnanmax<double>(x, lo), hi)
                            : (double)pnanmin<float>(pnanmax<float>((float)x, (float)lo), (float)hi);
        } else if (op == P_NOT) {
            st[sp - 1] = st[sp - 1] == 0;
        } else if (op == P_ADD || op == P_SUB || op == P_MUL || op == P_DIV || op == P_POW || op == P_MOD ||
                   op == P_FLOORDIV || op == P_ATAN2 || op == P_MIN || op == P_MAX || op == P_HYPOT ||
                   (op >= P_LT && op <= P_NE)) {
            const double b = st[--sp], x = st[--sp];
            st[sp++] = wide ? polar_binary<double>(op, x, b) : polar_binary<float>(op, (float)x, (float)b);
        } else {
            const double x = st[sp - 1];
            st[sp - 1] = wide ? polar_unary<double>(op, x) : polar_unary<float>(op, (float)x);
        }
    }
    return st[0];
}

// wide_out: bit 0 = sin/cos in float64 (the angle expression is float64 or a bare Python scalar: numpy.sin of
// one is a float64), bit 1 = the product in float64
__global__ void k_pp_polar(float2 *__restrict__ flow, size_t N, PolarProg pr, PolarProg pa, int wide_out)
{
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    const float2 f = flow[t];
    const float r = sqrtf(f.x * f.x + f.y * f.y); // numpy.linalg.norm of a float32 pair (filters.py:83)
    const float a = atan2f(f.y, f.x);              // :84
    const double R = polar_eval(pr, r, a), A = polar_eval(pa, r, a);
    double s, c;
    if (wide_out & 1) {
        s = sin(A);
        c = cos(A);
    } else {
        s = sinf((float)A);
        c = cosf((float)A);
    }
    float2 o;
    if (wide_out & 2) {
        o.y = (float)(R * s); // :87 then :88
        o.x = (float)(R * c);
    } else {
        o.y = (float)R * (float)s;
        o.x = (float)R * (float)c;
    }
    flow[t] = o;
}

} // namespace

TF_API int tf_flow_merge_dev(int kind, int n, const void *const *flows_dev, void *out_dev, size_t n_values)
{
    TF_REQUIRE(kind >= TF_MERGE_FIRST && kind <= TF_MERGE_ABSMAX, "tf_flow_merge: unknown kind %d", kind);
    TF_REQUIRE(n >= 1 && n <= TF_MAX_MERGE, "tf_flow_merge: %d flows (1..%d)", n, TF_MAX_MERGE);
    TF_REQUIRE(kind != TF_MERGE_ABSMAX || n == 2, "tf_flow_merge: absmax merges exactly two flows, got %d", n);
    TF_REQUIRE(flows_dev && (out_dev || n_values == 0), "tf_flow_merge: null pointer");
    TF_TRY(ensure_init());
    MergeArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    for (int i = 0; i < n; i++) {
        TF_REQUIRE(flows_dev[i] || n_values == 0, "tf_flow_merge: flow %d is null", i);
        a.in[i] = (const float *)flows_dev[i];
    }
    return launch("flow_merge", k_flow_merge, dim3(cdiv(n_values, BLOCK)), dim3(BLOCK), 0, a, (float *)out_dev, n_values,
                  kind);
}

TF_API int tf_flow_upscale_dev(const void *in_dev, void *out_dev, int width, int height, int wf, int hf)
{
    TF_REQUIRE(width >= 0 && height >= 0 && wf >= 1 && hf >= 1, "tf_flow_upscale: bad size %dx%d x(%d,%d)", width, height,
               wf, hf);
    TF_REQUIRE((in_dev && out_dev) || (size_t)width * height == 0, "tf_flow_upscale: null pointer");
    TF_TRY(ensure_init());
    const size_t n = (size_t)width * wf * height * hf;
    return launch("flow_upscale", k_flow_upscale, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0, (const float2 *)in_dev,
                  (float2 *)out_dev, width, height, wf, hf);
}

TF_API int tf_flow_convolve_dev(const void *flow_dev, const void *kernel_dev, int kh, int kw, int wide, void *out_dev,
                                int width, int height)
{
    TF_REQUIRE(kh >= 1 && kw >= 1 && width >= 0 && height >= 0, "tf_flow_convolve: bad sizes");
    TF_REQUIRE((flow_dev && out_dev && kernel_dev) || (size_t)width * height == 0, "tf_flow_convolve: null pointer");
    TF_TRY(ensure_init());
    const size_t n = (size_t)width * height;
    const size_t tile_bytes = (((size_t)(CV_TW + kw - 1) * (CV_TH + kh - 1) * sizeof(float2) + 15) & ~(size_t)15) +
                              (size_t)kh * kw * (wide ? 8 : 4);
    static const bool no_tiles = tune("TF_CONV_NO_TILES", 0) != 0;
    if (tile_bytes <= 64 * 1024 && !no_tiles && n) {
        dim3 grid(cdiv(width, CV_TW), cdiv(height, CV_TH));
        if (wide)
            return launch("flow_convolve_f64", k_flow_convolve_tiled<double>, grid, dim3(256), tile_bytes,
                          (const float2 *)flow_dev, (const double *)kernel_dev, kh, kw, (double2 *)out_dev, width, height);
        return launch("flow_convolve_f32", k_flow_convolve_tiled<float>, grid, dim3(256), tile_bytes,
                      (const float2 *)flow_dev, (const float *)kernel_dev, kh, kw, (float2 *)out_dev, width, height);
    }
    if (wide)
        return launch("flow_convolve_f64", k_flow_convolve<double>, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                      (const float2 *)flow_dev, (const double *)kernel_dev, kh, kw, (double2 *)out_dev, width, height);
    return launch("flow_convolve_f32", k_flow_convolve<float>, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                  (const float2 *)flow_dev, (const float *)kernel_dev, kh, kw, (float2 *)out_dev, width, height);
}

template <typename T2> static int pp_any(T2 *flow, int W, int H, int direction, int *winner)
{
    const size_t n = (size_t)W * H;
    dim3 grid(cdiv(n, BLOCK)), block(BLOCK);
    if (direction == 0) {
        TF_REQUIRE(winner || n == 0, "tf_flow_post_process: FORWARD needs the 4 B/px scratch buffer");
        TF_HIP(hipMemsetAsync(winner, 0xFF, n * 4, stream()));
        TF_TRY(launch("flow_pp_fwd_scatter", k_pp_fwd_scatter_t<T2>, grid, block, 0, (const T2 *)flow, winner, W, H));
        return launch("flow_pp_fwd_resolve", k_pp_fwd_resolve_t<T2>, grid, block, 0, flow, (const int *)winner, W, H);
    }
    return launch("flow_pp_clip", k_pp_clip_t<T2>, grid, block, 0, flow, W, H);
}

TF_API int tf_flow_post_process_dev(void *flow_dev, int wide, int width, int height, int direction, void *scratch_dev)
{
    TF_REQUIRE(direction == 0 || direction == 1, "tf_flow_post_process: direction must be 0 (FORWARD) or 1 (BACKWARD)");
    TF_REQUIRE(width >= 0 && height >= 0 && (long long)width * height < (1ll << 31), "tf_flow_post_process: bad size");
    TF_REQUIRE(flow_dev || (size_t)width * height == 0, "tf_flow_post_process: null pointer");
    TF_TRY(ensure_init());
    if (wide)
        return pp_any((double2 *)flow_dev, width, height, direction, (int *)scratch_dev);
    return pp_any((float2 *)flow_dev, width, height, direction, (int *)scratch_dev);
}

static Colors make_colors(const float *rgb, int n)
{
    Colors c;
    memset(&c, 0, sizeof(c));
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            c.c[i][k] = rgb[i * 3 + k];
    return c;
}

TF_API int tf_flow_render1d_dev(const void *arr_dev, void *rgb_dev, size_t n, float scale, const float colors_rgb[6],
                                int binary)
{
    TF_REQUIRE(colors_rgb && ((arr_dev && rgb_dev) || n == 0), "tf_flow_render1d: null pointer");
    TF_TRY(ensure_init());
    return launch("flow_render1d", k_render1d, dim3(cdiv((n + 3) / 4, BLOCK)), dim3(BLOCK), 0, (const float *)arr_dev,
                  (uint8_t *)rgb_dev, n, scale, make_colors(colors_rgb, 2), binary);
}

TF_API int tf_flow_render2d_dev(const void *flow_dev, void *rgb_dev, size_t n, float scale, const float colors_rgb[12])
{
    TF_REQUIRE(colors_rgb && ((flow_dev && rgb_dev) || n == 0), "tf_flow_render2d: null pointer");
    TF_TRY(ensure_init());
    return launch("flow_render2d", k_render2d, dim3(cdiv((n + 3) / 4, BLOCK)), dim3(BLOCK), 0, (const float2 *)flow_dev,
                  (uint8_t *)rgb_dev, n, scale, make_colors(colors_rgb, 4));
}

TF_API int tf_frame_grey_dev(const void *bgr_dev, int src_width, int src_height, void *grey_dev, int width, int height)
{
    TF_REQUIRE(src_width >= 1 && src_height >= 1 && width >= 0 && height >= 0, "tf_frame_grey: bad sizes");
    TF_REQUIRE((bgr_dev && grey_dev) || (size_t)width * height == 0, "tf_frame_grey: null pointer");
    TF_TRY(ensure_init());
    const size_t n = (size_t)width * height;
    // cv2.resize with dsize only: fx = dsize.width / src.cols; the nearest-neighbour source index is
    // floor(dst * (1 / fx))
    const double ifx = width ? 1.0 / ((double)width / src_width) : 1.0, ify = height ? 1.0 / ((double)height / src_height) : 1.0;
    if (width == src_width && height == src_height && (n & 3) == 0 && ((uintptr_t)bgr_dev & 3) == 0 &&
        ((uintptr_t)grey_dev & 3) == 0)
        return launch("frame_grey", k_bgr_to_grey4, dim3(cdiv(n / 4, BLOCK)), dim3(BLOCK), 0, (const uint32_t *)bgr_dev,
                      (uint32_t *)grey_dev, n / 4);
    return launch("frame_grey", k_bgr_to_grey, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0, (const uint8_t *)bgr_dev, src_width,
                  src_height, (uint8_t *)grey_dev, width, height, ifx, ify);
}

static int load_prog(PolarProg &p, int n, const tf_polar_step *steps, const char *what)
{
    TF_REQUIRE(n >= 1 && n <= TF_MAX_POLAR_STEPS && steps, "tf_flow_polar: %s program has %d steps (1..%d)", what, n,
               TF_MAX_POLAR_STEPS);
    memset(&p, 0, sizeof(p));
    p.n = n;
    int depth = 0;
    for (int i = 0; i < n; i++) {
        const int op = steps[i].op;
        TF_REQUIRE(op >= 0 && op < P_COUNT, "tf_flow_polar: %s step %d: unknown opcode %d", what, i, op);
        int pops = 1, pushes = 1;
        if (op == P_PUSH_R || op == P_PUSH_A || op == P_PUSH_CONST)
            pops = 0;
        else if (op == P_WHERE || op == P_CLIP)
            pops = 3;
        else if (op == P_ADD || op == P_SUB || op == P_MUL || op == P_DIV || op == P_POW || op == P_MOD ||
                 op == P_FLOORDIV || op == P_ATAN2 || op == P_MIN || op == P_MAX || op == P_HYPOT ||
                 (op >= P_LT && op <= P_NE))
            pops = 2;
        TF_REQUIRE(depth >= pops, "tf_flow_polar: %s step %d pops an empty stack", what, i);
        depth += pushes - pops;
        TF_REQUIRE(depth <= TF_MAX_POLAR_STACK, "tf_flow_polar: %s program needs more than %d stack slots", what,
                   TF_MAX_POLAR_STACK);
        p.op[i] = (unsigned char)op;
        p.wide[i] = steps[i].wide != 0;
        p.imm[i] = steps[i].imm;
    }
    TF_REQUIRE(depth == 1, "tf_flow_polar: %s program leaves %d values", what, depth);
    return TF_OK;
}

TF_API int tf_flow_polar_dev(void *flow_dev, size_t n_pixels, int n_radius, const tf_polar_step *radius, int n_theta,
                             const tf_polar_step *theta, int wide_trig, int wide_product)
{
    TF_REQUIRE(flow_dev || n_pixels == 0, "tf_flow_polar: null pointer");
    PolarProg pr, pa;
    TF_TRY(load_prog(pr, n_radius, radius, "radius"));
    TF_TRY(load_prog(pa, n_theta, theta, "theta"));
    TF_TRY(ensure_init());
    return launch("flow_polar", k_pp_polar, dim3(cdiv(n_pixels, BLOCK)), dim3(BLOCK), 0, (float2 *)flow_dev, n_pixels, pr,
                  pa, (wide_trig ? 1 : 0) | (wide_product ? 2 : 0));
}
