// Farnebäck dense optical flow on gfx950 (hand-written HIP, no library calls).
//
// Replaces cv2.calcOpticalFlowFarneback as the reference calls it
// (transflow/flow/sources/cv.py:479-490, parameters cv.py:273-281).  The stages are
// those of OpenCV 4.x's CPU path (modules/video/src/optflowgf.cpp), restated in
// SURVEY.md Appendix A:
//   A1 pre-blur at full resolution + bilinear resize to the pyramid level
//   A2 polynomial expansion (separable, float vertical pass, double horizontal pass)
//   A3 update-matrices (bilinear gather of R1 at x+flow, 2x2 system per pixel)
//   A4 box blur of the system (double sums) + 2x2 solve
//   A5 coarse-to-fine flow upsampling
// Kernels (DESIGN.md section 3 has the table):
//   A1+A2  k_level0_polyexp_t (level 0), k_level1_polyexp_t (half-size level), otherwise
//          k_level_rowpass + k_level_colpass (long blur kernels) or k_level_image, then k_polyexp_t
//   A3+A4(+A5)  k_flow_iter_pc: one whole iteration, the 2x2 systems never leave the CU (large levels);
//          k_update_matrices + k_blur_solve_wave on the small ones
//   option fb_exact_sums: k_flow_carry_pc<.., STORE> or k_update_matrices + k_exact_vsum, then k_exact_hsolve
//
// Arithmetic discipline: every float/double operation is written in the order of
// the CPU path and the file is compiled with -ffp-contract=off, so A1-A3 are
// bit-identical to the scalar CPU statement.  A4 keeps OpenCV's running column sums (one
// float-differenced chain per column from row 0, carried across row segments) and adds them across
// the window's columns directly in double where OpenCV slides a second running sum along the row
// (differences ~1e-16 relative); with option fb_exact_sums that sum is OpenCV's too and the flow
// is bit-identical.
//
// HBM layout (per handle, sized for `max_pairs` frame pairs):
//   frames   u8  [slot][H][W]
//   img      f32 [image][Hk*Wk]            level images (A1 output, levels without a fused A1+A2)
//   rowf     f32 [level][image][H][2*Wk]   row-pass planes of the long-kernel levels
//   R        f32 [image][5][Hk*Wk]         polynomial coefficients, planar (SoA); image = a frame of the batch
//   M        f32 [pair][5][Hk*Wk]          2x2 systems, planar (two-kernel iterations only)
//   exact_vsum f64 [pair][5][Hk*Wk]        option fb_exact_sums: the column sums of every row of the level being solved
//   lflow[5] f32 [pair][Hk*Wk][2]          per-level flow: three rotate, two hold the result of even / odd calls
// Stencils, gathers and 2x2 solves (<= ~60 flop/B, no dense contraction): MFMA is not applicable.
#include <type_traits>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "common.h"

using namespace tf;

namespace {

constexpr int MAX_POLY_N = 15;

// two floats at 4-byte alignment: one global_load_dwordx2 (the hardware takes unaligned dwordx2)
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

// ---------------------------------------------------------------------------------
// Layout of R, the polynomial coefficients of one image at one level (Nk pixels, 5 Nk floats, OpenCV's channel
// order): three planes -- (c0, c1) pairs [Nk][2], (c2, c3) pairs [Nk][2], c4 [Nk].  FarnebackUpdateMatrices pairs
// the channels exactly so ((c0, c1) feed h, (c2, c3) the diagonal of G, c4 its off-diagonal): the two taps of a
// bilinear row arrive as ONE 16-byte load per channel pair (x1 and x1 + 1 are adjacent pixels), a pixel's own
// coefficients as two 8-byte loads and a 4-byte one, and the arithmetic runs on register pairs as they were
// loaded: 10 instead of 16 loads per pixel and no shuffling between loads and packed math.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ size_t r_off23(size_t Nk) { return 2 * Nk; }
__device__ __forceinline__ size_t r_off4(size_t Nk) { return 4 * Nk; }

__device__ __forceinline__ void r_load_px(const float *__restrict__ R, size_t Nk, size_t o, float v[5])
{
    const float2u a = *reinterpret_cast<const float2u *>(R + 2 * o);
    const float2u b = *reinterpret_cast<const float2u *>(R + r_off23(Nk) + 2 * o);
    v[0] = a.x;
    v[1] = a.y;
    v[2] = b.x;
    v[3] = b.y;
    v[4] = R[r_off4(Nk) + o];
}
// the taps at pixels q and q + 1 (row y1) and q + Wk, q + Wk + 1 (row y1 + 1): per channel (left, right)
__device__ __forceinline__ void r_load_taps(const float *__restrict__ R, size_t Nk, size_t q, int Wk, float2u t[5], float2u b[5])
{
    const float4u t01 = *reinterpret_cast<const float4u *>(R + 2 * q), b01 = *reinterpret_cast<const float4u *>(R + 2 * (q + Wk));
    const float4u t23 = *reinterpret_cast<const float4u *>(R + r_off23(Nk) + 2 * q);
    const float4u b23 = *reinterpret_cast<const float4u *>(R + r_off23(Nk) + 2 * (q + Wk));
    const float2u t4 = *reinterpret_cast<const float2u *>(R + r_off4(Nk) + q), b4 = *reinterpret_cast<const float2u *>(R + r_off4(Nk) + q + Wk);
    t[0] = float2u{t01.x, t01.z};
    t[1] = float2u{t01.y, t01.w};
    t[2] = float2u{t23.x, t23.z};
    t[3] = float2u{t23.y, t23.w};
    t[4] = t4;
    b[0] = float2u{b01.x, b01.z};
    b[1] = float2u{b01.y, b01.w};
    b[2] = float2u{b23.x, b23.z};
    b[3] = float2u{b23.y, b23.w};
    b[4] = b4;
}
__device__ __forceinline__ void r_store_px(float *__restrict__ R, size_t Nk, size_t o, const float v[5])
{
    *reinterpret_cast<float2u *>(R + 2 * o) = float2u{v[0], v[1]};
    *reinterpret_cast<float2u *>(R + r_off23(Nk) + 2 * o) = float2u{v[2], v[3]};
    R[r_off4(Nk) + o] = v[4];
}
__device__ __forceinline__ void r_store_px2(float *__restrict__ R, size_t Nk, size_t o, const float v0[5], const float v1[5])
{
    *reinterpret_cast<float4u *>(R + 2 * o) = float4u{v0[0], v0[1], v1[0], v1[1]};
    *reinterpret_cast<float4u *>(R + r_off23(Nk) + 2 * o) = float4u{v0[2], v0[3], v1[2], v1[3]};
    *reinterpret_cast<float2u *>(R + r_off4(Nk) + o) = float2u{v0[4], v1[4]};
}

#ifndef BLUR_PREFETCH
#define BLUR_PREFETCH 3 // rows of M kept in flight per wave in the blur march (2: 1219 us, 3: 1177, 4: 1225 at 4K x16)
#endif

struct PolyConst {
    int n;
    float g[MAX_POLY_N + 1], xg[MAX_POLY_N + 1], xxg[MAX_POLY_N + 1];
    double ig11, ig03, ig33, ig55;
};

__host__ __device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// the same for lo <= hi as one v_med3_i32
__device__ __forceinline__ int med3i(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "s"(hi));
    return r;
}

// Blocks are dealt round-robin over the 8 XCDs, each with its own L2 (MI355X_MICROARCH.md, dispatch):
// renumber the blocks of a 2-D grid so that one XCD walks a contiguous raster range of tiles and
// spatial neighbours (shared halo columns, shared rows, the cache line a misaligned strip spills
// into) meet in the same L2.  Bijective for any grid size; affects speed only.
__device__ __forceinline__ void xcd_tile(unsigned &bx, unsigned &by)
{
    const unsigned nt = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = bid & 7, qn = nt >> 3, rn = nt & 7;
    const unsigned t = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    bx = t % gridDim.x;
    by = t / gridDim.x;
}

// The same for the one-kernel iteration's grid of (strip, segment, pair): XCD x takes a contiguous run of the PAIRS (pairs
// [x P / 8, (x + 1) P / 8)) and, of those, all strips and segments -- the strips of a pair share their halo columns in one
// L2, and consecutive pairs the frame they both read (R0 of one is R1 of the next).  Bijective for any grid.
__device__ __forceinline__ void xcd_pair_tile(unsigned &bx, unsigned &by, int &pair)
{
    const unsigned S = gridDim.x, G = gridDim.y, P = gridDim.z, nt = S * G * P;
    const unsigned lin = (blockIdx.z * G + blockIdx.y) * S + blockIdx.x;
    const unsigned xcd = lin & 7, qn = nt >> 3, rn = nt & 7;
    const unsigned u = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (lin >> 3); // place in XCD-major order
    const unsigned whole = u / (S * G); // whole pairs' worth of work in front of it
    unsigned l = 0;
    while (l < 7 && (l + 1) * P / 8 <= whole)
        l++;
    const unsigned p0 = l * P / 8, n = (l + 1) * P / 8 - p0, r = u - p0 * S * G; // within the run: (segment, strip, pair)
    by = r / (S * n);
    const unsigned rem = r - by * S * n;
    bx = rem / n;
    pair = (int)(p0 + (rem - bx * n));
}

// Barriers that order LDS traffic only.  __syncthreads() also carries a release fence on GLOBAL memory,
// i.e. `s_waitcnt vmcnt(0)`: in a marching loop that drains every prefetched load at every row.
// lds_barrier(): all waves of the workgroup; lds_wave_sync(): the lanes of one wave (single-wave exchange).
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// 1/d for a positive, normal double: the hardware estimate (good to 2^-26 or better) and one Newton step, which
// squares the error: 2^-52, in a quantity that leaves the kernel as a float and that the path needs to 1e-4.  (The
// compiler's IEEE division adds a second step, a residual correction and the scaling / fix-up of denormal and
// infinite operands, ~25 instructions; determinants that carry +1e-3 are never those.)
__device__ __forceinline__ double fast_recip(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// REFLECT_101 when the overshoot is known to be smaller than the image (one reflection suffices)
__device__ __forceinline__ int reflect101_once(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * len - 2 - p : p); }
// REFLECT_101 for an index at most one step outside [0, len): no loop (reflect101's `while` becomes a real loop
// with its own exec masking around every load that uses it)
__device__ __forceinline__ int reflect101_near(int p, int len) { return len < 2 ? 0 : reflect101_once(p, len); }

// ---------------------------------------------------------------------------------
// A1: level image = resize(GaussianBlur(float(frame)), level size), one kernel per level.
// A block produces a tile of TWo x THo level pixels.  It stages the uint8 source region
// those pixels depend on (with the blur's halo, REFLECT_101 applied while loading) in
// LDS, runs the row pass only at the two source columns each output column interpolates
// between, then the column pass at the two source rows of each output row, then the two
// lerps.  Tap order and float rounding are those of the CPU filters (row pass: paired
// taps for ksz <= 5, left-to-right otherwise; column pass: centre, then pairs outwards).
// In the row pass lanes walk source ROWS, so the byte reads of one instruction hit
// different LDS banks (pitch/4 is odd).
// ---------------------------------------------------------------------------------
struct ImgTile {
    int TWo, THo;           // output tile (TWo a power of two)
    int tw_shift;           // log2(TWo)
    int rstride;            // floats per staged row of the row-pass buffer: 2*TWo, or TWo for a copy-sized level
    int LW, LH;             // source columns / rows staged per tile (upper bounds)
    int pitch;              // bytes per staged source row, multiple of 4 with pitch/4 odd
    int same_size;          // level size == frame size: resize is a copy
    double scale_x, scale_y; // resize.cpp's 1/(dst/src) per axis
};

// resize.cpp's INTER_LINEAR source coordinate for destination index d (the statement order
// of the host's make_lerp: double product and difference, one rounding each, then float).
__device__ __forceinline__ int lerp_coord(int d, double scale, int src, bool zero_at_edges, float &frac)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (zero_at_edges) {
        if (s < 0) {
            f = 0.f;
            s = 0;
        }
        if (s >= src - 1) {
            f = 0.f;
            s = src - 1;
        }
    }
    frac = f;
    return s;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
k_level_image(const uint8_t *__restrict__ frames, const int2 *__restrict__ pairs, float *__restrict__ img, int W,
              int H, int Wk, int Hk, const float *__restrict__ kern, int ksz, ImgTile tl)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    uint8_t *sS = s_raw;                                                     // [LH][pitch] source bytes
    float *sR = reinterpret_cast<float *>(s_raw + (size_t)tl.LH * tl.pitch); // [LH][2*TWo] row-pass values
    float *sK = sR + (size_t)tl.LH * tl.rstride;                             // [ksz] blur taps
    __shared__ int sX[128], sY[32];     // source column / row of each output column / row of the tile
    __shared__ float sFx[128], sFy[32]; // and the lerp fractions
    const int r = ksz >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < ksz; i += blockDim.x)
        sK[i] = kern[i];
    const int pi = blockIdx.z;
    const int2 pr = pairs[pi >> 1];
    const uint8_t *src = frames + (size_t)((pi & 1) ? pr.y : pr.x) * W * H;
    const int dx0 = blockIdx.x * tl.TWo, dy0 = blockIdx.y * tl.THo;
    const int ndx = min(tl.TWo, Wk - dx0), ndy = min(tl.THo, Hk - dy0);
    float ftmp;
    int sx_first, sx_last, sy_first, sy_last;
    if (tl.same_size) { // resize of equal sizes is a copy: source == destination coordinates
        sx_first = dx0;
        sx_last = dx0 + ndx - 1;
        sy_first = dy0;
        sy_last = dy0 + ndy - 1;
    } else {
        sx_first = lerp_coord(dx0, tl.scale_x, W, true, ftmp);
        sx_last = lerp_coord(dx0 + ndx - 1, tl.scale_x, W, true, ftmp);
        sy_first = lerp_coord(dy0, tl.scale_y, H, false, ftmp);
        sy_last = lerp_coord(dy0 + ndy - 1, tl.scale_y, H, false, ftmp);
        if (threadIdx.x < ndx) {
            float f;
            sX[threadIdx.x] = lerp_coord(dx0 + threadIdx.x, tl.scale_x, W, true, f);
            sFx[threadIdx.x] = f;
        } else if (threadIdx.x >= 128 && threadIdx.x - 128 < ndy) {
            float f;
            sY[threadIdx.x - 128] = lerp_coord(dy0 + threadIdx.x - 128, tl.scale_y, H, false, f);
            sFy[threadIdx.x - 128] = f;
        }
    }
    // staged columns start at a multiple of 4 so interior tiles can be copied as dwords
    const int x_lo = (sx_first - r) & ~3, x_hi = min(sx_last + 1, W - 1) + r;
    const int y_lo = clampi(sy_first, 0, H - 1) - r, y_hi = clampi(sy_last + 1, 0, H - 1) + r;
    const int ncols = x_hi - x_lo + 1, nrows = y_hi - y_lo + 1;
    // ---- phase 1: stage the source region.  Each wave owns rows wave, wave+4, ...; loads are
    // issued eight rows at a time so their latencies overlap.
    constexpr int U = 8;
    const bool small_halo = r < H && r < W; // one reflection is enough
    const bool dwords = (W & 3) == 0 && x_lo >= 0 && x_lo + ((ncols + 3) & ~3) <= W;
    if (dwords) {
        const int nq = (ncols + 3) >> 2;
        for (int c = lane; c < nq; c += 64) {
            for (int j0 = wave; j0 < nrows; j0 += 4 * U) {
                uint32_t v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        v[u] = *reinterpret_cast<const uint32_t *>(
                            src + (size_t)(small_halo ? reflect101_once(y_lo + ry, H) : reflect101(y_lo + ry, H)) * W + x_lo + 4 * c);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        *reinterpret_cast<uint32_t *>(sS + ry * tl.pitch + 4 * c) = v[u];
                }
            }
        }
    } else {
        for (int c = lane; c < ncols; c += 64) {
            const int x = reflect101(x_lo + c, W);
            for (int j0 = wave; j0 < nrows; j0 += 4 * U) {
                uint8_t v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        v[u] = src[(size_t)reflect101(y_lo + ry, H) * W + x];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int ry = j0 + 4 * u;
                    if (ry < nrows)
                        sS[ry * tl.pitch + c] = v[u];
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: row pass at the needed columns.  With many staged rows the lanes walk rows
    // (conflict-free byte reads); with few (small kernels) they walk output columns.
    // A copy-sized level only needs column sx of each output (even slots).
    const int ostep = tl.same_size ? 2 : 1;
    const int no = 2 * ndx / ostep;
    const bool lanes_on_rows = nrows >= 48;
    const int n_a = lanes_on_rows ? nrows : no, n_b = lanes_on_rows ? no : nrows;
    const float k0 = sK[0], kc = sK[r], kc1 = ksz >= 3 ? sK[r + 1] : 0.f, kc2 = ksz >= 5 ? sK[r + 2] : 0.f;
    if (lanes_on_rows && ksz > 5) {
        // long kernels: a lane owns one staged row and four output columns at a time, so four
        // independent left-to-right sums are in flight and each tap is fetched once for the four
        // work item = (chunk of 64 staged rows, group of 4 output columns), dealt round-robin to
        // the four waves so none idles when a tile has few column groups
        const int ngroups = (no + 3) >> 2, nchunks = (nrows + 63) >> 6;
        for (int item = wave; item < ngroups * nchunks; item += 4) {
            const int b_ = item % ngroups, ry = (item / ngroups) * 64 + lane;
            // the four columns are two interpolation pairs (sx, sx+1): base columns A and B
            const int sxA = __builtin_amdgcn_readfirstlane(sX[min(4 * b_, no - 1) >> 1]);
            const int sxB = __builtin_amdgcn_readfirstlane(sX[min(4 * b_ + 2, no - 1) >> 1]);
            if (sxA + 1 < W && sxB + 1 < W) {
                // Column sx+1 reads the byte stream of column sx one tap later, so each pair shares one
                // stream: aligned dword reads (conflict-free: pitch/4 is odd) re-aligned to the
                // stream's first byte, each byte converted once, and the two streams carried as the
                // halves of float pairs so that a tap costs two packed multiplies and two packed adds
                // for four sums.  Every sum still adds its taps left to right.
                const int a0 = sxA - x_lo - r, b0 = sxB - x_lo - r;
                const int da = a0 >> 2, db = b0 >> 2;
                const unsigned sa = a0 & 3, sb = b0 & 3;
                if (ry < nrows) {
                    const uint32_t *q32 = reinterpret_cast<const uint32_t *>(sS + ry * tl.pitch);
                    uint32_t loA = q32[da], loB = q32[db], hiA = q32[da + 1], hiB = q32[db + 1];
                    uint32_t wA = __builtin_amdgcn_alignbyte(hiA, loA, sa), wB = __builtin_amdgcn_alignbyte(hiB, loB, sb);
                    f32x2 p0 = {(float)(wA & 0xff), (float)(wB & 0xff)};
                    f32x2 p1 = {(float)((wA >> 8) & 0xff), (float)((wB >> 8) & 0xff)};
                    f32x2 p2 = {(float)((wA >> 16) & 0xff), (float)((wB >> 16) & 0xff)};
                    f32x2 p3 = {(float)(wA >> 24), (float)(wB >> 24)};
                    f32x2 acc0, acc1; // {A, B} and {A+1, B+1}
                    int i = 0, t = 2;
                    for (; i + 4 <= ksz; i += 4, t++) {
                        loA = hiA;
                        loB = hiB;
                        hiA = q32[da + t];
                        hiB = q32[db + t];
                        wA = __builtin_amdgcn_alignbyte(hiA, loA, sa);
                        wB = __builtin_amdgcn_alignbyte(hiB, loB, sb);
                        const f32x2 c0 = {(float)(wA & 0xff), (float)(wB & 0xff)};
                        const f32x2 c1 = {(float)((wA >> 8) & 0xff), (float)((wB >> 8) & 0xff)};
                        const f32x2 c2 = {(float)((wA >> 16) & 0xff), (float)((wB >> 16) & 0xff)};
                        const f32x2 c3 = {(float)(wA >> 24), (float)(wB >> 24)};
                        const float t0 = sK[i], t1 = sK[i + 1], t2 = sK[i + 2], t3 = sK[i + 3];
                        if (i == 0) {
                            acc0 = t0 * p0;
                            acc1 = t0 * p1;
                        } else {
                            acc0 += t0 * p0;
                            acc1 += t0 * p1;
                        }
                        acc0 += t1 * p1;
                        acc1 += t1 * p2;
                        acc0 += t2 * p2;
                        acc1 += t2 * p3;
                        acc0 += t3 * p3;
                        acc1 += t3 * c0;
                        p0 = c0;
                        p1 = c1;
                        p2 = c2;
                        p3 = c3;
                    }
                    if (i < ksz) { // up to three taps left; they need p0..p3 only
                        float tt = sK[i];
                        acc0 += tt * p0;
                        acc1 += tt * p1;
                        if (i + 1 < ksz) {
                            tt = sK[i + 1];
                            acc0 += tt * p1;
                            acc1 += tt * p2;
                        }
                        if (i + 2 < ksz) {
                            tt = sK[i + 2];
                            acc0 += tt * p2;
                            acc1 += tt * p3;
                        }
                    }
                    float *out = sR + ry * tl.rstride + 4 * b_;
                    out[0] = acc0.x;
                    out[1] = acc1.x;
                    if (4 * b_ + 2 < no) {
                        out[2] = acc0.y;
                        out[3] = acc1.y;
                    }
                }
                continue;
            }
            // a pair at the right image border (sx + 1 clamps to sx): plain per-column streams
            int cofs[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int o = min(4 * b_ + j, no - 1);
                int sx = sX[o >> 1];
                cofs[j] = ((o & 1) ? min(sx + 1, W - 1) : sx) - x_lo - r;
            }
            if (ry < nrows) {
                const uint8_t *q = sS + ry * tl.pitch;
                float acc[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[j] = k0 * (float)q[cofs[j]];
#pragma unroll 4
                for (int i = 1; i < ksz; i++) {
                    const float t = sK[i];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[j] += t * (float)q[cofs[j] + i];
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (4 * b_ + j < no)
                        sR[ry * tl.rstride + 4 * b_ + j] = acc[j];
            }
        }
    } else {
    for (int b_ = wave; b_ < n_b; b_ += 4) {
        for (int a_ = lane; a_ < n_a; a_ += 64) {
            const int ry = lanes_on_rows ? a_ : b_, o = (lanes_on_rows ? b_ : a_) * ostep;
            int sx = tl.same_size ? dx0 + (o >> 1) : sX[o >> 1];
            int col = (o & 1) ? min(sx + 1, W - 1) : sx;
            const uint8_t *p = sS + ry * tl.pitch + (col - x_lo); // tap i sits at p[i - r]
            float acc;
            if (ksz == 3) {
                acc = (float)p[0] * kc + ((float)p[-1] + (float)p[1]) * kc1;
            } else if (ksz == 5) {
                acc = (float)p[0] * kc + ((float)p[-1] + (float)p[1]) * kc1 + ((float)p[-2] + (float)p[2]) * kc2;
            } else {
                const uint8_t *q = p - r;
                acc = k0 * (float)q[0];
                int i = 1;
                for (; i + 3 < ksz; i += 4) { // same left-to-right order, four taps per trip
                    float t0 = sK[i] * (float)q[i], t1 = sK[i + 1] * (float)q[i + 1];
                    float t2 = sK[i + 2] * (float)q[i + 2], t3 = sK[i + 3] * (float)q[i + 3];
                    acc += t0;
                    acc += t1;
                    acc += t2;
                    acc += t3;
                }
                for (; i < ksz; i++)
                    acc += sK[i] * (float)q[i];
            }
            sR[ry * tl.rstride + (tl.same_size ? (o >> 1) : o)] = acc;
        }
    }
    }
    __syncthreads();
    // ---- phase 3: column pass at the needed rows, then the lerps
    float *dst = img + (size_t)pi * Wk * Hk;
    const int st = tl.rstride;
    const int cs = tl.same_size ? 1 : 2; // row-pass slots per output column
    for (int idx = threadIdx.x; idx < (ndy << tl.tw_shift); idx += blockDim.x) {
        const int ty = idx >> tl.tw_shift, tx = idx & (tl.TWo - 1);
        if (tx >= ndx)
            continue;
        int dx = dx0 + tx, dy = dy0 + ty;
        const int sy = tl.same_size ? dy : sY[ty];
        int row0 = clampi(sy, 0, H - 1) - y_lo, row1 = clampi(sy + 1, 0, H - 1) - y_lo;
        const float *c0 = sR + row0 * st + cs * tx;
        const float *c1 = sR + row1 * st + cs * tx;
        float v00 = kc * c0[0];
        for (int i = 1; i <= r; i++)
            v00 += sK[r + i] * (c0[i * st] + c0[-i * st]);
        float out;
        if (tl.same_size) {
            out = v00;
        } else {
            float v01 = kc * c0[1];
            for (int i = 1; i <= r; i++)
                v01 += sK[r + i] * (c0[i * st + 1] + c0[-i * st + 1]);
            float v10 = v00, v11 = v01;
            if (row1 != row0) {
                v10 = kc * c1[0];
                v11 = kc * c1[1];
                for (int i = 1; i <= r; i++) {
                    v10 += sK[r + i] * (c1[i * st] + c1[-i * st]);
                    v11 += sK[r + i] * (c1[i * st + 1] + c1[-i * st + 1]);
                }
            }
            const float fx = sFx[tx], fy = sFy[ty];
            float h0, h1;
            if (sX[tx] >= W - 1) { // resize.cpp: dx >= xmax copies S[sx]
                h0 = v00;
                h1 = v10;
            } else {
                h0 = v00 * (1.f - fx) + v01 * fx;
                h1 = v10 * (1.f - fx) + v11 * fx;
            }
            out = h0 * (1.f - fy) + h1 * fy;
        }
        dst[(size_t)dy * Wk + dx] = out;
    }
}

// ---------------------------------------------------------------------------------
// A1 for the levels with long blur kernels (ksz >= 15: scale 1/8 and coarser), as two kernels.
// One tile of such a level depends on a frame region of (tile * scale + ksz)^2 bytes, so
// k_level_image's tiles shrink to a few dozen outputs and its fixed costs per workgroup dominate.
// Here the row pass runs over whole frame rows (k_level_rowpass: every lane busy, the frame read
// once) into a [H][2*Wk] float plane per image, and the column pass + both lerps read that plane
// through an LDS tile (k_level_colpass).  Same statements in the same order as k_level_image:
// row pass left to right, column pass centre then pairs outwards, horizontal then vertical lerp.
// ---------------------------------------------------------------------------------
// lanes = R rows x (64/R) groups of two interpolation pairs; with the row pitch = 1 (mod R) dwords and
// the groups s/2 dwords apart the 64 aligned-dword reads of one instruction fall in 64 banks.  One launch
// serves every split level: the frame rows are staged once (margin of the longest kernel) and each level
// runs its own taps over them into its own plane.
constexpr int RP_MAX_LEVELS = 4;
struct RowPassLevel {
    float *rowf;        // [image][H][NC]
    const int *colsrc;  // [NC]
    const float *kern;  // [ksz]
    int NC, ksz, rshift;
};
struct RowPassArgs {
    int n;
    RowPassLevel lv[RP_MAX_LEVELS];
};

__global__ void __launch_bounds__(256)
k_level_rowpass(const uint8_t *__restrict__ frames, const int2 *__restrict__ pairs, int W, int H, RowPassArgs args, int RB,
                int pitch, int r4, int rmax)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rp[];
    uint8_t *sS = s_rp;                                                   // [RB][pitch]: column c at byte r4 + c
    float *sKall = reinterpret_cast<float *>(s_rp + (size_t)RB * pitch);  // the levels' taps, one after the other
    const int pi = blockIdx.y;
    const int2 pr = pairs[pi >> 1];
    const uint8_t *src = frames + (size_t)((pi & 1) ? pr.y : pr.x) * W * H;
    const int y0 = blockIdx.x * RB, nrows = min(RB, H - y0);
    {
        int base = 0;
        for (int l = 0; l < args.n; l++) {
            for (int i = threadIdx.x; i < args.lv[l].ksz; i += 256)
                sKall[base + i] = args.lv[l].kern[i];
            base += args.lv[l].ksz;
        }
    }
    // stage nrows frame rows: the interior as dwords (W % 4 == 0 is required by the host), rmax reflected bytes each side
    const int nq = W >> 2;
    for (int idx = threadIdx.x; idx < nrows * nq; idx += 256) {
        const int i = idx / nq, c = idx - i * nq;
        *reinterpret_cast<uint32_t *>(sS + i * pitch + r4 + 4 * c) =
            *reinterpret_cast<const uint32_t *>(src + (size_t)(y0 + i) * W + 4 * c);
    }
    for (int idx = threadIdx.x; idx < nrows * 2 * rmax; idx += 256) {
        const int i = idx / (2 * rmax), j = idx - i * 2 * rmax;
        const int c = j < rmax ? j - rmax : W + (j - rmax); // -rmax..-1, W..W+rmax-1
        sS[i * pitch + r4 + c] = src[(size_t)(y0 + i) * W + reflect101(c, W)];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int kbase = 0;
    for (int l = 0; l < args.n; l++) {
        const RowPassLevel &L = args.lv[l];
        const float *sK = sKall + kbase;
        kbase += L.ksz;
        const int ksz = L.ksz, r = ksz >> 1, NC = L.NC, rshift = L.rshift;
        const int R = 1 << rshift, G = 64 >> rshift;
        const int li = lane & (R - 1), lg = lane >> rshift;
        const int ngroups = (NC + 3) >> 2;
        const int n_rb = (nrows + R - 1) >> rshift, n_gb = (ngroups + G - 1) / G;
        for (int item = wave; item < n_rb * n_gb; item += 4) {
            const int gb = item % n_gb, rb = item / n_gb;
            const int row = rb * R + li, grp = gb * G + lg;
            if (row >= nrows || grp >= ngroups)
                continue;
            const int oA = 4 * grp, oB = min(4 * grp + 2, NC - 2);
            const int cA = L.colsrc[oA], cB = L.colsrc[oB];
            const bool dupA = L.colsrc[oA + 1] == cA, dupB = L.colsrc[oB + 1] == cB; // pair at the right frame border: sx+1 clamps to sx
            const int a0 = r4 + cA - r, b0 = r4 + cB - r;
            const int da = a0 >> 2, db = b0 >> 2;
            const unsigned sa = a0 & 3, sb = b0 & 3;
            const uint32_t *q32 = reinterpret_cast<const uint32_t *>(sS + row * pitch);
            uint32_t loA = q32[da], loB = q32[db], hiA = q32[da + 1], hiB = q32[db + 1];
            uint32_t wA = __builtin_amdgcn_alignbyte(hiA, loA, sa), wB = __builtin_amdgcn_alignbyte(hiB, loB, sb);
            f32x2 p0 = {(float)(wA & 0xff), (float)(wB & 0xff)};
            f32x2 p1 = {(float)((wA >> 8) & 0xff), (float)((wB >> 8) & 0xff)};
            f32x2 p2 = {(float)((wA >> 16) & 0xff), (float)((wB >> 16) & 0xff)};
            f32x2 p3 = {(float)(wA >> 24), (float)(wB >> 24)};
            f32x2 acc0, acc1; // {A, B} and {A+1, B+1}
            int i = 0, t = 2;
            for (; i + 4 <= ksz; i += 4, t++) {
                loA = hiA;
                loB = hiB;
                hiA = q32[da + t];
                hiB = q32[db + t];
                wA = __builtin_amdgcn_alignbyte(hiA, loA, sa);
                wB = __builtin_amdgcn_alignbyte(hiB, loB, sb);
                const f32x2 c0 = {(float)(wA & 0xff), (float)(wB & 0xff)};
                const f32x2 c1 = {(float)((wA >> 8) & 0xff), (float)((wB >> 8) & 0xff)};
                const f32x2 c2 = {(float)((wA >> 16) & 0xff), (float)((wB >> 16) & 0xff)};
                const f32x2 c3 = {(float)(wA >> 24), (float)(wB >> 24)};
                const float t0 = sK[i], t1 = sK[i + 1], t2 = sK[i + 2], t3 = sK[i + 3];
                if (i == 0) {
                    acc0 = t0 * p0;
                    acc1 = t0 * p1;
                } else {
                    acc0 += t0 * p0;
                    acc1 += t0 * p1;
                }
                acc0 += t1 * p1;
                acc1 += t1 * p2;
                acc0 += t2 * p2;
                acc1 += t2 * p3;
                acc0 += t3 * p3;
                acc1 += t3 * c0;
                p0 = c0;
                p1 = c1;
                p2 = c2;
                p3 = c3;
            }
            if (i < ksz) { // up to three taps left; they need p0..p3 only
                float tt = sK[i];
                acc0 += tt * p0;
                acc1 += tt * p1;
                if (i + 1 < ksz) {
                    tt = sK[i + 1];
                    acc0 += tt * p1;
                    acc1 += tt * p2;
                }
                if (i + 2 < ksz) {
                    tt = sK[i + 2];
                    acc0 += tt * p2;
                    acc1 += tt * p3;
                }
            }
            // a clamped pair reads the same column twice: the same sum
            if (dupA)
                acc1.x = acc0.x;
            if (dupB)
                acc1.y = acc0.y;
            float *out = L.rowf + ((size_t)pi * H + y0 + row) * NC + 4 * grp;
            out[0] = acc0.x;
            out[1] = acc1.x;
            if (4 * grp + 2 < NC) {
                out[2] = acc0.y;
                out[3] = acc1.y;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_level_colpass(const float *__restrict__ rowf, float *__restrict__ img, int W, int H, int Wk, int Hk, int NC,
                const float *__restrict__ kern, int ksz, const int *__restrict__ xofs, const float *__restrict__ xfrac,
                const int *__restrict__ yofs, const float *__restrict__ yfrac)
{
    // One thread per level pixel, straight from the row-pass plane (round 2; the LDS-tiled form spent its time on
    // LDS reads, one per multiply-add): a pixel's two source columns sit side by side in the plane (one 8-byte
    // load per row, consecutive lanes consecutive pairs) and its two source rows sy, sy + 1 are one row apart, so
    // the taps of the second are the first's shifted by one: tap i of row sy needs rows sy + i and sy - i, tap i
    // of row sy + 1 needs sy + i + 1 and sy - i + 1 -- the upper one is loaded for the next tap of row sy anyway,
    // the lower one was the previous tap's.  Two loads and six packed operations per tap; the statements and
    // their order are k_level_image's (centre first, then pairs outwards; both lerps).
    extern __shared__ float s_taps[]; // [ksz]
    const int r = ksz >> 1;
    for (int i = threadIdx.x; i < ksz; i += 256)
        s_taps[i] = kern[i];
    __syncthreads();
    const int pi = blockIdx.z;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= Wk || dy >= Hk)
        return;
    const float *plane = rowf + (size_t)pi * H * NC + 2 * dx;
    // rows at most r + 1 outside the frame: one reflection does where the frame is taller than that (no loop around
    // the loads then: the whole column is written twice, under one uniform branch); the taps go four at a time, their
    // eight rows loaded before the first is used
    const int sy = yofs[dy];
    const int row0 = clampi(sy, 0, H - 1), row1 = clampi(sy + 1, 0, H - 1);
    float2u v0, v1;
    auto column = [&](auto reflect) {
        auto row = [&](int y) { return *reinterpret_cast<const float2u *>(plane + (size_t)reflect(y) * NC); };
        const float kc = s_taps[r];
        const float2u c0 = row(row0);
        float2u up = row(row0 + 1);   // U[1]
        float2u down_prev = c0;       // D[0]
        v0 = kc * c0;
        v1 = kc * up; // row1 == row0 + 1 wherever v1 is used: its centre is U[1]
        int i = 1;
        for (; i + 3 <= r; i += 4) {
            float2u dn[4], un[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                dn[q] = row(row0 - (i + q));     // D[i + q]
                un[q] = row(row0 + (i + q) + 1); // U[i + q + 1]
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float k = s_taps[r + i + q];
                v0 += k * (up + dn[q]);
                v1 += k * (un[q] + down_prev);
                up = un[q];
                down_prev = dn[q];
            }
        }
        for (; i <= r; i++) {
            const float k = s_taps[r + i];
            const float2u down = row(row0 - i);     // D[i]
            const float2u up_next = row(row0 + i + 1); // U[i + 1]
            v0 += k * (up + down);
            v1 += k * (up_next + down_prev);
            up = up_next;
            down_prev = down;
        }
    };
    if (H > r + 2)
        column([&](int y) { return reflect101_once(y, H); });
    else
        column([&](int y) { return reflect101(y, H); });
    if (row1 == row0) // both source rows clamp to the same frame row (above the first / below the last)
        v1 = v0;
    const float fx = xfrac[dx], fy = yfrac[dy];
    float h0, h1;
    if (xofs[dx] >= W - 1) { // resize.cpp: dx >= xmax copies S[sx]
        h0 = v0.x;
        h1 = v1.x;
    } else {
        h0 = v0.x * (1.f - fx) + v0.y * fx;
        h1 = v1.x * (1.f - fx) + v1.y * fx;
    }
    img[(size_t)pi * Wk * Hk + (size_t)dy * Wk + dx] = h0 * (1.f - fy) + h1 * fy;
}

// ---------------------------------------------------------------------------------
// A2: FarnebackPolyExp.  Tile 64x16 outputs; LDS holds the image tile with an
// n-pixel halo, then the three vertical-pass planes; the horizontal pass runs in
// double.  Clamped loads reproduce OpenCV's row clamping (vertical) and its
// replication of the edge triple (horizontal).
// ---------------------------------------------------------------------------------
constexpr int PX_TW = 64, PX_TH = 16;

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
    constexpr int TW = 64, TH = 16, LW = TW + 2 * N, LH = TH + 2 * N;
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

// rows of a tile of the fused expansion kernels (4K x 33 frames: level 0 with 12 / 16 / 20 rows 2.43 / 2.10 / 2.18 ms,
// level 1 with 16 / 20 / 24 / 32 rows 1.31 (before its 2 x 2 blocks) / 0.87 / 0.95 / 1.05 ms)
#ifndef TF_EXP_TH0
#define TF_EXP_TH0 16
#endif
#ifndef TF_EXP_TH1
#define TF_EXP_TH1 20
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

// OpenCV's border down-weighting table {0.14, 0.14, 0.4472, 0.4472, 0.4472} by distance d to an
// edge (1 beyond 5 px), as selects instead of a memory table: a table lookup is a global load,
// and its wait would drain every prefetched gather
__device__ __forceinline__ float border_weight(int d) { return d < 2 ? 0.14f : (d < 5 ? 0.4472f : 1.f); }
// The weight FarnebackUpdateMatrices applies: the product of the four edge weights, but only where its
// own test fires -- (unsigned)(x - 5) >= (unsigned)(W - 10) || the same in y.  For W, H >= 10 that is
// exactly where a weight differs from 1; below, W - 10 wraps and the test fires on fewer pixels than
// lie within 5 of an edge (W = 9: column 4 alone), and the others stay unweighted.  Kept as it is.
__device__ __forceinline__ float border_scale(int x, int y, int W, int H)
{
    const bool fires = (unsigned)(x - 5) >= (unsigned)(W - 10) || (unsigned)(y - 5) >= (unsigned)(H - 10);
    return fires ? border_weight(x) * border_weight(W - x - 1) * border_weight(y) * border_weight(H - y - 1) : 1.f;
}

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

// Flow source of the first update-matrices of a level.
//   mode 0: zero flow (coarsest scale, flags == 0)
//   mode 1: bilinear upsample of the coarser level's flow, times 1/pyr_scale (A5)
//   mode 2: explicit flow array at this level (stage tests)
struct FlowInit {
    int mode;
    const float2 *src; // coarse flow [pair][Hc*Wc] (mode 1) or level flow (mode 2)
    int Wc, Hc;
    const int *xofs, *yofs;
    const float *xfrac, *yfrac;
    float mul;
    // which two expansions (images of R, [image][5][Nk]) pair p compares: frames shared by the pairs of a
    // batch are expanded once.  Null: images 2p and 2p+1 (stage entry points).
    const int2 *rmap;
};

__device__ __forceinline__ int2 pair_images(const FlowInit &fi, int pair)
{
    return fi.rmap ? fi.rmap[pair] : make_int2(2 * pair, 2 * pair + 1);
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

// winsize 1 (m = 0).  FarnebackUpdateFlow_Blur primes its running sums with (m + 2) copies of the first
// row / column and takes one back when row m enters; with m = 0 the row that "enters" at y = 0 is row 0
// itself, the extra copy is never taken back, and every sum is first + current instead of current:
// G(y, x) = M(0,0) + M(y,0) + M(0,x) + M(y,x) (scale 1).  A setting nobody uses; kept as OpenCV computes it --
// including how: the column sums are OpenCV's chain (row 0 * 2 as a float product, then the float differences of
// consecutive rows accumulated in double), whose roundings a 1 x 1 "window" does nothing to average out.
// k_w1_vsum: one thread per column and channel walks the rows, V[pair][c][y][x]; k_w1_solve: per pixel.
__global__ void __launch_bounds__(64)
k_w1_vsum(const float *__restrict__ Min, double *__restrict__ V, int Wk, int Hk)
{
    const int x = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y, pair = blockIdx.z;
    if (x >= Wk)
        return;
    const size_t Nk = (size_t)Wk * Hk;
    const float *P = Min + ((size_t)pair * 5 + c) * Nk + x;
    double *o = V + ((size_t)pair * 5 + c) * Nk + x;
    double vs = (double)(P[0] * 2.f); // vsum[x] = srow0[x] * (m + 2)
    float prev = P[0];
#pragma unroll 8
    for (int y = 0; y < Hk; y++) {
        const float cur = P[(size_t)y * Wk];
        vs += (double)(cur - prev); // vsum[x] += srow1[x] - srow0[x]: rows y and max(y - 1, 0)
        prev = cur;
        o[(size_t)y * Wk] = vs;
    }
}
__global__ void k_w1_solve(const double *__restrict__ V, float2 *__restrict__ flow_out, int Wk, int Hk)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wk)
        return;
    const size_t Nk = (size_t)Wk * Hk;
    const double *Vp = V + (size_t)blockIdx.z * 5 * Nk + (size_t)y * Wk;
    double g[5];
#pragma unroll
    for (int c = 0; c < 5; c++)
        g[c] = Vp[c * Nk] + Vp[c * Nk + x]; // the row's running sum: column 0 twice, then the differences of neighbours
    const double idet = 1. / (g[0] * g[2] - g[1] * g[1] + 1e-3);
    flow_out[(size_t)blockIdx.z * Nk + (size_t)y * Wk + x] =
        make_float2((float)((g[0] * g[4] - g[1] * g[3]) * idet), (float)((g[2] * g[3] - g[1] * g[4]) * idet));
}

// ---------------------------------------------------------------------------------
// A4 sum for sum as FarnebackUpdateFlow_Blur runs it (option "fb_exact_sums").  OpenCV keeps ONE set of
// running sums for the whole image: per column a double that is primed with (m + 2) copies of the first row
// (a float product) and then, row after row from row 0, receives the FLOAT difference of the row that
// enters and the row that leaves; per row a double running sum of those across the columns, updated by
// double differences from column 0 on.  Every sum therefore carries the rounding history of everything
// above / left of it.  The marching kernels above and below restart their sums per segment and add across
// columns directly -- the same numbers up to ~1e-7 relative (the float differences' roundings), which is
// what decides FarnebackUpdateMatrices' discontinuous in-frame test for the rare border pixel whose sample
// point sits within that distance of the last row / column.  These two kernels repeat
// OpenCV's order exactly -- the same operations on the same operands, so the flow is bit-identical to the
// CPU path's -- at the price of its serial dependences.  (On a launch with many columns side by side the column sums come
// from k_flow_carry_pc<.., STORE> instead, straight from the expansions: launch_flow_iter.)
//   k_exact_vsum: one thread per column and channel walks all rows (coalesced across the wave: lanes are columns) and
//          stores every row's sums, vsum[pair][channel][y][x].
//   k_exact_hsolve: one WAVE takes ROWS rows and walks them together from column 0, 64 columns at a time: every lane
//          forms its column's double differences vsum[x + m] - vsum[x - m - 1] (coalesced loads, the next 64 columns'
//          loads in flight meanwhile) into LDS, lane r * 5 + c then runs the sum of row r, channel c over them IN ORDER
//          (its running sum stays in a register from chunk to chunk) and leaves the sums in their place, and every
//          lane solves its column's pixels of the ROWS rows (IEEE division, as the CPU path).  The serial part is 64
//          dependent additions per 64 x ROWS pixels; the rows of LDS are 65 doubles apart (conflict-free both ways).
// (Until round 4 the second kernel ran one THREAD per row -- 64 rows' lines per load -- over sums stored transposed, which
// the first kernel then wrote 8 bytes per line: 1.3 + 2.5 ms per iteration of one 4K pair.)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_exact_vsum(const float *__restrict__ Min, double *__restrict__ vsum, int Wk, int Hk, int m)
{
    const int x = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y, pair = blockIdx.z;
    if (x >= Wk)
        return;
    const size_t Nk = (size_t)Wk * Hk;
    const float *P = Min + ((size_t)pair * 5 + c) * Nk + x;
    double *V = vsum + ((size_t)pair * 5 + c) * Nk + x;
    double vs = (double)(P[0] * (float)(m + 2)); // vsum[x] = srow0[x] * (m + 2): a float product
    for (int y = 1; y < m; y++)
        vs += (double)P[(size_t)min(y, Hk - 1) * Wk];
#pragma unroll 8 // the loads of eight rows go out together: the chain through `vs` is one addition per row
    for (int y = 0; y < Hk; y++) {
        const float in = P[(size_t)min(y + m, Hk - 1) * Wk], out = P[(size_t)max(y - m - 1, 0) * Wk];
        vs += (double)(in - out); // vsum[x] += srow1[x] - srow0[x]
        V[(size_t)y * Wk] = vs;
    }
}

template <int ROWS>
__global__ void __launch_bounds__(64)
k_exact_hsolve(const double *__restrict__ vsum, float2 *__restrict__ flow_out, int Wk, int Hk, int m, double scale)
{
    constexpr int NCH = ROWS * 5, LDW = 65, GRP = 16;
    static_assert(NCH <= 64, "one chain per lane");
    __shared__ double D[NCH][LDW];
    const int lane = threadIdx.x, pair = blockIdx.z, y0 = blockIdx.x * ROWS;
    const size_t Nk = (size_t)Wk * Hk;
    const double *V = vsum + (size_t)pair * 5 * Nk + (size_t)y0 * Wk;
    float2 *out = flow_out + (size_t)pair * Nk + (size_t)y0 * Wk;
    const int nrows = min(ROWS, Hk - y0);
    const int kr = lane / 5, kc = lane - kr * 5; // the chain this lane runs: row kr, channel kc
    const bool chain = lane < nrows * 5;
    double g = 0.0;
    if (chain) { // the priming: column 0 (m + 2) times, then columns 1 .. m - 1
        const double *row = V + (size_t)kc * Nk + (size_t)kr * Wk;
        g = row[0] * (double)(m + 2);
        for (int x = 1; x < m; x++)
            g += row[min(x, Wk - 1)];
    }
    // the differences of the chunk at x0, one column per lane (rows past the level's last: row 0 of the tile again, unused)
    double d[NCH];
    auto fetch = [&](int x0) {
        const int xa = min(x0 + lane + m, Wk - 1), xb = min(max(x0 + lane - m - 1, 0), Wk - 1);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const double *row = V + (size_t)(r < nrows ? r : 0) * Wk;
#pragma unroll
            for (int c = 0; c < 5; c++)
                d[r * 5 + c] = row[(size_t)c * Nk + xa] - row[(size_t)c * Nk + xb];
        }
    };
    fetch(0);
    for (int x0 = 0; x0 < Wk; x0 += 64) {
#pragma unroll
        for (int k = 0; k < NCH; k++)
            D[k][lane] = d[k];
        if (x0 + 64 < Wk)
            fetch(x0 + 64); // in flight while this chunk is summed and solved
        lds_wave_sync();
        if (chain) {
            const int n = min(64, Wk - x0);
#pragma unroll
            for (int j0 = 0; j0 < 64; j0 += GRP) {
                double v[GRP];
#pragma unroll
                for (int q = 0; q < GRP; q++)
                    v[q] = D[lane][j0 + q];
                if (j0 + GRP <= n) {
                    v[0] = g + v[0];
#pragma unroll
                    for (int q = 1; q < GRP; q++)
                        v[q] = v[q - 1] + v[q];
                    g = v[GRP - 1];
                } else {
#pragma unroll
                    for (int q = 0; q < GRP; q++) {
                        if (j0 + q < n)
                            g += v[q];
                        v[q] = g;
                    }
                }
#pragma unroll
                for (int q = 0; q < GRP; q++)
                    D[lane][j0 + q] = v[q];
            }
        }
        lds_wave_sync();
        if (x0 + lane < Wk) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                if (r < nrows) {
                    const double g11 = D[r * 5 + 0][lane] * scale, g12 = D[r * 5 + 1][lane] * scale, g22 = D[r * 5 + 2][lane] * scale,
                                 h1 = D[r * 5 + 3][lane] * scale, h2 = D[r * 5 + 4][lane] * scale;
                    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                    out[(size_t)r * Wk + x0 + lane] =
                        make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
                }
            }
        }
        lds_wave_sync();
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

// ---------------------------------------------------------------------------------
// One pixel of A3 split into "issue the loads" and "finish the arithmetic" (one column per lane),
// so a marching wave can keep the gathers of later rows in flight: used by the producers of
// k_flow_iter_pc below.  Same statements as update_matrix_px, rounding for rounding (see gather1_finish).
// ---------------------------------------------------------------------------------
struct Gather1 {
    float2u r0a, r0b;            // R0 at the pixel: (c0, c1), (c2, c3)
    float r0c;                   // ... and c4
    float4u t01, t23, b01, b23;  // R1 on rows y1 / y1 + 1: a channel pair at x1 (.xy) and at x1 + 1 (.zw)
    float2u t4, b4;              // c4 at x1, x1 + 1
    float dx, dy, fx, fy;
    bool inb;
};

// The wave-uniform plane bases stay in SGPRs and every load is base + one 32-bit byte offset per lane: the
// pair planes share one offset (8 bytes per pixel), the c4 planes another (4 bytes per pixel).
struct PlaneBases {
    const float *r0_01, *r0_23, *r0_4;
    const float *r1_01, *r1_23, *r1_4;
    const float *r1_01b, *r1_23b, *r1_4b; // the same planes one row down: the bilinear bottom row shares the top row's offset
};
__device__ __forceinline__ PlaneBases plane_bases(const float *R0, const float *R1, size_t Nk, int Wk)
{
    return PlaneBases{R0, R0 + r_off23(Nk), R0 + r_off4(Nk), R1, R1 + r_off23(Nk), R1 + r_off4(Nk),
                      R1 + 2 * Wk, R1 + r_off23(Nk) + 2 * Wk, R1 + r_off4(Nk) + Wk};
}

__device__ __forceinline__ float ld_f32(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float2u ld_f32x2(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float2u *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float4u ld_f32x4(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float4u *>(reinterpret_cast<const char *>(base) + byte_off);
}

// The addresses and weights of one pixel's gathers (everything that depends on the flow), apart from the loads.
struct GatherPrep {
    unsigned o8, o4, q8, q4; // byte offsets: the pixel in an 8-byte / 4-byte plane of R0, the top-left tap in R1's
    float dx, dy, fx, fy;
    bool inb;
};
__device__ __forceinline__ GatherPrep gather1_prep(int Wk, int Hk, int x, int y, float2 fl)
{
    GatherPrep p;
    const unsigned o = (unsigned)y * Wk + x;
    p.o8 = o * 8u;
    p.o4 = o * 4u;
    const float fx = x + fl.x, fy = y + fl.y;
    const float flx = floorf(fx), fly = floorf(fy); // (float)(int)floor(f) == floor(f) wherever the int exists
    const int x1 = (int)flx, y1 = (int)fly;
    p.dx = fl.x;
    p.dy = fl.y;
    p.fx = fx - flx;
    p.fy = fy - fly;
    p.inb = (unsigned)x1 < (unsigned)(Wk - 1) && (unsigned)y1 < (unsigned)(Hk - 1);
    // out-of-frame taps load from a clamped (valid) address and are discarded: no branch around the loads
    // (rows and widths are far below 2^24: the 24-bit multiply-add is exact and a single full-rate instruction)
    const unsigned qt = __umul24((unsigned)med3i(y1, 0, Hk - 2), (unsigned)Wk) + (unsigned)med3i(x1, 0, Wk - 2);
    p.q8 = qt * 8u;
    p.q4 = qt * 4u;
    return p;
}
__device__ __forceinline__ void gather1_load(Gather1 &g, const PlaneBases &pb, const GatherPrep &p)
{
    g.r0a = ld_f32x2(pb.r0_01, p.o8);
    g.r0b = ld_f32x2(pb.r0_23, p.o8);
    g.r0c = ld_f32(pb.r0_4, p.o4);
    g.dx = p.dx;
    g.dy = p.dy;
    g.fx = p.fx;
    g.fy = p.fy;
    g.inb = p.inb;
    g.t01 = ld_f32x4(pb.r1_01, p.q8);
    g.b01 = ld_f32x4(pb.r1_01b, p.q8);
    g.t23 = ld_f32x4(pb.r1_23, p.q8);
    g.b23 = ld_f32x4(pb.r1_23b, p.q8);
    g.t4 = ld_f32x2(pb.r1_4, p.q4);
    g.b4 = ld_f32x2(pb.r1_4b, p.q4);
}
__device__ __forceinline__ void gather1_issue(Gather1 &g, const PlaneBases &pb, int Wk, int Hk, int x, int y, float2 fl)
{
    gather1_load(g, pb, gather1_prep(Wk, Hk, x, y, fl));
}

// update_matrix_px's statements on the channel pairs as they were loaded: per channel the bilinear sum is
// ((a00 * t(x1) + a01 * t(x1 + 1)) + a10 * b(x1)) + a11 * b(x1 + 1), two channels per packed instruction, every
// operation rounded on its own (no contraction: the file is built with -ffp-contract=off), so this M is
// k_update_matrices' and the CPU path's bit for bit.  (Round 2 let the compiler fuse these multiply-adds, +1.8 %
// frames/s at 4K; M then differed in the last bit, which was half of why border pixels flipped: DESIGN.md section 4.)
// wx, wy: the edge weights of the column and of the row.  FarnebackUpdateMatrices multiplies border[x] (x < 5),
// border[W-1-x] (x >= W-5), border[y], border[H-1-y]; from 10 x 10 up at most one factor per direction differs
// from 1 (and a factor of exactly 1 changes nothing), so the product is border(min(x, W-1-x)) * border(min(y,
// H-1-y)), the column's factor first as in the original: bit-identical, with the column's half a constant of the
// march and the row's half wave-uniform.
// UNWEIGHTED: the pixel is at least 5 pixels from every edge of the level (weight exactly 1: the five multiplications by
// it change no bit and are not issued).
template <bool UNWEIGHTED = false>
__device__ __forceinline__ void gather1_finish(const Gather1 &g, float wx, float wy, float m[5])
{
    const float fx = g.fx, fy = g.fy, dx = g.dx, dy = g.dy;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    float2u r23 = a00 * g.t01.xy + a01 * g.t01.zw + a10 * g.b01.xy + a11 * g.b01.zw;
    float2u r45 = a00 * g.t23.xy + a01 * g.t23.zw + a10 * g.b23.xy + a11 * g.b23.zw;
    float r6 = a00 * g.t4.x + a01 * g.t4.y + a10 * g.b4.x + a11 * g.b4.y;
    r45 = (g.r0b + r45) * 0.5f;
    r6 = (g.r0c + r6) * 0.25f;
    const float o6 = g.r0c * 0.5f;
    r23 = g.inb ? r23 : float2u{0.f, 0.f};
    r45 = g.inb ? r45 : g.r0b;
    r6 = g.inb ? r6 : o6;
    r23 = (g.r0a - r23) * 0.5f;
    float r2 = r23.x, r3 = r23.y, r4 = r45.x, r5 = r45.y;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if (!UNWEIGHTED) {
        const float scale = wx * wy;
        r2 *= scale;
        r3 *= scale;
        r4 *= scale;
        r5 *= scale;
        r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// ---------------------------------------------------------------------------------
// The vertical window sums, kept as FarnebackUpdateFlow_Blur keeps them.  OpenCV holds ONE running sum per column and
// channel for the whole image: a double primed with (m + 2) copies of row 0 -- a FLOAT product -- plus rows 1 .. m - 1,
// which then receives, row after row from row 0, the FLOAT difference of the row that enters the window and the row that
// leaves it (optflowgf.cpp, FarnebackUpdateFlow_Blur: `vsum[x] += srow1[x] - srow0[x]`).  Every sum so carries the roundings of all the float
// differences above it, ~1e-7 relative: a march that starts its sums afresh at a segment's first row gets other
// roundings, and that much decides FarnebackUpdateMatrices' in-frame test for the rare border pixel whose sample point
// sits within float resolution of the last row or column.  The marching kernels therefore run
// OpenCV's chain: the segment at the top of a column primes it as OpenCV does, every other segment starts from the
// chain's value after the row above it (its "carry").  Two ways to have that value:
//   mode 1 (hand-off inside the launch): the segments of a column run one after the other; a workgroup draws a ticket,
//          tickets are dealt segment-major, and a segment waits for the carry its predecessor -- an earlier ticket, so
//          resident or finished -- publishes when it is done.  The chain is then OpenCV's operation for operation: the
//          vertical sums are bit-identical to the CPU path's.  Costs nothing where a launch has more columns of
//          workgroups than the chip has slots (the predecessor is done before the successor is dispatched).
//   mode 0 (carries from a pre-pass): where a launch is small, segments must run side by side.  A first launch marches
//          every segment for its sum of differences alone (k_flow_carry_pc; k_blur_carry where M is in memory), a scan
//          adds them up along each column (k_carry_scan), and the segments read their carry.  Adding a segment's
//          differences up before adding them to the chain re-associates double additions: ~1e-16 relative.
// The host picks per launch (choose_march).
// ---------------------------------------------------------------------------------
struct ColumnCarry {
    int mode;             // 0: carry[((seg * pairs + pair) * 5 + c) * Wk + x], written by an earlier launch
                          // 1: carry[(((seg * pairs + pair) * strips + strip) * 5 + c) * 128 + column of the strip], handed over in the launch
    int segs, pairs, strips;
    double *carry;
    unsigned *flags;      // mode 1: [((seg * pairs + pair) * strips + strip) * 2 + producer wave] == epoch once that wave's carries are stored
    unsigned epoch;       // never 0; a handle counts its chained launches
    unsigned *ticket;     // mode 1: eight counters (one list of work per XCD), zeroed before the launch
    unsigned *fault;      // host-visible word (pinned): set if a wait for a carry gave up
};

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define TF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// One wave waits for the word its predecessor stores (cdna_hip_programming.md, guideline 16: the word is written by an
// agent-scope atomic store after the storing wave drained its payload stores; polled relaxed; the payload is then read
// with agent-scope loads, which pass the L1).  Bounded: after ~5 s the wave sets the handle's fault word and goes on
// (with a wrong carry -- the host turns the fault into an error), so every wave of the grid reaches its end.
__device__ __forceinline__ void wait_for_epoch(unsigned *flag, unsigned epoch, unsigned *fault)
{
    gu32 *f = (gu32 *)flag;
    if (__hip_atomic_load(f, TF_RLX_AGENT) != epoch) {
        const unsigned long long t0 = wall_clock64(); // 100 MHz
        while (__hip_atomic_load(f, TF_RLX_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > 500000000ull) {
                __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (no instruction: the loads that follow stay behind the poll)
}

// ---------------------------------------------------------------------------------
// One column per lane, row after row of the 2x2 systems M (A3, with A5 on the fly): what the producer waves of
// k_flow_iter_pc and the lanes of k_flow_carry_pc do.  The gathers of row e + 1 are in flight while row e is finished
// (two or three rows ahead were measured slower, on a full chip -- 2.94 -> 3.37 ms per level-0 launch at 4K x 32 -- and on a
// part-empty one alike -- 0.92 -> 1.08 -> 2.5 ms at level 1: a lone wave issues an instruction every ~8 cycles, and that, not
// memory latency, is what a step of ~130 instructions waits for), the flow of row e + 2 is the first load of a step, and for A5 the two
// lerps of a row's flow run one step after its four coarse loads.
// FLOW: where the iteration's input flow comes from -- 0: zero (coarsest scale), 1: flow_in, 2: A5 on the fly,
// resize(coarser flow, INTER_LINEAR) * 1/pyr_scale through `fi` (the statements of k_flow_upsample; the column's
// table entries are loaded once per lane, the row's are the same address for all lanes).
// ---------------------------------------------------------------------------------
template <int FLOW>
struct RowProducer {
    struct FlowRaw {
        float2 a, b, d, e; // FLOW == 2: the coarse flow at (sx, sy0), (sx + 1, sy0), (sx, sy1), (sx + 1, sy1); else a = the flow
        float fy;
    };
    int Wk, Hk, x;
    PlaneBases pb;
    const float2 *fin, *coarse;
    int Wc, Hc;
    const int *yofs;
    const float *yfrac;
    float mul;
    int up_sx, up_sx1;
    float up_fx;
    bool up_edge;
    float wx;          // the column's edge weight
    Gather1 G;         // the row being gathered
    FlowRaw F;         // the flow of the row after it
    int y_fin, y_iss;  // the (clamped) rows of G and F
    int sy_q;          // FLOW == 2: the table entries of the row whose flow is loaded next (scalar loads, fetched a step early)
    float fy_q;

    // rows_ofs / rows_frac: fi.yofs / fi.yfrac as __restrict__ kernel arguments of their own -- a march reads a row's
    // entries with SCALAR loads (the same address for all lanes), and the compiler only issues those for memory it can
    // prove nothing in the kernel writes; behind the ticket's atomic a pointer out of the by-value struct no longer
    // qualifies and the entries came as two vector loads per step (+6 % vector-memory instructions, +15 % time of an
    // A5 launch)
    __device__ __forceinline__ void init(const float *R, const float2 *flow_in, const FlowInit &fi, const int *rows_ofs,
                                         const float *rows_frac, int pair, size_t Nk, int Wk_, int Hk_, int x_)
    {
        Wk = Wk_;
        Hk = Hk_;
        x = x_;
        const int2 im = pair_images(fi, pair);
        pb = plane_bases(R + (size_t)im.x * 5 * Nk, R + (size_t)im.y * 5 * Nk, Nk, Wk);
        fin = FLOW == 1 ? flow_in + (size_t)pair * Nk : nullptr;
        coarse = FLOW == 2 ? fi.src + (size_t)pair * fi.Wc * fi.Hc : nullptr;
        Wc = fi.Wc;
        Hc = fi.Hc;
        yofs = rows_ofs;
        yfrac = rows_frac;
        mul = fi.mul;
        up_sx = up_sx1 = 0;
        up_fx = 0.f;
        up_edge = false;
        if (FLOW == 2) {
            up_sx = fi.xofs[x];
            up_fx = fi.xfrac[x];
            up_edge = up_sx >= Wc - 1; // resize.cpp: dx >= xmax copies S[sx]
            up_sx1 = min(up_sx + 1, Wc - 1);
        }
        wx = border_weight(min(x, Wk - 1 - x));
        sy_q = 0;
        fy_q = 0.f;
    }
    __device__ __forceinline__ void fetch_row_entries(int row)
    {
        if (FLOW == 2) {
            sy_q = yofs[row];
            fy_q = yfrac[row];
        }
    }
    __device__ __forceinline__ FlowRaw load_flow(int row) const // row: clamped to the level; FLOW == 2: its table entries in sy_q, fy_q
    {
        FlowRaw r;
        r.a = r.b = r.d = r.e = make_float2(0.f, 0.f);
        r.fy = 0.f;
        if (FLOW == 1) {
            const unsigned off = ((unsigned)row * Wk + x) * 8u;
            r.a = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(fin) + off);
        } else if (FLOW == 2) {
            const int sy = sy_q;
            r.fy = fy_q;
            const int sy0 = clampi(sy, 0, Hc - 1), sy1 = clampi(sy + 1, 0, Hc - 1);
            r.a = coarse[sy0 * Wc + up_sx];
            r.b = coarse[sy0 * Wc + up_sx1];
            r.d = coarse[sy1 * Wc + up_sx];
            r.e = coarse[sy1 * Wc + up_sx1];
        }
        return r;
    }
    __device__ __forceinline__ float2 flow_of(const FlowRaw &r) const
    {
        if (FLOW != 2)
            return r.a;
        const float a0 = 1.f - up_fx;
        float2 h0 = make_float2(r.a.x * a0 + r.b.x * up_fx, r.a.y * a0 + r.b.y * up_fx);
        float2 h1 = make_float2(r.d.x * a0 + r.e.x * up_fx, r.d.y * a0 + r.e.y * up_fx);
        if (up_edge) {
            h0 = r.a;
            h1 = r.d;
        }
        const float b0 = 1.f - r.fy;
        return make_float2((h0.x * b0 + h1.x * r.fy) * mul, (h0.y * b0 + h1.y * r.fy) * mul);
    }
    // before the first next(): the march starts at row e0 (rows outside the level are their nearest row: replicated border)
    __device__ __forceinline__ void start(int e0)
    {
        y_fin = clampi(e0, 0, Hk - 1);
        y_iss = clampi(e0 + 1, 0, Hk - 1);
        fetch_row_entries(y_fin);
        gather1_issue(G, pb, Wk, Hk, x, y_fin, flow_of(load_flow(y_fin)));
        fetch_row_entries(y_iss);
        F = load_flow(y_iss);
        fetch_row_entries(clampi(e0 + 2, 0, Hk - 1));
    }
    // Row e of M (the e-th call after start(e0) is for row e0 + e ...: the caller passes the unclamped row).  INTERIOR
    // (compile time): rows e .. e + 3 lie inside the level, row e at least 5 rows from its top and bottom and the column
    // at least 5 from its sides, so nothing is clamped and the pixel's edge weight is 1 (x * 1.f == x: the same bits) --
    // the scalar clamps and selects of the general step and the five multiplications by the weight are not issued.
    template <bool INTERIOR>
    __device__ __forceinline__ void next(int e, float m[5])
    {
        // the row's edge weight is wave-uniform: border_weight() as scalar selects on the floats' bits (0.14f, 0.4472f, 1.f)
        const int dyb = min(y_fin, Hk - 1 - y_fin);
        const unsigned wyb = INTERIOR ? 0x3f800000u : (dyb < 2 ? 0x3e0f5c29u : (dyb < 5 ? 0x3ee4f766u : 0x3f800000u));
        // the flow of the row after next is the first load of the step: when the step ends by moving it into
        // place the wave waits for a load a whole step old, not for one it has just issued
        const int y_flow = INTERIOR ? e + 2 : clampi(e + 2, 0, Hk - 1);
        const FlowRaw Fn = load_flow(y_flow);
        fetch_row_entries(INTERIOR ? e + 3 : clampi(e + 3, 0, Hk - 1));
        gather1_finish<INTERIOR>(G, wx, __uint_as_float(wyb), m);
        gather1_issue(G, pb, Wk, Hk, x, y_iss, flow_of(F));
        F = Fn;
        y_fin = y_iss;
        y_iss = y_flow;
    }
};

// ---------------------------------------------------------------------------------
// A3+A4 fused, roles split inside the workgroup (the default on large levels).  Four waves march a strip
// of 128 columns together: waves 0-1 are PRODUCERS (one column per lane: RowProducer makes row e of M
// from R0, R1 and the flow; the lane keeps the window's 2M+1 rows of its column in an LDS ring, runs OpenCV's
// vertical running sum over it in double and publishes that sum), waves 2-3 are CONSUMERS taking turns by row
// (two columns per lane: a consumer adds the sums across columns -- pair sums through LDS, as
// k_blur_solve_wave --, solves and writes the flow; it copies what it needs of a row's sums out of s_v before
// the step's barrier and then has two steps for the rest, so the producers set the pace).  One workgroup
// barrier per row, sums double-buffered by step parity.  M is never stored: the window costs 38 KB of LDS per
// 112 output columns, 53.8 KB per workgroup, 3 workgroups = 12 waves per CU.
// A workgroup marches rows r0 .. r1 - 1 of its strip.  Its first WIN = 2M+1 steps only fill the ring (rows
// r0 - M - 1 .. r0 + M - 1: what the window of row r0 - 1 held); then the chain continues from the carry (see
// ColumnCarry) and every step slides it down one row: vs += (double)(entering row - leaving row), the difference in float.
// ---------------------------------------------------------------------------------
#ifndef TF_PC_CONS
#define TF_PC_CONS 2 // consumer waves per workgroup: they take turns by row (1: 1701, 2: 1723 frames/s at 4K x 32)
#endif
#define TF_PC_THREADS (128 + 64 * TF_PC_CONS)
template <int M, int FLOW>
__global__ void __launch_bounds__(TF_PC_THREADS)
k_flow_iter_pc(const float *__restrict__ R, const float2 *__restrict__ flow_in, float2 *__restrict__ flow_out, int Wk,
               int Hk, double scale, int seg, FlowInit fi, const int *__restrict__ rows_ofs, const float *__restrict__ rows_frac,
               ColumnCarry cc)
{
    static_assert(M & 1, "the pair-sum window needs an odd half-width");
    // A strip's halo is M columns rounded up to whole lanes: 112 outputs per strip for M = 7, every strip starting on
    // a multiple of 8 columns = 64 bytes of the 8-byte planes (114 outputs on strips that start on odd columns were
    // measured 3 % slower: every 512-byte row piece a wave loads then straddles one more 128-byte line).
    constexpr int HALO = (M + 1) & ~1;
    constexpr int OUTC = 128 - 2 * HALO, WIN = 2 * M + 1;
    __shared__ float ring[WIN][5][128];   // the window's rows of M, one column per producer lane
    __shared__ double s_v[2][5][128];     // vertical window sums of the row just produced (double-buffered by step parity)
    constexpr int CONS = TF_PC_CONS;
    __shared__ double s_p[CONS][5][64]; // each consumer's pair sums
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned bx, by;
    int pair;
    if (cc.mode == 1) {
        // Tickets, one list per XCD.  The launch's pairs are dealt to the eight XCDs in contiguous runs, and the workgroups
        // an XCD receives (the hardware deals workgroup i to XCD i mod 8) draw from their XCD's list first: the strips of a
        // pair stand side by side in one L2 (their halo columns are read once), and so do consecutive pairs, which read
        // the frame they share -- R0 of one, R1 of the next -- at the same rows at about the same time.  A list is
        // segment-major (then strip, then pair): every segment of a column has a later ticket in the same list than the
        // segment above it, and whoever holds a ticket is resident, so a wait for a predecessor always ends, whatever
        // order the hardware dispatches in.  A workgroup whose own list is used up takes from the next one (lists
        // differ in length when the pairs do not divide by eight): every workgroup finds exactly one ticket.
        // (the ticket travels through a double of s_p, which the consumers first touch many barriers later: a word of
        // its own would be the 43rd LDS granule of 1280 bytes and cost the CU its third workgroup)
        unsigned *s_ticket = reinterpret_cast<unsigned *>(&s_p[0][0][0]);
        if (threadIdx.x == 0) {
            unsigned got = ~0u, list = 0;
            for (unsigned k = 0; k < 8 && got == ~0u; k++) {
                list = (blockIdx.x + k) & 7;
                const unsigned n = (list + 1) * (unsigned)cc.pairs / 8 - list * (unsigned)cc.pairs / 8;
                const unsigned len = n * (unsigned)(cc.segs * cc.strips);
                if (len == 0 || __hip_atomic_load(cc.ticket + list, TF_RLX_AGENT) >= len)
                    continue; // (a look first: a list that is used up is not counted up again by every passer-by)
                const unsigned t = __hip_atomic_fetch_add(cc.ticket + list, 1u, TF_RLX_AGENT);
                if (t < len)
                    got = t;
            }
            s_ticket[0] = got;
            s_ticket[1] = list;
        }
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket[0]), list = __builtin_amdgcn_readfirstlane(s_ticket[1]);
        __syncthreads(); // every wave has read it
        if (t == ~0u)
            return; // (no ticket left: the grid and the lists have parted -- touch nothing)
        const unsigned p0 = list * (unsigned)cc.pairs / 8, n = (list + 1) * (unsigned)cc.pairs / 8 - p0;
        const unsigned per_seg = n * (unsigned)cc.strips;
        by = t / per_seg;
        const unsigned rem = t - by * per_seg;
        bx = rem / n;
        pair = (int)(p0 + (rem - bx * n));
    } else {
        xcd_pair_tile(bx, by, pair);
    }
    const size_t Nk = (size_t)Wk * Hk;
    const int r0 = by * seg, r1 = min(r0 + seg, Hk);
    // step s: the producers make row e = r0 - M - 1 + s of M (s < n_rows) and, from s = WIN on, the window sums of row
    // e - M; a consumer turns the sums of step s - 1 into the flow of row r0 + (s - 1) - WIN
    const int e0 = r0 - M - 1, n_rows = (r1 - r0) + WIN, nsteps = n_rows + 1;
    if (wave < 2) {
        // the producers set a step's pace: where a SIMD holds a producer and consumers (or another kernel's waves) the
        // producer issues first (4K x 32, one batch in flight: level 0 2859 -> 2806 us, level 1 894 -> 851, 1750 -> 1784
        // frames/s; with two batches in flight nothing changes)
        __builtin_amdgcn_s_setprio(1);
        const int col = wave * 64 + lane;
        const int x = clampi((int)bx * OUTC - HALO + col, 0, Wk - 1); // replicated border columns
        RowProducer<FLOW> P;
        P.init(R, flow_in, fi, rows_ofs, rows_frac, pair, Nk, Wk, Hk, x);
        P.start(e0);
        double vs[5] = {0, 0, 0, 0, 0};
        int slot = 0;
        // The ring's first WIN rows.  The segment at the top of the level also primes the chain, statement for statement
        // as FarnebackUpdateFlow_Blur does: vsum = row 0 * (m + 2), a float product; vsum += row y for y = 1 .. m - 1.
        for (int s = 0; s < WIN; s++) {
            float m[5];
            P.template next<false>(e0 + s, m);
#pragma unroll
            for (int c = 0; c < 5; c++)
                ring[slot][c][col] = m[c];
            if (r0 == 0) {
                if (s == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++)
                        vs[c] = (double)(m[c] * (float)(M + 2));
                } else if (s >= M + 2 && s <= 2 * M) { // rows 1 .. M - 1 (beyond the last row: the last row again)
#pragma unroll
                    for (int c = 0; c < 5; c++)
                        vs[c] += (double)m[c];
                }
            }
            slot = slot + 1 == WIN ? 0 : slot + 1;
            lds_barrier();
        }
        if (r0 != 0) { // the chain's value after row r0 - 1
            if (cc.mode == 1) {
                const size_t item = ((size_t)by * cc.pairs + pair) * cc.strips + bx;
                wait_for_epoch(cc.flags + item * 2 + wave, cc.epoch, cc.fault);
                gu64 *C = (gu64 *)(cc.carry + item * (5 * 128) + col);
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] = __longlong_as_double((long long)__hip_atomic_load(C + c * 128, TF_RLX_AGENT));
            } else {
                const double *C = cc.carry + (((size_t)by * cc.pairs + pair) * 5) * Wk + x;
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] = C[(size_t)c * Wk];
            }
        }
        // One step of the march proper.  A march runs three loops: the rows at the top of the level, the interior (see
        // RowProducer::next), the rows at the bottom.
        auto step = [&](int s, auto interior) {
            constexpr bool INTERIOR = decltype(interior)::value;
            float m[5];
            P.template next<INTERIOR>(e0 + s, m);
            // the row that leaves the window sits in the slot the new row takes (e - WIN == e mod WIN);
            // only this lane ever touches its column of the ring
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float old = ring[slot][c][col];
                ring[slot][c][col] = m[c];
                vs[c] += (double)(m[c] - old); // vsum[x] += srow1[x] - srow0[x]: a float difference accumulated in double
                s_v[s & 1][c][col] = vs[c];
            }
            slot = slot + 1 == WIN ? 0 : slot + 1;
            lds_barrier();
        };
        // steps whose row e = e0 + s lies in [5, Hk - 6] (then e + 3 <= Hk - 1 too)
        int s_in0 = min(max(5 - e0, WIN), n_rows), s_in1 = min(max(Hk - 5 - e0, s_in0), n_rows);
        // ... and only in strips whose 128 columns all lie at least 5 pixels inside the level (wave-uniform): there the
        // interior step also drops the edge weight's five multiplications
        const int strip0 = (int)bx * OUTC - HALO;
        if (strip0 < 5 || strip0 + 127 > Wk - 6)
            s_in0 = s_in1 = WIN;
        int s = WIN;
        for (; s < s_in0; s++)
            step(s, std::false_type{});
        for (; s < s_in1; s++)
            step(s, std::true_type{});
        for (; s < n_rows; s++)
            step(s, std::false_type{});
        if (cc.mode == 1 && (int)by + 1 < cc.segs) {
            // vs is the chain after row r1 - 1: the carry of the segment below.  Write-through stores, drained, then this
            // wave's flag (each producer wave hands over its own 64 columns: no barrier between the two).
            const size_t item = ((size_t)(by + 1) * cc.pairs + pair) * cc.strips + bx;
            gu64 *C = (gu64 *)(cc.carry + item * (5 * 128) + col);
#pragma unroll
            for (int c = 0; c < 5; c++)
                __hip_atomic_store(C + c * 128, (unsigned long long)__double_as_longlong(vs[c]), TF_RLX_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store((gu32 *)(cc.flags + item * 2 + wave), cc.epoch, TF_RLX_AGENT);
        }
        lds_barrier(); // step n_rows: the consumers' last row
    } else {
        // The consumers take turns: wave 2 + k serves the steps with s % TF_PC_CONS == k.  In its step a consumer first
        // takes what it needs of the row's sums out of s_v (the sum of its two columns and one single column at each end of
        // the window) -- that much must be done before the step's barrier, after which the producers overwrite the
        // buffer -- and then has until its next turn for the exchange of pair sums, the solve and the store.
        const int who = wave - 2;
        double(*sp)[64] = s_p[who];
        // A lane's two columns are strip columns 2 * lane and the next one; the halo is whole lanes
        const int c0 = (int)bx * OUTC - HALO + 2 * lane;
        constexpr int first_out = HALO / 2, last_out = (128 - HALO) / 2 - 1; // lanes whose two columns are outputs
        static_assert(last_out - first_out + 1 == OUTC / 2, "outputs are whole lanes");
        const bool is_out = lane >= first_out && lane <= last_out && c0 < Wk;
        const double eps = 1e-3 / (scale * scale);
        constexpr int kk = (M + 1) / 2;
        const int lo = max(lane - kk, 0), hi = min(lane + kk, 63);
        for (int s = 0; s < nsteps; s++) {
            const int y = r0 + (s - 1) - WIN; // the row whose window the producers completed in step s - 1
            const bool mine = (s % CONS) == who && y >= r0; // wave-uniform
            double p[5], left[5], right[5];
            if (mine) {
                const double(*sv)[128] = s_v[(s - 1) & 1];
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    p[c] = sv[c][2 * lane] + sv[c][2 * lane + 1];
                    left[c] = sv[c][2 * lo + 1];
                    right[c] = sv[c][2 * hi];
                }
            }
            lds_barrier();
            if (mine) {
                // The M pair sums of a window through sums of three: T[l] = P[l-1] + P[l] + P[l+1] replaces P in
                // LDS (a lane keeps its own P), and the window is T[l] (M = 3), T[l-1] + T[l+1] - P[l] (M = 5) or
                // T[l-2] + P[l] + T[l+2] (M = 7): four LDS accesses and four additions per channel instead of
                // eight and six.  T of lanes 0 and 63 is not a sum of three and no output lane reads it.
                // (The same exchange as whole-wave DPP shifts was measured 4 % slower: tools/variants/.)
                static_assert(M == 3 || M == 5 || M == 7, "window sums from sums of three");
                double t[5];
#pragma unroll
                for (int c = 0; c < 5; c++)
                    sp[c][lane] = p[c];
                lds_wave_sync();
                const int lm = max(lane - 1, 0), lp = min(lane + 1, 63);
#pragma unroll
                for (int c = 0; c < 5; c++)
                    t[c] = (sp[c][lm] + p[c]) + sp[c][lp];
                if (M > 3) {
                    lds_wave_sync();
#pragma unroll
                    for (int c = 0; c < 5; c++)
                        sp[c][lane] = t[c];
                    lds_wave_sync();
                }
                if (is_out) {
                    double g0[5], g1[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double common;
                        if (M == 3)
                            common = t[c];
                        else if (M == 5)
                            common = (sp[c][lane - 1] + sp[c][lane + 1]) - p[c];
                        else
                            common = (sp[c][lane - 2] + p[c]) + sp[c][lane + 2];
                        g0[c] = left[c] + common;
                        g1[c] = common + right[c];
                    }
                    float2 *o = flow_out + (size_t)pair * Nk + (size_t)y * Wk + c0;
                    {
                        // the solve with its multiply-adds fused and one Newton step on v_rcp_f64 (a float leaves here):
                        // 10 fp64 instructions less per row, +1 % frames/s; k_blur_solve_wave keeps the separate
                        // operations and the second step
#pragma clang fp contract(fast)
                        const double idet0 = fast_recip(g0[0] * g0[2] - g0[1] * g0[1] + eps);
                        const double idet1 = fast_recip(g1[0] * g1[2] - g1[1] * g1[1] + eps);
                        const float4u f = {(float)((g0[0] * g0[4] - g0[1] * g0[3]) * idet0),
                                           (float)((g0[2] * g0[3] - g0[1] * g0[4]) * idet0),
                                           (float)((g1[0] * g1[4] - g1[1] * g1[3]) * idet1),
                                           (float)((g1[2] * g1[3] - g1[1] * g1[4]) * idet1)};
                        if (c0 + 1 < Wk) // both columns in one 16-byte store (8-byte aligned where the level's width is odd)
                            *reinterpret_cast<float4u *>(o) = f;
                        else
                            o[0] = make_float2(f.x, f.y);
                    }
                }
                lds_wave_sync();
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// The pre-pass of mode 0 for the one-kernel iteration (M is never in memory there): every lane marches ONE column of one
// segment through the same rows of M the iteration will make, with the same ring, for the chain's increments alone --
// no halo columns, no window sums across columns, no barrier: the waves run free.  Segment 0 delivers the chain's value
// after its last row (primed as OpenCV primes it), the others the sum of their rows' increments from zero; k_carry_scan
// turns that into each segment's carry.  S: [segment][pair][channel][Wk].
// ---------------------------------------------------------------------------------
// STORE (option fb_exact_sums on a large launch): one segment = the whole column, and what leaves is not the segment's
// last value but the chain's value at EVERY row, V[pair][channel][y][x] -- k_exact_vsum's output without M ever being in
// memory (k_exact_hsolve takes it from there).
template <int M, int FLOW, bool STORE = false>
__global__ void __launch_bounds__(128)
k_flow_carry_pc(const float *__restrict__ R, const float2 *__restrict__ flow_in, int Wk, int Hk, int seg, FlowInit fi,
                const int *__restrict__ rows_ofs, const float *__restrict__ rows_frac, double *__restrict__ S)
{
    constexpr int WIN = 2 * M + 1;
    __shared__ float ring[WIN][5][128];
    const int col = threadIdx.x, xr = blockIdx.x * 128 + col, x = min(xr, Wk - 1);
    const int by = blockIdx.y, pair = blockIdx.z;
    const size_t Nk = (size_t)Wk * Hk;
    const int r0 = by * seg, r1 = min(r0 + seg, Hk);
    const int e0 = r0 - M - 1, n_rows = (r1 - r0) + WIN;
    RowProducer<FLOW> P;
    P.init(R, flow_in, fi, rows_ofs, rows_frac, pair, Nk, Wk, Hk, x);
    P.start(e0);
    double vs[5] = {0, 0, 0, 0, 0};
    int slot = 0;
    for (int s = 0; s < WIN; s++) {
        float m[5];
        P.template next<false>(e0 + s, m);
#pragma unroll
        for (int c = 0; c < 5; c++)
            ring[slot][c][col] = m[c];
        if (r0 == 0) {
            if (s == 0) {
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] = (double)(m[c] * (float)(M + 2));
            } else if (s >= M + 2 && s <= 2 * M) {
#pragma unroll
                for (int c = 0; c < 5; c++)
                    vs[c] += (double)m[c];
            }
        }
        slot = slot + 1 == WIN ? 0 : slot + 1;
    }
    double *V = S + (size_t)pair * 5 * Nk + (size_t)r0 * Wk + x; // (STORE)
    for (int s = WIN; s < n_rows; s++) {
        float m[5];
        P.template next<false>(e0 + s, m);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const float old = ring[slot][c][col];
            ring[slot][c][col] = m[c];
            vs[c] += (double)(m[c] - old);
            if (STORE && xr < Wk)
                V[(size_t)c * Nk] = vs[c];
        }
        V += Wk;
        slot = slot + 1 == WIN ? 0 : slot + 1;
    }
    if (!STORE && xr < Wk) {
        double *o = S + (((size_t)by * gridDim.z + pair) * 5) * Wk + xr;
#pragma unroll
        for (int c = 0; c < 5; c++)
            o[(size_t)c * Wk] = vs[c];
    }
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
struct AreaTabs {
    const int *xsi, *xstart; // x entries: source column; first entry of every destination column (Wc + 1)
    const float *xalpha;
    const int *ysi, *ystart;
    const float *yalpha;
    int ix, iy;              // > 0: the integer-factor path
};
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
// ---------------------------------------------------------------------------------
// B1: FlowSource.post_process (source.py:337-363)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float2 clip_to_frame(float2 f, int i, int j, int W, int H)
{
    f.x = clip_nan(f.x, (float)(-j), (float)(W - 1 - j));
    f.y = clip_nan(f.y, (float)(-i), (float)(H - 1 - i));
    return f;
}

__global__ void k_pp_clip(float2 *flow, int W, int H, FastDiv dw)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * H)
        return;
    const int i = (int)fast_div((uint32_t)t, dw);
    flow[t] = clip_to_frame(flow[t], i, t - i * W, W, H);
}

// source.py:350-358: every moving source p claims target p+d; numpy.put writes in
// ascending p, so the largest p wins -> atomicMax on the source index.
__global__ void k_pp_fwd_scatter(const float2 *__restrict__ flow, int *__restrict__ winner, int W, int H, FastDiv dw)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = W * H;
    if (t >= N)
        return;
    const int i = (int)fast_div((uint32_t)t, dw);
    float2 f = clip_to_frame(flow[t], i, t - i * W, W, H);
    int ix = (int)rintf(f.x), iy = (int)rintf(f.y);
    int d = iy * W + ix;
    if (d == 0)
        return;
    int target = clampi(t + d, 0, N - 1); // mode="clip"
    atomicMax(&winner[target], t);
}

__global__ void k_pp_fwd_resolve(float2 *__restrict__ flow, const int *__restrict__ winner, int W, int H, FastDiv dw)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * H)
        return;
    int w = winner[t];
    int src = w >= 0 ? w : t;
    const int i = (int)fast_div((uint32_t)t, dw), j = t - i * W;
    const int si = (int)fast_div((uint32_t)src, dw);
    float2 f = make_float2((float)(src - si * W - j), (float)(si - i)); // source.py:359-360
    flow[t] = clip_to_frame(f, i, j, W, H);                            // :361-362
}

// The optional pre-steps of post_process: filters.py:36-72 and the mask multiply of
// source.py:342-343, per pixel, in numpy's arithmetic (float32 for weak scalars, float64 for
// numpy.float64 values; numpy.linalg.norm of a float32 pair is sqrt(x*x + y*y) in float32).
struct FlowOps {
    int n;
    tf_flow_op op[TF_MAX_FLOW_OPS];
};

__global__ void k_pp_ops(float2 *__restrict__ flow, const float *__restrict__ mask, int N, FlowOps ops)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N)
        return;
    float2 f = flow[t];
    for (int i = 0; i < ops.n; i++) {
        const int kind = ops.op[i].kind, wide = ops.op[i].wide;
        const double v = ops.op[i].value;
        if (kind == TF_FLOW_SCALE) {
            if (wide) {
                f.x = (float)((double)f.x * v);
                f.y = (float)((double)f.y * v);
            } else {
                f.x = f.x * (float)v;
                f.y = f.y * (float)v;
            }
        } else {
            const float norm = sqrtf(f.x * f.x + f.y * f.y);
            if (kind == TF_FLOW_THRESHOLD) {
                const bool hit = wide ? ((double)norm <= v) : (norm <= (float)v);
                if (hit)
                    f = make_float2(0.f, 0.f);
            } else { // clip: factors stay 1.0 (float64) where the norm is below the threshold
                const bool hit = wide ? ((double)norm >= v) : (norm >= (float)v);
                if (hit) {
                    const double factor = wide ? v / (double)norm : (double)((float)v / norm);
                    f.x = (float)((double)f.x * factor);
                    f.y = (float)((double)f.y * factor);
                }
            }
        }
    }
    if (mask) {
        const float m = mask[t];
        f.x = m * f.x;
        f.y = m * f.y;
    }
    flow[t] = f;
}

// host-side layout converters for the stage entry points
__global__ void k_interleaved_to_planar5(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    for (int c = 0; c < 5; c++)
        dst[c * n + t] = src[t * 5 + c];
}

// host [n][5] interleaved <-> the channel-pair layout of R
__global__ void k_interleaved_to_rpairs(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        r_store_px(dst, n, i, src + i * 5);
}
__global__ void k_rpairs_to_interleaved(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v[5];
        r_load_px(src, n, i, v);
#pragma unroll
        for (int c = 0; c < 5; c++)
            dst[i * 5 + c] = v[c];
    }
}

__global__ void k_planar5_to_interleaved(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    for (int c = 0; c < 5; c++)
        dst[t * 5 + c] = src[c * n + t];
}

// ---- host-side constant preparation ----------------------------------------------
inline int cv_round(double v) { return (int)lrint(v); }

std::vector<float> gaussian_kernel(int n, double sigma)
{
    std::vector<float> k((size_t)n);
    if (sigma <= 0 && (n == 1 || n == 3 || n == 5 || n == 7)) {
        static const float t1[] = {1.f};
        static const float t3[] = {0.25f, 0.5f, 0.25f};
        static const float t5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
        static const float t7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
        const float *t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : t7;
        for (int i = 0; i < n; i++)
            k[i] = t[i];
        return k;
    }
    double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2x = -0.5 / (sx * sx);
    std::vector<double> v((size_t)n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        v[i] = std::exp(scale2x * x * x);
        sum += v[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++)
        k[i] = (float)(v[i] * sum);
    return k;
}

// Symmetric 6x6 solve for the four entries of G^-1 the expansion needs.
void invert6(const double G[36], double inv[36])
{
    double L[36] = {0};
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = G[i * 6 + j];
            for (int k = 0; k < j; k++)
                s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = (i == j) ? std::sqrt(s) : s / L[j * 6 + j];
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

PolyConst make_poly_const(int n, double sigma)
{
    PolyConst pc;
    memset(&pc, 0, sizeof(pc));
    pc.n = n;
    if (sigma < FLT_EPSILON)
        sigma = n * 0.3;
    std::vector<float> gb(2 * n + 1), xgb(2 * n + 1), xxgb(2 * n + 1);
    float *g = gb.data() + n, *xg = xgb.data() + n, *xxg = xxgb.data() + n;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[36] = {0};
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            float gg = g[y] * g[x];
            G[0] += gg;
            G[7] += gg * x * x;
            G[21] += gg * x * x * x * x;
            G[35] += gg * x * x * y * y;
        }
    G[14] = G[3] = G[4] = G[18] = G[24] = G[7];
    G[28] = G[21];
    G[22] = G[27] = G[35];
    double inv[36];
    invert6(G, inv);
    pc.ig11 = inv[7];
    pc.ig03 = inv[3];
    pc.ig33 = inv[21];
    pc.ig55 = inv[35];
    for (int k = 0; k <= n; k++) {
        pc.g[k] = g[k];
        pc.xg[k] = xg[k];
        pc.xxg[k] = xxg[k];
    }
    return pc;
}

// resize.cpp's INTER_LINEAR coefficient tables for one axis
void make_lerp(int src, int dst, bool zero_at_edges, std::vector<int> &ofs, std::vector<float> &frac)
{
    ofs.resize((size_t)dst);
    frac.resize((size_t)dst);
    double inv_scale = (double)dst / src;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
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
        ofs[d] = s;
        frac[d] = f;
    }
}

struct LerpDev {
    DevBuf xofs, xfrac, yofs, yfrac;
    int upload_tabs(int sw, int sh, int dw, int dh)
    {
        std::vector<int> o;
        std::vector<float> f;
        make_lerp(sw, dw, true, o, f);
        TF_TRY(xofs.alloc(o.size() * 4));
        TF_TRY(xfrac.alloc(f.size() * 4));
        TF_HIP(hipMemcpy(xofs.p, o.data(), o.size() * 4, hipMemcpyHostToDevice));
        TF_HIP(hipMemcpy(xfrac.p, f.data(), f.size() * 4, hipMemcpyHostToDevice));
        make_lerp(sh, dh, false, o, f);
        TF_TRY(yofs.alloc(o.size() * 4));
        TF_TRY(yfrac.alloc(f.size() * 4));
        TF_HIP(hipMemcpy(yofs.p, o.data(), o.size() * 4, hipMemcpyHostToDevice));
        TF_HIP(hipMemcpy(yfrac.p, f.data(), f.size() * 4, hipMemcpyHostToDevice));
        return TF_OK;
    }
};

struct Level {
    int W, H, ksz;
    double sigma;
    ImgTile tile;
    std::vector<float> kern_host;
    DevBuf kern;
    DevBuf img, R;     // this level's image / polynomial coefficients (levels >= 1; level 0 uses the handle's)
    LerpDev img_lerp;  // frame -> this level (unused when sizes are equal)
    // long blur kernels: row pass over whole frame rows, then column pass + lerps (k_level_rowpass / _colpass)
    bool split = false;
    DevBuf colsrc;          // source column of each of the NC = 2*W row-pass columns
    int NC = 0, rp_rshift = 0;
    size_t rowf_off = 0;    // this level's plane inside tf_fb::rowf (floats)
    LerpDev flow_lerp; // level k+1 -> this level
};

} // namespace

struct tf_fb {
    int W = 0, H = 0;
    tf_fb_params prm;
    int K = 0; // scales K..0
    int slots = 0, max_pairs = 0;
    PolyConst pc;
    std::vector<Level *> lv;
    DevBuf frames, img, R, M, lflow[5], pairs, winner, scratch; // lflow[3..4]: the result of even / odd calls
    DevBuf rowf; // row-pass planes of the split levels, [level][image][H][NC]
    int rp_RB = 0, rp_pitch = 0, rp_r4 = 0, rp_rmax = 0, rp_first = -1; // one k_level_rowpass launch serves them all
    int nsets = 1, cur = 0;                    // result buffers in rotation / the one this call writes
    hipStream_t chain_stream = nullptr;        // everything a call launches; the library stream only waits for its end, so
                                               // what the caller queues after a call (its remap) runs beside the NEXT call
    hipEvent_t entry[2] = {nullptr, nullptr};  // position of the library stream when call (parity) was issued
    bool entry_pending[2] = {false, false};
    hipEvent_t chain_done = nullptr;           // end of the latest call's work on chain_stream
    int2 *pairs_host = nullptr;                // pinned staging: the call's image list (slots, up to 4P ints), then its pair -> image map (P int2)
    const int2 *rmap_dev = nullptr;            // that map on the device while a batch is being issued; null: images 2p, 2p+1
    std::vector<int> image_of;                 // slot -> index in the image list of the call being issued
    hipEvent_t pairs_copied = nullptr;
    bool pairs_pending = false;
    int last_pairs = 0;
    int final_buf = 0; // which lflow buffer holds the level-0 result
    // fb_flags: OPTFLOW_USE_INITIAL_FLOW (4) and OPTFLOW_FARNEBACK_GAUSSIAN (256)
    DevBuf init_flow;            // [P][H][W] float2: the caller's initial flow of every pair (flag 4)
    DevBuf area_i, area_f;       // resize(INTER_AREA) tables to the coarsest scale: ints, then weights
    AreaTabs area{};
    DevBuf gauss_taps;           // winsize / 2 + 1 taps of the Gaussian window (flag 256)
    DevBuf bgr_stage;            // tf_fb_set_frame_bgr: the decoded frame on its way to a slot
    DevBuf exact_vsum;           // option fb_exact_sums: OpenCV's column sums of the level being solved, [pair][5][y][x] doubles
    // OpenCV's column sums across row segments (ColumnCarry)
    DevBuf col_carry;            // the chain's value in front of every segment of the launch being issued
    DevBuf chain_words;          // words 0-7: the ticket counters of a launch; from word 16 on: the hand-off flags
    unsigned chain_epoch = 0;
    unsigned *chain_fault = nullptr; // pinned, device-visible: a wait for a carry gave up (k_flow_iter_pc)
    // tf_fb_async_io: uploads of frames and downloads of results on copy streams of their own, so that a streaming
    // caller's next frame goes up and its previous flow comes down while the current pair is being computed
    bool async_io = false;
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    hipEvent_t call_done[2] = {nullptr, nullptr};      // end of a call's kernels, by the call's number mod 2
    hipEvent_t result_ready[2] = {nullptr, nullptr};   // the library stream's position when a download of result set s was asked for
    hipEvent_t download_done[2] = {nullptr, nullptr};  // ... and its end
    bool download_pending[2] = {false, false};
    std::vector<long> slot_read_call;                  // the last call that read each slot's frame bytes (expanded it)
    long n_calls = 0;
    int exact = -1;              // tf_fb_set_exact: 1 / 0 = this handle sums the box window in OpenCV's own order or not; -1 = as option "fb_exact_sums" says at each call
    tf_fb *lane_of = nullptr;    // tf_fb_create_lane: the handle whose frame slots these are
    int lanes = 0;               // ... and how many lanes read this handle's
    bool destroy_with_lanes = false; // tf_fb_destroy came while lanes were alive: the last lane's destroy releases it
    bool use_initial() const { return (prm.flags & 4) != 0; }
    bool gaussian() const { return (prm.flags & 256) != 0; }
    // A3+A4 of one iteration as ONE kernel (k_flow_iter_pc: M never stored) on levels big enough to fill
    // the chip with its 3-wave workgroups, as two kernels (k_update_matrices, k_blur_solve_wave)
    // otherwise.  Option "fb_fused" = 0 / 1 forces never / always (read at tf_fb_create).
    int fused = (int)option(OPT_FB_FUSED);
    float *Rk(int k) { return (k <= 0 ? R : lv[k]->R).as<float>(); }
    // where the expansion launches being issued write and which part of the image list they read
    // (tf_fb_calc_slots; zero outside it)
    int prep_image0 = 0, prep_list0 = 0;
    float *Rk_out(int k) { return Rk(k) + (size_t)prep_image0 * 5 * (k <= 0 ? (size_t)W * H : (size_t)lv[k]->W * lv[k]->H); }
    const int2 *image_list() { return pairs.as<int2>() + prep_list0; }
    // tf_fb_keep_expansions: R is indexed by frame slot and an expansion stays valid until its slot is written
    bool keep = false;
    std::vector<char> expanded, external;
    float *imgk(int k) { return (k <= 0 ? img : lv[k]->img).as<float>(); }
    ~tf_fb()
    {
        for (auto *l : lv)
            delete l;
        if (chain_done)
            (void)hipEventDestroy(chain_done);
        if (pairs_copied)
            (void)hipEventDestroy(pairs_copied);
        if (pairs_host)
            (void)hipHostFree(pairs_host);
        if (chain_fault)
            (void)hipHostFree(chain_fault);
        for (int i = 0; i < 2; i++)
            for (hipEvent_t e : {call_done[i], result_ready[i], download_done[i]})
                if (e)
                    (void)hipEventDestroy(e);
        for (auto e : entry)
            if (e)
                (void)hipEventDestroy(e);
        // chain_stream is the library's side stream (runtime.hip), not ours to destroy
    }
};

// The exact mode is a property of the HANDLE (tf_fb_set_exact); a handle that was never told follows the process-wide
// option, read at each call.
static bool fb_exact(const tf_fb *fb) { return fb->exact >= 0 ? fb->exact != 0 : option(OPT_FB_EXACT_SUMS) != 0; }

// read when a handle is created: option "fb_no_overlap" = 1 keeps everything on the library stream
static bool fb_overlap_enabled() { return option(OPT_FB_NO_OVERLAP) == 0; }

// Profiler labels: with option "prof_levels" = 1 every Farneback launch is
// labelled with its pyramid level ("fb_polyexp.k2"), otherwise by kernel only.
static const char *lvl_name(const char *base, int k)
{
    const bool per_level = option(OPT_PROF_LEVELS) != 0;
    if (!per_level || k < 0)
        return base;
    static std::map<std::string, std::string> names; // (entries are never erased: the strings stay where they are)
    static std::mutex names_mu;                       // handles of several threads label their launches
    std::lock_guard<std::mutex> lk(names_mu);
    std::string key = std::string(base) + ".k" + std::to_string(k);
    auto it = names.find(key);
    if (it == names.end())
        it = names.emplace(key, key).first;
    return it->second.c_str();
}

// `standalone`: a single level is wanted (stage entry points): run the shared row pass regardless of the order
static int fb_level_image(tf_fb *fb, int k, int n_images, bool standalone = false)
{
    Level &L = *fb->lv[k];
    if (L.split) {
        if (k == fb->rp_first || standalone) { // the coarsest split level comes first in the preparation: row pass of all of them now
            RowPassArgs a;
            memset(&a, 0, sizeof(a));
            size_t taps = 0;
            for (int j = fb->K; j >= 1; j--) {
                Level &S = *fb->lv[j];
                if (!S.split)
                    continue;
                RowPassLevel &rl = a.lv[a.n++];
                rl.rowf = fb->rowf.as<float>() + S.rowf_off;
                rl.colsrc = S.colsrc.as<int>();
                rl.kern = S.kern.as<float>();
                rl.NC = S.NC;
                rl.ksz = S.ksz;
                rl.rshift = S.rp_rshift;
                taps += (size_t)S.ksz;
            }
            const size_t smem_rp = (size_t)fb->rp_RB * fb->rp_pitch + taps * sizeof(float);
            TF_TRY(launch(lvl_name("fb_level_rowpass", -1), k_level_rowpass, dim3(cdiv(fb->H, fb->rp_RB), n_images),
                          dim3(256), smem_rp, (const uint8_t *)fb->frames.as<uint8_t>(),
                          fb->image_list(), fb->W, fb->H, a, fb->rp_RB, fb->rp_pitch, fb->rp_r4,
                          fb->rp_rmax));
        }
        return launch(lvl_name("fb_level_colpass", k), k_level_colpass, dim3(cdiv(L.W, 64), cdiv(L.H, 4), n_images), dim3(256),
                      (size_t)L.ksz * sizeof(float), (const float *)(fb->rowf.as<float>() + L.rowf_off), fb->imgk(k), fb->W, fb->H,
                      L.W, L.H, L.NC, (const float *)L.kern.as<float>(), L.ksz, (const int *)L.img_lerp.xofs.as<int>(),
                      (const float *)L.img_lerp.xfrac.as<float>(), (const int *)L.img_lerp.yofs.as<int>(),
                      (const float *)L.img_lerp.yfrac.as<float>());
    }
    const ImgTile &t = L.tile;
    dim3 grid(cdiv(L.W, t.TWo), cdiv(L.H, t.THo), n_images);
    size_t smem = (size_t)t.LH * t.pitch + (size_t)t.LH * t.rstride * sizeof(float) + (size_t)L.ksz * sizeof(float);
    return launch(lvl_name("fb_level_image", k), k_level_image, grid, dim3(256), smem,
                  (const uint8_t *)fb->frames.as<uint8_t>(), fb->image_list(), fb->imgk(k),
                  fb->W, fb->H, L.W, L.H, (const float *)L.kern.as<float>(), L.ksz, t);
}

// Plans the two-kernel form of A1 for a level with a long blur kernel (returns false where it does not
// apply: short kernels, frame widths that are not a multiple of 4, frames too wide to stage 8 rows).
static bool plan_split_level(int W, int H, Level &L, std::vector<int> &colsrc)
{
    static const bool off = tune("TF_IMG_NO_SPLIT", 0) != 0;
    static const int min_ksz = (int)tune("TF_IMG_SPLIT_MIN_KSZ", 9);
    if (off || L.ksz < min_ksz || L.ksz <= 5 || (W & 3) != 0 || (L.W == W && L.H == H))
        return false;
    std::vector<int> xo, yo;
    std::vector<float> fr;
    make_lerp(W, L.W, true, xo, fr);
    make_lerp(H, L.H, false, yo, fr);
    L.NC = 2 * L.W;
    colsrc.resize((size_t)L.NC);
    for (int x = 0; x < L.W; x++) {
        colsrc[2 * x] = xo[x];
        colsrc[2 * x + 1] = std::min(xo[x] + 1, W - 1);
    }
    // lanes = R rows x 64/R groups; one group = two level columns = s/2 dwords of a frame row
    const int s_ = std::max(1, W / std::max(1, L.W));
    int rshift = 1;
    while ((1 << rshift) < std::min(8, std::max(2, s_ / 2)))
        rshift++;
    L.rp_rshift = rshift;
    return true;
}

// Output tile of a level: as large as fits ~60 KB of LDS, given the source extent a tile needs.
static ImgTile choose_tile(int W, int H, int Wk, int Hk, int ksz, int level)
{
    std::vector<int> xo, yo;
    std::vector<float> fr;
    make_lerp(W, Wk, true, xo, fr);
    make_lerp(H, Hk, false, yo, fr);
    const int r = ksz / 2;
    auto extent = [&](const std::vector<int> &ofs, int n, int len, int tile, bool align4) {
        int worst = 0;
        for (int d0 = 0; d0 < n; d0 += tile) {
            int d1 = std::min(n, d0 + tile) - 1;
            int lo = std::max(0, std::min(ofs[d0], len - 1)) - r, hi = std::max(0, std::min(ofs[d1] + 1, len - 1)) + r;
            if (align4)
                lo &= ~3;
            worst = std::max(worst, hi - lo + 1);
        }
        return worst;
    };
    int s = std::max(1, (W + Wk - 1) / Wk);
    static const size_t lds_cap = (size_t)tune("TF_IMG_LDS_KB", 60) * 1024;
    ImgTile t;
    t.same_size = (W == Wk && H == Hk);
    t.scale_x = 1. / ((double)Wk / W);
    t.scale_y = 1. / ((double)Hk / H);
    auto fill = [&](int two, int tho) {
        t.TWo = two;
        t.THo = tho;
        t.LW = extent(xo, Wk, W, two, true) + 3; // dword copies may run up to 3 bytes past the last column
        t.LH = extent(yo, Hk, H, tho, false);
        t.pitch = (t.LW + 3) & ~3;
        if (((t.pitch / 4) & 1) == 0)
            t.pitch += 4;
        t.rstride = t.same_size ? two : 2 * two;
        t.tw_shift = 0;
        while ((1 << t.tw_shift) < two)
            t.tw_shift++;
        return (size_t)t.LH * t.pitch + (size_t)t.LH * t.rstride * sizeof(float) + (size_t)ksz * sizeof(float);
    };
    if (const char *ov = tune_str("TF_IMG_TILES")) { // "level:TWo:THo,..." experiment override
        for (const char *p = ov; p && *p;) {
            int l = 0, a = 0, b = 0;
            if (sscanf(p, "%d:%d:%d", &l, &a, &b) == 3 && l == level && fill(a, b) <= 64 * 1024)
                return t;
            p = strchr(p, ',');
            if (p)
                p++;
        }
    }
    if (ksz > 5) {
        // long kernels (measured on MI355X, tools/tile_sweep.sh): tiles spanning ~128 source columns,
        // as many output rows as fill whole rounds of 64 staged rows (the row pass costs
        // ceil(LH/64) lane-rounds per column group) within ~40 KB of LDS so several blocks share a CU
        const size_t cap = std::min<size_t>(lds_cap, 40 * 1024);
        int btw = 4;
        while (btw * 2 <= std::max(4, 128 / s))
            btw *= 2;
        int bth = 1;
        double best = 1e30;
        for (int tho = 1; tho <= 32; tho++) {
            size_t smem = fill(btw, tho);
            if (smem > cap && tho > 1)
                break;
            double rounds = (double)((t.LH + 63) / 64) * 64 / tho; // lane-rows per output row
            if (rounds <= best) {
                best = rounds;
                bth = tho;
            }
        }
        fill(btw, bth);
        return t;
    }
    int two = 8;
    while (two * 2 <= std::min(128, 256 / s))
        two *= 2;
    int tho = std::max(2, std::min(32, 128 / s));
    for (;;) {
        size_t smem = fill(two, tho);
        if (smem <= std::min<size_t>(lds_cap, 32 * 1024) || (two <= 2 && tho <= 1))
            break;
        if (tho > 1 && (tho >= two / 4 || two <= 2))
            tho = std::max(1, tho / 2);
        else
            two = std::max(2, two / 2);
    }
    return t;
}

static int fb_polyexp(tf_fb *fb, int w, int h, int n_images, int k = -1)
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
static bool fb_can_fuse_level(tf_fb *fb, int k)
{
    static const bool off = tune("TF_FB_NO_A1A2", 0) != 0;
    const Level &L = *fb->lv[k];
    return !off && L.W == fb->W && L.H == fb->H && L.ksz == 3 && (fb->pc.n == 5 || fb->pc.n == 7);
}

// ... and to a level that is exactly half the frame (k_level1_polyexp_t)
static bool fb_can_fuse_half_level(tf_fb *fb, int k)
{
    static const bool off = tune("TF_FB_NO_A1A2", 0) != 0;
    const Level &L = *fb->lv[k];
    return !off && 2 * L.W == fb->W && 2 * L.H == fb->H && L.ksz == 3 && (fb->pc.n == 5 || fb->pc.n == 7);
}

static int fb_level1_polyexp(tf_fb *fb, int k, int n_images)
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

static int fb_level0_polyexp(tf_fb *fb, int k, int n_images)
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

static int fb_update_matrices(tf_fb *fb, int w, int h, int n_pairs, const FlowInit &fi, int k = -1)
{
    dim3 grid(cdiv(w, UM_TW), cdiv(h, UM_TH), n_pairs);
    FlowInit f = fi;
    f.rmap = fb->rmap_dev;
    return launch(lvl_name("fb_update_matrices", k), k_update_matrices, grid, dim3(256), 0, (const float *)fb->Rk(k),
                  fb->M.as<float>(), w, h, f);
}

// ---------------------------------------------------------------------------------
// How a marching launch is cut into row segments, and where its segments get the column sums' carries from
// (ColumnCarry).  `columns` = strips x pairs workgroups stand side by side; a launch of `segs` segments has segs x columns
// workgroups of (h / segs + warm-up) steps each, `slots` of them resident at a time.
//   hand-off inside the launch (mode 1): the segments of a column run one after the other.  With at least as many
//     columns as slots that costs nothing -- by the time a segment is dispatched the one above it is done -- and the
//     segment count is the one that minimises rounds x steps.  With fewer columns the segments would only queue up
//     behind each other, so the column is marched whole (one segment, no hand-off) on a part-empty chip.
//   pre-pass (mode 0): segments side by side as before, for a second launch that makes the rows of M for their carries
//     (prepass_cost x the march's time; k_blur_carry, which reads M, is cheap) and a third that adds them up: two
//     launches of a fixed cost each (prepass_steps, in steps of the march) that a short column does not repay.
// Times are in units of one workgroup step at full residency; a step is faster on a part-empty chip (step_time).
// ---------------------------------------------------------------------------------
struct March {
    int mode, segs, seg; // seg: rows per segment
};
static double step_time(double wgs_per_cu, int slots_per_cu)
{
    // measured on MI355X for k_flow_iter_pc (3 slots per CU): a step takes 0.69 / 0.83 / 0.88 us with 1.1 / 2.25 / 3
    // workgroups per CU on average (a wave's ~130 instructions per step at one issue every ~8 cycles, not memory latency,
    // set the pace, so company costs little)
    const double full = slots_per_cu, o = std::min(std::max(wgs_per_cu, 1.0), full);
    static const double alone = tune("TF_STEP_ALONE_PCT", 78) / 100.0;
    return full <= 1 ? 1.0 : alone + (1.0 - alone) * (o - 1.0) / (full - 1.0);
}
static March choose_march(long columns, int h, int warm, long slots, int slots_per_cu, double prepass_cost, double prepass_steps, int min_rows,
                          bool has_company = false)
{
    const long forced_segs = option(OPT_FB_SEGS), forced_mode = option(OPT_FB_CHAIN);
    const long cus = std::max(1l, slots / slots_per_cu);
    auto rounds_cost = [&](long sg, double *cost) {
        const long rows = (h + sg - 1) / sg;
        const long wgs = columns * sg, rounds = (wgs + slots - 1) / slots;
        *cost = (double)rounds * (double)(rows + warm + 1) * step_time((double)std::min(wgs, slots) / cus, slots_per_cu);
        return rows;
    };
    // the best segment count for segments that run side by side
    long best_segs = 1;
    double best_cost = 1e300;
    for (long sg = 1; sg <= 64 && sg <= h; sg++) {
        double cost;
        const long rows = rounds_cost(sg, &cost);
        if (rows < min_rows && sg > 1)
            break;
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best_segs = sg;
        }
    }
    if (forced_segs > 0) {
        best_segs = std::min<long>(forced_segs, h);
        rounds_cost(best_segs, &best_cost);
    }
    // A handle with a lane (tf_fb_create_lane) has the other lane's batch for company: what a whole-column march leaves idle
    // is not lost, while a pre-pass's second making of M is work the chip does not get back -- the pre-pass must win by more
    // (measured at 4K x 8, two lanes: 1444 frames/s with whole columns, 1124 with the pre-pass the lone-launch model picks)
    static const double company = tune("TF_PC_COMPANY_PCT", 70) / 100.0;
    March m;
    double whole;
    rounds_cost(1, &whole);
    if (best_segs == 1) {
        m.mode = 0;
        m.segs = 1;
    } else if (forced_mode == 1 || (forced_mode < 0 && columns >= slots)) {
        m.mode = 1;
        m.segs = (int)best_segs;
    } else if (forced_mode == 0 || forced_segs > 0 || best_cost * (1.0 + prepass_cost) + prepass_steps < whole * (has_company ? company : 1.0)) {
        m.mode = 0;
        m.segs = (int)best_segs;
    } else {
        m.mode = 0;
        m.segs = 1;
    }
    m.seg = (h + m.segs - 1) / m.segs;
    m.segs = (h + m.seg - 1) / m.seg;
    return m;
}

// room for the carries of a launch (and, for hand-offs inside it, its flags); the fault word
static int fb_carry_room(tf_fb *fb, size_t carry_doubles, size_t flags)
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
static int fb_check_fault(tf_fb *fb, const char *where)
{
    if (fb->chain_fault && *(volatile unsigned *)fb->chain_fault) {
        *fb->chain_fault = 0;
        return set_error(TF_ERR_HIP, "%s: a segment of k_flow_iter_pc waited more than 5 s for the column sums of the segment "
                                     "above it; the flow of that call is invalid", where);
    }
    return TF_OK;
}

static int fb_carry_scan(tf_fb *fb, int w, int n_pairs, int segs, int k)
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

// The row walker of option fb_exact_sums over fb->exact_vsum (k_exact_hsolve).
static int fb_exact_hsolve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k)
{
    const int m = fb->prm.winsize / 2;
    const double scale = 1. / ((double)fb->prm.winsize * fb->prm.winsize);
    // three rows per wave (6: the same at 32 pairs of 4K, 202 against 165 us for one pair; 12: 3.7 x slower -- their
    // differences in flight take every register a lane has)
    return launch(lvl_name("fb_exact_hsolve", k), k_exact_hsolve<3>, dim3(cdiv(h, 3), 1, n_pairs), dim3(64), 0,
                  (const double *)fb->exact_vsum.as<double>(), flow_out, w, h, m, scale);
}

static int fb_blur_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k = -1)
{
    const int m = fb->prm.winsize / 2;
    const double scale = 1. / ((double)fb->prm.winsize * fb->prm.winsize);
    if (fb_exact(fb)) { // OpenCV's own running sums, in its order (k_exact_vsum's note)
        const size_t need = (size_t)n_pairs * 5 * w * h * sizeof(double);
        if (fb->exact_vsum.bytes < need && fb->exact_vsum.alloc(need) != TF_OK)
            return set_error(TF_ERR_HIP, "fb_exact_sums: no room for the column sums of %d pairs of %d x %d pixels (%zu bytes of doubles; "
                                         "fewer pairs per call need less)", n_pairs, w, h, need);
        TF_TRY(launch(lvl_name("fb_exact_vsum", k), k_exact_vsum, dim3(cdiv(w, 64), 5, n_pairs), dim3(64), 0,
                      (const float *)fb->M.as<float>(), fb->exact_vsum.as<double>(), w, h, m));
        return fb_exact_hsolve(fb, w, h, n_pairs, flow_out, k);
    }
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
    if (m == 0) {
        const size_t need = (size_t)n_pairs * 5 * w * h * sizeof(double);
        if (fb->exact_vsum.bytes < need)
            TF_TRY(fb->exact_vsum.alloc(need));
        TF_TRY(launch("fb_w1_vsum", k_w1_vsum, dim3(cdiv(w, 64), 5, n_pairs), dim3(64), 0, (const float *)fb->M.as<float>(),
                      fb->exact_vsum.as<double>(), w, h));
        return launch("fb_w1_solve", k_w1_solve, dim3(cdiv(w, 256), h, n_pairs), dim3(256), 0,
                      (const double *)fb->exact_vsum.as<double>(), flow_out, w, h);
    }
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

// `up`: the first iteration of a level below the coarsest takes its flow from the coarser level (A5 fused)
template <int M>
static int launch_flow_iter(tf_fb *fb, int w, int h, int n_pairs, const float2 *flow_in, float2 *flow_out, int k,
                            const FlowInit *up)
{
    const double scale = 1. / ((double)fb->prm.winsize * fb->prm.winsize);
    const float *R = fb->Rk(k);
    constexpr int OUTC = 128 - 2 * ((M + 1) & ~1), WIN = 2 * M + 1;
    const unsigned strips = cdiv(w, OUTC);
    FlowInit f;
    memset(&f, 0, sizeof(f));
    if (up)
        f = *up;
    f.rmap = fb->rmap_dev;
    if (fb_exact(fb)) {
        // fb_exact_sums on a large launch: the column sums of EVERY row straight from R0, R1 and the flow (the pre-pass
        // kernel with one segment, storing as it goes: M is never in memory), then the row walker.  4K x 32, level 0: 4.3 +
        // 2.7 ms against 3.3 + 3.1 + 2.8 through update-matrices (and 9.0 for an exact form of k_flow_iter_pc whose
        // consumers handed the rows' sums from strip to strip, tried in round 4: profiles/NOTES.md)
        const size_t need = (size_t)n_pairs * 5 * w * h * sizeof(double);
        if (fb->exact_vsum.bytes < need && fb->exact_vsum.alloc(need) != TF_OK)
            return set_error(TF_ERR_HIP, "fb_exact_sums: no room for the column sums of %d pairs of %d x %d pixels (%zu bytes of doubles; "
                                         "fewer pairs per call need less)", n_pairs, w, h, need);
        const dim3 pgrid(cdiv(w, 128), 1, n_pairs);
        double *V = fb->exact_vsum.as<double>();
        const char *name = lvl_name("fb_flow_vsum", k);
        if (up)
            TF_TRY(launch(name, k_flow_carry_pc<M, 2, true>, pgrid, dim3(128), 0, R, flow_in, w, h, h, f, f.yofs, f.yfrac, V));
        else if (flow_in)
            TF_TRY(launch(name, k_flow_carry_pc<M, 1, true>, pgrid, dim3(128), 0, R, flow_in, w, h, h, f, f.yofs, f.yfrac, V));
        else
            TF_TRY(launch(name, k_flow_carry_pc<M, 0, true>, pgrid, dim3(128), 0, R, flow_in, w, h, h, f, f.yofs, f.yfrac, V));
        return fb_exact_hsolve(fb, w, h, n_pairs, flow_out, k);
    }
    // 3 workgroups per CU are resident (768 on the chip) and all take the same time: the launch runs in
    // rounds of 768, each as long as a segment plus its 2M+1 warm-up steps and the drain step (choose_march)
    static const long slots = tune("TF_PC_SLOTS", 768);
    static const double prepass_cost = tune("TF_PC_PREPASS_PCT", 80) / 100.0; // k_flow_carry_pc: 0.76 of the march it serves (4K x 32, level 2)
    March mc = choose_march((long)strips * n_pairs, h, WIN, slots, 3, prepass_cost, 14, 2 * WIN, fb->lane_of || fb->lanes); // (steps of ~0.9 us)
    ColumnCarry cc;
    memset(&cc, 0, sizeof(cc));
    cc.mode = mc.mode;
    cc.segs = mc.segs;
    cc.pairs = n_pairs;
    cc.strips = (int)strips;
    dim3 grid(strips, mc.segs, n_pairs);
    if (mc.segs > 1 && mc.mode == 1) {
        const size_t items = (size_t)mc.segs * n_pairs * strips;
        TF_TRY(fb_carry_room(fb, items * 5 * 128, items * 2));
        if (++fb->chain_epoch == 0) { // 2^32 chained launches later: the flags start over
            TF_HIP(hipMemsetAsync(fb->chain_words.as<unsigned>() + 16, 0, fb->chain_words.bytes - 64, stream()));
            fb->chain_epoch = 1;
        }
        cc.mode = 1;
        cc.carry = fb->col_carry.as<double>();
        cc.flags = fb->chain_words.as<unsigned>() + 16;
        cc.epoch = fb->chain_epoch;
        cc.ticket = fb->chain_words.as<unsigned>();
        TF_HIP(hipMemsetAsync(cc.ticket, 0, 8 * sizeof(unsigned), stream()));
        cc.fault = fb->chain_fault;
        grid = dim3((unsigned)items);
    } else if (mc.segs > 1) {
        TF_TRY(fb_carry_room(fb, (size_t)mc.segs * n_pairs * 5 * w, 0));
        cc.carry = fb->col_carry.as<double>();
        cc.fault = fb->chain_fault;
        const dim3 pgrid(cdiv(w, 128), mc.segs, n_pairs);
        if (up)
            TF_TRY(launch(lvl_name("fb_flow_carry", k), k_flow_carry_pc<M, 2>, pgrid, dim3(128), 0, R, flow_in, w, h, mc.seg, f, f.yofs, f.yfrac, cc.carry));
        else if (flow_in)
            TF_TRY(launch(lvl_name("fb_flow_carry", k), k_flow_carry_pc<M, 1>, pgrid, dim3(128), 0, R, flow_in, w, h, mc.seg, f, f.yofs, f.yfrac, cc.carry));
        else
            TF_TRY(launch(lvl_name("fb_flow_carry", k), k_flow_carry_pc<M, 0>, pgrid, dim3(128), 0, R, flow_in, w, h, mc.seg, f, f.yofs, f.yfrac, cc.carry));
        TF_TRY(fb_carry_scan(fb, w, n_pairs, mc.segs, k));
    } else {
        cc.mode = 0;
    }
    int rc;
    const int kind = up ? 2 : (flow_in ? 1 : 0);
    const char *name = lvl_name("fb_flow_iter", k);
#define TF_PC_LAUNCH(FLOWK)                                                                                                    \
    launch(name, k_flow_iter_pc<M, FLOWK>, grid, dim3(TF_PC_THREADS), 0, R, flow_in, flow_out, w, h, scale, mc.seg, f, f.yofs,   \
           f.yfrac, cc)
    rc = kind == 2 ? TF_PC_LAUNCH(2) : (kind == 1 ? TF_PC_LAUNCH(1) : TF_PC_LAUNCH(0));
#undef TF_PC_LAUNCH
    return rc;
}

// true if the fused iteration kernel exists for this window; launches it
static bool fb_flow_iter(tf_fb *fb, int w, int h, int n_pairs, const float2 *flow_in, float2 *flow_out, int k, int &rc,
                         const FlowInit *up = nullptr)
{
    if (w < 10 || h < 10) // border_scale: below 10 x 10 the two-kernel form carries OpenCV's edge test
        return false;
    switch (fb->prm.winsize / 2) { // odd half-widths: the pair-sum window
    case 3: rc = launch_flow_iter<3>(fb, w, h, n_pairs, flow_in, flow_out, k, up); return true;
    case 5: rc = launch_flow_iter<5>(fb, w, h, n_pairs, flow_in, flow_out, k, up); return true;
    case 7: rc = launch_flow_iter<7>(fb, w, h, n_pairs, flow_in, flow_out, k, up); return true;
    default: return false;
    }
}

static int fb_gauss_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k = -1)
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

static int fb_setup_flags(tf_fb *fb)
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
static int fb_initial_flow(tf_fb *fb, int n_pairs, float2 *out)
{
    const Level &C = *fb->lv[fb->K];
    double scale = 1;
    for (int i = 0; i < fb->K; i++)
        scale *= fb->prm.pyr_scale;
    dim3 block(64), grid(cdiv(C.W, 64), C.H, n_pairs);
    return launch("fb_initial_flow", k_flow_area_init, grid, block, 0, (const float2 *)fb->init_flow.as<float2>(), out, fb->W,
                  fb->H, C.W, C.H, fb->area, (float)scale);
}

static int fb_validate_params(const tf_fb_params *p, int width, int height)
{
    TF_REQUIRE(width > 0 && height > 0 && (long long)width * height < (1ll << 30), "tf_fb_create: bad size %dx%d", width,
               height);
    TF_REQUIRE(p->pyr_scale > 0 && p->pyr_scale < 1, "tf_fb_create: pyr_scale must be in (0,1), got %g", p->pyr_scale);
    TF_REQUIRE(p->levels >= 0 && p->levels <= 30, "tf_fb_create: bad levels %d", p->levels);
    TF_REQUIRE(p->winsize >= 1 && p->winsize / 2 <= 100, "tf_fb_create: bad winsize %d", p->winsize);
    TF_REQUIRE(p->iterations >= 1, "tf_fb_create: iterations must be >= 1, got %d", p->iterations);
    TF_REQUIRE(p->poly_n >= 1 && p->poly_n <= MAX_POLY_N, "tf_fb_create: poly_n must be in [1,%d], got %d", MAX_POLY_N,
               p->poly_n);
    if ((p->flags & ~(4 | 256)) != 0)
        return set_error(TF_ERR_UNSUPPORTED, "tf_fb_create: flags=%d not supported (OPTFLOW_USE_INITIAL_FLOW = 4 and "
                                             "OPTFLOW_FARNEBACK_GAUSSIAN = 256 are)", p->flags);
    if ((p->flags & 256) && p->winsize / 2 > 31)
        return set_error(TF_ERR_UNSUPPORTED, "tf_fb_create: the Gaussian window serves winsize <= 63, got %d", p->winsize);
    return TF_OK;
}

// `share`: the handle whose frame slots the new one uses (tf_fb_create_lane)
static int fb_create(tf_fb **out, int width, int height, const tf_fb_params *params, int frame_slots, int max_pairs, tf_fb *share)
{
    TF_REQUIRE(out && params, "tf_fb_create: null pointer");
    TF_TRY(fb_validate_params(params, width, height));
    TF_REQUIRE(frame_slots >= 2 && max_pairs >= 1, "tf_fb_create: need >= 2 frame slots and >= 1 pair");
    TF_TRY(ensure_init());
    tf_fb *fb = new tf_fb;
    auto fail = [&](int rc) {
        delete fb;
        return rc;
    };
    fb->W = width;
    fb->H = height;
    fb->prm = *params;
    fb->slots = frame_slots;
    fb->max_pairs = max_pairs;
    fb->pc = make_poly_const(params->poly_n, params->poly_sigma);
    // A.1 driver: number of usable coarse scales
    {
        int k;
        double scale = 1;
        for (k = 0; k < params->levels; k++) {
            scale *= params->pyr_scale;
            if (width * scale < 32 || height * scale < 32)
                break;
        }
        fb->K = k;
    }
    int rc;
    for (int k = 0; k <= fb->K; k++) {
        Level *L = new Level;
        fb->lv.push_back(L);
        double scale = 1;
        for (int i = 0; i < k; i++)
            scale *= params->pyr_scale;
        L->sigma = (1. / scale - 1) * 0.5;
        int sm = cv_round(L->sigma * 5) | 1;
        L->ksz = std::max(sm, 3);
        L->W = cv_round(width * scale);
        L->H = cv_round(height * scale);
        if (L->W < 1 || L->H < 1)
            return fail(set_error(TF_ERR_ARG, "tf_fb_create: level %d is empty", k));
        std::vector<float> kern = gaussian_kernel(L->ksz, L->sigma);
        L->kern_host = kern;
        if ((rc = L->kern.alloc(kern.size() * 4)))
            return fail(rc);
        if (hipMemcpy(L->kern.p, kern.data(), kern.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return fail(set_error(TF_ERR_HIP, "hipMemcpy failed"));
        if ((rc = L->img_lerp.upload_tabs(width, height, L->W, L->H)))
            return fail(rc);
        L->tile = choose_tile(width, height, L->W, L->H, L->ksz, k);
        {
            std::vector<int> colsrc;
            L->split = plan_split_level(width, height, *L, colsrc);
            if (L->split) {
                if ((rc = L->colsrc.alloc(colsrc.size() * 4)))
                    return fail(rc);
                if (hipMemcpy(L->colsrc.p, colsrc.data(), colsrc.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return fail(set_error(TF_ERR_HIP, "hipMemcpy failed"));
            }
        }
        if (tune("TF_DEBUG_TILES", 0))
            fprintf(stderr, "level %d: %dx%d ksz=%d tile %dx%d LW=%d LH=%d pitch=%d\n", k, L->W, L->H, L->ksz,
                    L->tile.TWo, L->tile.THo, L->tile.LW, L->tile.LH, L->tile.pitch);
        {
            size_t smem = (size_t)L->tile.LH * L->tile.pitch + (size_t)L->tile.LH * L->tile.rstride * sizeof(float) +
                          (size_t)L->ksz * sizeof(float);
            if (smem > 64 * 1024)
                return fail(set_error(TF_ERR_UNSUPPORTED, "tf_fb_create: level %d needs %zu bytes of LDS per tile "
                                                          "(blur kernel %d taps)", k, smem, L->ksz));
        }
    }
    for (int k = 0; k < fb->K; k++) {
        Level &L = *fb->lv[k], &C = *fb->lv[k + 1];
        if ((rc = L.flow_lerp.upload_tabs(C.W, C.H, L.W, L.H)))
            return fail(rc);
    }
    if ((rc = fb_setup_flags(fb)))
        return fail(rc);
    const size_t N0 = (size_t)width * height, P = (size_t)max_pairs;
    fb->nsets = (fb_overlap_enabled() && fb->K > 0) ? 2 : 1; // a single scale: every launch fills the chip anyway
    if (share)
        fb->frames.borrow(share->frames.p, N0 * frame_slots);
    if ((!share && (rc = fb->frames.alloc(N0 * frame_slots))) || (rc = fb->img.alloc(P * 2 * N0 * 4)) ||
        (rc = fb->R.alloc(P * 10 * N0 * 4)) ||
        (rc = fb->M.alloc(P * 5 * N0 * 4)) ||
        (rc = fb->lflow[0].alloc(P * N0 * 8)) || (rc = fb->lflow[1].alloc(P * N0 * 8)) ||
        (rc = fb->lflow[2].alloc(P * N0 * 8)) ||
        (fb->nsets > 1 && ((rc = fb->lflow[3].alloc(P * N0 * 8)) || (rc = fb->lflow[4].alloc(P * N0 * 8)))) ||
        (rc = fb->pairs.alloc(3 * P * 8)) || (rc = fb->winner.alloc(N0 * 4)) || (rc = fb->scratch.alloc(N0 * 20)))
        return fail(rc);
    for (int k = 1; k <= fb->K; k++) {
        Level &L = *fb->lv[k];
        const size_t nk = (size_t)L.W * L.H;
        if ((rc = L.img.alloc(P * 2 * nk * 4)) || (rc = L.R.alloc(P * 10 * nk * 4)))
            return fail(rc);
    }
    {
        // one row-pass launch for all split levels: rows staged with the longest kernel's margin, pitch = 1
        // (mod 8) dwords (conflict-free for every lane mapping R = 2, 4, 8), ~40 KB of rows per workgroup
        int nsplit = 0, rmax = 0;
        size_t floats = 0;
        for (int k = fb->K; k >= 1; k--) {
            Level &L = *fb->lv[k];
            if (!L.split)
                continue;
            if (nsplit == RP_MAX_LEVELS) { // more split levels than one launch carries: the rest keep k_level_image
                L.split = false;
                continue;
            }
            if (nsplit++ == 0)
                fb->rp_first = k;
            rmax = std::max(rmax, L.ksz / 2);
            L.rowf_off = floats;
            floats += P * 2 * (size_t)height * L.NC;
        }
        if (nsplit) {
            fb->rp_rmax = rmax;
            fb->rp_r4 = (rmax + 3) & ~3;
            int pitch = (fb->rp_r4 + width + rmax + 8 + 3) & ~3;
            while (((pitch / 4) % 8) != 1)
                pitch += 4;
            fb->rp_pitch = pitch;
            const int RB = (int)((40 * 1024) / pitch) & ~7;
            if (RB < 8) { // frame rows too long to stage eight of them
                for (int k = 1; k <= fb->K; k++)
                    fb->lv[k]->split = false;
                fb->rp_first = -1;
            } else {
                fb->rp_RB = std::min(RB, 64);
                if ((rc = fb->rowf.alloc(floats * 4)))
                    return fail(rc);
            }
        }
    }
    if (hipEventCreateWithFlags(&fb->chain_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&fb->pairs_copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&fb->entry[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&fb->entry[1], hipEventDisableTiming) != hipSuccess ||
        side_stream(share ? 2 : 1, &fb->chain_stream) != TF_OK ||
        hipHostMalloc((void **)&fb->pairs_host, 3 * P * sizeof(int2), hipHostMallocDefault) != hipSuccess)
        return fail(set_error(TF_ERR_HIP, "creating the handle's events and staging buffer failed"));
    if (share) {
        fb->lane_of = share;
        fb->exact = share->exact;
        share->lanes++;
    }
    *out = fb;
    return TF_OK;
}

TF_API int tf_fb_create(tf_fb **out, int width, int height, const tf_fb_params *params, int frame_slots, int max_pairs)
{
    return fb_create(out, width, height, params, frame_slots, max_pairs, nullptr);
}

// A second lane for the calls of `first`: a handle of the same size and parameters that reads first's frame slots and
// queues its calls on the library's OTHER call stream.  A caller that sends batches alternately to the two keeps two calls
// in flight: the part-empty launches of one (the coarse levels; the tail of every launch) run beside the full ones of the
// other.  Results are per handle, as ever: tf_fb_flow_ptr etc. on the handle that ran the call.
TF_API int tf_fb_create_lane(tf_fb **out, tf_fb *first)
{
    TF_REQUIRE(out && first, "tf_fb_create_lane: null pointer");
    TF_REQUIRE(!first->lane_of, "tf_fb_create_lane: the handle is itself a lane");
    TF_REQUIRE(!first->keep, "tf_fb_create_lane: the handle keeps its expansions between calls (tf_fb_keep_expansions); lanes do not share them");
    return fb_create(out, first->W, first->H, &first->prm, first->slots, first->max_pairs, first);
}

// A handle whose frame slots lanes still read outlives its own tf_fb_destroy: it is released with the last of them
// (the lanes' `frames` is first's memory, and their lane_of points at it).
TF_API void tf_fb_destroy(tf_fb *fb)
{
    if (!fb)
        return;
    if (fb->lanes > 0) {
        fb->destroy_with_lanes = true;
        return;
    }
    tf_fb *first = fb->lane_of;
    delete fb;
    if (first && --first->lanes == 0 && first->destroy_with_lanes)
        delete first;
}

// Per-handle exactness (cv.py:479-490: one call, one result -- no state outside the handle decides what a call returns).
TF_API int tf_fb_set_exact(tf_fb *fb, int mode)
{
    TF_REQUIRE(fb, "tf_fb_set_exact: null handle");
    TF_REQUIRE(mode >= -1 && mode <= 1, "tf_fb_set_exact: mode %d (1: exact, 0: default, -1: as option fb_exact_sums says)", mode);
    fb->exact = mode;
    return TF_OK;
}

TF_API int tf_fb_level_count(tf_fb *fb, int *n_scales)
{
    TF_REQUIRE(fb && n_scales, "tf_fb_level_count: null pointer");
    *n_scales = fb->K + 1;
    return TF_OK;
}

TF_API int tf_fb_level_size(tf_fb *fb, int level, int *w, int *h)
{
    TF_REQUIRE(fb && w && h, "tf_fb_level_size: null pointer");
    TF_REQUIRE(level >= 0 && level <= fb->K, "tf_fb_level_size: level %d out of range", level);
    *w = fb->lv[level]->W;
    *h = fb->lv[level]->H;
    return TF_OK;
}

// tf_fb_async_io: a frame goes up on the upload stream, which first waits for the last call that read the slot's old
// frame (a streaming caller's slot was read two calls ago: no wait in practice).  Without async io: nothing to do, the
// upload is ordered on the caller's stream.
static int fb_upload_scope_begin(tf_fb *fb, int slot)
{
    if (!fb->async_io)
        return TF_OK;
    const long last = fb->slot_read_call[(size_t)slot];
    if (last >= 0 && fb->call_done[last & 1])
        TF_HIP(hipStreamWaitEvent(fb->up_stream, fb->call_done[last & 1], 0)); // (that call's event, or a later call's of the same parity)
    return TF_OK;
}

// Streaming callers (transflow/flow/sources/cv.py:460-490 per frame): with async io on, tf_fb_set_frame / _bgr put the
// frame up on a copy stream of the library's (they still return with the frame in place, but no longer wait for the
// handle's kernels in flight), and tf_fb_get_flow_begin / _end bring a result down on another, beside the next call.
TF_API int tf_fb_async_io(tf_fb *fb, int on)
{
    TF_REQUIRE(fb, "tf_fb_async_io: null handle");
    TF_TRY(ensure_init());
    if (on && !fb->up_stream) {
        TF_TRY(side_stream(3, &fb->up_stream));
        TF_TRY(side_stream(4, &fb->down_stream));
        for (int i = 0; i < 2; i++) {
            TF_HIP(hipEventCreateWithFlags(&fb->call_done[i], hipEventDisableTiming));
            TF_HIP(hipEventCreateWithFlags(&fb->result_ready[i], hipEventDisableTiming));
            TF_HIP(hipEventCreateWithFlags(&fb->download_done[i], hipEventDisableTiming));
        }
        fb->slot_read_call.assign((size_t)fb->slots, -1);
    }
    if (!on)
        for (int i = 0; i < 2; i++)
            if (fb->download_pending[i]) {
                TF_HIP(hipEventSynchronize(fb->download_done[i]));
                fb->download_pending[i] = false;
            }
    fb->async_io = on != 0;
    return TF_OK;
}

TF_API int tf_fb_set_frame(tf_fb *fb, int slot, const uint8_t *grey, ptrdiff_t stride)
{
    TF_REQUIRE(fb && grey, "tf_fb_set_frame: null pointer");
    TF_REQUIRE(slot >= 0 && slot < fb->slots, "tf_fb_set_frame: slot %d out of range (%d slots)", slot, fb->slots);
    TF_REQUIRE(stride >= fb->W, "tf_fb_set_frame: stride %td smaller than width %d", stride, fb->W);
    TF_TRY(ensure_init());
    uint8_t *dst = fb->frames.as<uint8_t>() + (size_t)slot * fb->W * fb->H;
    if (fb->keep)
        fb->expanded[slot] = 0;
    TF_TRY(fb_upload_scope_begin(fb, slot));
    StreamScope up(fb->async_io ? fb->up_stream : stream());
    TF_HIP(hipMemcpy2DAsync(dst, fb->W, grey, (size_t)stride, fb->W, fb->H, hipMemcpyHostToDevice, stream()));
    TF_HIP(hipStreamSynchronize(stream())); // the host frame is borrowed for this call only
    return TF_OK;
}

// cv.py:461-466 on the device: the decoded BGR frame goes up as it is, cv2.resize(INTER_NEAREST) to the
// handle's size and cv2.cvtColor(COLOR_BGR2GRAY) run as one kernel straight into the frame slot.
TF_API int tf_fb_set_frame_bgr(tf_fb *fb, int slot, const uint8_t *bgr, int src_width, int src_height, ptrdiff_t stride)
{
    TF_REQUIRE(fb && bgr, "tf_fb_set_frame_bgr: null pointer");
    TF_REQUIRE(slot >= 0 && slot < fb->slots, "tf_fb_set_frame_bgr: slot %d out of range (%d slots)", slot, fb->slots);
    TF_REQUIRE(src_width >= 1 && src_height >= 1 && (long long)src_width * src_height < (1ll << 31),
               "tf_fb_set_frame_bgr: bad source size %dx%d", src_width, src_height);
    TF_REQUIRE(stride >= (ptrdiff_t)3 * src_width, "tf_fb_set_frame_bgr: stride %td smaller than a row of %d BGR pixels", stride,
               src_width);
    TF_TRY(ensure_init());
    const size_t row = (size_t)3 * src_width, need = row * src_height;
    if (fb->bgr_stage.bytes < need)
        TF_TRY(fb->bgr_stage.alloc(need));
    uint8_t *dst = fb->frames.as<uint8_t>() + (size_t)slot * fb->W * fb->H;
    if (fb->keep)
        fb->expanded[slot] = 0;
    TF_TRY(fb_upload_scope_begin(fb, slot));
    StreamScope up(fb->async_io ? fb->up_stream : stream());
    TF_HIP(hipMemcpy2DAsync(fb->bgr_stage.p, row, bgr, (size_t)stride, row, src_height, hipMemcpyHostToDevice, stream()));
    TF_TRY(tf_frame_grey_dev(fb->bgr_stage.p, src_width, src_height, dst, fb->W, fb->H));
    TF_HIP(hipStreamSynchronize(stream())); // the host frame is borrowed for this call only
    return TF_OK;
}

TF_API int tf_fb_frame_ptr(tf_fb *fb, int slot, void **dev)
{
    TF_REQUIRE(fb && dev, "tf_fb_frame_ptr: null pointer");
    TF_REQUIRE(slot >= 0 && slot < fb->slots, "tf_fb_frame_ptr: slot %d out of range", slot);
    if (fb->keep)
        fb->external[slot] = 1; // written behind the library's back from now on: expanded on every call
    *dev = fb->frames.as<uint8_t>() + (size_t)slot * fb->W * fb->H;
    return TF_OK;
}

TF_API int tf_fb_keep_expansions(tf_fb *fb, int on)
{
    TF_REQUIRE(fb, "tf_fb_keep_expansions: null handle");
    TF_REQUIRE(!on || (!fb->lane_of && fb->lanes == 0), "tf_fb_keep_expansions: not with lanes (tf_fb_create_lane): expansions are per handle");
    TF_REQUIRE(!on || fb->slots <= 2 * fb->max_pairs, "tf_fb_keep_expansions: %d frame slots need room for %d expansions, "
                                                      "the handle holds %d (2 x max_pairs)", fb->slots, fb->slots,
               2 * fb->max_pairs);
    fb->keep = on != 0;
    fb->expanded.assign((size_t)fb->slots, 0);
    fb->external.assign((size_t)fb->slots, 0);
    return TF_OK;
}

TF_API int tf_fb_calc_slots(tf_fb *fb, int n_pairs, const int *prev_slots, const int *next_slots)
{
    TF_REQUIRE(fb && prev_slots && next_slots, "tf_fb_calc_slots: null pointer");
    TF_REQUIRE(n_pairs >= 1 && n_pairs <= fb->max_pairs, "tf_fb_calc_slots: n_pairs %d not in [1,%d]", n_pairs,
               fb->max_pairs);
    TF_TRY(ensure_init());
    for (int i = 0; i < n_pairs; i++)
        TF_REQUIRE(prev_slots[i] >= 0 && prev_slots[i] < fb->slots && next_slots[i] >= 0 && next_slots[i] < fb->slots,
                   "tf_fb_calc_slots: pair %d uses a slot outside [0,%d)", i, fb->slots);
    TF_TRY(fb_check_fault(fb, "tf_fb_calc_slots (an earlier call)"));
    if (!fb_exact(fb) && fb->exact_vsum.p && fb->prm.winsize / 2 != 0)
        fb->exact_vsum.release(); // the exact mode's column sums (40 bytes per pixel and pair): not kept once it is off
    if (fb->pairs_pending) { // the previous call's copy out of the staging buffer (long done in practice)
        TF_HIP(hipEventSynchronize(fb->pairs_copied));
        fb->pairs_pending = false;
    }
    // A1+A2 depend on the frame alone, and consecutive pairs of a video share one: every slot the batch
    // names is expanded once (16 consecutive pairs: 17 expansions, not 32) and each pair carries the
    // indices of its two.  TF_FB_NO_SHARE=1: one expansion per pair and side, as separate calls would do.
    // With tf_fb_keep_expansions an image IS its slot and survives the call: only slots written since
    // their last expansion are listed, in runs of consecutive slots (a run = one set of launches).
    const bool no_share = option(OPT_FB_NO_SHARE) != 0; // read per call
    const int P = fb->max_pairs;
    int *image_slot = reinterpret_cast<int *>(fb->pairs_host); // [4P]: the runs, each padded to an even length
    int2 *rmap_host = fb->pairs_host + 2 * P;
    struct Run {
        int image0, list0, n; // first image index written, offset into image_slot (even), images
    };
    std::vector<Run> runs;
    if (fb->keep) {
        std::vector<int> need;
        fb->image_of.assign((size_t)fb->slots, -1);
        for (int i = 0; i < n_pairs; i++) {
            for (int s : {prev_slots[i], next_slots[i]})
                if (fb->image_of[s] < 0 && (!fb->expanded[s] || fb->external[s])) {
                    fb->image_of[s] = 1;
                    need.push_back(s);
                }
            rmap_host[i] = make_int2(prev_slots[i], next_slots[i]);
        }
        std::sort(need.begin(), need.end());
        int off = 0;
        for (size_t q = 0; q < need.size();) {
            size_t e = q + 1;
            while (e < need.size() && need[e] == need[e - 1] + 1)
                e++;
            runs.push_back({need[q], off, (int)(e - q)});
            for (size_t j = q; j < e; j++)
                image_slot[off++] = need[j];
            if (off & 1)
                image_slot[off++] = need[q];
            q = e;
        }
    } else {
        int n_images = 0;
        fb->image_of.assign((size_t)fb->slots, -1);
        auto image = [&](int slot) {
            if (no_share || fb->image_of[slot] < 0) {
                fb->image_of[slot] = n_images;
                image_slot[n_images++] = slot;
            }
            return fb->image_of[slot];
        };
        for (int i = 0; i < n_pairs; i++) {
            const int a = image(prev_slots[i]);
            rmap_host[i] = make_int2(a, image(next_slots[i]));
        }
        if (n_images & 1)
            image_slot[n_images] = image_slot[0]; // the unused half of the last int2
        runs.push_back({0, 0, n_images});
    }
    const int m = fb->prm.winsize / 2;
    const bool fusable = m == 3 || m == 5 || m == 7; // the pair-sum window of the fused kernel
    // A call's work goes to a stream of its own.  The library stream -- where the caller's work on the
    // result goes (post_process, the remap of each pair) -- waits for its end, and the NEXT call does
    // not wait for that work: the result lands in one of two buffers by call parity, and this call
    // only waits for what the library stream had been given when the PREVIOUS call was issued (all
    // that could still read this parity's buffer).  So the remap of call i (a serial chain of
    // per-pair launches) runs beside the coarse levels of call i+1 (small launches, mostly idle chip).
    // Within a call everything is in order on the one stream: running the expansion of the next batch
    // beside the flow chain on a third stream was measured slower at every size.
    const bool overlap = fb->nsets > 1;
    const int set = fb->cur;
    hipStream_t cs = overlap ? fb->chain_stream : main_stream();
    if (overlap) {
        TF_HIP(hipEventRecord(fb->entry[set], main_stream()));
        fb->entry_pending[set] = true;
        if (fb->entry_pending[set ^ 1])
            TF_HIP(hipStreamWaitEvent(cs, fb->entry[set ^ 1], 0));
    }
    // Frames: tf_fb_set_frame returns with the frame in place; a caller writing frames on the device
    // orders that itself (tfhip.h).
    if (fb->async_io) {
        if (fb->download_pending[set]) // this call's result takes the place of the one still on its way down
            TF_HIP(hipStreamWaitEvent(cs, fb->download_done[set], 0));
        for (const Run &run : runs) // the slots whose frame bytes this call reads
            for (int j = 0; j < run.n; j++)
                fb->slot_read_call[(size_t)image_slot[run.list0 + j]] = fb->n_calls;
    }
    TF_HIP(hipMemcpyAsync(fb->pairs.p, fb->pairs_host, (size_t)3 * P * sizeof(int2), hipMemcpyHostToDevice, cs));
    TF_HIP(hipEventRecord(fb->pairs_copied, cs));
    fb->pairs_pending = true;
    struct MapScope { // the launchers below read these from the handle; stage entry points run without them
        tf_fb *fb;
        ~MapScope()
        {
            fb->rmap_dev = nullptr;
            fb->prep_image0 = fb->prep_list0 = 0;
        }
    } map_scope{fb};
    fb->rmap_dev = fb->pairs.as<int2>() + 2 * P;
    StreamScope chain_scope(cs);
    // A1+A2 of every level (they depend on the frames only), coarse level first: the shared row pass
    // of the long-kernel levels is launched with the coarsest of them
    for (const Run &run : runs) {
        fb->prep_image0 = run.image0;
        fb->prep_list0 = run.list0 / 2;
        for (int k = fb->K; k >= 0; k--) {
            Level &L = *fb->lv[k];
            if (fb_can_fuse_level(fb, k)) {
                TF_TRY(fb_level0_polyexp(fb, k, run.n)); // A1+A2 in one kernel: the level image stays on chip
            } else if (fb_can_fuse_half_level(fb, k)) {
                TF_TRY(fb_level1_polyexp(fb, k, run.n));
            } else {
                TF_TRY(fb_level_image(fb, k, run.n));
                TF_TRY(fb_polyexp(fb, L.W, L.H, run.n, k));
            }
        }
    }
    fb->prep_image0 = fb->prep_list0 = 0;
    if (fb->keep) // valid from here on (until tf_fb_set_frame writes the slot)
        for (const Run &run : runs)
            for (int j = 0; j < run.n; j++)
                fb->expanded[run.image0 + j] = 1;
    int coarse = -1; // lflow buffer holding the coarser level's result
    for (int k = fb->K; k >= 0; k--) {
        Level &L = *fb->lv[k];
        FlowInit fi;
        memset(&fi, 0, sizeof(fi));
        if (k < fb->K) {
            Level &C = *fb->lv[k + 1];
            fi.mode = 1;
            fi.src = fb->lflow[coarse].as<float2>();
            fi.Wc = C.W;
            fi.Hc = C.H;
            fi.xofs = L.flow_lerp.xofs.as<int>();
            fi.yofs = L.flow_lerp.yofs.as<int>();
            fi.xfrac = L.flow_lerp.xfrac.as<float>();
            fi.yfrac = L.flow_lerp.yfrac.as<float>();
            fi.mul = (float)(1. / fb->prm.pyr_scale);
        }
        // two buffers other than `coarse` for this level's iterations
        int a = (coarse + 1) % 3, b = (coarse + 2) % 3;
        if (coarse < 0) {
            a = 0;
            b = 1;
        }
        const int out_buf = overlap ? 3 + set : -1; // where the full-resolution result of this call lives
        const long fuse_min_px = option(OPT_FB_FUSE_MIN_PX);
        // (the fused kernel multiplies the edge weights unconditionally: identical from 10 x 10 up, border_scale)
        // fb_exact_sums: column sums straight from R (k_flow_carry_pc<.., STORE>, one march per column) where many columns
        // stand side by side, through M in memory otherwise (4K: 7.0 against 9.3 ms per level-0 iteration with 32 pairs,
        // 2.9 against 2.5 with 8, 1.8 against 0.5 with one)
        const bool exact_one_kernel = !fb_exact(fb) || fb->fused > 0 || (long)cdiv(L.W, 112) * n_pairs >= 560;
        const bool fused_here = fusable && !fb->gaussian() && exact_one_kernel && L.W >= 10 && L.H >= 10 && (size_t)L.W * L.H < (1u << 29) && L.W < (1 << 24) && L.H < (1 << 24) &&
                                (fb->fused > 0 || (fb->fused < 0 && (long)L.W * L.H * n_pairs >= fuse_min_px));
        if (k == 0 && overlap && !fused_here)
            a = out_buf;
        int result;
        // one kernel per iteration where the level fills the chip (>= 4M pixels over the batch), two otherwise
        if (fused_here) {
            // the iterations ping-pong between two buffers, ordered so that the LAST one writes p: the
            // call's result buffer at full resolution, otherwise any buffer the coarser level is not in
            const int I = fb->prm.iterations;
            const int p = (k == 0 && overlap) ? out_buf : a, q = (k == 0 && overlap) ? a : b;
            auto buf_of = [&](int i) { return ((I - i) & 1) ? q : p; }; // what iteration i (1..I) writes
            const float2 *src = nullptr; // zero flow at the coarsest scale ...
            if (k == fb->K && fb->use_initial()) { // ... or the caller's, shrunk to it (a buffer neither p nor q)
                int c = 0;
                while (c == p || c == q)
                    c++;
                TF_TRY(fb_initial_flow(fb, n_pairs, fb->lflow[c].as<float2>()));
                src = fb->lflow[c].as<float2>();
            }
            for (int i = 1; i <= I; i++) {
                int rc = TF_OK;
                // the first iteration below the coarsest scale upsamples the coarser level's flow itself
                if (!fb_flow_iter(fb, L.W, L.H, n_pairs, src, fb->lflow[buf_of(i)].as<float2>(), k, rc,
                                  (i == 1 && k < fb->K) ? &fi : nullptr))
                    return set_error(TF_ERR_STATE, "tf_fb_calc_slots: no one-kernel iteration for level %d", k);
                TF_TRY(rc);
                src = fb->lflow[buf_of(i)].as<float2>();
            }
            result = p;
        } else {
            FlowInit fl;
            memset(&fl, 0, sizeof(fl));
            fl.mode = 2;
            fl.src = fb->lflow[a].as<float2>();
            if (k == fb->K && fb->use_initial()) { // the caller's flow, shrunk to the coarsest scale, instead of zero
                TF_TRY(fb_initial_flow(fb, n_pairs, fb->lflow[a].as<float2>()));
                TF_TRY(fb_update_matrices(fb, L.W, L.H, n_pairs, fl, k));
            } else {
                TF_TRY(fb_update_matrices(fb, L.W, L.H, n_pairs, fi, k));
            }
            for (int i = 0; i < fb->prm.iterations; i++) {
                if (fb->gaussian())
                    TF_TRY(fb_gauss_solve(fb, L.W, L.H, n_pairs, fb->lflow[a].as<float2>(), k));
                else
                    TF_TRY(fb_blur_solve(fb, L.W, L.H, n_pairs, fb->lflow[a].as<float2>(), k));
                if (i < fb->prm.iterations - 1) // M is a pure function of (R0, R1, flow): rebuild it in place
                    TF_TRY(fb_update_matrices(fb, L.W, L.H, n_pairs, fl, k));
            }
            result = a;
        }
        coarse = result;
        fb->final_buf = result;
    }
    if (overlap) {
        TF_HIP(hipEventRecord(fb->chain_done, cs));
        TF_HIP(hipStreamWaitEvent(main_stream(), fb->chain_done, 0));
    }
    if (fb->async_io)
        TF_HIP(hipEventRecord(fb->call_done[fb->n_calls & 1], cs));
    fb->n_calls++;
    fb->cur = (set + 1) % fb->nsets;
    fb->last_pairs = n_pairs;
    return TF_OK;
}

TF_API int tf_fb_flow_ptr(tf_fb *fb, int pair, void **dev)
{
    TF_REQUIRE(fb && dev, "tf_fb_flow_ptr: null pointer");
    TF_REQUIRE(pair >= 0 && pair < fb->max_pairs, "tf_fb_flow_ptr: pair %d out of range", pair);
    *dev = fb->lflow[fb->final_buf].as<float2>() + (size_t)pair * fb->W * fb->H;
    return TF_OK;
}

TF_API int tf_fb_get_flow(tf_fb *fb, int pair, float *flow_out)
{
    TF_REQUIRE(fb && flow_out, "tf_fb_get_flow: null pointer");
    TF_REQUIRE(pair >= 0 && pair < fb->last_pairs, "tf_fb_get_flow: pair %d was not computed by the last call", pair);
    TF_TRY(ensure_init());
    void *src;
    TF_TRY(tf_fb_flow_ptr(fb, pair, &src));
    TF_HIP(hipMemcpyAsync(flow_out, src, (size_t)fb->W * fb->H * 8, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return fb_check_fault(fb, "tf_fb_get_flow");
}

// The flow of `pair` on its way to `flow_out` beside whatever is queued next: the copy starts when the caller's stream
// reaches this point (after its post_process), on the library's download stream; *token names the transfer for
// tf_fb_get_flow_end, which returns once the array is filled.  flow_out should be page-locked (tf_host_alloc).
TF_API int tf_fb_get_flow_begin(tf_fb *fb, int pair, float *flow_out, int *token)
{
    TF_REQUIRE(fb && flow_out && token, "tf_fb_get_flow_begin: null pointer");
    TF_REQUIRE(fb->async_io, "tf_fb_get_flow_begin: tf_fb_async_io is off");
    TF_REQUIRE(pair >= 0 && pair < fb->last_pairs, "tf_fb_get_flow_begin: pair %d was not computed by the last call", pair);
    TF_TRY(ensure_init());
    const int set = (fb->cur + fb->nsets - 1) % fb->nsets; // the last call's result set
    if (fb->download_pending[set]) {
        // A handle with ONE result set in rotation (a single scale, or option fb_no_overlap): the download a streaming
        // caller still holds is of this very set.  The call in between already waited for it on the device before it
        // wrote the set again (tf_fb_calc_slots), so it is over or about to be: end it here; its tf_fb_get_flow_end then
        // finds nothing left to wait for but this download.
        TF_HIP(hipEventSynchronize(fb->download_done[set]));
        fb->download_pending[set] = false;
    }
    void *src;
    TF_TRY(tf_fb_flow_ptr(fb, pair, &src));
    TF_HIP(hipEventRecord(fb->result_ready[set], stream()));
    TF_HIP(hipStreamWaitEvent(fb->down_stream, fb->result_ready[set], 0));
    TF_HIP(hipMemcpyAsync(flow_out, src, (size_t)fb->W * fb->H * 8, hipMemcpyDeviceToHost, fb->down_stream));
    TF_HIP(hipEventRecord(fb->download_done[set], fb->down_stream));
    fb->download_pending[set] = true;
    *token = set;
    return TF_OK;
}

TF_API int tf_fb_get_flow_end(tf_fb *fb, int token)
{
    TF_REQUIRE(fb, "tf_fb_get_flow_end: null handle");
    TF_REQUIRE(token == 0 || token == 1, "tf_fb_get_flow_end: token %d", token);
    if (fb->download_pending[token]) {
        TF_HIP(hipEventSynchronize(fb->download_done[token]));
        fb->download_pending[token] = false;
    }
    return fb_check_fault(fb, "tf_fb_get_flow_end");
}

TF_API int tf_fb_set_initial_flow(tf_fb *fb, int pair, const float *flow)
{
    TF_REQUIRE(fb && flow, "tf_fb_set_initial_flow: null pointer");
    TF_REQUIRE(fb->use_initial(), "tf_fb_set_initial_flow: the handle was created without OPTFLOW_USE_INITIAL_FLOW (flags & 4)");
    TF_REQUIRE(pair >= 0 && pair < fb->max_pairs, "tf_fb_set_initial_flow: pair %d of %d", pair, fb->max_pairs);
    TF_TRY(ensure_init());
    const size_t N0 = (size_t)fb->W * fb->H;
    TF_HIP(hipMemcpyAsync(fb->init_flow.as<float2>() + (size_t)pair * N0, flow, N0 * 8, hipMemcpyHostToDevice, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_fb_initial_flow_ptr(tf_fb *fb, int pair, void **dev)
{
    TF_REQUIRE(fb && dev, "tf_fb_initial_flow_ptr: null pointer");
    TF_REQUIRE(fb->use_initial(), "tf_fb_initial_flow_ptr: the handle was created without OPTFLOW_USE_INITIAL_FLOW (flags & 4)");
    TF_REQUIRE(pair >= 0 && pair < fb->max_pairs, "tf_fb_initial_flow_ptr: pair %d of %d", pair, fb->max_pairs);
    *dev = fb->init_flow.as<float2>() + (size_t)pair * fb->W * fb->H;
    return TF_OK;
}

TF_API int tf_fb_calc(tf_fb *fb, const uint8_t *prev, ptrdiff_t prev_stride, const uint8_t *next, ptrdiff_t next_stride,
                      float *flow_out)
{
    TF_REQUIRE(fb && prev && next && flow_out, "tf_fb_calc: null pointer");
    if (fb->use_initial()) // cv2's `flow` is an in/out array: with OPTFLOW_USE_INITIAL_FLOW it is read first
        TF_TRY(tf_fb_set_initial_flow(fb, 0, flow_out));
    TF_TRY(tf_fb_set_frame(fb, 0, prev, prev_stride));
    TF_TRY(tf_fb_set_frame(fb, 1, next, next_stride));
    int a = 0, b = 1;
    TF_TRY(tf_fb_calc_slots(fb, 1, &a, &b));
    return tf_fb_get_flow(fb, 0, flow_out);
}

static int pp_run(tf_fb *fb, float2 *flow, int direction)
{
    TF_REQUIRE(direction == 0 || direction == 1, "post_process: direction must be 0 (FORWARD) or 1 (BACKWARD), got %d",
               direction);
    const int N = fb->W * fb->H;
    dim3 grid(cdiv((size_t)N, 256)), block(256);
    const FastDiv dw = fast_div_setup((uint32_t)fb->W);
    if (direction == 0) {
        TF_HIP(hipMemsetAsync(fb->winner.p, 0xFF, (size_t)N * 4, stream()));
        TF_TRY(launch("pp_fwd_scatter", k_pp_fwd_scatter, grid, block, 0, (const float2 *)flow, fb->winner.as<int>(),
                      fb->W, fb->H, dw));
        return launch("pp_fwd_resolve", k_pp_fwd_resolve, grid, block, 0, flow, (const int *)fb->winner.as<int>(), fb->W,
                      fb->H, dw);
    }
    return launch("pp_clip", k_pp_clip, grid, block, 0, flow, fb->W, fb->H, dw);
}

static int pp_ops_run(tf_fb *fb, float2 *flow, int n_ops, const tf_flow_op *ops, const float *mask_dev)
{
    TF_REQUIRE(n_ops >= 0 && n_ops <= TF_MAX_FLOW_OPS, "post_process: at most %d flow filters, got %d", TF_MAX_FLOW_OPS,
               n_ops);
    TF_REQUIRE(n_ops == 0 || ops, "post_process: null filter list");
    if (n_ops == 0 && !mask_dev)
        return TF_OK;
    FlowOps fo;
    memset(&fo, 0, sizeof(fo));
    fo.n = n_ops;
    for (int i = 0; i < n_ops; i++) {
        TF_REQUIRE(ops[i].kind >= TF_FLOW_SCALE && ops[i].kind <= TF_FLOW_CLIP, "post_process: unknown filter kind %d",
                   ops[i].kind);
        fo.op[i] = ops[i];
    }
    const int N = fb->W * fb->H;
    return launch("pp_ops", k_pp_ops, dim3(cdiv((size_t)N, 256)), dim3(256), 0, flow, mask_dev, N, fo);
}

TF_API int tf_fb_post_process_ex(tf_fb *fb, int pair, int direction, int n_ops, const tf_flow_op *ops,
                                 const void *mask_dev)
{
    TF_REQUIRE(fb, "tf_fb_post_process_ex: null handle");
    TF_REQUIRE(pair >= 0 && pair < fb->max_pairs, "tf_fb_post_process_ex: pair %d out of range", pair);
    TF_TRY(ensure_init());
    void *p;
    TF_TRY(tf_fb_flow_ptr(fb, pair, &p));
    TF_TRY(pp_ops_run(fb, (float2 *)p, n_ops, ops, (const float *)mask_dev));
    return pp_run(fb, (float2 *)p, direction);
}

TF_API int tf_fb_post_process_host_ex(tf_fb *fb, float *flow_inout, int direction, int n_ops, const tf_flow_op *ops,
                                      const float *mask)
{
    TF_REQUIRE(fb && flow_inout, "tf_fb_post_process_host_ex: null pointer");
    TF_TRY(ensure_init());
    const size_t n = (size_t)fb->W * fb->H;
    TF_HIP(hipMemcpyAsync(fb->scratch.p, flow_inout, n * 8, hipMemcpyHostToDevice, stream()));
    const float *mask_dev = nullptr;
    if (mask) { // scratch holds 20 B/px: the flow takes 8, the mask the next 4
        float *m = fb->scratch.as<float>() + n * 2;
        TF_HIP(hipMemcpyAsync(m, mask, n * 4, hipMemcpyHostToDevice, stream()));
        mask_dev = m;
    }
    TF_TRY(pp_ops_run(fb, fb->scratch.as<float2>(), n_ops, ops, mask_dev));
    if (direction >= 0)
        TF_TRY(pp_run(fb, fb->scratch.as<float2>(), direction));
    TF_HIP(hipMemcpyAsync(flow_inout, fb->scratch.p, n * 8, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_fb_post_process(tf_fb *fb, int pair, int direction)
{
    TF_REQUIRE(fb, "tf_fb_post_process: null handle");
    TF_REQUIRE(pair >= 0 && pair < fb->max_pairs, "tf_fb_post_process: pair %d out of range", pair);
    TF_TRY(ensure_init());
    void *p;
    TF_TRY(tf_fb_flow_ptr(fb, pair, &p));
    return pp_run(fb, (float2 *)p, direction);
}

TF_API int tf_fb_post_process_scatter(tf_fb *fb, int pair, void **winners_dev)
{
    TF_REQUIRE(fb && winners_dev, "tf_fb_post_process_scatter: null pointer");
    TF_REQUIRE(pair >= 0 && pair < fb->max_pairs, "tf_fb_post_process_scatter: pair %d out of range", pair);
    TF_TRY(ensure_init());
    void *p;
    TF_TRY(tf_fb_flow_ptr(fb, pair, &p));
    const int N = fb->W * fb->H;
    TF_HIP(hipMemsetAsync(fb->winner.p, 0xFF, (size_t)N * 4, stream()));
    TF_TRY(launch("pp_fwd_scatter", k_pp_fwd_scatter, dim3(cdiv((size_t)N, 256)), dim3(256), 0, (const float2 *)p,
                  fb->winner.as<int>(), fb->W, fb->H, fast_div_setup((uint32_t)fb->W)));
    *winners_dev = fb->winner.p;
    return TF_OK;
}

TF_API int tf_fb_post_process_host(tf_fb *fb, float *flow_inout, int direction)
{
    TF_REQUIRE(fb && flow_inout, "tf_fb_post_process_host: null pointer");
    TF_TRY(ensure_init());
    size_t bytes = (size_t)fb->W * fb->H * 8;
    TF_HIP(hipMemcpyAsync(fb->scratch.p, flow_inout, bytes, hipMemcpyHostToDevice, stream()));
    TF_TRY(pp_run(fb, fb->scratch.as<float2>(), direction));
    TF_HIP(hipMemcpyAsync(flow_inout, fb->scratch.p, bytes, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

// ---- stage entry points (parity tests drive single kernels through these) ---------
TF_API int tf_fb_stage_level_image(tf_fb *fb, const uint8_t *grey, ptrdiff_t stride, int level, float *out)
{
    TF_REQUIRE(fb && grey && out, "tf_fb_stage_level_image: null pointer");
    TF_REQUIRE(level >= 0 && level <= fb->K, "tf_fb_stage_level_image: level %d out of range", level);
    TF_TRY(tf_fb_set_frame(fb, 0, grey, stride));
    int2 pr = make_int2(0, 0);
    TF_HIP(hipMemcpy(fb->pairs.p, &pr, 8, hipMemcpyHostToDevice));
    TF_TRY(fb_level_image(fb, level, 2, true));
    Level &L = *fb->lv[level];
    TF_HIP(hipMemcpyAsync(out, fb->imgk(level), (size_t)L.W * L.H * 4, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

static int check_stage_size(tf_fb *fb, int w, int h)
{
    TF_REQUIRE(w >= 1 && h >= 1 && (size_t)w * h <= (size_t)fb->W * fb->H, "stage: %dx%d exceeds the handle's %dx%d", w,
               h, fb->W, fb->H);
    return TF_OK;
}

static int upload_planar5(float *dst_planar, const float *host_interleaved, size_t n, DevBuf &staging)
{
    TF_HIP(hipMemcpyAsync(staging.p, host_interleaved, n * 20, hipMemcpyHostToDevice, stream()));
    return launch("stage_to_planar", k_interleaved_to_planar5, dim3(cdiv(n, 256)), dim3(256), 0,
                  (const float *)staging.as<float>(), dst_planar, n);
}

static int upload_rpairs(float *dst_r, const float *host_interleaved, size_t n, DevBuf &staging)
{
    TF_HIP(hipMemcpyAsync(staging.p, host_interleaved, n * 20, hipMemcpyHostToDevice, stream()));
    return launch("stage_to_rpairs", k_interleaved_to_rpairs, dim3(cdiv(n, 256)), dim3(256), 0,
                  (const float *)staging.as<float>(), dst_r, n);
}

static int download_rpairs(float *host_interleaved, const float *src_r, size_t n, DevBuf &staging)
{
    TF_TRY(launch("stage_from_rpairs", k_rpairs_to_interleaved, dim3(cdiv(n, 256)), dim3(256), 0, src_r,
                  staging.as<float>(), n));
    TF_HIP(hipMemcpyAsync(host_interleaved, staging.p, n * 20, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

static int download_planar5(float *host_interleaved, const float *src_planar, size_t n, DevBuf &staging)
{
    TF_TRY(launch("stage_to_interleaved", k_planar5_to_interleaved, dim3(cdiv(n, 256)), dim3(256), 0, src_planar,
                  staging.as<float>(), n));
    TF_HIP(hipMemcpyAsync(host_interleaved, staging.p, n * 20, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_fb_stage_level_polyexp(tf_fb *fb, const uint8_t *grey, ptrdiff_t stride, int level, float *r_out)
{
    TF_REQUIRE(fb && grey && r_out, "tf_fb_stage_level_polyexp: null pointer");
    TF_REQUIRE(level >= 0 && level <= fb->K, "tf_fb_stage_level_polyexp: level %d out of range", level);
    TF_TRY(tf_fb_set_frame(fb, 0, grey, stride));
    int2 pr = make_int2(0, 0);
    TF_HIP(hipMemcpy(fb->pairs.p, &pr, 8, hipMemcpyHostToDevice));
    Level &L = *fb->lv[level];
    if (fb_can_fuse_level(fb, level)) {
        TF_TRY(fb_level0_polyexp(fb, level, 2));
    } else if (fb_can_fuse_half_level(fb, level)) {
        TF_TRY(fb_level1_polyexp(fb, level, 2));
    } else {
        TF_TRY(fb_level_image(fb, level, 2, true));
        TF_TRY(fb_polyexp(fb, L.W, L.H, 2, level));
    }
    return download_rpairs(r_out, fb->Rk(level), (size_t)L.W * L.H, fb->scratch);
}

TF_API int tf_fb_stage_polyexp(tf_fb *fb, const float *img, int w, int h, float *r_out)
{
    TF_REQUIRE(fb && img && r_out, "tf_fb_stage_polyexp: null pointer");
    TF_TRY(check_stage_size(fb, w, h));
    TF_TRY(ensure_init());
    size_t n = (size_t)w * h;
    TF_HIP(hipMemcpyAsync(fb->img.p, img, n * 4, hipMemcpyHostToDevice, stream()));
    TF_TRY(fb_polyexp(fb, w, h, 1));
    return download_rpairs(r_out, fb->Rk(0), n, fb->scratch);
}

TF_API int tf_fb_stage_update_matrices(tf_fb *fb, const float *r0, const float *r1, const float *flow, int w, int h,
                                       float *m_out)
{
    TF_REQUIRE(fb && r0 && r1 && flow && m_out, "tf_fb_stage_update_matrices: null pointer");
    TF_TRY(check_stage_size(fb, w, h));
    TF_TRY(ensure_init());
    size_t n = (size_t)w * h;
    TF_TRY(upload_rpairs(fb->Rk(0), r0, n, fb->scratch));
    TF_HIP(hipStreamSynchronize(stream()));
    TF_TRY(upload_rpairs(fb->Rk(0) + 5 * n, r1, n, fb->scratch));
    TF_HIP(hipMemcpyAsync(fb->lflow[0].p, flow, n * 8, hipMemcpyHostToDevice, stream()));
    FlowInit fi;
    memset(&fi, 0, sizeof(fi));
    fi.mode = 2;
    fi.src = fb->lflow[0].as<float2>();
    TF_TRY(fb_update_matrices(fb, w, h, 1, fi));
    return download_planar5(m_out, fb->M.as<float>(), n, fb->scratch);
}

// A5 + A3 as the pyramid runs them at `level` (< K): the coarser level's flow is upsampled
// (resize INTER_LINEAR, x 1/pyr_scale) inside the kernel that builds the matrices.
TF_API int tf_fb_stage_upsampled_matrices(tf_fb *fb, int level, const float *r0, const float *r1, const float *coarse_flow,
                                          float *m_out)
{
    TF_REQUIRE(fb && r0 && r1 && coarse_flow && m_out, "tf_fb_stage_upsampled_matrices: null pointer");
    TF_REQUIRE(level >= 0 && level < fb->K, "tf_fb_stage_upsampled_matrices: level %d has no coarser level (K = %d)", level,
               fb->K);
    TF_TRY(ensure_init());
    Level &L = *fb->lv[level];
    Level &C = *fb->lv[level + 1];
    const size_t n = (size_t)L.W * L.H, nc = (size_t)C.W * C.H;
    TF_TRY(upload_rpairs(fb->Rk(0), r0, n, fb->scratch));
    TF_HIP(hipStreamSynchronize(stream()));
    TF_TRY(upload_rpairs(fb->Rk(0) + 5 * n, r1, n, fb->scratch));
    TF_HIP(hipMemcpyAsync(fb->lflow[1].p, coarse_flow, nc * 8, hipMemcpyHostToDevice, stream()));
    FlowInit fi;
    memset(&fi, 0, sizeof(fi));
    fi.mode = 1;
    fi.src = fb->lflow[1].as<float2>();
    fi.Wc = C.W;
    fi.Hc = C.H;
    fi.xofs = L.flow_lerp.xofs.as<int>();
    fi.yofs = L.flow_lerp.yofs.as<int>();
    fi.xfrac = L.flow_lerp.xfrac.as<float>();
    fi.yfrac = L.flow_lerp.yfrac.as<float>();
    fi.mul = (float)(1. / fb->prm.pyr_scale);
    TF_TRY(fb_update_matrices(fb, L.W, L.H, 1, fi));
    return download_planar5(m_out, fb->M.as<float>(), n, fb->scratch);
}

// OPTFLOW_USE_INITIAL_FLOW's first step alone: flow [H][W][2] -> the coarsest scale's starting flow
// [Hc][Wc][2] = resize(flow, INTER_AREA) * pyr_scale^K.
TF_API int tf_fb_stage_initial_flow(tf_fb *fb, const float *flow, float *coarse_out)
{
    TF_REQUIRE(fb && flow && coarse_out, "tf_fb_stage_initial_flow: null pointer");
    TF_TRY(tf_fb_set_initial_flow(fb, 0, flow));
    const Level &C = *fb->lv[fb->K];
    TF_TRY(fb_initial_flow(fb, 1, fb->lflow[0].as<float2>()));
    TF_HIP(hipMemcpyAsync(coarse_out, fb->lflow[0].p, (size_t)C.W * C.H * 8, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_fb_stage_blur_solve(tf_fb *fb, const float *m, int w, int h, float *flow_out)
{
    TF_REQUIRE(fb && m && flow_out, "tf_fb_stage_blur_solve: null pointer");
    TF_TRY(check_stage_size(fb, w, h));
    TF_TRY(ensure_init());
    size_t n = (size_t)w * h;
    TF_TRY(upload_planar5(fb->M.as<float>(), m, n, fb->scratch));
    if (fb->gaussian()) // FarnebackUpdateFlow_GaussianBlur's window on a handle created with flags & 256
        TF_TRY(fb_gauss_solve(fb, w, h, 1, fb->lflow[0].as<float2>()));
    else
        TF_TRY(fb_blur_solve(fb, w, h, 1, fb->lflow[0].as<float2>()));
    TF_HIP(hipMemcpyAsync(flow_out, fb->lflow[0].p, n * 8, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}
