// A1 of the Farneback path: the level images (pre-blur + resize), DESIGN.md section 3 -- the general tile kernel, the two-kernel
// form for long blur kernels (row pass over whole frame rows into a plane, column pass + lerps from it) and the one-kernel
// form for the scale-1/4 level (imgproc's GaussianBlur + resize as calcOpticalFlowFarneback calls them, cv.py:479-490).
#include "fb_common.h"

namespace {

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

// One level's row pass for a kernel length known at compile time (round 5).  The general loop below spends two thirds
// of its instructions outside the taps -- an item's set-up (four table loads, two divisions), the peeled first four
// taps, the up-to-three left over, the register moves that slide the window -- and at ksz = 9 an item is one trip
// through the loop.  Here the taps are unrolled (the window's bytes are converted once each, straight from the
// aligned dwords; no moves, no remainder), the set-up is done once per column group and reused for every row block of
// the workgroup, and the item order needs no division.  Same lanes (R rows x 64/R groups), same LDS reads, the same
// statements per output: acc = t0 * w0, then acc += t_i * w_i left to right.
// threads of a row-pass workgroup (512 with registers capped at 64 for eight waves per SIMD: 1.80 -> 3.13 ms, spills and idle waves)
constexpr int RP_THREADS = 256, RP_WAVES = RP_THREADS / 64;

template <int KSZ, int RSHIFT>
__device__ __forceinline__ void rowpass_level_t(const RowPassLevel &L, const uint8_t *sS, const float *sK, int nrows, int pitch,
                                                int r4, size_t out_row0, int lane, int wave)
{
    constexpr int R = 1 << RSHIFT, G = 64 >> RSHIFT, r = KSZ >> 1;
    constexpr int NB = KSZ + 1;        // bytes of a window: outputs at columns c and c + 1
    constexpr int NW = (NB + 3) / 4;   // aligned words of it
    const int NC = L.NC;
    const int li = lane & (R - 1), lg = lane >> RSHIFT;
    const int ngroups = (NC + 3) >> 2;
    const int n_gb = (ngroups + G - 1) / G, n_rb = (nrows + R - 1) >> RSHIFT;
    for (int gb = wave; gb < n_gb; gb += RP_WAVES) {
        const int grp = gb * G + lg;
        if (grp >= ngroups)
            continue;
        const int oA = 4 * grp, oB = min(4 * grp + 2, NC - 2);
        const int cA = L.colsrc[oA], cB = L.colsrc[oB];
        const bool dupA = L.colsrc[oA + 1] == cA, dupB = L.colsrc[oB + 1] == cB; // pair at the right frame border: sx+1 clamps to sx
        const int a0 = r4 + cA - r, b0 = r4 + cB - r;
        const int da = a0 >> 2, db = b0 >> 2;
        const unsigned sa = a0 & 3, sb = b0 & 3;
        const bool four = 4 * grp + 2 < NC;
        for (int rb = 0; rb < n_rb; rb++) {
            const int row = rb * R + li;
            if (row >= nrows)
                continue;
            const uint32_t *q32 = reinterpret_cast<const uint32_t *>(sS + row * pitch);
            uint32_t rA[NW + 1], rB[NW + 1];
#pragma unroll
            for (int j = 0; j <= NW; j++) {
                rA[j] = q32[da + j];
                rB[j] = q32[db + j];
            }
            f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}; // {A, B} and {A+1, B+1}
#pragma unroll
            for (int wv = 0; wv < NW; wv++) {
                const uint32_t wA = __builtin_amdgcn_alignbyte(rA[wv + 1], rA[wv], sa);
                const uint32_t wB = __builtin_amdgcn_alignbyte(rB[wv + 1], rB[wv], sb);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int i = 4 * wv + b; // byte i of the window: tap i of the left output, tap i - 1 of the right one
                    if (i < NB) {
                        const f32x2 p = {(float)((wA >> (8 * b)) & 0xff), (float)((wB >> (8 * b)) & 0xff)};
                        if (i < KSZ) {
                            if (i == 0)
                                acc0 = sK[0] * p;
                            else
                                acc0 += sK[i] * p;
                        }
                        if (i >= 1) {
                            if (i == 1)
                                acc1 = sK[0] * p;
                            else
                                acc1 += sK[i - 1] * p;
                        }
                    }
                }
            }
            // a clamped pair reads the same column twice: the same sum
            if (dupA)
                acc1.x = acc0.x;
            if (dupB)
                acc1.y = acc0.y;
            float *out = L.rowf + (out_row0 + row) * NC + 4 * grp;
            out[0] = acc0.x; // (one 16-byte store per lane instead of four: 1.80 -> 1.89 ms)
            out[1] = acc1.x;
            if (four) {
                out[2] = acc0.y;
                out[3] = acc1.y;
            }
        }
    }
}

__global__ void __launch_bounds__(RP_THREADS)
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
            for (int i = threadIdx.x; i < args.lv[l].ksz; i += RP_THREADS)
                sKall[base + i] = args.lv[l].kern[i];
            base += args.lv[l].ksz;
        }
    }
    // stage nrows frame rows: the interior as dwords (W % 4 == 0 is required by the host), rmax reflected bytes each side
    const int nq = W >> 2;
    {
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
        constexpr int U = 4;
        for (int i = wave_; i < nrows; i += RP_WAVES) { // a wave copies whole rows: 256 contiguous bytes per instruction, U in flight
            const uint32_t *g = reinterpret_cast<const uint32_t *>(src + (size_t)(y0 + i) * W);
            uint32_t *d = reinterpret_cast<uint32_t *>(sS + i * pitch + r4);
            for (int c0 = lane_; c0 < nq; c0 += 64 * U) {
                uint32_t v[U];
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (c0 + 64 * u < nq)
                        v[u] = g[c0 + 64 * u];
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (c0 + 64 * u < nq)
                        d[c0 + 64 * u] = v[u];
            }
        }
    }
    for (int idx = threadIdx.x; idx < nrows * 2 * rmax; idx += RP_THREADS) {
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
        const size_t out_row0 = (size_t)pi * H + y0;
        // the kernel lengths of a pyr_scale = 0.5 pyramid (scales 1/4 .. 1/32) with their lane mappings
        if (L.ksz == 9 && L.rshift == 1) {
            rowpass_level_t<9, 1>(L, sS, sK, nrows, pitch, r4, out_row0, lane, wave);
            continue;
        }
        if (L.ksz == 19 && L.rshift == 2) {
            rowpass_level_t<19, 2>(L, sS, sK, nrows, pitch, r4, out_row0, lane, wave);
            continue;
        }
        if (L.ksz == 39 && L.rshift == 3) {
            rowpass_level_t<39, 3>(L, sS, sK, nrows, pitch, r4, out_row0, lane, wave);
            continue;
        }
        if (L.ksz == 79 && L.rshift == 3) {
            rowpass_level_t<79, 3>(L, sS, sK, nrows, pitch, r4, out_row0, lane, wave);
            continue;
        }
        const int ksz = L.ksz, r = ksz >> 1, NC = L.NC, rshift = L.rshift;
        const int R = 1 << rshift, G = 64 >> rshift;
        const int li = lane & (R - 1), lg = lane >> rshift;
        const int ngroups = (NC + 3) >> 2;
        const int n_rb = (nrows + R - 1) >> rshift, n_gb = (ngroups + G - 1) / G;
        for (int item = wave; item < n_rb * n_gb; item += RP_WAVES) {
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
// A1 for a level that is exactly a QUARTER of the frame in both directions, whose blur has 9 taps (scale 1/4 of a
// pyr_scale = 0.5 pyramid: sigma 1.5), in one kernel and without the row-pass plane (round 5).  As two kernels this level
// alone writes and reads 2 x 16.6 MB of plane per 4K frame -- 4.3 GB per pass of 129 frames, moved at 1.8 - 2.5 TB/s:
// 0.73 ms of the row pass and the 0.97 ms of its column pass are that traffic.  resize.cpp's coordinates are
// (4X + 1.5, 4Y + 1.5): a level pixel is the lerp (both fractions exactly 0.5; the host checks its tables) of the blurred
// frame at columns 4X + 1, 4X + 2 and rows 4Y + 1, 4Y + 2, so it needs the row pass at those two columns on frame rows
// 4Y - 3 .. 4Y + 6, and the pixel below it needs six of the same ten rows.  A lane owns a level column and walks down
// eight level rows with the row-pass values of ten frame rows in registers: four new frame rows per level row (two at
// a time, packed), then the column pass (centre, then pairs outwards) and both lerps -- k_level_image's statements in
// k_level_image's order.  Tile 64 x 32 level pixels = 264 x 134 staged bytes; a wave per eight level rows.
// ---------------------------------------------------------------------------------
#ifndef TF_QI_TH
#define TF_QI_TH 16
#endif
constexpr int QI_TW = 64, QI_TH = TF_QI_TH;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_level_quarter_image(const uint8_t *__restrict__ frames, const int2 *__restrict__ pairs, float *__restrict__ img, int W, int H,
                      const float *__restrict__ kern, const float *__restrict__ xfrac, const float *__restrict__ yfrac)
{
    constexpr int S = 4, K = 9, r = 4, SEG = QI_TH / 4;
    constexpr int NR = S * (QI_TH - 1) + 2 + 2 * r; // 134 frame rows behind 32 level rows
    constexpr int PD = QI_TW + 2;                   // dwords per staged row: lane l reads dwords l .. l + 2
    __shared__ uint32_t sS[NR * PD];
    const int Wk = W >> 2, Hk = H >> 2;
    // neighbouring tiles share six frame rows and the cache lines at their sides: one after the other on the same XCD
    unsigned tbx, tby, tbz;
    xcd_tile3(tbx, tby, tbz);
    const int pi = tbz;
    const int2 pr = pairs[pi >> 1];
    const uint8_t *src = frames + (size_t)((pi & 1) ? pr.y : pr.x) * W * H;
    const int X0 = tbx * QI_TW, Y0 = tby * QI_TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // staged row j = frame row 4 Y0 - 3 + j; staged byte b = frame column 4 X0 - 4 + b (REFLECT_101 outside the frame):
    // level column X0 + l reads bytes 4 l + 1 .. 4 l + 10
    const int xs = S * X0 - 4, ys = S * Y0 - 3;
    const bool dwords = xs >= 0 && xs + 4 * PD <= W && ys >= 0 && ys + NR <= H;
    if (dwords) {
        constexpr int U = (NR + 3) / 4; // all of a wave's rows in flight at once: one round trip to memory per tile
        for (int j0 = wave; j0 < NR; j0 += 4 * U) {
            uint32_t v[U], v2[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = min(j0 + 4 * u, NR - 1);
                const uint32_t *g = reinterpret_cast<const uint32_t *>(src + (size_t)(ys + j) * W + xs);
                v[u] = g[lane];
                v2[u] = g[64 + (lane & 1)];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + 4 * u;
                if (j < NR) {
                    sS[j * PD + lane] = v[u];
                    if (lane < 2)
                        sS[j * PD + 64 + lane] = v2[u];
                }
            }
        }
    } else {
        uint8_t *s8 = reinterpret_cast<uint8_t *>(sS);
        for (int j = wave; j < NR; j += 4) {
            const uint8_t *g = src + (size_t)reflect101(ys + j, H) * W;
            for (int b = lane; b < 4 * PD; b += 64)
                s8[j * 4 * PD + b] = g[reflect101(xs + b, W)];
        }
    }
    __syncthreads();
    const int X = X0 + lane, Yw = Y0 + SEG * wave;
    if (X >= Wk || Yw >= Hk)
        return;
    float t[K];
#pragma unroll
    for (int i = 0; i < K; i++)
        t[i] = kern[i];
    const float fx = xfrac[X];
    const uint32_t *q = sS + (S * SEG * wave) * PD + lane; // the wave's first staged row
    constexpr int NROWS = S * (SEG - 1) + 2 + 2 * r;       // 38 frame rows behind a wave's eight level rows
    float c0[NROWS], c1[NROWS];                            // row pass at columns 4X + 1 and 4X + 2
    // the row pass of staged rows j and j + 1 (of the wave), the two rows in the halves of packed operations
    auto row_pair = [&](int j) {
        const uint32_t *qa = q + j * PD, *qb = qa + PD;
        const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
        const uint32_t wa[3] = {__builtin_amdgcn_alignbyte(a1, a0, 1), __builtin_amdgcn_alignbyte(a2, a1, 1), a2 >> 8};
        const uint32_t wb[3] = {__builtin_amdgcn_alignbyte(b1, b0, 1), __builtin_amdgcn_alignbyte(b2, b1, 1), b2 >> 8};
        f32x2 p[K + 1];
#pragma unroll
        for (int i = 0; i <= K; i++)
            p[i] = f32x2{(float)((wa[i >> 2] >> (8 * (i & 3))) & 0xff), (float)((wb[i >> 2] >> (8 * (i & 3))) & 0xff)};
        f32x2 A0 = t[0] * p[0], A1 = t[0] * p[1];
#pragma unroll
        for (int i = 1; i < K; i++) {
            A0 += t[i] * p[i];
            A1 += t[i] * p[i + 1];
        }
        c0[j] = A0.x;
        c0[j + 1] = A0.y;
        c1[j] = A1.x;
        c1[j + 1] = A1.y;
    };
#pragma unroll
    for (int j = 0; j < 6; j += 2)
        row_pair(j);
    float *dst = img + (size_t)pi * Wk * Hk + X;
#pragma unroll
    for (int y = 0; y < SEG; y++) {
        if (Yw + y >= Hk)
            break;
        row_pair(S * y + 6);
        row_pair(S * y + 8);
        // rows 4y .. 4y + 9 of the wave are frame rows 4Y - 3 .. 4Y + 6: the pixel's source rows are 4y + 4 and 4y + 5
        const int m = S * y + r;
        float v00 = t[r] * c0[m], v01 = t[r] * c1[m], v10 = t[r] * c0[m + 1], v11 = t[r] * c1[m + 1];
#pragma unroll
        for (int i = 1; i <= r; i++) {
            const float k = t[r + i];
            v00 += k * (c0[m + i] + c0[m - i]);
            v01 += k * (c1[m + i] + c1[m - i]);
            v10 += k * (c0[m + 1 + i] + c0[m + 1 - i]);
            v11 += k * (c1[m + 1 + i] + c1[m + 1 - i]);
        }
        const float fy = yfrac[Yw + y];
        const float h0 = v00 * (1.f - fx) + v01 * fx, h1 = v10 * (1.f - fx) + v11 * fx;
        dst[(size_t)(Yw + y) * Wk] = h0 * (1.f - fy) + h1 * fy;
    }
}

} // namespace

namespace tf {
namespace fb {

// `standalone`: a single level is wanted (stage entry points): run the shared row pass regardless of the order
int fb_level_image(tf_fb *fb, int k, int n_images, bool standalone)
{
    Level &L = *fb->lv[k];
    if (L.quarter)
        return launch(lvl_name("fb_level_image", k), k_level_quarter_image, dim3(cdiv(L.W, QI_TW), cdiv(L.H, QI_TH), n_images),
                      dim3(256), 0, (const uint8_t *)fb->frames.as<uint8_t>(), fb->image_list(), fb->imgk(k), fb->W, fb->H,
                      (const float *)L.kern.as<float>(), (const float *)L.img_lerp.xfrac.as<float>(),
                      (const float *)L.img_lerp.yfrac.as<float>());
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
                          dim3(RP_THREADS), smem_rp, (const uint8_t *)fb->frames.as<uint8_t>(),
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

// Does k_level_quarter_image apply?  The level is the frame / 4 in both directions with the 9-tap blur, and resize.cpp's
// tables say what the kernel assumes: source column 4X + 1 (row 4Y + 1), fraction 0.5, nothing clamped.
bool plan_quarter_level(int W, int H, const Level &L)
{
    static const bool off = tune("TF_IMG_NO_QUARTER", 0) != 0;
    if (off || L.ksz != 9 || 4 * L.W != W || 4 * L.H != H)
        return false;
    std::vector<int> o;
    std::vector<float> f;
    make_lerp(W, L.W, true, o, f);
    for (int x = 0; x < L.W; x++)
        if (o[x] != 4 * x + 1 || f[x] != 0.5f)
            return false;
    make_lerp(H, L.H, false, o, f);
    for (int y = 0; y < L.H; y++)
        if (o[y] != 4 * y + 1 || f[y] != 0.5f)
            return false;
    return true;
}

// Plans the two-kernel form of A1 for a level with a long blur kernel (returns false where it does not
// apply: short kernels, frame widths that are not a multiple of 4, frames too wide to stage 8 rows).
bool plan_split_level(int W, int H, Level &L, std::vector<int> &colsrc)
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
ImgTile choose_tile(int W, int H, int Wk, int Hk, int ksz, int level)
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

} // namespace fb
} // namespace tf
