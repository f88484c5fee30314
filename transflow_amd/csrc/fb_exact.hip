// A4 summed exactly as FarnebackUpdateFlow_Blur sums it -- the handle's exact mode (tf_fb_set_exact / option
// fb_exact_sums): flows bit-identical to the CPU path's -- and the winsize = 1 window, which OpenCV's priming turns into
// something else than a 1 x 1 sum (optflowgf.cpp; cv.py:479-490).
#include "fb_common.h"

namespace {

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

} // namespace

namespace tf {
namespace fb {

// room for the column sums of one level of the batch, [pair][5][y][x] doubles
int fb_exact_room(tf_fb *fb, int w, int h, int n_pairs)
{
    const size_t need = (size_t)n_pairs * 5 * w * h * sizeof(double);
    if (fb->exact_vsum.bytes < need && fb->exact_vsum.alloc(need) != TF_OK)
        return set_error(TF_ERR_HIP, "fb_exact_sums: no room for the column sums of %d pairs of %d x %d pixels (%zu bytes of doubles; "
                                     "fewer pairs per call need less)", n_pairs, w, h, need);
    return TF_OK;
}

// The row walker of option fb_exact_sums over fb->exact_vsum (k_exact_hsolve).
int fb_exact_hsolve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k)
{
    const int m = fb->prm.winsize / 2;
    const double scale = 1. / ((double)fb->prm.winsize * fb->prm.winsize);
    // three rows per wave (6: the same at 32 pairs of 4K, 202 against 165 us for one pair; 12: 3.7 x slower -- their
    // differences in flight take every register a lane has)
    return launch(lvl_name("fb_exact_hsolve", k), k_exact_hsolve<3>, dim3(cdiv(h, 3), 1, n_pairs), dim3(64), 0,
                  (const double *)fb->exact_vsum.as<double>(), flow_out, w, h, m, scale);
}

// M in memory (k_update_matrices): OpenCV's column sums of every row (k_exact_vsum's note), then the row walker
int fb_exact_from_matrices(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k)
{
    TF_TRY(fb_exact_room(fb, w, h, n_pairs));
    TF_TRY(launch(lvl_name("fb_exact_vsum", k), k_exact_vsum, dim3(cdiv(w, 64), 5, n_pairs), dim3(64), 0,
                  (const float *)fb->M.as<float>(), fb->exact_vsum.as<double>(), w, h, fb->prm.winsize / 2));
    return fb_exact_hsolve(fb, w, h, n_pairs, flow_out, k);
}

int fb_w1_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out)
{
    const size_t need = (size_t)n_pairs * 5 * w * h * sizeof(double);
    if (fb->exact_vsum.bytes < need)
        TF_TRY(fb->exact_vsum.alloc(need));
    TF_TRY(launch("fb_w1_vsum", k_w1_vsum, dim3(cdiv(w, 64), 5, n_pairs), dim3(64), 0, (const float *)fb->M.as<float>(),
                  fb->exact_vsum.as<double>(), w, h));
    return launch("fb_w1_solve", k_w1_solve, dim3(cdiv(w, 256), h, n_pairs), dim3(256), 0,
                  (const double *)fb->exact_vsum.as<double>(), flow_out, w, h);
}

} // namespace fb
} // namespace tf
