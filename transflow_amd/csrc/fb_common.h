// Shared by the translation units of the Farneback path (fb_*.hip, farneback.hip): the device helpers several kernels
// use, the layout of R, the per-level and per-handle host state, and the stage launchers one unit offers the others.
//   fb_pyramid.hip      A1 + A2: level images, polynomial expansion (and their fused forms)
//   fb_matrices.hip     A3 / A4 as two kernels: update-matrices, the marching blur + solve, Gaussian window, INTER_AREA init
//   fb_iterate.hip      A3 + A4 (+ A5) as ONE kernel: k_flow_iter_pc, its pre-pass, the march planner
//   fb_exact.hip        A4 in OpenCV's own summation order (tf_fb_set_exact / option fb_exact_sums), winsize 1
//   fb_postprocess.hip  B1: FlowSource.post_process and the flow filters
//   fb_stages.hip       single-stage entry points for the parity tests
//   farneback.hip       constants, the handle, the pyramid driver tf_fb_calc_slots, frames in / flows out
#pragma once
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

#include <mutex>

struct tf_fb;

namespace tf {
namespace fb {

constexpr int MAX_POLY_N = 15;

// two floats at 4-byte alignment: one global_load_dwordx2 (the hardware takes unaligned dwordx2)
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

// ---------------------------------------------------------------------------------
// Layout of R, the polynomial coefficients of one image at one level (Nk pixels, 5 Nk floats, OpenCV's channel
// order): two planes -- (c0, c1, c2, c3) quads [Nk][4] and c4 [Nk].  FarnebackUpdateMatrices pairs the channels
// ((c0, c1) feed h, (c2, c3) the diagonal of G, c4 its off-diagonal), and the halves of a quad are those pairs: a
// pixel's own coefficients arrive as ONE 16-byte load and a 4-byte one, the two taps of a bilinear row as two
// 16-byte loads (x1 and x1 + 1 are adjacent pixels) and an 8-byte one, and the arithmetic runs on register pairs as
// they were loaded: 9 loads per pixel (rounds 2 - 4 kept (c0, c1) and (c2, c3) in planes of their own: 10 loads; the
// producers of the one-kernel iteration are bound by the instructions they issue, and the load less bought 0.6 % of
// the whole step in same-box runs, round 5).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ size_t r_off4(size_t Nk) { return 4 * Nk; }

__device__ __forceinline__ void r_load_px(const float *__restrict__ R, size_t Nk, size_t o, float v[5])
{
    const float4u a = *reinterpret_cast<const float4u *>(R + 4 * o);
    v[0] = a.x;
    v[1] = a.y;
    v[2] = a.z;
    v[3] = a.w;
    v[4] = R[r_off4(Nk) + o];
}
__device__ __forceinline__ void r_load_taps(const float *__restrict__ R, size_t Nk, size_t q, int Wk, float2u t[5], float2u b[5])
{
    const float4u t0 = *reinterpret_cast<const float4u *>(R + 4 * q), t1 = *reinterpret_cast<const float4u *>(R + 4 * q + 4);
    const float4u b0 = *reinterpret_cast<const float4u *>(R + 4 * (q + Wk)), b1 = *reinterpret_cast<const float4u *>(R + 4 * (q + Wk) + 4);
    const float2u t4 = *reinterpret_cast<const float2u *>(R + r_off4(Nk) + q), b4 = *reinterpret_cast<const float2u *>(R + r_off4(Nk) + q + Wk);
    t[0] = float2u{t0.x, t1.x};
    t[1] = float2u{t0.y, t1.y};
    t[2] = float2u{t0.z, t1.z};
    t[3] = float2u{t0.w, t1.w};
    t[4] = t4;
    b[0] = float2u{b0.x, b1.x};
    b[1] = float2u{b0.y, b1.y};
    b[2] = float2u{b0.z, b1.z};
    b[3] = float2u{b0.w, b1.w};
    b[4] = b4;
}
__device__ __forceinline__ void r_store_px(float *__restrict__ R, size_t Nk, size_t o, const float v[5])
{
    *reinterpret_cast<float4u *>(R + 4 * o) = float4u{v[0], v[1], v[2], v[3]};
    R[r_off4(Nk) + o] = v[4];
}
__device__ __forceinline__ void r_store_px2(float *__restrict__ R, size_t Nk, size_t o, const float v0[5], const float v1[5])
{
    *reinterpret_cast<float4u *>(R + 4 * o) = float4u{v0[0], v0[1], v0[2], v0[3]};
    *reinterpret_cast<float4u *>(R + 4 * o + 4) = float4u{v1[0], v1[1], v1[2], v1[3]};
    *reinterpret_cast<float2u *>(R + r_off4(Nk) + o) = float2u{v0[4], v1[4]};
}
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

// The same over a whole 3-D grid (x fastest, then y, then z = image): the hardware deals workgroups to the XCDs by their
// place in the whole launch, so with gridDim.x * gridDim.y not a multiple of 8 the per-slice form above groups the
// wrong blocks from the second image on.  Each XCD walks a contiguous eighth of the launch in (z, y, x) order.
__device__ __forceinline__ void xcd_tile3(unsigned &bx, unsigned &by, unsigned &bz)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, nt = gx * gy * gridDim.z;
    const unsigned lin = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const unsigned xcd = lin & 7, qn = nt >> 3, rn = nt & 7;
    const unsigned t = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (lin >> 3);
    bz = t / (gx * gy);
    const unsigned rem = t - bz * gx * gy;
    by = rem / gx;
    bx = rem - by * gx;
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

struct ImgTile {
    int TWo, THo;           // output tile (TWo a power of two)
    int tw_shift;           // log2(TWo)
    int rstride;            // floats per staged row of the row-pass buffer: 2*TWo, or TWo for a copy-sized level
    int LW, LH;             // source columns / rows staged per tile (upper bounds)
    int pitch;              // bytes per staged source row, multiple of 4 with pitch/4 odd
    int same_size;          // level size == frame size: resize is a copy
    double scale_x, scale_y; // resize.cpp's 1/(dst/src) per axis
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RP_MAX_LEVELS = 4; // split levels one k_level_rowpass launch serves (fb_pyramid.hip)

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

// resize(INTER_AREA) tables of OPTFLOW_USE_INITIAL_FLOW (k_flow_area_init, fb_matrices.hip)
struct AreaTabs {
    const int *xsi, *xstart; // x entries: source column; first entry of every destination column (Wc + 1)
    const float *xalpha;
    const int *ysi, *ystart;
    const float *yalpha;
    int ix, iy;              // > 0: the integer-factor path
};

// resize.cpp's INTER_LINEAR coefficient tables for one axis (farneback.hip)
void make_lerp(int src, int dst, bool zero_at_edges, std::vector<int> &ofs, std::vector<float> &frac);

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
    bool quarter = false;   // exactly the frame / 4 with the 9-tap blur: one kernel, no plane (k_level_quarter_image)
    bool split = false;
    DevBuf colsrc;          // source column of each of the NC = 2*W row-pass columns
    int NC = 0, rp_rshift = 0;
    size_t rowf_off = 0;    // this level's plane inside tf_fb::rowf (floats)
    LerpDev flow_lerp; // level k+1 -> this level
};

} // namespace fb
} // namespace tf

using namespace tf;
using namespace tf::fb;

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

namespace tf {
namespace fb {

// The exact mode is a property of the HANDLE (tf_fb_set_exact); a handle that was never told follows the process-wide
// option, read at each call.
inline bool fb_exact(const tf_fb *fb) { return fb->exact >= 0 ? fb->exact != 0 : option(OPT_FB_EXACT_SUMS) != 0; }

// How a marching launch is cut into row segments, and where the segments get their column sums' carries from
// (0: a pre-pass, 1: handed down inside the launch; fb_iterate.hip: choose_march)
struct March {
    int mode, segs, seg; // seg: rows per segment
};

// Profiler labels: with option "prof_levels" = 1 every Farneback launch is labelled with its pyramid level (farneback.hip)
const char *lvl_name(const char *base, int k);

// ---- fb_pyramid.hip ----
int fb_level_image(tf_fb *fb, int k, int n_images, bool standalone = false);
bool plan_split_level(int W, int H, Level &L, std::vector<int> &colsrc);
bool plan_quarter_level(int W, int H, const Level &L);
ImgTile choose_tile(int W, int H, int Wk, int Hk, int ksz, int level);
int fb_polyexp(tf_fb *fb, int w, int h, int n_images, int k = -1);
bool fb_can_fuse_level(tf_fb *fb, int k);
bool fb_can_fuse_half_level(tf_fb *fb, int k);
int fb_level1_polyexp(tf_fb *fb, int k, int n_images);
int fb_level0_polyexp(tf_fb *fb, int k, int n_images);
// ---- fb_matrices.hip ----
int fb_m_room(tf_fb *fb, int w, int h, int n_pairs);
int fb_update_matrices(tf_fb *fb, int w, int h, int n_pairs, const FlowInit &fi, int k = -1);
int fb_carry_room(tf_fb *fb, size_t carry_doubles, size_t flags);
int fb_check_fault(tf_fb *fb, const char *where);
int fb_carry_scan(tf_fb *fb, int w, int n_pairs, int segs, int k);
int fb_blur_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k = -1);
int fb_gauss_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k = -1);
int fb_setup_flags(tf_fb *fb);
int fb_initial_flow(tf_fb *fb, int n_pairs, float2 *out);
// ---- fb_exact.hip ----
int fb_exact_room(tf_fb *fb, int w, int h, int n_pairs);                               // fb->exact_vsum for one level of the batch
int fb_exact_from_matrices(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k); // M in memory -> column sums -> rows + solve
int fb_exact_hsolve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out, int k);     // rows + solve over fb->exact_vsum
int fb_w1_solve(tf_fb *fb, int w, int h, int n_pairs, float2 *flow_out);                // winsize 1
// ---- fb_iterate.hip ----
// true if the one-kernel iteration exists for this window; launches it.  `up`: the first iteration of a level below the
// coarsest takes its flow from the coarser level (A5 fused)
bool fb_flow_iter(tf_fb *fb, int w, int h, int n_pairs, const float2 *flow_in, float2 *flow_out, int k, int &rc,
                  const FlowInit *up = nullptr);

} // namespace fb
} // namespace tf
