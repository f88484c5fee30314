// Compositor remap on gfx950: the `moveref` layer of transflow as HIP kernels.
//
// Reference semantics (paths relative to the reference tree):
//   MovementLayer._update_flow/_update_move   transflow/compositor/layers/movement.py:20-60
//   ReferenceLayer._update_reset_*            transflow/compositor/layers/reference.py:58-91
//   ReferenceLayer._update_rgba               transflow/compositor/layers/reference.py:93-105
//   Layer.render / Compositor.render          layers/layer.py:32-34, compositor/compositor.py:31-40
//
// HBM layout: `data` is int4 per pixel (i_ref, j_ref, alpha, source), two buffers
// (the move is a gather from a snapshot: movement.py:51-52); `rgba` uchar4 per
// pixel; masks one byte / one float per pixel, absent (nullptr) when default.
// All kernels are bandwidth-bound, one pixel per lane, 16-byte accesses on `data`.
#include <cstring>
#include <type_traits>

#include "common.h"

using namespace tf;

namespace {

constexpr int BLOCK = 256;

struct MoveFlags {
    int transparent_can_move;
    int to_empty;
    int to_filled;
    int leave_empty;
};

// d = rint(fy)*W + rint(fx): numpy.round is half-to-even, like v_rndne_f32 (movement.py:22-23)
__device__ __forceinline__ long long flow_offset(float2 f, int W)
{
    int fx = (int)rintf(f.x), fy = (int)rintf(f.y);
    return (long long)fy * W + fx;
}

// Membership of target t in the move set T (movement.py:27-48) given the snapshot.
// Returns true and the source pixel's record when t receives a pixel.
__device__ __forceinline__ bool move_target(long long t, const float2 *__restrict__ flow, const int4 *__restrict__ old,
                                            const uint8_t *__restrict__ msrc, const uint8_t *__restrict__ mdst, int N,
                                            int W, MoveFlags fl, int4 me, int4 &src_rec, long long &s, bool &oob)
{
    long long d = flow_offset(flow[t], W);
    oob = false;
    if (d == 0)
        return false;
    s = t + d;
    if (s < 0 || s >= N) {
        oob = true;
        return false;
    }
    src_rec = old[s];
    bool src_filled = src_rec.z != 0;
    bool ms = (msrc ? msrc[s] != 0 : true) && (fl.transparent_can_move || src_filled);
    bool md = (mdst ? mdst[t] != 0 : true) && (fl.to_empty || me.z != 0) && (fl.to_filled || me.z == 0);
    return ms && md;
}

__global__ void k_remap_init(int4 *data, int N, int W)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    data[t] = make_int4(t / W, t % W, 1, 0); // reference.py:40-41
}

// reference.py:46-52: for s in order: data.source[mask_s] = s  (highest set mask wins)
__global__ void k_remap_set_sources(int4 *data, const uint8_t *masks, int n_sources, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int src = -1;
    for (int s = 0; s < n_sources; s++)
        if (masks[(size_t)s * N + t])
            src = s;
    if (src >= 0)
        data[t].w = src;
}

// movement.py:25-60 as a per-target gather from the snapshot `old` into `neu`.
__global__ void k_remap_move(const float2 *__restrict__ flow, const int4 *__restrict__ old, int4 *__restrict__ neu,
                             const uint8_t *__restrict__ msrc, const uint8_t *__restrict__ mdst, int N, int W,
                             MoveFlags fl, int *err)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int4 me = old[t];
    int4 out = me, so;
    long long s;
    bool oob;
    if (move_target(t, flow, old, msrc, mdst, N, W, fl, me, so, s, oob)) {
        out = so;                                  // putn(..., DEPTH)            :51-52
        if (!fl.transparent_can_move || so.z != 0) // alpha[target] = 1           :55-60
            out.z = 1;
    }
    if (oob)
        atomicOr(err, 1);
    neu[t] = out;
}

// moving_pixels_leave_empty_spot (movement.py:53-54): alpha[source] = 0 for every
// moved pixel, applied BEFORE alpha[target] = 1, so a source that is itself a
// filled target stays 1.  Every writer stores the same value: benign race.
__global__ void k_remap_leave_empty(const float2 *__restrict__ flow, const int4 *__restrict__ old,
                                    int4 *__restrict__ neu, const uint8_t *__restrict__ msrc,
                                    const uint8_t *__restrict__ mdst, int N, int W, MoveFlags fl)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int4 so, so2;
    long long s, s2;
    bool oob;
    if (!move_target(t, flow, old, msrc, mdst, N, W, fl, old[t], so, s, oob))
        return;
    // is the source pixel s itself a target whose alpha is forced to 1?
    bool s_is_one = move_target(s, flow, old, msrc, mdst, N, W, fl, old[s], so2, s2, oob) &&
                    (!fl.transparent_can_move || so2.z != 0);
    if (!s_is_one)
        neu[s].z = 0;
}

// ---- Philox2x32-10 (counter-based, Salmon et al. 2011: Random123): one 2x32 block per pixel and frame
// gives the 64 bits a float64 uniform needs.  Counter = (pixel, frame), key = the seed's halves mixed.
// (The 4x32 form costs twice the multiplies for 128 bits of which 64 went unused; this kernel is bound
// by its instructions, not by HBM.)
__device__ __forceinline__ double philox_uniform(uint32_t pixel, uint64_t frame, uint64_t seed)
{
    const uint32_t M = 0xD256D193u;
    uint32_t c0 = pixel, c1 = (uint32_t)frame ^ ((uint32_t)(frame >> 32) * 0x85EBCA6Bu);
    uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi = __umulhi(M, c0), lo = M * c0;
        c0 = hi ^ k ^ c1;
        c1 = lo;
        k += 0x9E3779B9u;
    }
    // 53-bit mantissa, as numpy's random_sample builds it from two 32-bit draws
    return ((double)(c0 >> 5) * 67108864.0 + (double)(c1 >> 6)) * (1.0 / 9007199254740992.0);
}

// reference.py:58-67.  thr = float32(factor) * reset_mask (numpy: python scalar times
// float32 array stays float32); the comparison is in float64 (u is float64).
__global__ void k_remap_reset_random(int4 *data, const double *__restrict__ u, const float *__restrict__ reset_mask,
                                     float factor, int reset_source, const uint8_t *__restrict__ intro, int n_sources,
                                     int N, int W, uint64_t seed, uint64_t frame)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    float thr = reset_mask ? factor * reset_mask[t] : factor;
    double uu = u ? u[t] : philox_uniform((uint32_t)t, frame, seed);
    if (!(uu < (double)thr))
        return;
    int4 d = data[t];
    d.x = t / W;
    d.y = t % W;
    d.z = 1;
    if (reset_source)
        for (int s = 0; s < n_sources; s++)
            if (intro[(size_t)s * N + t])
                d.w = s;
    data[t] = d;
}

// reference.py:69-79, float32 throughout (dij_base.astype(float32)).
__global__ void k_remap_reset_constant(int4 *data, const float *__restrict__ reset_mask, float step, int N, int W)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int4 d = data[t];
    float bi = (float)(t / W - d.x), bj = (float)(t % W - d.y);
    float di = bi, dj = bj;
    float norm_base = fmaxf(fabsf(di), fabsf(dj));
    if (norm_base != 0.f) {
        di = di / norm_base;
        dj = dj / norm_base;
    }
    float k = reset_mask ? step * reset_mask[t] : step;
    di *= k;
    dj *= k;
    float norm_scaled = fmaxf(fabsf(di), fabsf(dj));
    if (norm_scaled > norm_base) {
        di = bi;
        dj = bj;
    }
    d.x += (int)rintf(di);
    d.y += (int)rintf(dj);
    data[t] = d;
}

// reference.py:81-83: python float * int32 array is float64, times the float32 mask -> float64.
__global__ void k_remap_reset_linear(int4 *data, const float *__restrict__ reset_mask, double factor, int N, int W)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int4 d = data[t];
    double di = factor * (double)(t / W - d.x), dj = factor * (double)(t % W - d.y);
    double m = reset_mask ? (double)reset_mask[t] : 1.0;
    d.x += (int)rint(m * di);
    d.y += (int)rint(m * dj);
    data[t] = d;
}

// reference.py:94-105 for one source.
template <int C>
__global__ void k_remap_gather(const int4 *__restrict__ data, uchar4 *__restrict__ rgba,
                               const uint8_t *__restrict__ pixmap, int source_index, int N, int H, int W)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int4 d = data[t];
    bool sel = d.w == source_index && d.z != 0;
    if (C == 4) {
        if (sel) {
            int i = min(max(d.x, 0), H - 1), j = min(max(d.y, 0), W - 1);
            rgba[t] = reinterpret_cast<const uchar4 *>(pixmap)[(size_t)i * W + j];
        }
    } else {
        uchar4 px = rgba[t];
        if (sel) {
            int i = min(max(d.x, 0), H - 1), j = min(max(d.y, 0), W - 1);
            const uint8_t *p = pixmap + ((size_t)i * W + j) * 3;
            px.x = p[0];
            px.y = p[1];
            px.z = p[2];
        }
        px.w = sel ? 1 : 0; // rgba[:,:,3] = 0 everywhere, 1 on the selection  :103-105
        rgba[t] = px;
    }
}

// layer.py:32-34 then compositor.py:36-39.
__global__ void k_remap_render(uchar4 *__restrict__ rgba, const float *__restrict__ mask_alpha,
                               uint8_t *__restrict__ image, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    uchar4 px = rgba[t];
    if (mask_alpha) {
        unsigned char a = (unsigned char)(int)(mask_alpha[t] * (float)px.w);
        if (a != px.w) {
            px.w = a;
            rgba[t].w = a;
        }
    }
    if (px.w != 0) {
        uint8_t *o = image + (size_t)t * 3;
        o[0] = px.x;
        o[1] = px.y;
        o[2] = px.z;
    }
}

// The resident path's single launch per frame: [clip of post_process] + move + random reset +
// gather of source 0 + Layer.render + Compositor.render for a one-layer compositor.  Same
// statements as the separate kernels above, applied per pixel in the reference's order; valid
// when nothing needs a second pass over `neu` (no leave-empty scatter).  RGB output is staged
// through LDS so the 3-byte pixels leave as whole dwords.
struct StepParams {
    MoveFlags fl;
    FastDiv div;       // t / W without a division (W is the same for every pixel of a launch)
    int clip_flow;     // 1: apply source.py:361-362 to the flow in registers (BACKWARD post_process);
                       // 2: `flow` is the winner map of a FORWARD post_process: source.py:359-362 in registers
    int reset_random;  // reset_mode == random
    float factor;
    int reset_source;
    int n_sources;
    uint64_t seed, frame;
    uchar4 bg;
    // tf_remap_steps_dev, every step but the last: where *rgba_dead != 0 the layer's rgba is not stored -- the next step
    // of the same call overwrites every pixel of it without reading it (see there).  Null: always stored.
    const int *rgba_dead;
};

// The layer state in HBM: int32 x 4 per pixel as the reference keeps it (data.py:6-17), or -- while only
// this kernel touches it -- int16 x 4: row, column, alpha and source index all fit, and the kernel is
// HBM-bound with the state as 48 of its ~58 bytes per pixel.
struct short4s {
    short x, y, z, w;
};
__device__ __forceinline__ int4 state_load(const int4 *p, size_t t) { return p[t]; }
__device__ __forceinline__ int4 state_load(const short4s *p, size_t t)
{
    const uint2 v = reinterpret_cast<const uint2 *>(p)[t];
    return make_int4((short)(v.x & 0xffff), (short)(v.x >> 16), (short)(v.y & 0xffff), (short)(v.y >> 16));
}
__device__ __forceinline__ void state_store(int4 *p, size_t t, int4 d) { p[t] = d; }
__device__ __forceinline__ void state_store(short4s *p, size_t t, int4 d)
{
    reinterpret_cast<uint2 *>(p)[t] = make_uint2(((unsigned)d.x & 0xffffu) | ((unsigned)d.y << 16),
                                                 ((unsigned)d.z & 0xffffu) | ((unsigned)d.w << 16));
}

// ... or, where they fit, ONE 32-bit word (round 5): row 13 bits, column 13 bits, alpha 1 bit, source index 5 bits --
// frames up to 8192 x 8192, up to 32 sources, alpha 0 or 1 (all the layer itself ever writes: 1 at creation, on a move
// and on a reset, 0 where a moving pixel leaves an empty spot).  The step reads the state twice per pixel (at the pixel
// and at its source) and writes it once: 12 of its ~34 bytes per pixel instead of 24 of ~44.
struct packed32 {
    unsigned v;
};
__device__ __forceinline__ int4 state_load(const packed32 *p, size_t t)
{
    const unsigned v = p[t].v;
    return make_int4((int)(v & 0x1fffu), (int)((v >> 13) & 0x1fffu), (int)((v >> 26) & 1u), (int)(v >> 27));
}
__device__ __forceinline__ void state_store(packed32 *p, size_t t, int4 d)
{
    p[t].v = ((unsigned)d.x & 0x1fffu) | (((unsigned)d.y & 0x1fffu) << 13) | (((unsigned)d.z & 1u) << 26) | ((unsigned)d.w << 27);
}

template <typename S>
__global__ void k_state_pack(const int4 *__restrict__ src, S *__restrict__ dst, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < N)
        state_store(dst, t, src[t]);
}

template <typename S>
__global__ void k_state_unpack(const S *__restrict__ src, int4 *__restrict__ dst, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < N)
        dst[t] = state_load(src, t);
}

template <int C, typename S>
__global__ void __launch_bounds__(BLOCK)
k_remap_step(const float2 *__restrict__ flow, const S *__restrict__ old, S *__restrict__ neu,
             const uint8_t *__restrict__ msrc, const uint8_t *__restrict__ mdst, const double *__restrict__ u,
             const float *__restrict__ reset_mask, const uint8_t *__restrict__ intro, uchar4 *__restrict__ rgba,
             const uint8_t *__restrict__ pixmap, const float *__restrict__ mask_alpha, uint8_t *__restrict__ image,
             int N, int H, int W, StepParams sp, int *err)
{
    __shared__ uint32_t s_rgb[BLOCK * 3 / 4];
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    uint8_t *s8 = reinterpret_cast<uint8_t *>(s_rgb);
    if (t < N) {
        const int i = (int)fast_div((uint32_t)t, sp.div), j = t - i * W;
        float2 f;
        if (sp.clip_flow == 2) {
            const int w = reinterpret_cast<const int *>(flow)[t];
            const int src = w >= 0 ? w : t;
            const int si = (int)fast_div((uint32_t)src, sp.div);
            f = make_float2((float)(src - si * W - j), (float)(si - i)); // source.py:359-360
        } else {
            f = flow[t];
        }
        if (sp.clip_flow) {
            f.x = clip_nan(f.x, (float)(-j), (float)(W - 1 - j));
            f.y = clip_nan(f.y, (float)(-i), (float)(H - 1 - i));
        }
        // --- move (movement.py:20-60)
        int4 me = state_load(old, t);
        int4 d = me;
        long long off = flow_offset(f, W);
        if (off != 0) {
            long long s = t + off;
            if (s < 0 || s >= N) {
                atomicOr(err, 1);
            } else {
                int4 so = state_load(old, (size_t)s);
                bool src_filled = so.z != 0;
                bool ms = (msrc ? msrc[s] != 0 : true) && (sp.fl.transparent_can_move || src_filled);
                bool md = (mdst ? mdst[t] != 0 : true) && (sp.fl.to_empty || me.z != 0) && (sp.fl.to_filled || me.z == 0);
                if (ms && md) {
                    d = so;
                    if (!sp.fl.transparent_can_move || src_filled)
                        d.z = 1;
                }
            }
        }
        // --- random reset (reference.py:58-67)
        if (sp.reset_random) {
            float thr = reset_mask ? sp.factor * reset_mask[t] : sp.factor;
            double uu = u ? u[t] : philox_uniform((uint32_t)t, sp.frame, sp.seed);
            if (uu < (double)thr) {
                d.x = i;
                d.y = j;
                d.z = 1;
                if (sp.reset_source)
                    for (int s = 0; s < sp.n_sources; s++)
                        if (intro[(size_t)s * N + t])
                            d.w = s;
            }
        }
        state_store(neu, t, d);
        // --- gather of source 0 (reference.py:94-105)
        bool sel = d.w == 0 && d.z != 0;
        uchar4 px;
        if (C == 4) {
            if (sel) {
                int gi = min(max(d.x, 0), H - 1), gj = min(max(d.y, 0), W - 1);
                px = reinterpret_cast<const uchar4 *>(pixmap)[(size_t)gi * W + gj];
            } else {
                px = rgba[t];
            }
        } else {
            if (sel) {
                int gi = min(max(d.x, 0), H - 1), gj = min(max(d.y, 0), W - 1);
                const uint8_t *p = pixmap + ((size_t)gi * W + gj) * 3;
                px = make_uchar4(p[0], p[1], p[2], 1);
            } else {
                px = rgba[t];
                px.w = 0;
            }
        }
        // --- Layer.render (layer.py:32-34)
        if (mask_alpha)
            px.w = (unsigned char)(int)(mask_alpha[t] * (float)px.w);
        if (!(sp.rgba_dead && *sp.rgba_dead))
            rgba[t] = px;
        // --- Compositor.render over the background (compositor.py:35-39)
        uchar4 o = px.w != 0 ? px : sp.bg;
        s8[threadIdx.x * 3 + 0] = o.x;
        s8[threadIdx.x * 3 + 1] = o.y;
        s8[threadIdx.x * 3 + 2] = o.z;
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * BLOCK * 3; // multiple of 4
    const size_t total = (size_t)N * 3;
    if (threadIdx.x < BLOCK * 3 / 4) {
        size_t b = base + (size_t)threadIdx.x * 4;
        if (b + 4 <= total) {
            *reinterpret_cast<uint32_t *>(image + b) = s_rgb[threadIdx.x];
        } else {
            for (size_t q = b; q < total; q++)
                image[q] = s8[q - base];
        }
    }
}

// The same step with PX pixels per thread, written in phases (all loads addressed by the pixel, then all
// loads addressed by its source, then the pixmap gathers) and without branches around loads.  The
// one-pixel form above waits on counters 83 % of its wave cycles at full occupancy -- three dependent
// loads per pixel: flow -> moved state -> pixmap -- so what helps is more of those chains in flight
// per wave, not fewer bytes.  Same statements, same results.
template <int C, typename S, int PX>
__global__ void __launch_bounds__(BLOCK)
k_remap_step_px(const float2 *__restrict__ flow, const S *__restrict__ old, S *__restrict__ neu,
                const uint8_t *__restrict__ msrc, const uint8_t *__restrict__ mdst, const double *__restrict__ u,
                const float *__restrict__ reset_mask, const uint8_t *__restrict__ intro, uchar4 *__restrict__ rgba,
                const uint8_t *__restrict__ pixmap, const float *__restrict__ mask_alpha, uint8_t *__restrict__ image,
                int N, int H, int W, StepParams sp, int *err)
{
    __shared__ uint32_t s_rgb[PX * BLOCK * 3 / 4];
    uint8_t *s8 = reinterpret_cast<uint8_t *>(s_rgb);
    // Blocks are dealt round-robin over the 8 XCDs, each with its own L2: renumbered so that one XCD walks a contiguous
    // eighth of the frame, the state a pixel reads at its SOURCE -- a few rows and columns away -- is the state the
    // neighbouring blocks of the same XCD read at their own pixels: one fetch per L2 instead of two (round 5).
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, qn = nb >> 3, rn = nb & 7;
    const unsigned bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int t0 = bid * (PX * BLOCK) + threadIdx.x;
    int t[PX], tc[PX];
    bool live[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
        t[p] = t0 + p * BLOCK;
        live[p] = t[p] < N;
        tc[p] = live[p] ? t[p] : N - 1; // a dead lane of the last block reads a valid pixel and stores nothing
    }
    // --- phase 1: what the pixel itself addresses
    float2 f[PX];
    int wv[PX];
    int4 me[PX];
    uint8_t mdv[PX];
    float rm[PX], ma[PX];
    double uv[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
        if (sp.clip_flow == 2)
            wv[p] = reinterpret_cast<const int *>(flow)[tc[p]];
        else
            f[p] = flow[tc[p]];
        me[p] = state_load(old, (size_t)tc[p]);
        rm[p] = (sp.reset_random && reset_mask) ? reset_mask[tc[p]] : 1.f;
        uv[p] = (sp.reset_random && u) ? u[tc[p]] : 0.0;
        ma[p] = mask_alpha ? mask_alpha[tc[p]] : 0.f;
    }
    // (an optional mask is one uniform branch around the loads of all PX pixels: taken per pixel, each load would be
    // waited for before the next pixel's loads were issued)
    if (mdst) {
#pragma unroll
        for (int p = 0; p < PX; p++)
            mdv[p] = mdst[tc[p]];
    } else {
#pragma unroll
        for (int p = 0; p < PX; p++)
            mdv[p] = 1;
    }
    // --- phase 2: the source pixel of the move (movement.py:20-48)
    int pi[PX], pj[PX], sidx[PX];
    bool moved[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
        const int i = (int)fast_div((uint32_t)tc[p], sp.div), j = tc[p] - i * W;
        pi[p] = i;
        pj[p] = j;
        float2 g;
        if (sp.clip_flow == 2) {
            const int src = wv[p] >= 0 ? wv[p] : tc[p];
            const int si = (int)fast_div((uint32_t)src, sp.div);
            g = make_float2((float)(src - si * W - j), (float)(si - i)); // source.py:359-360
        } else {
            g = f[p];
        }
        if (sp.clip_flow) {
            g.x = clip_nan(g.x, (float)(-j), (float)(W - 1 - j));
            g.y = clip_nan(g.y, (float)(-i), (float)(H - 1 - i));
        }
        const long long off = flow_offset(g, W);
        const long long s = tc[p] + off;
        const bool inside = s >= 0 && s < N;
        if (off != 0 && !inside && live[p])
            atomicOr(err, 1);
        moved[p] = off != 0 && inside;
        sidx[p] = moved[p] ? (int)s : tc[p];
    }
    // --- phase 3: what the source addresses
    int4 so[PX];
    uint8_t msv[PX];
#pragma unroll
    for (int p = 0; p < PX; p++)
        so[p] = state_load(old, (size_t)sidx[p]);
    if (msrc) {
#pragma unroll
        for (int p = 0; p < PX; p++)
            msv[p] = msrc[sidx[p]];
    } else {
#pragma unroll
        for (int p = 0; p < PX; p++)
            msv[p] = 1;
    }
    // --- phase 4: the new state (move, then the random reset of reference.py:58-67), the gather address
    int4 d[PX];
    bool sel[PX];
    size_t gidx[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
        d[p] = me[p];
        const bool src_filled = so[p].z != 0;
        const bool ms = msv[p] != 0 && (sp.fl.transparent_can_move || src_filled);
        const bool md = mdv[p] != 0 && (sp.fl.to_empty || me[p].z != 0) && (sp.fl.to_filled || me[p].z == 0);
        if (moved[p] && ms && md) {
            d[p] = so[p];
            if (!sp.fl.transparent_can_move || src_filled)
                d[p].z = 1;
        }
        if (sp.reset_random) {
            const float thr = sp.factor * rm[p]; // factor * 1.f == factor where no mask is set
            const double uu = u ? uv[p] : philox_uniform((uint32_t)tc[p], sp.frame, sp.seed);
            if (uu < (double)thr) {
                d[p].x = pi[p];
                d[p].y = pj[p];
                d[p].z = 1;
                if (sp.reset_source)
                    for (int q = 0; q < sp.n_sources; q++)
                        if (intro[(size_t)q * N + tc[p]])
                            d[p].w = q;
            }
        }
        if (live[p])
            state_store(neu, (size_t)t[p], d[p]);
        sel[p] = d[p].w == 0 && d[p].z != 0;
        const int gi = min(max(d[p].x, 0), H - 1), gj = min(max(d[p].y, 0), W - 1);
        gidx[p] = sel[p] ? (size_t)gi * W + gj : 0;
    }
    // --- phase 5: gather of source 0 (reference.py:94-105); pixels not selected keep their previous colour
    const bool store_rgba = !(sp.rgba_dead && *sp.rgba_dead);
    uchar4 px[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
        if (C == 4) {
            px[p] = reinterpret_cast<const uchar4 *>(pixmap)[gidx[p]];
        } else {
            const uint8_t *q = pixmap + gidx[p] * 3;
            px[p] = make_uchar4(q[0], q[1], q[2], 1);
        }
    }
#pragma unroll
    for (int p = 0; p < PX; p++) {
        if (!sel[p]) {
            px[p] = rgba[tc[p]];
            if (C == 3)
                px[p].w = 0;
        }
        // --- Layer.render (layer.py:32-34)
        if (mask_alpha)
            px[p].w = (unsigned char)(int)(ma[p] * (float)px[p].w);
        if (live[p] && store_rgba)
            rgba[t[p]] = px[p];
        // --- Compositor.render over the background (compositor.py:35-39)
        const uchar4 o = px[p].w != 0 ? px[p] : sp.bg;
        const int k = (p * BLOCK + threadIdx.x) * 3;
        s8[k + 0] = o.x;
        s8[k + 1] = o.y;
        s8[k + 2] = o.z;
    }
    __syncthreads();
    const size_t base = (size_t)bid * (PX * BLOCK) * 3; // multiple of 4
    const size_t total = (size_t)N * 3;
    for (int idx = threadIdx.x; idx < PX * BLOCK * 3 / 4; idx += BLOCK) {
        const size_t b = base + (size_t)idx * 4;
        if (b + 4 <= total) {
            *reinterpret_cast<uint32_t *>(image + b) = s_rgb[idx];
        } else {
            for (size_t q = b; q < total; q++)
                image[q] = s8[q - base];
        }
    }
}

__global__ void k_remap_clip_flow(float2 *flow, int W, int H)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= W * H)
        return;
    int i = t / W, j = t % W;
    float2 f = flow[t];
    f.x = clip_nan(f.x, (float)(-j), (float)(W - 1 - j));
    f.y = clip_nan(f.y, (float)(-i), (float)(H - 1 - i));
    flow[t] = f;
}

// source.py:359-362 alone: the flow a FORWARD winner map stands for (the unfused step's first launch)
__global__ void k_remap_winner_flow(const int *__restrict__ winner, float2 *__restrict__ flow, int W, int H)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= W * H)
        return;
    int i = t / W, j = t % W;
    const int w = winner[t];
    const int src = w >= 0 ? w : t;
    float2 f = make_float2((float)(src % W - j), (float)(src / W - i));
    f.x = clip_nan(f.x, (float)(-j), (float)(W - 1 - j));
    f.y = clip_nan(f.y, (float)(-i), (float)(H - 1 - i));
    flow[t] = f;
}

__global__ void k_comp_fill(uint8_t *image, int N, uchar4 bg)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    uint8_t *o = image + (size_t)t * 3;
    o[0] = bg.x;
    o[1] = bg.y;
    o[2] = bg.z;
}


// ---------------------------------------------------------------------------------
// The other layer classes (layer.py:44-56)
// ---------------------------------------------------------------------------------
// SumLayer._update_sum (sum.py:10): (i, j) += floor(flow) -- channel 0 to i, channel 1 to j, as written
__global__ void k_layer_sum(int4 *data, const float2 *__restrict__ flow, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    float2 f = flow[t];
    int4 d = data[t];
    d.x += (int)floorf(f.x);
    d.y += (int)floorf(f.y);
    data[t] = d;
}

// StaticLayer.__init__ (static.py:11): alpha = 1
__global__ void k_layer_static_init(uchar4 *rgba, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < N)
        rgba[t] = make_uchar4(0, 0, 0, 1);
}

// StaticLayer.update for one source (static.py:14-17): rgba[..., :C][mask] = pixmap[mask]
template <int C>
__global__ void k_layer_static_put(uchar4 *__restrict__ rgba, const uint8_t *__restrict__ pixmap,
                                   const uint8_t *__restrict__ intro, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N || !intro[t])
        return;
    if (C == 4) {
        rgba[t] = reinterpret_cast<const uchar4 *>(pixmap)[t];
    } else {
        const uint8_t *p = pixmap + (size_t)t * 3;
        uchar4 px = rgba[t];
        px.x = p[0];
        px.y = p[1];
        px.z = p[2];
        rgba[t] = px;
    }
}

// IntroductionLayer: the canvas is two int4 per pixel, lo = (r, g, b, alpha), hi = (source, i, j, frame)
// (introduction.py:10-14), i.e. the reference's int32 [H][W][8] as it lies in memory.
__device__ __forceinline__ bool intro_move_target(long long t, const float2 *__restrict__ flow,
                                                  const int4 *__restrict__ old, const uint8_t *__restrict__ msrc,
                                                  const uint8_t *__restrict__ mdst, int N, int W, MoveFlags fl,
                                                  int my_alpha, long long &s, int &src_alpha, bool &oob)
{
    long long d = flow_offset(flow[t], W);
    oob = false;
    if (d == 0)
        return false;
    s = t + d;
    if (s < 0 || s >= N) {
        oob = true;
        return false;
    }
    src_alpha = old[2 * s].w;
    bool ms = (msrc ? msrc[s] != 0 : true) && (fl.transparent_can_move || src_alpha != 0);
    bool md = (mdst ? mdst[t] != 0 : true) && (fl.to_empty || my_alpha != 0) && (fl.to_filled || my_alpha == 0);
    return ms && md;
}

// movement.py:25-60 with DEPTH = 8, INDEX_ALPHA = 3
__global__ void k_intro_move(const float2 *__restrict__ flow, const int4 *__restrict__ old, int4 *__restrict__ neu,
                             const uint8_t *__restrict__ msrc, const uint8_t *__restrict__ mdst, int N, int W,
                             MoveFlags fl, int *err)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int4 lo = old[2 * (size_t)t], hi = old[2 * (size_t)t + 1];
    long long s;
    int sa;
    bool oob;
    if (intro_move_target(t, flow, old, msrc, mdst, N, W, fl, lo.w, s, sa, oob)) {
        lo = old[2 * s];
        hi = old[2 * s + 1];
        if (!fl.transparent_can_move || sa != 0)
            lo.w = 1;
    }
    if (oob)
        atomicOr(err, 1);
    neu[2 * (size_t)t] = lo;
    neu[2 * (size_t)t + 1] = hi;
}

// movement.py:53-54 on the 8-channel canvas (see k_remap_leave_empty)
__global__ void k_intro_leave_empty(const float2 *__restrict__ flow, const int4 *__restrict__ old,
                                    int4 *__restrict__ neu, const uint8_t *__restrict__ msrc,
                                    const uint8_t *__restrict__ mdst, int N, int W, MoveFlags fl)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    long long s, s2;
    int sa, sa2;
    bool oob;
    if (!intro_move_target(t, flow, old, msrc, mdst, N, W, fl, old[2 * (size_t)t].w, s, sa, oob))
        return;
    bool s_is_one = intro_move_target(s, flow, old, msrc, mdst, N, W, fl, old[2 * s].w, s2, sa2, oob) &&
                    (!fl.transparent_can_move || sa2 != 0);
    if (!s_is_one)
        neu[2 * s].w = 0;
}

// introduction.py:24-44: which targets may receive an introduced pixel, from the canvas after the
// move.  on_empty_spots / unmoving_pixels / the write of on_all_empty_spots select nothing in the
// reference (tfhip.h), so three conditions remain.
__global__ void k_intro_mask(const int4 *__restrict__ data, const float2 *__restrict__ flow, uint8_t *__restrict__ mask,
                             int N, int W, int on_filled, int moving, int all_filled)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    const bool filled = data[2 * (size_t)t].w != 0;
    bool m = true;
    if (!on_filled && filled)
        m = false;
    if (!moving && flow_offset(flow[t], W) != 0)
        m = false;
    if (all_filled && filled)
        m = true;
    mask[t] = m;
}

// introduction.py:46-63 for one source
template <int C>
__global__ void k_intro_put(int4 *__restrict__ data, const float2 *__restrict__ flow, const uint8_t *__restrict__ mask,
                            const uint8_t *__restrict__ intro, const uint8_t *__restrict__ pixmap, int N, int W,
                            int consider_flow, int source_index, int frame_number, int *err)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N || !mask[t] || !intro[t])
        return;
    long long s = t;
    if (consider_flow) {
        s += flow_offset(flow[t], W);
        if (s < 0 || s >= N) {
            atomicOr(err, 1);
            return;
        }
    }
    int4 lo;
    if (C == 4) {
        uchar4 px = reinterpret_cast<const uchar4 *>(pixmap)[s];
        lo = make_int4(px.x, px.y, px.z, px.w);
    } else {
        const uint8_t *p = pixmap + (size_t)s * 3;
        lo = make_int4(p[0], p[1], p[2], 1);
    }
    data[2 * (size_t)t] = lo;
    data[2 * (size_t)t + 1] = make_int4(source_index, (int)(s / W), (int)(s % W), frame_number);
}

// Layer.render on the int32 view rgba = data[..., :4] (introduction.py:65-66, layer.py:32-34):
// alpha := int32(float64(mask_alpha) * alpha) written back into the canvas, the returned image is
// clip(rgba, 0, 255) as uint8; then Compositor.render's paint (compositor.py:36-39).
__global__ void k_intro_render(int4 *__restrict__ data, const float *__restrict__ mask_alpha,
                               uint8_t *__restrict__ image, int N)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= N)
        return;
    int4 lo = data[2 * (size_t)t];
    if (mask_alpha) {
        int a = (int)((double)mask_alpha[t] * (double)lo.w);
        if (a != lo.w) {
            lo.w = a;
            data[2 * (size_t)t].w = a;
        }
    }
    if (min(max(lo.w, 0), 255) != 0) {
        uint8_t *o = image + (size_t)t * 3;
        o[0] = (uint8_t)min(max(lo.x, 0), 255);
        o[1] = (uint8_t)min(max(lo.y, 0), 255);
        o[2] = (uint8_t)min(max(lo.z, 0), 255);
    }
}

} // namespace

struct tf_comp {
    int H, W, N;
    uchar4 bg;
    DevBuf image;
};

struct tf_remap {
    int H, W, N;
    tf_layer_cfg cfg;
    MoveFlags fl;
    DevBuf data[2];
    int cur = 0;
    DevBuf rgba;
    DevBuf mask_src, mask_dst, mask_alpha, reset_mask;
    DevBuf intro;
    DevBuf intro_sel;  // introduction layer: the mask of introduction.py:24-44 for the current frame
    DevBuf last_flow;  // introduction layer: the flow of the current frame (device copy when it came from the host)
    const float2 *flow_for_intro = nullptr;
    int n_sources = 0;
    int depth() const { return cfg.layer_class == TF_LAYER_INTRODUCTION ? 8 : (cfg.layer_class == TF_LAYER_STATIC ? 0 : 4); }
    DevBuf err;
    DevBuf scratch_flow, scratch_u, scratch_pix;
    DevBuf flow_scratch; // tf_remap_step_dev's unfused form on a winner map: the flow it stands for
    DevBuf sel_flag;     // tf_remap_steps_dev: 1 while every pixel is selected by source 0
    uint64_t frame = 0;
    // the fused step keeps the state as one 32-bit word or as int16 x 4 (k_remap_step's note); every other entry point
    // that touches `data` converts it back first (state_unpacked)
    DevBuf pdata[2];
    int pcur = 0;
    int packed = 0;            // 0: data[cur] is current; 1: pdata[pcur] as int16 x 4; 2: pdata[pcur] as one word (data[cur] stale)
    bool state_fits = true;    // false after a set_state with values outside int16
    bool state_fits32 = true;  // false after a set_state with a row / column outside [0, 8191], an alpha other than 0 / 1 or a source index outside [0, 31]
    // tf_remap_gather_beside: the pixmap goes up on the library's upload stream, beside the update kernel queued before it
    hipEvent_t pix_up = nullptr, pix_used = nullptr; // the upload's end; the end of the last kernel that read scratch_pix
    bool pix_used_pending = false;
    int4 *cur_data() { return data[cur].as<int4>(); }
    ~tf_remap()
    {
        for (hipEvent_t e : {pix_up, pix_used})
            if (e)
                (void)hipEventDestroy(e);
    }
};

// *flag := 0 if any pixel of the state is NOT "selected" by source 0 (alpha != 0 and source index 0); the caller sets it to 1 first
template <typename S>
__global__ void k_state_all_selected(const S *__restrict__ state, int N, int *flag)
{
    bool bad = false;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < (size_t)N; t += (size_t)gridDim.x * BLOCK) {
        const int4 d = state_load(state, t);
        bad = bad || !(d.w == 0 && d.z != 0);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0)
        atomicAnd(flag, 0);
}

// Makes data[cur] (int32) the current state.
static int state_unpacked(tf_remap *L)
{
    if (!L->packed)
        return TF_OK;
    const dim3 grid(cdiv((size_t)L->N, BLOCK)), block(BLOCK);
    if (L->packed == 2)
        TF_TRY(launch("remap_state_unpack", k_state_unpack<packed32>, grid, block, 0,
                      (const packed32 *)L->pdata[L->pcur].as<packed32>(), L->cur_data(), L->N));
    else
        TF_TRY(launch("remap_state_unpack", k_state_unpack<short4s>, grid, block, 0,
                      (const short4s *)L->pdata[L->pcur].as<short4s>(), L->cur_data(), L->N));
    L->packed = 0;
    return TF_OK;
}

// How the fused step may keep the state: 2 = one word per pixel, 1 = int16 x 4, 0 = as it is (option remap_no_pack: 1 = never
// packed, 2 = never as one word)
static int state_can_pack(const tf_remap *L)
{
    const long off = option(OPT_REMAP_NO_PACK);
    if (off == 1)
        return 0;
    if (off != 2 && L->state_fits32 && L->H <= 8192 && L->W <= 8192 && L->n_sources <= 32)
        return 2;
    return (L->state_fits && L->H <= 32767 && L->W <= 32767 && L->n_sources <= 32767) ? 1 : 0;
}

// Makes pdata[pcur] in form `kind` (1: int16 x 4, 2: one word) the current state.
static int state_packed(tf_remap *L, int kind)
{
    if (L->packed == kind)
        return TF_OK;
    TF_TRY(state_unpacked(L));
    for (auto &b : L->pdata)
        if (!b.p)
            TF_TRY(b.alloc((size_t)L->N * sizeof(short4s))); // (room for either form)
    const dim3 grid(cdiv((size_t)L->N, BLOCK)), block(BLOCK);
    if (kind == 2)
        TF_TRY(launch("remap_state_pack", k_state_pack<packed32>, grid, block, 0, (const int4 *)L->cur_data(),
                      L->pdata[L->pcur].as<packed32>(), L->N));
    else
        TF_TRY(launch("remap_state_pack", k_state_pack<short4s>, grid, block, 0, (const int4 *)L->cur_data(),
                      L->pdata[L->pcur].as<short4s>(), L->N));
    L->packed = kind;
    return TF_OK;
}

static int upload(DevBuf &buf, const void *host, size_t bytes)
{
    if (buf.bytes < bytes)
        TF_TRY(buf.alloc(bytes));
    TF_HIP(hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, stream()));
    return TF_OK;
}

static int comp_create(tf_comp **out, int height, int width, const uint8_t background_rgb[3], void *image_dev)
{
    TF_REQUIRE(height >= 0 && width >= 0 && (long long)height * width < (1ll << 31), "tf_comp_create: bad size %dx%d",
               width, height);
    TF_TRY(ensure_init());
    tf_comp *c = new tf_comp;
    c->H = height;
    c->W = width;
    c->N = height * width;
    c->bg = make_uchar4(background_rgb[0], background_rgb[1], background_rgb[2], 0);
    int rc = TF_OK;
    if (image_dev)
        c->image.borrow(image_dev, (size_t)c->N * 3);
    else
        rc = c->image.alloc((size_t)c->N * 3);
    if (rc != TF_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return tf_comp_begin(c);
}

TF_API int tf_comp_create(tf_comp **out, int height, int width, const uint8_t background_rgb[3])
{
    TF_REQUIRE(out && background_rgb, "tf_comp_create: null pointer");
    return comp_create(out, height, width, background_rgb, nullptr);
}

TF_API int tf_comp_create_on(tf_comp **out, int height, int width, const uint8_t background_rgb[3], void *image_dev)
{
    TF_REQUIRE(out && background_rgb, "tf_comp_create_on: null pointer");
    TF_REQUIRE(image_dev || (long long)height * width == 0, "tf_comp_create_on: null image");
    return comp_create(out, height, width, background_rgb, image_dev);
}

TF_API void tf_comp_destroy(tf_comp *comp) { delete comp; }

TF_API int tf_comp_begin(tf_comp *comp)
{
    TF_REQUIRE(comp, "tf_comp_begin: null handle");
    TF_TRY(ensure_init());
    return launch("comp_fill", k_comp_fill, dim3(cdiv(comp->N, BLOCK)), dim3(BLOCK), 0, comp->image.as<uint8_t>(),
                  comp->N, comp->bg);
}

TF_API int tf_comp_download(tf_comp *comp, uint8_t *rgb_out)
{
    TF_REQUIRE(comp && rgb_out, "tf_comp_download: null pointer");
    TF_TRY(ensure_init());
    if (comp->N) {
        TF_HIP(hipMemcpyAsync(rgb_out, comp->image.p, (size_t)comp->N * 3, hipMemcpyDeviceToHost, stream()));
    }
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_comp_image_ptr(tf_comp *comp, void **dev)
{
    TF_REQUIRE(comp && dev, "tf_comp_image_ptr: null pointer");
    *dev = comp->image.p;
    return TF_OK;
}

TF_API int tf_remap_create(tf_remap **out, int height, int width, const tf_layer_cfg *cfg, const uint8_t *mask_src,
                           const uint8_t *mask_dst, const float *mask_alpha, const float *reset_mask)
{
    TF_REQUIRE(out && cfg, "tf_remap_create: null pointer");
    TF_REQUIRE(height >= 0 && width >= 0 && (long long)height * width < (1ll << 31),
               "tf_remap_create: bad size %dx%d", width, height);
    TF_REQUIRE(cfg->reset_mode >= 0 && cfg->reset_mode <= 3, "tf_remap_create: unknown reset mode %d", cfg->reset_mode);
    TF_REQUIRE(cfg->layer_class >= TF_LAYER_MOVEREF && cfg->layer_class <= TF_LAYER_INTRODUCTION,
               "tf_remap_create: unknown layer class %d", cfg->layer_class);
    TF_TRY(ensure_init());
    tf_remap *L = new tf_remap;
    L->H = height;
    L->W = width;
    L->N = height * width;
    L->cfg = *cfg;
    L->fl = MoveFlags{cfg->transparent_pixels_can_move != 0, cfg->pixels_can_move_to_empty_spot != 0,
                      cfg->pixels_can_move_to_filled_spot != 0, cfg->moving_pixels_leave_empty_spot != 0};
    size_t n = (size_t)L->N;
    int rc = TF_OK;
    auto fail = [&](int code) {
        delete L;
        return code;
    };
    const size_t px_bytes = (size_t)L->depth() * 4;
    if ((rc = L->data[0].alloc(n * px_bytes)) || (rc = L->data[1].alloc(n * px_bytes)) || (rc = L->rgba.alloc(n * 4)) ||
        (rc = L->err.alloc(4)))
        return fail(rc);
    if (cfg->layer_class == TF_LAYER_INTRODUCTION && (rc = L->intro_sel.alloc(n)))
        return fail(rc);
    if (mask_src && (rc = upload(L->mask_src, mask_src, n)))
        return fail(rc);
    if (mask_dst && (rc = upload(L->mask_dst, mask_dst, n)))
        return fail(rc);
    if (mask_alpha && (rc = upload(L->mask_alpha, mask_alpha, n * 4)))
        return fail(rc);
    if (reset_mask && (rc = upload(L->reset_mask, reset_mask, n * 4)))
        return fail(rc);
    hipError_t e;
    if (n && (e = hipMemsetAsync(L->rgba.p, 0, n * 4, stream())) != hipSuccess)
        return fail(set_error(TF_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e)));
    if ((e = hipMemsetAsync(L->err.p, 0, 4, stream())) != hipSuccess)
        return fail(set_error(TF_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e)));
    switch (cfg->layer_class) {
    case TF_LAYER_STATIC:
        rc = launch("layer_static_init", k_layer_static_init, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0, L->rgba.as<uchar4>(),
                    L->N);
        break;
    case TF_LAYER_INTRODUCTION: // data.py:17: an all-zero canvas
        if (n && (e = hipMemsetAsync(L->data[0].p, 0, n * px_bytes, stream())) != hipSuccess)
            return fail(set_error(TF_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e)));
        break;
    default:
        rc = launch("remap_init", k_remap_init, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0, L->cur_data(), L->N, L->W);
    }
    if (rc)
        return fail(rc);
    if ((e = hipStreamSynchronize(stream())) != hipSuccess) // host mask buffers are borrowed for this call only
        return fail(set_error(TF_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e)));
    *out = L;
    return TF_OK;
}

TF_API void tf_remap_destroy(tf_remap *layer) { delete layer; }

TF_API int tf_remap_set_sources(tf_remap *L, int n_sources, const uint8_t *const *introduction_masks)
{
    TF_REQUIRE(L, "tf_remap_set_sources: null handle");
    TF_REQUIRE(n_sources >= 0 && (n_sources == 0 || introduction_masks), "tf_remap_set_sources: bad arguments");
    TF_TRY(ensure_init());
    size_t n = (size_t)L->N;
    L->n_sources = n_sources;
    if (n_sources == 0 || n == 0)
        return TF_OK;
    if (L->intro.bytes < n * n_sources)
        TF_TRY(L->intro.alloc(n * n_sources));
    for (int s = 0; s < n_sources; s++) {
        TF_REQUIRE(introduction_masks[s], "tf_remap_set_sources: mask %d is null", s);
        TF_HIP(hipMemcpyAsync(L->intro.as<uint8_t>() + n * s, introduction_masks[s], n, hipMemcpyHostToDevice,
                              stream()));
    }
    TF_TRY(state_unpacked(L));
    if (L->depth() == 4) // ReferenceLayer.set_sources (reference.py:54-56); the other classes only keep the masks
        TF_TRY(launch("remap_set_sources", k_remap_set_sources, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0, L->cur_data(),
                      L->intro.as<uint8_t>(), n_sources, L->N));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

// Queues move (+ leave-empty) + reset on the stream, writing the other data buffer.
static int queue_update(tf_remap *L, const float2 *flow, const double *u, uint64_t seed)
{
    TF_TRY(state_unpacked(L));
    int N = L->N;
    dim3 grid(cdiv((size_t)N, BLOCK)), block(BLOCK);
    const int4 *old = L->data[L->cur].as<int4>();
    int4 *neu = L->data[L->cur ^ 1].as<int4>();
    const uint8_t *msrc = L->mask_src.as<uint8_t>(), *mdst = L->mask_dst.as<uint8_t>();
    if (L->cfg.layer_class == TF_LAYER_INTRODUCTION) {
        TF_TRY(launch("intro_move", k_intro_move, grid, block, 0, flow, old, neu, msrc, mdst, N, L->W, L->fl,
                      L->err.as<int>()));
        if (L->fl.leave_empty)
            TF_TRY(launch("intro_leave_empty", k_intro_leave_empty, grid, block, 0, flow, old, neu, msrc, mdst, N, L->W,
                          L->fl));
        L->flow_for_intro = flow;
        return launch("intro_mask", k_intro_mask, grid, block, 0, (const int4 *)neu, flow, L->intro_sel.as<uint8_t>(), N,
                      L->W, L->cfg.introduce_pixels_on_filled_spots, L->cfg.introduce_moving_pixels,
                      L->cfg.introduce_on_all_filled_spots);
    }
    if (L->cfg.layer_class == TF_LAYER_SUM) {
        // no snapshot needed: every pixel updates itself; the "new" buffer is the current one
        neu = L->data[L->cur].as<int4>();
        TF_TRY(launch("layer_sum", k_layer_sum, grid, block, 0, neu, flow, N));
    } else {
        TF_TRY(launch("remap_move", k_remap_move, grid, block, 0, flow, old, neu, msrc, mdst, N, L->W, L->fl,
                      L->err.as<int>()));
        if (L->fl.leave_empty)
            TF_TRY(launch("remap_leave_empty", k_remap_leave_empty, grid, block, 0, flow, old, neu, msrc, mdst, N, L->W,
                          L->fl));
    }
    const float *rmask = L->reset_mask.as<float>();
    switch (L->cfg.reset_mode) {
    case 1:
        TF_TRY(launch("remap_reset_random", k_remap_reset_random, grid, block, 0, neu, u, rmask,
                      (float)L->cfg.reset_random_factor, L->cfg.reset_source, L->intro.as<uint8_t>(), L->n_sources, N,
                      L->W, seed, L->frame));
        break;
    case 2:
        TF_TRY(launch("remap_reset_constant", k_remap_reset_constant, grid, block, 0, neu, rmask,
                      (float)L->cfg.reset_constant_step, N, L->W));
        break;
    case 3:
        TF_TRY(launch("remap_reset_linear", k_remap_reset_linear, grid, block, 0, neu, rmask,
                      L->cfg.reset_linear_factor, N, L->W));
        break;
    default:
        break;
    }
    return TF_OK;
}

TF_API int tf_remap_update_dev(tf_remap *L, const void *flow_dev, const void *uniform_dev, uint64_t seed)
{
    TF_REQUIRE(L && (flow_dev || L->N == 0), "tf_remap_update_dev: null pointer");
    TF_TRY(ensure_init());
    if (L->N == 0 || L->cfg.layer_class == TF_LAYER_STATIC) // static.py:13 ignores the flow
        return TF_OK;
    TF_TRY(queue_update(L, (const float2 *)flow_dev, (const double *)uniform_dev, seed));
    if (L->cfg.layer_class != TF_LAYER_SUM)
        L->cur ^= 1;
    L->frame++;
    return TF_OK;
}

TF_API int tf_remap_update(tf_remap *L, const float *flow, const double *uniform, uint64_t seed)
{
    TF_REQUIRE(L && (flow || L->N == 0), "tf_remap_update: null pointer");
    TF_TRY(ensure_init());
    if (L->N == 0 || L->cfg.layer_class == TF_LAYER_STATIC)
        return TF_OK;
    size_t n = (size_t)L->N;
    TF_TRY(upload(L->scratch_flow, flow, n * 8));
    const double *u = nullptr;
    if (uniform && L->cfg.reset_mode == 1) {
        TF_TRY(upload(L->scratch_u, uniform, n * 8));
        u = L->scratch_u.as<double>();
    }
    TF_HIP(hipMemsetAsync(L->err.p, 0, 4, stream()));
    TF_TRY(queue_update(L, L->scratch_flow.as<float2>(), u, seed));
    int err = 0;
    TF_HIP(hipMemcpyAsync(&err, L->err.p, 4, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    if (err)
        return set_error(TF_ERR_INDEX, "tf_remap_update: a rounded flow vector leaves the %dx%d frame "
                                       "(run post_process first); layer state unchanged",
                         L->W, L->H);
    if (L->cfg.layer_class != TF_LAYER_SUM)
        L->cur ^= 1;
    L->frame++;
    return TF_OK;
}

TF_API int tf_remap_check(tf_remap *L, int *out_of_frame)
{
    TF_REQUIRE(L && out_of_frame, "tf_remap_check: null pointer");
    TF_TRY(ensure_init());
    int err = 0;
    TF_HIP(hipMemcpyAsync(&err, L->err.p, 4, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipMemsetAsync(L->err.p, 0, 4, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    *out_of_frame = err != 0;
    return TF_OK;
}

__global__ void k_remap_uniform(double *__restrict__ u, int N, uint64_t seed, uint64_t frame)
{
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < N)
        u[t] = philox_uniform((uint32_t)t, frame, seed);
}

TF_API int tf_remap_uniform_dev(tf_remap *L, uint64_t seed, void *uniform_dev)
{
    TF_REQUIRE(L && (uniform_dev || L->N == 0), "tf_remap_uniform_dev: null pointer");
    TF_TRY(ensure_init());
    return launch("remap_uniform", k_remap_uniform, dim3(cdiv((size_t)L->N, BLOCK)), dim3(BLOCK), 0, (double *)uniform_dev,
                  L->N, seed, L->frame);
}

TF_API int tf_remap_gather_dev(tf_remap *L, int source_index, const void *pixmap_dev, int channels)
{
    TF_REQUIRE(L && (pixmap_dev || L->N == 0), "tf_remap_gather_dev: null pointer");
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_gather: pixmap must have 3 or 4 channels, got %d", channels);
    TF_REQUIRE(L->cfg.layer_class != TF_LAYER_INTRODUCTION, "tf_remap_gather: an introduction layer takes its pixmaps "
                                                            "through tf_remap_introduce");
    TF_TRY(ensure_init());
    TF_TRY(state_unpacked(L));
    dim3 grid(cdiv((size_t)L->N, BLOCK)), block(BLOCK);
    if (L->cfg.layer_class == TF_LAYER_STATIC) {
        TF_REQUIRE(source_index >= 0 && source_index < L->n_sources, "tf_remap_gather: source %d of %d (set_sources first)",
                   source_index, L->n_sources);
        const uint8_t *intro = L->intro.as<uint8_t>() + (size_t)L->N * source_index;
        if (channels == 4)
            return launch("layer_static_put_rgba", k_layer_static_put<4>, grid, block, 0, L->rgba.as<uchar4>(),
                          (const uint8_t *)pixmap_dev, intro, L->N);
        return launch("layer_static_put_rgb", k_layer_static_put<3>, grid, block, 0, L->rgba.as<uchar4>(),
                      (const uint8_t *)pixmap_dev, intro, L->N);
    }
    if (channels == 4)
        return launch("remap_gather_rgba", k_remap_gather<4>, grid, block, 0, (const int4 *)L->cur_data(),
                      L->rgba.as<uchar4>(), (const uint8_t *)pixmap_dev, source_index, L->N, L->H, L->W);
    return launch("remap_gather_rgb", k_remap_gather<3>, grid, block, 0, (const int4 *)L->cur_data(),
                  L->rgba.as<uchar4>(), (const uint8_t *)pixmap_dev, source_index, L->N, L->H, L->W);
}

TF_API int tf_remap_gather(tf_remap *L, int source_index, const uint8_t *pixmap, int channels)
{
    TF_REQUIRE(L && (pixmap || L->N == 0), "tf_remap_gather: null pointer");
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_gather: pixmap must have 3 or 4 channels, got %d", channels);
    TF_TRY(ensure_init());
    if (L->N == 0)
        return TF_OK;
    TF_TRY(upload(L->scratch_pix, pixmap, (size_t)L->N * channels));
    TF_TRY(tf_remap_gather_dev(L, source_index, L->scratch_pix.p, channels));
    TF_HIP(hipStreamSynchronize(stream())); // the host pixmap is borrowed for this call only
    return TF_OK;
}

// The same with the pixmap going up on the library's upload stream instead of the caller's: it does not wait for what the
// caller queued before it -- the layer's update kernel, which may itself sit behind another thread's long kernels -- and the
// host only waits for the COPY (the pixmap is borrowed for this call only), not for the kernels.  For an update whose flow
// was on the device already (nothing else of this thread is on the link): 4K beside a prefetching flow source 0.78 ->
// 0.65 ms per call, 617 -> 650 frames/s; behind an update that uploaded its flow it only adds a second stream to the link
// (359 -> 289 frames/s), so the compositor asks for it per update.
// The pixmap into the layer's staging buffer, on the library's upload stream (`beside`) or on the caller's; returns with
// the host pixmap consumed and the caller's stream ordered behind the copy.
static int stage_pixmap(tf_remap *L, const uint8_t *pixmap, int channels, bool beside)
{
    const size_t bytes = (size_t)L->N * channels;
    if (L->scratch_pix.bytes < bytes)
        TF_TRY(L->scratch_pix.alloc(bytes));
    if (!L->pix_up) {
        TF_HIP(hipEventCreateWithFlags(&L->pix_up, hipEventDisableTiming));
        TF_HIP(hipEventCreateWithFlags(&L->pix_used, hipEventDisableTiming));
    }
    hipStream_t up = stream();
    if (beside)
        TF_TRY(side_stream(3, &up)); // (a streaming flow source's frames go up on it too: one after the other at the link's rate)
    if (beside && L->pix_used_pending) // the copy waits, on the device, for the last kernel that read the staging buffer
        TF_HIP(hipStreamWaitEvent(up, L->pix_used, 0));
    TF_HIP(hipMemcpyAsync(L->scratch_pix.p, pixmap, bytes, hipMemcpyHostToDevice, up));
    TF_HIP(hipEventRecord(L->pix_up, up));
    if (beside)
        TF_HIP(hipStreamWaitEvent(stream(), L->pix_up, 0));
    TF_HIP(hipEventSynchronize(L->pix_up));
    return TF_OK;
}

TF_API int tf_remap_gather_beside(tf_remap *L, int source_index, const uint8_t *pixmap, int channels)
{
    TF_REQUIRE(L && (pixmap || L->N == 0), "tf_remap_gather_beside: null pointer");
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_gather_beside: pixmap must have 3 or 4 channels, got %d", channels);
    TF_TRY(ensure_init());
    if (L->N == 0)
        return TF_OK;
    TF_TRY(stage_pixmap(L, pixmap, channels, true));
    TF_TRY(tf_remap_gather_dev(L, source_index, L->scratch_pix.p, channels));
    TF_HIP(hipEventRecord(L->pix_used, stream()));
    L->pix_used_pending = true;
    return TF_OK;
}

TF_API int tf_remap_stage_pixmap(tf_remap *L, const uint8_t *pixmap, int channels, int beside, void **pixmap_dev)
{
    TF_REQUIRE(L && pixmap_dev && (pixmap || L->N == 0), "tf_remap_stage_pixmap: null pointer");
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_stage_pixmap: pixmap must have 3 or 4 channels, got %d", channels);
    TF_TRY(ensure_init());
    *pixmap_dev = nullptr;
    if (L->N == 0)
        return TF_OK;
    TF_TRY(stage_pixmap(L, pixmap, channels, beside != 0));
    *pixmap_dev = L->scratch_pix.p;
    return TF_OK;
}

TF_API int tf_remap_staged_used(tf_remap *L)
{
    TF_REQUIRE(L, "tf_remap_staged_used: null handle");
    if (L->pix_up) {
        TF_HIP(hipEventRecord(L->pix_used, stream()));
        L->pix_used_pending = true;
    }
    return TF_OK;
}

TF_API int tf_remap_render(tf_remap *L, tf_comp *comp)
{
    TF_REQUIRE(L && comp, "tf_remap_render: null handle");
    TF_REQUIRE(L->H == comp->H && L->W == comp->W, "tf_remap_render: layer is %dx%d, compositor %dx%d", L->W, L->H,
               comp->W, comp->H);
    TF_TRY(ensure_init());
    if (L->cfg.layer_class == TF_LAYER_INTRODUCTION)
        return launch("intro_render", k_intro_render, dim3(cdiv((size_t)L->N, BLOCK)), dim3(BLOCK), 0, L->cur_data(),
                      (const float *)L->mask_alpha.as<float>(), comp->image.as<uint8_t>(), L->N);
    return launch("remap_render", k_remap_render, dim3(cdiv((size_t)L->N, BLOCK)), dim3(BLOCK), 0,
                  L->rgba.as<uchar4>(), (const float *)L->mask_alpha.as<float>(), comp->image.as<uint8_t>(), L->N);
}

TF_API int tf_remap_introduce_dev(tf_remap *L, int source_index, const void *pixmap_dev, int channels, int frame_number)
{
    TF_REQUIRE(L && (pixmap_dev || L->N == 0), "tf_remap_introduce_dev: null pointer");
    TF_REQUIRE(L->cfg.layer_class == TF_LAYER_INTRODUCTION, "tf_remap_introduce: not an introduction layer");
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_introduce: pixmap must have 3 or 4 channels, got %d", channels);
    TF_REQUIRE(source_index >= 0 && source_index < L->n_sources, "tf_remap_introduce: source %d of %d (set_sources first)",
               source_index, L->n_sources);
    TF_REQUIRE(L->flow_for_intro || L->N == 0, "tf_remap_introduce: no tf_remap_update yet");
    TF_TRY(ensure_init());
    if (L->N == 0)
        return TF_OK;
    dim3 grid(cdiv((size_t)L->N, BLOCK)), block(BLOCK);
    const int consider_flow = !(L->cfg.introduce_on_all_filled_spots || L->cfg.introduce_on_all_empty_spots);
    const uint8_t *intro = L->intro.as<uint8_t>() + (size_t)L->N * source_index;
    if (channels == 4)
        return launch("intro_put_rgba", k_intro_put<4>, grid, block, 0, L->cur_data(), L->flow_for_intro,
                      (const uint8_t *)L->intro_sel.as<uint8_t>(), intro, (const uint8_t *)pixmap_dev, L->N, L->W,
                      consider_flow, source_index, frame_number, L->err.as<int>());
    return launch("intro_put_rgb", k_intro_put<3>, grid, block, 0, L->cur_data(), L->flow_for_intro,
                  (const uint8_t *)L->intro_sel.as<uint8_t>(), intro, (const uint8_t *)pixmap_dev, L->N, L->W,
                  consider_flow, source_index, frame_number, L->err.as<int>());
}

TF_API int tf_remap_introduce(tf_remap *L, int source_index, const uint8_t *pixmap, int channels, int frame_number)
{
    TF_REQUIRE(L && (pixmap || L->N == 0), "tf_remap_introduce: null pointer");
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_introduce: pixmap must have 3 or 4 channels, got %d", channels);
    TF_TRY(ensure_init());
    if (L->N == 0)
        return TF_OK;
    TF_TRY(upload(L->scratch_pix, pixmap, (size_t)L->N * channels));
    TF_TRY(tf_remap_introduce_dev(L, source_index, L->scratch_pix.p, channels, frame_number));
    TF_HIP(hipStreamSynchronize(stream())); // the host pixmap is borrowed for this call only
    return TF_OK;
}

TF_API int tf_remap_data_depth(tf_remap *L, int *depth)
{
    TF_REQUIRE(L && depth, "tf_remap_data_depth: null pointer");
    *depth = L->depth();
    return TF_OK;
}

TF_API int tf_remap_get_state(tf_remap *L, int32_t *data, uint8_t *rgba)
{
    TF_REQUIRE(L, "tf_remap_get_state: null handle");
    TF_TRY(ensure_init());
    TF_TRY(state_unpacked(L));
    size_t n = (size_t)L->N;
    if (data && n && L->depth())
        TF_HIP(hipMemcpyAsync(data, L->cur_data(), n * 4 * L->depth(), hipMemcpyDeviceToHost, stream()));
    if (rgba && n)
        TF_HIP(hipMemcpyAsync(rgba, L->rgba.p, n * 4, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_remap_set_state(tf_remap *L, const int32_t *data, const uint8_t *rgba)
{
    TF_REQUIRE(L, "tf_remap_set_state: null handle");
    TF_TRY(ensure_init());
    TF_TRY(state_unpacked(L));
    size_t n = (size_t)L->N;
    if (data && n && L->depth() == 4) { // a checkpoint may hold anything: the int16 form only for values that fit
        bool fits = true, fits32 = true;
        for (size_t q = 0; q < n * 4 && (fits || fits32); q++) {
            const int32_t v = data[q];
            fits = fits && v >= -32768 && v <= 32767;
            const int c = (int)(q & 3);
            fits32 = fits32 && v >= 0 && v <= (c < 2 ? 8191 : (c == 2 ? 1 : 31));
        }
        L->state_fits = fits;
        L->state_fits32 = fits32;
    }
    if (data && n && L->depth())
        TF_HIP(hipMemcpyAsync(L->cur_data(), data, n * 4 * L->depth(), hipMemcpyHostToDevice, stream()));
    if (rgba && n)
        TF_HIP(hipMemcpyAsync(L->rgba.p, rgba, n * 4, hipMemcpyHostToDevice, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

static bool step_fusable(const tf_remap *L) { return !L->fl.leave_empty && (L->cfg.reset_mode == 0 || L->cfg.reset_mode == 1); }

static int step_dev_impl(tf_remap *L, tf_comp *comp, const void *flow_dev, int clip_flow, const void *uniform_dev, uint64_t seed,
                         const void *pixmap_dev, int channels, const int *rgba_dead)
{
    TF_REQUIRE(L && comp && (flow_dev || L->N == 0) && (pixmap_dev || L->N == 0), "tf_remap_step_dev: null pointer");
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_step_dev: pixmap must have 3 or 4 channels, got %d", channels);
    TF_REQUIRE(L->H == comp->H && L->W == comp->W, "tf_remap_step_dev: layer is %dx%d, compositor %dx%d", L->W, L->H,
               comp->W, comp->H);
    TF_REQUIRE(L->cfg.layer_class == TF_LAYER_MOVEREF, "tf_remap_step_dev: moveref layers only");
    // reference.py:94-105 loops over every source; this call gathers source 0 alone
    TF_REQUIRE(L->n_sources == 1, "tf_remap_step_dev: the layer has %d sources; the one-call step serves exactly one "
               "(use tf_remap_update_dev + tf_remap_gather_dev per source)", L->n_sources);
    TF_TRY(ensure_init());
    if (L->N == 0)
        return TF_OK;
    TF_REQUIRE(clip_flow >= 0 && clip_flow <= 2, "tf_remap_step_dev: clip_flow must be 0, 1 or 2, got %d", clip_flow);
    const bool fusable = step_fusable(L);
    if (!fusable) { // same statements, one launch each
        if (clip_flow == 2) { // the winner map becomes a flow array first (kept in the layer's own scratch)
            if (!L->flow_scratch.p)
                TF_TRY(L->flow_scratch.alloc((size_t)L->N * 8));
            TF_TRY(launch("remap_winner_flow", k_remap_winner_flow, dim3(cdiv((size_t)L->N, BLOCK)), dim3(BLOCK), 0,
                          (const int *)flow_dev, L->flow_scratch.as<float2>(), L->W, L->H));
            flow_dev = L->flow_scratch.p;
        } else if (clip_flow)
            TF_TRY(launch("remap_clip_flow", k_remap_clip_flow, dim3(cdiv((size_t)L->N, BLOCK)), dim3(BLOCK), 0,
                          (float2 *)const_cast<void *>(flow_dev), L->W, L->H));
        TF_TRY(tf_remap_update_dev(L, flow_dev, uniform_dev, seed));
        TF_TRY(tf_remap_gather_dev(L, 0, pixmap_dev, channels));
        TF_TRY(tf_comp_begin(comp));
        return tf_remap_render(L, comp);
    }
    StepParams sp;
    sp.fl = L->fl;
    sp.clip_flow = clip_flow;
    sp.div = fast_div_setup((uint32_t)L->W);
    sp.reset_random = L->cfg.reset_mode == 1;
    sp.factor = (float)L->cfg.reset_random_factor;
    sp.reset_source = L->cfg.reset_source;
    sp.n_sources = L->n_sources;
    sp.seed = seed;
    sp.frame = L->frame;
    sp.bg = comp->bg;
    sp.rgba_dead = rgba_dead;
    const int px_per_thread = (int)option(OPT_REMAP_PX);
    dim3 block(BLOCK);
    auto run = [&](auto *old, auto *neu) {
        using S = typename std::remove_const<typename std::remove_pointer<decltype(old)>::type>::type;
        auto go = [&](auto kernel, const char *name, int px) {
            return launch(name, kernel, dim3(cdiv((size_t)L->N, (size_t)px * BLOCK)), block, 0, (const float2 *)flow_dev, old, neu,
                          (const uint8_t *)L->mask_src.as<uint8_t>(), (const uint8_t *)L->mask_dst.as<uint8_t>(),
                          (const double *)uniform_dev, (const float *)L->reset_mask.as<float>(),
                          (const uint8_t *)L->intro.as<uint8_t>(), L->rgba.as<uchar4>(), (const uint8_t *)pixmap_dev,
                          (const float *)L->mask_alpha.as<float>(), comp->image.as<uint8_t>(), L->N, L->H, L->W, sp,
                          L->err.as<int>());
        };
        if (px_per_thread >= 4)
            return channels == 4 ? go(k_remap_step_px<4, S, 4>, "remap_step_rgba", 4)
                                 : go(k_remap_step_px<3, S, 4>, "remap_step_rgb", 4);
        if (px_per_thread >= 2)
            return channels == 4 ? go(k_remap_step_px<4, S, 2>, "remap_step_rgba", 2)
                                 : go(k_remap_step_px<3, S, 2>, "remap_step_rgb", 2);
        return channels == 4 ? go(k_remap_step<4, S>, "remap_step_rgba", 1) : go(k_remap_step<3, S>, "remap_step_rgb", 1);
    };
    if (const int kind = state_can_pack(L)) {
        TF_TRY(state_packed(L, kind));
        if (kind == 2)
            TF_TRY(run((const packed32 *)L->pdata[L->pcur].as<packed32>(), L->pdata[L->pcur ^ 1].as<packed32>()));
        else
            TF_TRY(run((const short4s *)L->pdata[L->pcur].as<short4s>(), L->pdata[L->pcur ^ 1].as<short4s>()));
        L->pcur ^= 1;
        L->frame++;
        return TF_OK;
    }
    TF_TRY(state_unpacked(L));
    TF_TRY(run((const int4 *)L->data[L->cur].as<int4>(), L->data[L->cur ^ 1].as<int4>()));
    L->cur ^= 1;
    L->frame++;
    return TF_OK;
}

TF_API int tf_remap_step_dev(tf_remap *L, tf_comp *comp, const void *flow_dev, int clip_flow, const void *uniform_dev,
                             uint64_t seed, const void *pixmap_dev, int channels)
{
    return step_dev_impl(L, comp, flow_dev, clip_flow, uniform_dev, seed, pixmap_dev, channels, nullptr);
}

// n consecutive tf_remap_step_dev calls in one: step i takes flows[i], paints comps[i] from pixmaps[i] (and draws from
// uniforms[i], if given).  Same results, state and frames, as the n calls.  What the one call can do that n cannot: it
// knows which stores of a step nobody will read.  The layer's rgba (reference.py:93-105) is read by the one-kernel step
// only at pixels that source 0 does NOT select (alpha 0, or another source's index: they keep their previous colour),
// and a step stores it at every pixel.  If every pixel is selected before the first step -- checked on the device, one
// pass over the state -- it stays so through these steps (a move copies a selected pixel's state, the random reset of a
// one-source layer writes alpha 1 and leaves the index 0; leave_empty and the other reset modes do not take the
// one-kernel step at all), so each step's rgba is overwritten whole by the next without having been read: steps
// 0 .. n-2 do not store it (4 of the step's 30 bytes per pixel), step n-1 does, and the layer leaves the call as the n
// calls leave it.
TF_API int tf_remap_steps_dev(tf_remap *L, int n, tf_comp *const *comps, const void *const *flows_dev, int clip_flow,
                              const void *const *uniforms_dev, uint64_t seed, const void *const *pixmaps_dev, int channels)
{
    TF_REQUIRE(L && n >= 0 && (n == 0 || (comps && flows_dev && pixmaps_dev)), "tf_remap_steps_dev: null argument");
    TF_TRY(ensure_init());
    const int *dead = nullptr;
    if (n >= 2 && L->N > 0 && L->cfg.layer_class == TF_LAYER_MOVEREF && L->n_sources == 1 && step_fusable(L) &&
        option(OPT_REMAP_KEEP_RGBA) == 0) {
        if (!L->sel_flag.p)
            TF_TRY(L->sel_flag.alloc(sizeof(int)));
        TF_HIP(hipMemsetD32Async((hipDeviceptr_t)L->sel_flag.p, 1, 1, main_stream()));
        const dim3 grid((unsigned)std::min<size_t>(cdiv((size_t)L->N, BLOCK), 4096)), block(BLOCK);
        if (const int kind = state_can_pack(L)) { // the form the steps will keep the state in
            TF_TRY(state_packed(L, kind));
            if (kind == 2)
                TF_TRY(launch("remap_all_selected", k_state_all_selected<packed32>, grid, block, 0,
                              (const packed32 *)L->pdata[L->pcur].as<packed32>(), L->N, L->sel_flag.as<int>()));
            else
                TF_TRY(launch("remap_all_selected", k_state_all_selected<short4s>, grid, block, 0,
                              (const short4s *)L->pdata[L->pcur].as<short4s>(), L->N, L->sel_flag.as<int>()));
        } else {
            TF_TRY(state_unpacked(L));
            TF_TRY(launch("remap_all_selected", k_state_all_selected<int4>, grid, block, 0, (const int4 *)L->cur_data(), L->N,
                          L->sel_flag.as<int>()));
        }
        dead = L->sel_flag.as<int>();
    }
    for (int i = 0; i < n; i++)
        TF_TRY(step_dev_impl(L, comps[i], flows_dev[i], clip_flow, uniforms_dev ? uniforms_dev[i] : nullptr, seed, pixmaps_dev[i],
                             channels, i + 1 < n ? dead : nullptr));
    return TF_OK;
}
