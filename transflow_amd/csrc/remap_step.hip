// The resident path's remap step as ONE kernel per frame (tf_remap_step_dev, tf_remap_steps_dev) and the compact forms it
// keeps the layer state in between its own launches.  Reference semantics as remap.hip's: movement.py:20-60,
// reference.py:58-105, layer.py:32-34, compositor.py:31-40, source.py:359-362.
#include "remap_common.h"

using namespace tf;
using namespace tf::remap;

namespace {

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

} // namespace

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
int tf::remap::state_unpacked(tf_remap *L)
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
    // every step's arguments are checked BEFORE the first step runs: once stores of the layer's rgba have been left out
    // (below) no later step may fail on an argument and leave the layer's state ahead of its rgba
    TF_REQUIRE(channels == 3 || channels == 4, "tf_remap_steps_dev: pixmap must have 3 or 4 channels, got %d", channels);
    TF_REQUIRE(clip_flow >= 0 && clip_flow <= 2, "tf_remap_steps_dev: clip_flow must be 0, 1 or 2, got %d", clip_flow);
    TF_REQUIRE(n == 0 || L->cfg.layer_class == TF_LAYER_MOVEREF, "tf_remap_steps_dev: moveref layers only");
    TF_REQUIRE(n == 0 || L->n_sources == 1, "tf_remap_steps_dev: the layer has %d sources; the one-call step serves exactly one",
               L->n_sources);
    for (int i = 0; i < n; i++) {
        TF_REQUIRE(comps[i] && (L->N == 0 || (flows_dev[i] && pixmaps_dev[i])), "tf_remap_steps_dev: null pointer in step %d", i);
        TF_REQUIRE(L->H == comps[i]->H && L->W == comps[i]->W, "tf_remap_steps_dev: layer is %dx%d, step %d's compositor %dx%d",
                   L->W, L->H, i, comps[i]->W, comps[i]->H);
    }
    TF_TRY(ensure_init());
    const int *dead = nullptr;
    if (n >= 2 && L->N > 0 && L->cfg.layer_class == TF_LAYER_MOVEREF && L->n_sources == 1 && step_fusable(L) &&
        option(OPT_REMAP_KEEP_RGBA) == 0) {
        if (!L->sel_flag.p)
            TF_TRY(L->sel_flag.alloc(sizeof(int)));
        TF_HIP(hipMemsetD32Async((hipDeviceptr_t)L->sel_flag.p, 1, 1, stream())); // the stream the check and the steps run on
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
