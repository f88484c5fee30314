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


#include "remap_common.h"

using namespace tf;
using namespace tf::remap;

namespace {

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

TF_API int tf_comp_download_begin(tf_comp *comp, uint8_t *rgb_out)
{
    TF_REQUIRE(comp && rgb_out, "tf_comp_download_begin: null pointer");
    TF_TRY(ensure_init());
    TF_TRY(tf_comp_download_end(comp)); // one transfer per image at a time
    if (!comp->N)
        return TF_OK;
    if (!comp->image_ready) {
        TF_HIP(hipEventCreateWithFlags(&comp->image_ready, hipEventDisableTiming));
        TF_HIP(hipEventCreateWithFlags(&comp->download_done, hipEventDisableTiming));
    }
    hipStream_t down;
    TF_TRY(tf::side_stream(4, &down));
    TF_HIP(hipEventRecord(comp->image_ready, stream()));
    TF_HIP(hipStreamWaitEvent(down, comp->image_ready, 0));
    TF_HIP(hipMemcpyAsync(rgb_out, comp->image.p, (size_t)comp->N * 3, hipMemcpyDeviceToHost, down));
    TF_HIP(hipEventRecord(comp->download_done, down));
    comp->download_pending = true;
    return TF_OK;
}

TF_API int tf_comp_download_end(tf_comp *comp)
{
    TF_REQUIRE(comp, "tf_comp_download_end: null handle");
    if (comp->download_pending) {
        TF_HIP(hipEventSynchronize(comp->download_done));
        comp->download_pending = false;
    }
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
