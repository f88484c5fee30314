// B1: FlowSource.post_process (source.py:337-363) on the device -- the clip, the FORWARD scatter (last write wins) and
// its resolve -- and the flow filters / mask multiply that precede it (filters.py:36-72).
#include "fb_common.h"

namespace {

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

} // namespace

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
