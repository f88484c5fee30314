// Single-stage entry points (tf_fb_stage_*): the parity tests drive one kernel at a time through these and compare with
// the oracle's stage of the same name.
#include "fb_common.h"

namespace {

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

} // namespace

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
    TF_TRY(fb_m_room(fb, w, h, 1));
    TF_TRY(upload_planar5(fb->M.as<float>(), m, n, fb->scratch));
    if (fb->gaussian()) // FarnebackUpdateFlow_GaussianBlur's window on a handle created with flags & 256
        TF_TRY(fb_gauss_solve(fb, w, h, 1, fb->lflow[0].as<float2>()));
    else
        TF_TRY(fb_blur_solve(fb, w, h, 1, fb->lflow[0].as<float2>()));
    TF_HIP(hipMemcpyAsync(flow_out, fb->lflow[0].p, n * 8, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}
