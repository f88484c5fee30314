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
#include "fb_common.h"

// ---- host-side constant preparation ----------------------------------------------
static inline int cv_round(double v) { return (int)lrint(v); }

static std::vector<float> gaussian_kernel(int n, double sigma)
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
static void invert6(const double G[36], double inv[36])
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

static PolyConst make_poly_const(int n, double sigma)
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

namespace tf {
namespace fb {

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

// Profiler labels: with option "prof_levels" = 1 every Farneback launch is
// labelled with its pyramid level ("fb_polyexp.k2"), otherwise by kernel only.
const char *lvl_name(const char *base, int k)
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

} // namespace fb
} // namespace tf

// read when a handle is created: option "fb_no_overlap" = 1 keeps everything on the library stream
static bool fb_overlap_enabled() { return option(OPT_FB_NO_OVERLAP) == 0; }

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
            L->quarter = plan_quarter_level(width, height, *L);
            L->split = !L->quarter && plan_split_level(width, height, *L, colsrc);
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
            int pitch = (fb->rp_r4 + width + rmax + 16 + 3) & ~3; // (the unrolled row pass reads whole aligned dwords up to 10 bytes past a window)
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
        const bool fused_here = fusable && !fb->gaussian() && exact_one_kernel && L.W >= 10 && L.H >= 10 && (size_t)L.W * L.H < (1u << 28) && L.W < (1 << 24) && L.H < (1 << 24) &&
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
