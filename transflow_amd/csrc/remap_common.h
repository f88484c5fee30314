// What the two units of the compositor remap share: the layer and compositor-image handles, the move flags, the
// generator of the random reset's uniform field (remap.hip: the layers and their entry points; remap_step.hip: the
// one-kernel step of the resident path and the state formats it keeps between its launches).
#pragma once
#include <cstring>
#include <type_traits>

#include "common.h"

namespace tf {
namespace remap {

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

} // namespace remap
} // namespace tf

struct tf_comp {
    using DevBuf = tf::DevBuf;
    int H, W, N;
    uchar4 bg;
    DevBuf image;
    // tf_comp_download_begin / _end: the image on its way down on the library's download stream
    hipEvent_t image_ready = nullptr, download_done = nullptr;
    bool download_pending = false;
    ~tf_comp()
    {
        if (download_pending)
            (void)hipEventSynchronize(download_done);
        if (image_ready)
            (void)hipEventDestroy(image_ready);
        if (download_done)
            (void)hipEventDestroy(download_done);
    }
};

struct tf_remap {
    using DevBuf = tf::DevBuf;
    int H, W, N;
    tf_layer_cfg cfg;
    tf::remap::MoveFlags fl;
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

namespace tf {
namespace remap {
// remap_step.hip: makes data[cur] (int32 x 4, the reference's layout) the current state again after one-kernel steps
int state_unpacked(tf_remap *L);
} // namespace remap
} // namespace tf
