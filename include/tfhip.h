/*
 * tfhip.h -- C ABI of libtfhip.so: the MI355X (gfx950) backend for transflow's
 * per-frame hot loop (Farnebäck dense optical flow + compositor remap).
 *
 * This is the drop-in boundary.  Plain C types only: pointers,
 * sizes, POD structs.  Host pointers are borrowed for the duration of a call;
 * the library owns all device memory, tied to the opaque handles.  Every
 * function returns TF_OK (0) or a negative tf_status; the message of the last
 * failure on the calling thread is tf_last_error().  No HIP call happens at
 * load time: the first tf_init()/tf_*_create() initialises the runtime (the
 * reference runs the flow source in a forked child, pipeline.py:56-64).
 *
 * Each entry point cites the reference interface it replaces; paths are
 * relative to the reference tree (ychalier/transflow v1.11.1).
 */
#ifndef TFHIP_H
#define TFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFHIP_ABI_VERSION 1

typedef enum tf_status {
    TF_OK = 0,
    TF_ERR_ARG = -1,     /* bad argument (ValueError on the Python side) */
    TF_ERR_HIP = -2,     /* a HIP runtime call failed (RuntimeError) */
    TF_ERR_INDEX = -3,   /* a flow vector leaves the frame: the reference raises IndexError
                            at compositor/layers/movement.py:33,39 */
    TF_ERR_STATE = -4,   /* call order / handle state */
    TF_ERR_UNSUPPORTED = -5
} tf_status;

/* ---- runtime ------------------------------------------------------------------ */
int tf_abi_version(void);
/* Select the GPU for this process/thread and create the library stream.  Idempotent. */
int tf_init(int device);
/* 1 once this process has initialised the HIP runtime through the library (a process forked after
   that point cannot use the GPU: fork first, as transflow/pipeline.py does). */
int tf_is_initialized(void);
int tf_device_count(int *count);
/* Run-time options of the library (process-wide; defaults are the measured best):
     "fb_fused"       -1  the Farnebäck iteration as one kernel on levels of >= fb_fuse_min_px pixels over the
                          batch, as two kernels below; 0 = two kernels everywhere, 1 = one kernel wherever it
                          can run (read by tf_fb_create)
     "fb_fuse_min_px" 4000000
     "fb_no_share"    0   1 = pairs of a call that share a frame expand it once each (read per call)
     "fb_no_overlap"  0   1 = a call's kernels stay on the library stream (read by tf_fb_create)
     "remap_px"       4   pixels per thread of tf_remap_step_dev's kernel: 1, 2 or 4
     "remap_no_pack"  0   0 = tf_remap_step_dev keeps the layer state as ONE 32-bit word per pixel between steps where row,
                          column, alpha and source index fit 13 + 13 + 1 + 5 bits (frames up to 8192 x 8192, 32
                          sources), as int16 x 4 otherwise; 2 = as int16 x 4 at most; 1 = as int32 x 4 (never packed)
     "remap_keep_rgba" 0  1 = tf_remap_steps_dev stores the layer's rgba in every step (it stores it in the last one alone where
                          every pixel is selected by source 0: nothing reads the others')
     "prof_levels"    0   1 = profiler labels carry the pyramid level
     "fb_exact_sums"  0   1 = the box window (flags without OPTFLOW_FARNEBACK_GAUSSIAN) is summed exactly as
                          FarnebackUpdateFlow_Blur sums it -- one set of running sums per image, float-differenced
                          down the columns from row 0, double-differenced along the rows from column 0 -- so the
                          flow is bit-identical to the CPU path's instead of within 1e-4 of it; about twice the
                          default's time, plus 40 bytes per pixel and pair of device memory (read per call)
     "fb_chain"       -1  the marching kernels keep FarnebackUpdateFlow_Blur's column sums (one running sum per column
                          from row 0).  A column cut into row segments needs the sum's value at each cut: 1 = every
                          segment waits for the one above it inside the launch (the sums are then the CPU path's bit
                          for bit), 0 = a pre-pass computes the values (equal up to the association of double
                          additions, ~1e-16), -1 = whichever is cheaper for the launch (read per call)
     "fb_segs"        0   > 0: that many row segments per column (0: chosen per launch; read per call)
   Unknown names and out-of-range values return TF_ERR_ARG.  The environment is never read. */
int tf_set_option(const char *name, long value);
int tf_get_option(const char *name, long *value);
const char *tf_last_error(void);
/* Block until everything queued on the library stream has finished. */
int tf_sync(void);
/* The hipStream_t all kernels of this library are launched on (for callers that
   record their own events or enqueue RCCL work in order with it). */
int tf_stream(void **hip_stream);

/* Events on the library stream (HIP events; bench.py times with these). */
typedef struct tf_event tf_event;
int tf_event_create(tf_event **ev);
int tf_event_record(tf_event *ev);
int tf_event_elapsed_ms(tf_event *start, tf_event *stop, float *ms); /* synchronises on stop */
void tf_event_destroy(tf_event *ev);
/* A flow that stays on the device between the flow source and the compositor (the seam transflow/pipeline.py:85-86,
   326, 562-567 crosses with a pickled host array): the producer records an event behind the flow's last kernel on ITS
   stream (tf_event_record), the consumer's stream waits for it on the device before its kernels read the flow
   (tf_stream_wait_event: no host synchronisation), or the host waits for it before the memory is handed to another
   process (tf_event_synchronize). */
int tf_stream_wait_event(tf_event *ev);
int tf_event_synchronize(tf_event *ev);
/* The same seam across the reference's process boundary (pipeline.py:56-64: the flow source is a child process, its
   items travel through a multiprocessing queue): a device allocation of this process (tf_dev_alloc) exported as 64
   opaque bytes (hipIpcGetMemHandle), opened in the consumer's process (hipIpcOpenMemHandle; one mapping per allocation,
   kept until tf_ipc_close) -- what crosses the queue is the 64 bytes instead of 66 MB per 4K flow.  The exporter
   keeps the allocation alive and untouched until the consumer has read it. */
#define TF_IPC_HANDLE_BYTES 64
int tf_ipc_export(void *dev, void *handle_out);
int tf_ipc_open(const void *handle, void **dev);
int tf_ipc_close(void *dev);

/* Per-kernel timing: when enabled every launch is bracketed by HIP events on the
   library stream.  tf_prof_report writes lines "name count total_ms" into buf. */
int tf_prof_enable(int on);
/* Only time kernels whose name contains `substring` (NULL or "" = all). */
int tf_prof_set_filter(const char *substring);
int tf_prof_reset(void);
int tf_prof_report(char *buf, size_t buf_size);

/* Page-locked host memory (hipHostMalloc): arrays the drop-in path hands across the boundary come from a pool of
   these, so that the copies behind tf_fb_get_flow / tf_comp_download (transflow/flow/sources/cv.py:490's result,
   transflow/output/ffmpeg.py:32-54's frame) and the upload of a flow into the compositor run at the link's rate
   instead of through the driver's staging of pageable memory. */
int tf_host_alloc(void **host, size_t bytes);
int tf_host_free(void *host);
/* The calling THREAD's library stream: 0 = the library stream (default), 1..3 = one of three further streams.
   Everything the thread then queues through the library -- uploads, a flow source's calls and downloads -- is ordered
   on that stream and runs beside other threads' work (the reference runs its flow source in a process of its own,
   transflow/pipeline.py:56-64; a prefetching flow source in a worker thread gets the same concurrency here). */
int tf_thread_stream(int which);

/* Raw device buffers, for harnesses that keep inputs resident in HBM. */
int tf_dev_alloc(void **dev, size_t bytes);
int tf_dev_free(void *dev);
int tf_dev_upload(void *dev, const void *host, size_t bytes);
int tf_dev_download(void *host, const void *dev, size_t bytes);
int tf_dev_copy(void *dst_dev, const void *src_dev, size_t bytes); /* device to device, on the library stream */
/* One 64-bit word stored at `dev` (8-byte aligned) in stream order on the calling thread's library stream: the
   generation word at the head of a flow buffer that crosses the reference's queue as an IPC token
   (transflow/pipeline.py:85-86, 326) -- bumped before the buffer is written again, compared by the consumer after
   its copy, so that a buffer rewritten too early raises instead of passing for the flow it no longer holds. */
int tf_dev_store_u64(void *dev, uint64_t value);
/* The same as one kernel, 16 bytes per lane: bench.py measures the practical HBM ceiling with it. */
int tf_dev_stream_copy(void *dst_dev, const void *src_dev, size_t bytes);

/* ---- Farnebäck dense optical flow ----------------------------------------------
 * Replaces cv2.calcOpticalFlowFarneback as called at
 * transflow/flow/sources/cv.py:479-490; parameter struct = the fb_* fields of
 * CvFlowConfig (cv.py:273-281).  flags (fb_flags, passed straight through at cv.py:489): 0 (transflow's
 * default), OPTFLOW_USE_INITIAL_FLOW = 4 (the flow array is read first: cv.py:478 fills it with the
 * previous output), OPTFLOW_FARNEBACK_GAUSSIAN = 256 (Gaussian instead of box window), or both.
 */
#define TF_OPTFLOW_USE_INITIAL_FLOW 4
#define TF_OPTFLOW_FARNEBACK_GAUSSIAN 256
typedef struct tf_fb_params {
    double pyr_scale;  /* fb_pyr_scale  (0.5)  */
    int levels;        /* fb_levels     (3)    */
    int winsize;       /* fb_winsize    (15)   */
    int iterations;    /* fb_iterations (3)    */
    int poly_n;        /* fb_poly_n     (5)    */
    double poly_sigma; /* fb_poly_sigma (1.2)  */
    int flags;         /* fb_flags      (0)    */
} tf_fb_params;

typedef struct tf_fb tf_fb;

/* frame_slots: how many uint8 grey frames the handle keeps resident in HBM;
   max_pairs: how many frame pairs one tf_fb_calc_slots call may process. */
int tf_fb_create(tf_fb **out, int width, int height, const tf_fb_params *params, int frame_slots, int max_pairs);
/* A second lane for first's calls (no counterpart in the reference, whose one call per frame is synchronous,
   cv.py:479-490): a handle of the same size and parameters that READS FIRST'S FRAME SLOTS and queues its calls on the
   library's other call stream.  Batches sent alternately to the two handles are in flight together, so the launches of
   one that leave the chip part-empty (coarse pyramid levels, the tail of every launch) run beside those of the other.
   Everything else is per handle: results are read from the handle that ran the call.  `first` destroyed while lanes
   still read its slots is released with the last of them (it must not be used after its own tf_fb_destroy).
   Not with tf_fb_keep_expansions.  A lane starts with first's tf_fb_set_exact mode. */
int tf_fb_create_lane(tf_fb **out, tf_fb *first);
void tf_fb_destroy(tf_fb *fb);
/* Which summation this HANDLE's calls use for the box window (cv.py:479-490: one call, one result; nothing outside
   the handle decides it): 1 = OpenCV's own order along the rows too (flow bit-identical to the CPU path's, ~2x the
   time), 0 = the default (window added across columns directly, every pixel within 1e-4 relative), -1 = whatever the
   process-wide option "fb_exact_sums" says when a call is issued (the state a new handle is in).  Takes effect with the
   next call; two handles of one process may differ, whichever threads drive them. */
int tf_fb_set_exact(tf_fb *fb, int mode);

/* One pair, host in / host out: flow_out is float32 [height][width][2] (x=dx, y=dy),
   exactly the array cv.py:479-490 produces.  Strides in bytes.  With TF_OPTFLOW_USE_INITIAL_FLOW the
   array is cv2's in/out `flow`: read as the initial flow, then overwritten with the result. */
int tf_fb_calc(tf_fb *fb, const uint8_t *prev, ptrdiff_t prev_stride, const uint8_t *next, ptrdiff_t next_stride,
               float *flow_out);

/* Resident path: upload grey frames into slots, compute n pairs in one pass
   (pairs are independent with flags == 0: cv.py:478,489), read results back or
   hand the device pointer on.  A slot named by several pairs of one call (consecutive pairs of a
   video share a frame) is expanded once for all of them; nothing is kept between calls.
   tf_fb_calc_slots only enqueues, on a stream of the library's
   own that does not wait for what the library stream was given last (the caller's remap of the
   previous result keeps running beside it), so a caller that fills frame slots on the device itself (through tf_fb_frame_ptr) calls tf_sync() before the next tf_fb_calc_slots;
   tf_fb_set_frame already returns with the frame in place. */
int tf_fb_set_frame(tf_fb *fb, int slot, const uint8_t *grey, ptrdiff_t stride);
/* The same from a decoded BGR frame, transflow/flow/sources/cv.py:461-466: uint8 [src_height][src_width][3]
   (row stride in bytes) is uploaded as it is and cv2.resize(INTER_NEAREST) to the handle's size +
   cv2.cvtColor(COLOR_BGR2GRAY) run on the device (tf_frame_grey_dev) straight into the slot. */
int tf_fb_set_frame_bgr(tf_fb *fb, int slot, const uint8_t *bgr, int src_width, int src_height, ptrdiff_t stride);
/* Resident path with TF_OPTFLOW_USE_INITIAL_FLOW: the initial flow of `pair` for the next tf_fb_calc_slots
   (float32 [height][width][2]; kept until written again, zero at creation), from the host or -- through
   its device address -- filled on the GPU (e.g. copied from a previous result). */
int tf_fb_set_initial_flow(tf_fb *fb, int pair, const float *flow);
int tf_fb_initial_flow_ptr(tf_fb *fb, int pair, void **dev);
/* Streaming use (cv.py:460-490 holds prev_gray and reads one new frame per call): with `on`, the
   pyramid levels and polynomial expansion of a slot are kept from call to call and redone only
   after tf_fb_set_frame wrote the slot, so the frame that was "next" in one call costs nothing as
   "prev" in the following one.  Slots handed out by tf_fb_frame_ptr are expanded on every call.
   Needs frame_slots <= 2 * max_pairs.  Off by default: a call then depends on nothing but the frames. */
int tf_fb_keep_expansions(tf_fb *fb, int on);
int tf_fb_frame_ptr(tf_fb *fb, int slot, void **dev);
int tf_fb_calc_slots(tf_fb *fb, int n_pairs, const int *prev_slots, const int *next_slots);
int tf_fb_get_flow(tf_fb *fb, int pair, float *flow_out);
/* Streaming callers, one new frame and one flow per call (cv.py:460-490): with async io on, tf_fb_set_frame / _bgr put
   the frame up on a copy stream of the library's (they still return with the frame in place, but do not wait for the
   handle's kernels in flight), and tf_fb_get_flow_begin / _end bring the last call's flow down on another copy stream:
   _begin queues the copy behind what the caller's stream holds at that moment (its post_process), *token names the
   transfer; _end returns once flow_out is filled.  So frame t + 1 goes up and flow t - 1 comes down while pair t is being
   computed.  flow_out should be page-locked (tf_host_alloc) and stay untouched between _begin and _end. */
int tf_fb_async_io(tf_fb *fb, int on);
int tf_fb_get_flow_begin(tf_fb *fb, int pair, float *flow_out, int *token);
int tf_fb_get_flow_end(tf_fb *fb, int token);
int tf_fb_flow_ptr(tf_fb *fb, int pair, void **dev);

/* FlowSource.post_process (transflow/flow/sources/source.py:337-363) without the
   optional filter/mask/kernel pre-steps: direction 0 = FORWARD (clip, round,
   scatter-invert with last-write-wins, :349-360), 1 = BACKWARD; both end with the
   clip to the frame (:361-362).  In place. */
int tf_fb_post_process(tf_fb *fb, int pair, int direction);                 /* device flow of `pair` */
/* The first half of FORWARD post_process alone (source.py:349-358: clip, round, every moving source
   claims its target, the highest source index wins): *winners_dev = int32 [H][W], the winning source
   index per target or -1.  The handle owns the map; the next scatter or post_process overwrites it.
   tf_remap_step_dev(clip_flow = 2) takes it in place of the flow and does the rest (:359-362) in
   registers, which saves writing and re-reading the flow. */
int tf_fb_post_process_scatter(tf_fb *fb, int pair, void **winners_dev);
int tf_fb_post_process_host(tf_fb *fb, float *flow_inout, int direction);   /* host array, same H,W */

/* The optional pre-steps of post_process (source.py:339-345), applied before the direction
   handling: flow filters `scale`, `threshold`, `clip` (transflow/flow/filters.py:36-72) with the
   value the filter's lambda returned for this frame, then the flow mask multiply.  `wide` tells
   how numpy typed that value: 0 = Python float/int (NEP 50 weak scalar: arithmetic in float32),
   1 = numpy.float64 (arithmetic in float64, result cast to float32).  mask: float32 [H][W] or NULL. */
typedef enum tf_flow_op_kind { TF_FLOW_SCALE = 0, TF_FLOW_THRESHOLD = 1, TF_FLOW_CLIP = 2 } tf_flow_op_kind;
typedef struct tf_flow_op {
    int kind;
    int wide;
    double value;
} tf_flow_op;
#define TF_MAX_FLOW_OPS 8
int tf_fb_post_process_ex(tf_fb *fb, int pair, int direction, int n_ops, const tf_flow_op *ops, const void *mask_dev);
int tf_fb_post_process_host_ex(tf_fb *fb, float *flow_inout, int direction, int n_ops, const tf_flow_op *ops,
                               const float *mask); /* direction -1: pre-steps only */

/* ---- flow-array steps either side of the path (device pointers, no handle) ---------
 * Flows are float32 [H][W][2] unless `wide` says float64.  All launches go to the library stream. */

/* Pipeline.FLOW_MERGING_FUNCTIONS (transflow/pipeline.py:149-158; helpers transflow/utils.py:359-381):
   n flows of n_values floats each, combined left to right in float32 as numpy does. */
enum { TF_MERGE_FIRST = 0, TF_MERGE_SUM, TF_MERGE_AVERAGE, TF_MERGE_DIFFERENCE, TF_MERGE_PRODUCT, TF_MERGE_MASKBIN,
       TF_MERGE_MASKLIN, TF_MERGE_ABSMAX };
#define TF_MAX_MERGE 8
int tf_flow_merge_dev(int kind, int n, const void *const *flows_dev, void *out_dev, size_t n_values);

/* utils.upscale_array (utils.py:417-418): out [H*hf][W*wf][2] = (x*wf, y*hf) of the nearest source pixel. */
int tf_flow_upscale_dev(const void *in_dev, void *out_dev, int width, int height, int wf, int hf);

/* The convolution-kernel pre-step of post_process (source.py:344-348):
   scipy.signal.convolve2d(channel, kernel, mode="same", boundary="fill", fillvalue=0) on both
   channels.  wide = 1: kernel float64 [kh][kw], out float64 [H][W][2] (numpy.result_type of a float32
   flow with a float64 or integer kernel); wide = 0: kernel and out float32. */
int tf_flow_convolve_dev(const void *flow_dev, const void *kernel_dev, int kh, int kw, int wide, void *out_dev, int width,
                         int height);

/* source.py:349-362 on a flow of either type, in place (after a convolution the reference carries
   on in the convolution's type).  FORWARD needs scratch_dev: 4 bytes per pixel. */
int tf_flow_post_process_dev(void *flow_dev, int wide, int width, int height, int direction, void *scratch_dev);

/* The `polar` flow filter, transflow/flow/filters.py:75-87, in place on a float32 flow: r = |v|,
   a = atan2(vy, vx), then v = (R cos A, R sin A) with R and A the filter's two user expressions of
   (t, r, a).  The expressions arrive as postfix programs (transflow_amd/exprs.py compiles them; parts
   that are not arrays are evaluated on the host per frame and arrive as constants); a step computes in
   float32 unless `wide`, as numpy types the same expression.  wide_trig: sin/cos of A in float64;
   wide_product: the products in float64 (then cast to the float32 flow). */
typedef struct tf_polar_step {
    int op;      /* enum PolarOp of flowops.hip == index into transflow_amd/exprs.py OPS */
    int wide;
    double imm;  /* push_const only */
} tf_polar_step;
#define TF_MAX_POLAR_STEPS 48
#define TF_MAX_POLAR_STACK 12
int tf_flow_polar_dev(void *flow_dev, size_t n_pixels, int n_radius, const tf_polar_step *radius, int n_theta,
                      const tf_polar_step *theta, int wide_trig, int wide_product);

/* Flow visualisation, transflow/output/render.py:9-27 and :30-48: arr float32 [n] / flow float32
   [n][2] -> rgb uint8 [n][3].  colors_rgb: 2 (render1d) or 4 (render2d) colours as float triples. */
int tf_flow_render1d_dev(const void *arr_dev, void *rgb_dev, size_t n, float scale, const float colors_rgb[6], int binary);
int tf_flow_render2d_dev(const void *flow_dev, void *rgb_dev, size_t n, float scale, const float colors_rgb[12]);

/* Frame ingest, transflow/flow/sources/cv.py:461-466: cv2.resize(INTER_NEAREST) to (width, height)
   then cv2.cvtColor(COLOR_BGR2GRAY), fused: bgr uint8 [src_height][src_width][3] -> grey uint8
   [height][width].  OpenCV's arithmetic is recalled, not verifiable here (parity unpinned). */
int tf_frame_grey_dev(const void *bgr_dev, int src_width, int src_height, void *grey_dev, int width, int height);

/* Stage-level entry points (debug/parity tests): run one stage of the pyramid on
   host arrays with exactly the kernels the full path uses.  Layouts as OpenCV's:
   R and M are [H][W][5] interleaved on the host side. */
int tf_fb_stage_level_image(tf_fb *fb, const uint8_t *grey, ptrdiff_t stride, int level, float *out /*[Hk][Wk]*/);
int tf_fb_stage_polyexp(tf_fb *fb, const float *img, int w, int h, float *r_out /*[h][w][5]*/);
/* A1 then A2 of one frame at one level, through whichever kernels the full path uses there
   (level 0 runs them fused): r_out [Hk][Wk][5]. */
int tf_fb_stage_level_polyexp(tf_fb *fb, const uint8_t *grey, ptrdiff_t stride, int level, float *r_out);
int tf_fb_stage_update_matrices(tf_fb *fb, const float *r0, const float *r1, const float *flow, int w, int h,
                                float *m_out /*[h][w][5]*/);
/* A5 then A3 at `level` (0 <= level < K) as the pyramid runs them: coarse_flow [Hc][Wc][2] of level + 1 is
   upsampled (resize INTER_LINEAR to the level's size, times 1/pyr_scale) inside the matrix kernel. */
int tf_fb_stage_upsampled_matrices(tf_fb *fb, int level, const float *r0, const float *r1, const float *coarse_flow,
                                   float *m_out /*[Hk][Wk][5]*/);
/* A4: the box window (FarnebackUpdateFlow_Blur) -- or, on a handle with TF_OPTFLOW_FARNEBACK_GAUSSIAN, the
   Gaussian one (FarnebackUpdateFlow_GaussianBlur) -- and the 2x2 solve. */
int tf_fb_stage_blur_solve(tf_fb *fb, const float *m, int w, int h, float *flow_out /*[h][w][2]*/);
/* The first step of TF_OPTFLOW_USE_INITIAL_FLOW: flow [H][W][2] -> [Hc][Wc][2] of the coarsest scale,
   resize(INTER_AREA) times pyr_scale^K. */
int tf_fb_stage_initial_flow(tf_fb *fb, const float *flow, float *coarse_out);
int tf_fb_level_count(tf_fb *fb, int *n_scales); /* K+1 */
int tf_fb_level_size(tf_fb *fb, int level, int *w, int *h);

/* ---- compositor layers -----------------------------------------------------------
 * One handle = one layer of the compositor.  layer_class selects which of the reference's
 * layer classes it is (Layer.from_args, transflow/compositor/layers/layer.py:44-56):
 *   TF_LAYER_MOVEREF       MoveReferenceLayer (move_reference.py:6-14): MovementLayer.update
 *                          (movement.py:20-64) then ReferenceLayer.update (reference.py:58-109)
 *   TF_LAYER_SUM           SumLayer (sum.py:7-14): (i, j) += floor(flow), then ReferenceLayer.update
 *   TF_LAYER_STATIC        StaticLayer (static.py:7-17): pixmaps copied where introduced
 *   TF_LAYER_INTRODUCTION  IntroductionLayer (introduction.py:8-73): 8-channel canvas
 *                          (r, g, b, alpha, source, i, j, frame), moved by the flow, then fed
 * and Layer.render (layer.py:32-34) for all of them.  tf_layer_cfg carries the LayerConfig
 * fields the layers read (transflow/config.py:88-105).
 */
enum { TF_LAYER_MOVEREF = 0, TF_LAYER_SUM = 1, TF_LAYER_STATIC = 2, TF_LAYER_INTRODUCTION = 3 };

typedef struct tf_layer_cfg {
    int transparent_pixels_can_move;    /* False */
    int pixels_can_move_to_empty_spot;  /* True  */
    int pixels_can_move_to_filled_spot; /* True  */
    int moving_pixels_leave_empty_spot; /* False */
    int reset_mode;                     /* 0 off, 1 random, 2 constant, 3 linear (reference.py:16-21) */
    double reset_random_factor;         /* 1   */
    double reset_constant_step;         /* 1   */
    double reset_linear_factor;         /* 0.1 */
    int reset_source;                   /* False */
    int layer_class;                    /* TF_LAYER_*; 0 = moveref */
    /* introduction.py:24-44.  As the reference is written, on_empty_spots, unmoving_pixels and
       the mask write of on_all_empty_spots index with the Python value False and select nothing;
       they are carried for completeness, on_all_empty_spots still turns `consider_flow` off (:40). */
    int introduce_pixels_on_empty_spots;  /* True  (no effect) */
    int introduce_pixels_on_filled_spots; /* True  */
    int introduce_moving_pixels;          /* True  */
    int introduce_unmoving_pixels;        /* True  (no effect) */
    int introduce_on_all_filled_spots;    /* False */
    int introduce_on_all_empty_spots;     /* False */
} tf_layer_cfg;

typedef struct tf_remap tf_remap;
typedef struct tf_comp tf_comp;

/* Masks are [height][width]; NULL = the reference's default (mask_src/mask_dst all
   true, mask_alpha/reset_mask all 1.0: layer.py:24, movement.py:14-15, reference.py:44).
   moveref/sum: data starts as (i, j, 1, 0) per pixel (reference.py:40-42); introduction: an
   all-zero 8-channel canvas (data.py:17); static: no data, rgba alpha = 1 (static.py:11). */
int tf_remap_create(tf_remap **out, int height, int width, const tf_layer_cfg *cfg, const uint8_t *mask_src,
                    const uint8_t *mask_dst, const float *mask_alpha, const float *reset_mask);
void tf_remap_destroy(tf_remap *layer);

/* ReferenceLayer.set_sources (reference.py:54-56): remembers the introduction masks
   (uint8 [height][width] each) and writes source index s where mask s is set. */
int tf_remap_set_sources(tf_remap *layer, int n_sources, const uint8_t *const *introduction_masks);

/* The flow-dependent part of Layer.update.  moveref: MovementLayer.update +
   ReferenceLayer._update_reset; sum: the accumulation (sum.py:10) + _update_reset;
   introduction: MovementLayer.update on the 8-channel canvas and the introduction mask
   (introduction.py:24-44) for the tf_remap_introduce calls that follow; static: nothing
   (static.py:13 ignores the flow).  `flow` float32 [H][W][2]
   (already post-processed).  `uniform`: float64 [H][W] in [0,1) -- the field the
   reference draws with numpy.random.random (reference.py:59) -- or NULL to draw it
   on the GPU from `seed` and the handle's frame counter.  Returns TF_ERR_INDEX if a
   rounded flow vector leaves the frame (state is then unchanged). */
int tf_remap_update(tf_remap *layer, const float *flow, const double *uniform, uint64_t seed);
int tf_remap_update_dev(tf_remap *layer, const void *flow_dev, const void *uniform_dev, uint64_t seed);
/* The float64 [H][W] field the NEXT tf_remap_update(_dev) / tf_remap_step_dev call on this layer would
   draw for uniform == NULL with this seed (it depends on the seed, the pixel and the layer's frame
   counter): written to uniform_dev so that a checker can hand the very same field to a CPU statement
   of reference.py:58-67 -- on-GPU draws are then comparable bit for bit. */
int tf_remap_uniform_dev(tf_remap *layer, uint64_t seed, void *uniform_dev);
/* Deferred form of the range check for the resident path: 1 if any update_dev since
   the last call saw an out-of-frame vector (those updates were skipped). */
int tf_remap_check(tf_remap *layer, int *out_of_frame);

/* One iteration of the per-source loop for source `source_index`; pixmap is uint8
   [H][W][channels], channels 3 or 4.  moveref/sum: ReferenceLayer._update_rgba
   (reference.py:94-105); static: the masked copy of static.py:14-17. */
int tf_remap_gather(tf_remap *layer, int source_index, const uint8_t *pixmap, int channels);
int tf_remap_gather_dev(tf_remap *layer, int source_index, const void *pixmap_dev, int channels);
/* tf_remap_gather with the pixmap going up on the library's upload stream, beside whatever the caller queued before it
   (the same iteration of reference.py:94-105 / static.py:14-17; the call still returns with the host pixmap consumed):
   for an update whose flow was on the device already (transflow/pipeline.py:562-567 with a DeviceFlow), where the
   upload would otherwise wait for the update kernel. */
int tf_remap_gather_beside(tf_remap *layer, int source_index, const uint8_t *pixmap, int channels);
/* The upload alone: the pixmap of `source.next()` (reference.py:99) into the layer's staging buffer -- on the library's
   upload stream if `beside` -- for a caller that runs the frame's kernels later in one launch (tf_remap_step_dev with
   *pixmap_dev: HipCompositor does, between update() and render(), compositor.py:27-40).  Returns with the host pixmap
   consumed.  tf_remap_staged_used: call after queueing the last kernel that reads the staging buffer (the next
   upload waits for it on the device). */
int tf_remap_stage_pixmap(tf_remap *layer, const uint8_t *pixmap, int channels, int beside, void **pixmap_dev);
int tf_remap_staged_used(tf_remap *layer);

/* Introduction layer: one iteration of introduction.py:46-63 -- every target selected by the
   mask tf_remap_update left and by the source's introduction mask takes the record
   (pixmap[s], [1 if RGB], source_index, i(s), j(s), frame_number) of s = target + rounded flow
   (s = target when an introduce_on_all_* flag is set).  introduce_once is the caller's business
   (introduction.py:21-23 returns before the sources are even asked for a frame). */
int tf_remap_introduce(tf_remap *layer, int source_index, const uint8_t *pixmap, int channels, int frame_number);
int tf_remap_introduce_dev(tf_remap *layer, int source_index, const void *pixmap_dev, int channels, int frame_number);

/* Layer.render (layer.py:32-34) + this layer's turn in Compositor.render
   (compositor.py:36-39): alpha := uint8(mask_alpha*alpha) in place, then paint the
   opaque pixels onto the compositor image. */
int tf_remap_render(tf_remap *layer, tf_comp *comp);

/* The resident path's one call per frame, same result as
     [clip of post_process BACKWARD on the flow, if clip_flow]; tf_remap_update_dev;
     tf_remap_gather_dev(source 0); tf_comp_begin; tf_remap_render
   for a compositor with this single layer and a single source.  Runs as ONE kernel when
   the layer needs no second pass (moving_pixels_leave_empty_spot off, reset off/random),
   otherwise as those separate kernels.  clip_flow: 0 = flow_dev is a post-processed flow; 1 = clip it
   first (BACKWARD post_process is the clip alone; written back only in the unfused form, the fused
   form clips in registers); 2 = flow_dev is the winner map of tf_fb_post_process_scatter, the flow
   it stands for (source.py:359-362) is formed in registers. */
int tf_remap_step_dev(tf_remap *layer, tf_comp *comp, const void *flow_dev, int clip_flow, const void *uniform_dev,
                      uint64_t seed, const void *pixmap_dev, int channels);
/* n consecutive tf_remap_step_dev calls as one: step i takes flows_dev[i], paints comps[i] from pixmaps_dev[i] and
   draws from uniforms_dev[i] (uniforms_dev NULL: the generator, as a NULL uniform_dev above).  The layer state and the n
   frames are those of the n calls (pipeline.py:565 + :518 for n consecutive flows of a clip whose frames stay on
   the device).  Knowing the steps that follow, the call leaves out stores nothing reads: while every pixel is
   selected by source 0 -- checked on the device before the first step; a one-source moveref layer that takes the
   one-kernel step keeps it so -- a step's rgba (reference.py:93-105) is overwritten whole by the next step without
   having been read, so only the last step stores it (option "remap_keep_rgba" = 1 stores it every time). */
int tf_remap_steps_dev(tf_remap *layer, int n, tf_comp *const *comps, const void *const *flows_dev, int clip_flow,
                       const void *const *uniforms_dev, uint64_t seed, const void *const *pixmaps_dev, int channels);

/* State exchange for checkpoints (pipeline.py:225-242 pickles the compositor) and
   for extra/control.py:146-162 which reads layer.data.  data int32 [H][W][4]
   (i, j, alpha, source) -- [H][W][8] for an introduction layer, absent for a static one;
   rgba uint8 [H][W][4] (an introduction layer's rgba IS data[..., :4]: pass NULL).
   Either pointer may be NULL.  tf_remap_data_depth: 4, 8 or 0. */
int tf_remap_data_depth(tf_remap *layer, int *depth);
int tf_remap_get_state(tf_remap *layer, int32_t *data, uint8_t *rgba);
int tf_remap_set_state(tf_remap *layer, const int32_t *data, const uint8_t *rgba);

/* Compositor image (compositor.py:17-40): background colour + output frame. */
int tf_comp_create(tf_comp **out, int height, int width, const uint8_t background_rgb[3]);
/* The same with the image in the caller's device memory (H*W*3 bytes at image_dev, which must outlive
   the handle): the frames of a batch side by side in one buffer are what one tf_batch_gather sends
   (pipeline.py:518 hands every finished frame to one output). */
int tf_comp_create_on(tf_comp **out, int height, int width, const uint8_t background_rgb[3], void *image_dev);
void tf_comp_destroy(tf_comp *comp);
int tf_comp_begin(tf_comp *comp);                      /* image = background.copy() (:35) */
int tf_comp_download(tf_comp *comp, uint8_t *rgb_out); /* uint8 [H][W][3] (:40) */
/* The same download beside whatever the caller queues next (pipeline.py:518 hands the frame to the outputs' queue,
   pipeline.py:565 goes on to the next update; output/ffmpeg.py:32-54 writes it to the encoder's pipe): the copy starts
   when the caller's stream reaches this point, on the library's download stream; tf_comp_download_end returns once
   rgb_out (page-locked: tf_host_alloc) is filled.  The image must not be written again before _end: a caller that
   wants frame t on its way down while it renders frame t + 1 alternates two images.  _end without _begin: no-op. */
int tf_comp_download_begin(tf_comp *comp, uint8_t *rgb_out);
int tf_comp_download_end(tf_comp *comp);
int tf_comp_image_ptr(tf_comp *comp, void **dev);

/* ---- batch-of-frames mode over the GPUs of one node ----------------------
 * With flags == 0 every Farnebäck pair is independent (transflow/flow/sources/cv.py:478-490: the
 * `flow=` argument is an output buffer only), so ranks take contiguous ranges of pairs and the path
 * needs no data-path collective.  What is exchanged, through RCCL over xGMI on the library stream:
 * the shared inputs once (broadcast: the pixmap every rank's compositor gathers from,
 * compositor/layers/reference.py:93-105, and the reset mask, reference.py:44) and finished frames to
 * one rank (gather: what Pipeline hands to its output, pipeline.py:518).  One process per GPU; the
 * communicator binds to the device tf_init selected.  librccl is loaded on the first tf_batch_* call.
 * The 128-byte id is made on rank 0 and carried to the other ranks by the host (file or socket). */
#define TF_BATCH_ID_BYTES 128
typedef struct tf_batch tf_batch;
int tf_batch_unique_id(uint8_t *id /*[TF_BATCH_ID_BYTES]*/);
int tf_batch_init(tf_batch **out, int rank, int world, const uint8_t *id);
void tf_batch_destroy(tf_batch *batch);
int tf_batch_info(tf_batch *batch, int *rank, int *world, int *rccl_version); /* any pointer may be NULL */
/* In place: root's `bytes` at dev reach every rank's dev. */
int tf_batch_broadcast(tf_batch *batch, void *dev, size_t bytes, int root);
/* Rank r's send_bytes land on root at recv_dev + sum(recv_bytes[0..r)); recv_bytes NULL = every rank
   sends send_bytes.  recv_dev / recv_bytes are read on root only.  Point-to-point sends into the root
   (all its inbound links at once), not a ring. */
int tf_batch_gather(tf_batch *batch, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *recv_bytes,
                    int root);
/* The gather with explicit places: rank r's send_bytes land on root at recv_dev + recv_offsets[r] (ranges inside
   recv_capacity, not overlapping; a rank with nothing to send passes 0).  "Flows to root" (mode F): the
   flows of a rank's pass land at the clip position of the pass's first pair, and the root's one compositor consumes
   the clip's flows in order, as the reference's one compositor does (transflow/pipeline.py:565 hands every flow of
   the clip to Compositor.update in turn; compositor/layers/movement.py:51-52 is the recurrence that makes the
   order matter). */
int tf_batch_gather_at(tf_batch *batch, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *recv_bytes,
                       const size_t *recv_offsets, size_t recv_capacity, int root);
/* The same gather beside the library stream's next work (transflow/pipeline.py:518 takes one frame at a time; a batch of
   finished frames need not hold up the next batch): _begin starts it, on a stream of the communicator's own, once the
   library stream has reached the point of the call; _end makes the library stream wait for it on the device.  One at
   a time; call _end before the send buffer is written again. */
int tf_batch_gather_begin(tf_batch *batch, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *recv_bytes,
                          int root);
int tf_batch_gather_end(tf_batch *batch);
/* values[n] (host) := sum (op 0) or max (op 1) over ranks; n <= 63.  Waits for the library stream on
   every rank: with n = 0 it is the barrier a timed region is bracketed with. */
int tf_batch_reduce(tf_batch *batch, double *values, int n, int op);

#ifdef __cplusplus
}
#endif
#endif /* TFHIP_H */
