// Runtime of libtfhip.so: device selection, the library stream, errors, events,
// the per-kernel profiler and raw device buffers.
#include <cstring>
#include <map>
#include <mutex>

#include "common.h"

#include <cstdlib>

namespace tf {

static thread_local std::string g_last_error;
static std::mutex g_mu;
static bool g_inited = false;
static int g_device = 0;
static hipStream_t g_stream = nullptr;

int set_error(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// ---- options ----------------------------------------------------------------------
struct OptDef {
    const char *name;
    long value, lo, hi;
};
static OptDef g_opts[OPT_COUNT] = {
    {"fb_fused", -1, -1, 1},                   // -1: one-kernel iteration on levels of >= fb_fuse_min_px pixels; 0 never; 1 always
    {"fb_fuse_min_px", 4000000, 0, 1l << 40},  // pixels of a level over the batch (two 1080p levels, 4.15M, are in)
    {"fb_no_share", 0, 0, 1},                  // 1: expand per pair and side even when pairs share a frame
    {"fb_no_overlap", 0, 0, 1},                // 1: a call's work stays on the library stream (read at tf_fb_create)
    {"remap_px", 4, 1, 4},                     // pixels per thread of the one-kernel remap step: 1, 2 or 4 (4: 1.98 ms per 32 4K frames, 2: 2.04, 1: 2.45)
    {"remap_no_pack", 0, 0, 2},                // 1: the one-kernel remap step keeps its state as int32 x 4; 2: as int16 x 4 at most (0: one 32-bit word where row, column, alpha and source index fit 13 + 13 + 1 + 5 bits)
    {"remap_keep_rgba", 0, 0, 1},              // 1: tf_remap_steps_dev stores the layer's rgba in every step (A/B of the elided stores)
    {"prof_levels", 0, 0, 1},                  // 1: profiler labels carry the pyramid level ("fb_polyexp.k2")
    {"fb_exact_sums", 0, 0, 1},                // 1: the box window's sums in OpenCV's own order (bit-identical flow, ~5x slower; read per call)
    {"fb_chain", -1, -1, 1},                   // how a segmented march gets OpenCV's column sums: -1 the cheaper way per launch; 0 a pre-pass; 1 handed down inside the launch
    {"fb_segs", 0, 0, 64},                     // > 0: row segments per column of the marching kernels (0: chosen per launch)
};
long option(Opt which) { return g_opts[which].value; }

static thread_local hipStream_t g_override = nullptr;
// tf_thread_stream: the calling thread's own "library stream" (a flow source prefetching in a worker thread queues its
// uploads, kernels and downloads there, beside the compositor's on the library stream proper)
static thread_local hipStream_t g_thread_main = nullptr;
static hipStream_t g_thread_streams[3] = {nullptr, nullptr, nullptr};

hipStream_t main_stream() { return g_thread_main ? g_thread_main : g_stream; }
hipStream_t stream() { return g_override ? g_override : main_stream(); }

// Side streams are the library's, not a handle's: HIP multiplexes streams onto a few hardware queues
// (four by default), and streams that share a queue serialise -- with per-handle side streams a second
// handle in the process cost the first its overlap (measured: -17 % at 1080p with an idle 4K handle).
static hipStream_t g_side[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
int side_stream(int which, hipStream_t *out)
{
    TF_REQUIRE(which >= 0 && which <= 4, "side_stream: bad index");
    TF_TRY(ensure_init());
    std::lock_guard<std::mutex> lk(g_mu); // handles are created from any thread
    if (!g_side[which]) {
        int least = 0, greatest = 0;
        TF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        TF_HIP(hipStreamCreateWithPriority(&g_side[which], hipStreamNonBlocking, (which == 0 || which >= 3) ? least : greatest));
    }
    *out = g_side[which];
    return TF_OK;
}

StreamScope::StreamScope(hipStream_t s) : prev(g_override) { g_override = s; }
StreamScope::~StreamScope() { g_override = prev; }

static int init_device(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_inited) {
        if (device != g_device)
            return set_error(TF_ERR_STATE, "tf_init(%d): library already bound to device %d", device, g_device);
        TF_HIP(hipSetDevice(g_device));
        return TF_OK;
    }
    int n = 0;
    TF_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n)
        return set_error(TF_ERR_ARG, "tf_init: device %d out of range (%d visible)", device, n);
    TF_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    TF_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(TF_ERR_UNSUPPORTED, "tf_init: device %d is %s; this library is built for gfx950 only",
                         device, prop.gcnArchName);
    {
        // the library stream carries the dependent chain of every job, side streams (tf_fb's preparation
        // stream) are meant to fill its gaps: highest priority here, lowest there (measured on MI355X:
        // no effect either way, the two queues share the CUs evenly)
        int least = 0, greatest = 0;
        TF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        TF_HIP(hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, greatest));
    }
    g_device = device;
    g_inited = true;
    return TF_OK;
}

int ensure_init()
{
    if (g_inited)
        return hipSetDevice(g_device) == hipSuccess ? TF_OK : set_error(TF_ERR_HIP, "hipSetDevice failed");
    return init_device(0);
}

int DevBuf::alloc(size_t n)
{
    release();
    if (n == 0)
        return TF_OK;
    TF_HIP(hipMalloc(&p, n));
    bytes = n;
    return TF_OK;
}

void DevBuf::borrow(void *ptr, size_t n)
{
    release();
    p = ptr;
    bytes = n;
    borrowed = true;
}

void DevBuf::release()
{
    if (p && !borrowed)
        (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    borrowed = false;
}

// ---- profiler ---------------------------------------------------------------------
struct ProfRec {
    const char *name;
    hipEvent_t a, b;
};
static bool g_prof = false;
static std::mutex g_prof_mu; // g_prof_filter, g_recs, g_free_events, g_acc
static std::string g_prof_filter;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_free_events;
static std::map<std::string, std::pair<long, double>> g_acc;

bool prof_enabled() { return g_prof; }

static hipEvent_t get_event_locked()
{
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

static void prof_drain_locked()
{
    if (g_recs.empty())
        return;
    (void)hipDeviceSynchronize();
    for (auto &r : g_recs) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        auto &acc = g_acc[r.name];
        acc.first += 1;
        acc.second += ms;
        g_free_events.push_back(r.a);
        g_free_events.push_back(r.b);
    }
    g_recs.clear();
}

ProfScope::ProfScope(const char *name_) : name(name_), a(nullptr), b(nullptr)
{
    if (!g_prof)
        return;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_filter.empty() && !strstr(name, g_prof_filter.c_str()))
            return;
        if (g_recs.size() >= 4096)
            prof_drain_locked();
        a = get_event_locked();
        b = get_event_locked();
    }
    (void)hipEventRecord(a, stream());
}

ProfScope::~ProfScope()
{
    if (!a)
        return;
    (void)hipEventRecord(b, stream());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.push_back(ProfRec{name, a, b});
}

} // namespace tf

using namespace tf;

struct tf_event {
    hipEvent_t ev;
};

TF_API int tf_abi_version(void) { return TFHIP_ABI_VERSION; }

TF_API int tf_init(int device) { return init_device(device); }

TF_API int tf_set_option(const char *name, long value)
{
    TF_REQUIRE(name, "tf_set_option: null name");
    for (auto &o : g_opts)
        if (strcmp(o.name, name) == 0) {
            TF_REQUIRE(value >= o.lo && value <= o.hi, "tf_set_option: %s takes %ld..%ld, got %ld", name, o.lo, o.hi, value);
            o.value = value;
            return TF_OK;
        }
    return set_error(TF_ERR_ARG, "tf_set_option: unknown option '%s'", name);
}

TF_API int tf_get_option(const char *name, long *value)
{
    TF_REQUIRE(name && value, "tf_get_option: null argument");
    for (auto &o : g_opts)
        if (strcmp(o.name, name) == 0) {
            *value = o.value;
            return TF_OK;
        }
    return set_error(TF_ERR_ARG, "tf_get_option: unknown option '%s'", name);
}

TF_API int tf_is_initialized(void) { return g_inited ? 1 : 0; }

TF_API int tf_device_count(int *count)
{
    TF_REQUIRE(count, "tf_device_count: null pointer");
    TF_HIP(hipGetDeviceCount(count));
    return TF_OK;
}

TF_API const char *tf_last_error(void) { return g_last_error.c_str(); }

TF_API int tf_sync(void)
{
    TF_TRY(ensure_init());
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_stream(void **hip_stream)
{
    TF_REQUIRE(hip_stream, "tf_stream: null pointer");
    TF_TRY(ensure_init());
    *hip_stream = (void *)stream();
    return TF_OK;
}

TF_API int tf_event_create(tf_event **ev)
{
    TF_REQUIRE(ev, "tf_event_create: null pointer");
    TF_TRY(ensure_init());
    tf_event *e = new tf_event;
    hipError_t rc = hipEventCreate(&e->ev);
    if (rc != hipSuccess) {
        delete e;
        return set_error(TF_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(rc));
    }
    *ev = e;
    return TF_OK;
}

TF_API int tf_event_record(tf_event *ev)
{
    TF_REQUIRE(ev, "tf_event_record: null event");
    TF_HIP(hipEventRecord(ev->ev, stream()));
    return TF_OK;
}

TF_API int tf_event_elapsed_ms(tf_event *start, tf_event *stop, float *ms)
{
    TF_REQUIRE(start && stop && ms, "tf_event_elapsed_ms: null argument");
    TF_HIP(hipEventSynchronize(stop->ev));
    TF_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return TF_OK;
}

TF_API int tf_stream_wait_event(tf_event *ev)
{
    TF_REQUIRE(ev, "tf_stream_wait_event: null event");
    TF_TRY(ensure_init());
    TF_HIP(hipStreamWaitEvent(stream(), ev->ev, 0));
    return TF_OK;
}

TF_API int tf_event_synchronize(tf_event *ev)
{
    TF_REQUIRE(ev, "tf_event_synchronize: null event");
    TF_HIP(hipEventSynchronize(ev->ev));
    return TF_OK;
}

static_assert(sizeof(hipIpcMemHandle_t) == TF_IPC_HANDLE_BYTES, "tfhip.h states the size of an IPC handle");

TF_API int tf_ipc_export(void *dev, void *handle_out)
{
    TF_REQUIRE(dev && handle_out, "tf_ipc_export: null pointer");
    TF_TRY(ensure_init());
    hipIpcMemHandle_t h;
    TF_HIP(hipIpcGetMemHandle(&h, dev));
    memcpy(handle_out, &h, sizeof(h));
    return TF_OK;
}

TF_API int tf_ipc_open(const void *handle, void **dev)
{
    TF_REQUIRE(handle && dev, "tf_ipc_open: null pointer");
    TF_TRY(ensure_init());
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    TF_HIP(hipIpcOpenMemHandle(dev, h, hipIpcMemLazyEnablePeerAccess));
    return TF_OK;
}

TF_API int tf_ipc_close(void *dev)
{
    if (dev)
        TF_HIP(hipIpcCloseMemHandle(dev));
    return TF_OK;
}

TF_API void tf_event_destroy(tf_event *ev)
{
    if (!ev)
        return;
    (void)hipEventDestroy(ev->ev);
    delete ev;
}

TF_API int tf_prof_enable(int on)
{
    TF_TRY(ensure_init());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!on)
        prof_drain_locked();
    g_prof = on != 0;
    return TF_OK;
}

TF_API int tf_prof_set_filter(const char *substring)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_filter = substring ? substring : "";
    return TF_OK;
}

TF_API int tf_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked();
    g_acc.clear();
    return TF_OK;
}

TF_API int tf_prof_report(char *buf, size_t buf_size)
{
    TF_REQUIRE(buf && buf_size > 0, "tf_prof_report: null buffer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked();
    size_t off = 0;
    buf[0] = 0;
    for (auto &kv : g_acc) {
        int n = snprintf(buf + off, buf_size - off, "%s %ld %.6f\n", kv.first.c_str(), kv.second.first,
                         kv.second.second);
        if (n < 0 || (size_t)n >= buf_size - off)
            return set_error(TF_ERR_ARG, "tf_prof_report: buffer too small");
        off += (size_t)n;
    }
    return TF_OK;
}

TF_API int tf_thread_stream(int which)
{
    TF_REQUIRE(which >= 0 && which <= 3, "tf_thread_stream: stream %d not in 0..3", which);
    TF_TRY(ensure_init());
    if (which == 0) {
        g_thread_main = nullptr;
        return TF_OK;
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_thread_streams[which - 1]) {
            int least = 0, greatest = 0;
            TF_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            TF_HIP(hipStreamCreateWithPriority(&g_thread_streams[which - 1], hipStreamNonBlocking, greatest));
        }
    }
    g_thread_main = g_thread_streams[which - 1];
    return TF_OK;
}

TF_API int tf_host_alloc(void **host, size_t bytes)
{
    TF_REQUIRE(host, "tf_host_alloc: null pointer");
    TF_TRY(ensure_init());
    TF_HIP(hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocDefault));
    return TF_OK;
}

TF_API int tf_host_free(void *host)
{
    if (host)
        TF_HIP(hipHostFree(host));
    return TF_OK;
}

TF_API int tf_dev_alloc(void **dev, size_t bytes)
{
    TF_REQUIRE(dev, "tf_dev_alloc: null pointer");
    TF_TRY(ensure_init());
    TF_HIP(hipMalloc(dev, bytes ? bytes : 1));
    return TF_OK;
}

TF_API int tf_dev_free(void *dev)
{
    if (dev)
        TF_HIP(hipFree(dev));
    return TF_OK;
}

TF_API int tf_dev_upload(void *dev, const void *host, size_t bytes)
{
    TF_REQUIRE(dev && host, "tf_dev_upload: null pointer");
    TF_TRY(ensure_init());
    TF_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

TF_API int tf_dev_download(void *host, const void *dev, size_t bytes)
{
    TF_REQUIRE(dev && host, "tf_dev_download: null pointer");
    TF_TRY(ensure_init());
    TF_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream()));
    TF_HIP(hipStreamSynchronize(stream()));
    return TF_OK;
}

// 16 bytes per lane, one pass: the streaming copy the HBM ceiling is quoted from (MI355X_MICROARCH.md)
__global__ void k_stream_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = src[i];
}

TF_API int tf_dev_stream_copy(void *dst_dev, const void *src_dev, size_t bytes)
{
    TF_REQUIRE(dst_dev && src_dev && (bytes & 15) == 0, "tf_dev_stream_copy: null pointer or size not a multiple of 16");
    TF_TRY(ensure_init());
    const size_t n = bytes / 16;
    return tf::launch("stream_copy", k_stream_copy, dim3(tf::cdiv(n, 256)), dim3(256), 0, (const float4 *)src_dev,
                      (float4 *)dst_dev, n);
}

__global__ void k_store_u64(unsigned long long *dst, unsigned long long v)
{
    *dst = v;
}

TF_API int tf_dev_store_u64(void *dev, uint64_t value)
{
    TF_REQUIRE(dev && ((uintptr_t)dev & 7) == 0, "tf_dev_store_u64: null or misaligned pointer");
    TF_TRY(ensure_init());
    return tf::launch("store_u64", k_store_u64, dim3(1), dim3(1), 0, (unsigned long long *)dev, (unsigned long long)value);
}

TF_API int tf_dev_copy(void *dst_dev, const void *src_dev, size_t bytes)
{
    TF_REQUIRE(dst_dev && src_dev, "tf_dev_copy: null pointer");
    TF_TRY(ensure_init());
    TF_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, stream()));
    return TF_OK;
}
