// Internal helpers shared by the translation units of libtfhip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/tfhip.h"

#define TF_API extern "C" __attribute__((visibility("default")))

namespace tf {

int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
hipStream_t stream();      // the stream launches go to: the library stream unless a StreamScope is active
hipStream_t main_stream(); // the library stream (tf_stream)
int side_stream(int which, hipStream_t *out); // 0: background work (lowest priority), 1, 2: a call's kernels (highest), 3, 4: uploads / downloads of a streaming caller (3 also a compositor layer's pixmap)
int ensure_init();

// Documented run-time options (tf_set_option / tf_get_option, include/tfhip.h).
enum Opt { OPT_FB_FUSED = 0, OPT_FB_FUSE_MIN_PX, OPT_FB_NO_SHARE, OPT_FB_NO_OVERLAP, OPT_REMAP_PX, OPT_REMAP_NO_PACK, OPT_REMAP_KEEP_RGBA,
           OPT_PROF_LEVELS, OPT_FB_EXACT_SUMS, OPT_FB_CHAIN, OPT_FB_SEGS, OPT_COUNT };
long option(Opt which);

// Kernel-experiment knobs (tile sizes, segment counts ...) are compile-time constants in the shipped
// library; a build with -DTF_EXPERIMENT reads them from the environment (tools/build_variant.sh).
#ifdef TF_EXPERIMENT
inline long tune(const char *env, long dflt)
{
    const char *v = getenv(env);
    return v ? atol(v) : dflt;
}
inline const char *tune_str(const char *env) { return getenv(env); }
#else
constexpr long tune(const char *, long dflt) { return dflt; }
constexpr const char *tune_str(const char *) { return nullptr; }
#endif

// Routes the launches (and profiler events) of the enclosing scope to another stream.
struct StreamScope {
    explicit StreamScope(hipStream_t s);
    ~StreamScope();
    hipStream_t prev;
};

#define TF_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return tf::set_error(TF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                             \
    } while (0)

#define TF_TRY(expr)        \
    do {                    \
        int _rc = (expr);   \
        if (_rc != TF_OK)   \
            return _rc;     \
    } while (0)

#define TF_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond))                                  \
            return tf::set_error(TF_ERR_ARG, __VA_ARGS__); \
    } while (0)

// Per-kernel profiler: brackets a launch with events when enabled.  Any thread may launch (a flow source prefetching
// in a worker thread beside the compositor's thread): a scope owns its two events until it ends and the shared record
// lists are only touched under the profiler's lock.
struct ProfScope {
    explicit ProfScope(const char *name);
    ~ProfScope();
    const char *name;
    hipEvent_t a, b;
};
bool prof_enabled();

// All launches go through this so that profiling and error capture are uniform.
template <typename K, typename... Args>
inline int launch(const char *name, K kernel, dim3 grid, dim3 block, size_t smem, Args... args)
{
    if (grid.x == 0 || grid.y == 0 || grid.z == 0)
        return TF_OK;
    {
        ProfScope ps(name);
        hipLaunchKernelGGL(kernel, grid, block, smem, stream(), args...);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return set_error(TF_ERR_HIP, "launch %s failed: %s", name, hipGetErrorString(e));
    return TF_OK;
}

// Simple owning device buffer.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool borrowed = false; // p belongs to the caller (tf_comp_create_on): never freed here
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n);
    void borrow(void *ptr, size_t n);
    void release();
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// numpy.clip semantics: minimum(maximum(v, lo), hi) with NaN propagated (fminf/fmaxf drop it)
// Unsigned division by a launch-wide constant d >= 1 (Granlund-Montgomery, round-up form): with
// s = ceil(log2 d), m = floor(2^32 (2^s - d) / d) + 1:  q = (((t - hi) >> min(s,1)) + hi) >> max(s-1,0),
// hi = umulhi(t, m).  Exact for every 32-bit t; 5 instructions instead of the ~40 of a division.
struct FastDiv {
    uint32_t m;
    int s1, s2;
};
inline FastDiv fast_div_setup(uint32_t d)
{
    int s = 0;
    while ((1ull << s) < d)
        s++;
    FastDiv f;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    f.s1 = s < 1 ? s : 1;
    f.s2 = s > 1 ? s - 1 : 0;
    return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t t, FastDiv f)
{
    const uint32_t hi = __umulhi(t, f.m);
    return (((t - hi) >> f.s1) + hi) >> f.s2;
}

__device__ __forceinline__ float clip_nan(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }

} // namespace tf
