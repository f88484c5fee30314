// Batch-of-frames mode over the GPUs of one node: the only exchanges the path has are
// a one-off broadcast of the shared inputs (pixmap, reset mask) from rank 0 and a gather of finished
// frames to rank 0.  Both go through RCCL (xGMI) on the library stream.  librccl is loaded on first
// use (dlopen), so the drop-in flow source / compositor never map it.
#include <dlfcn.h>

#include <cstring>

#include "common.h"

// The few RCCL types and constants the dlsym'ed entry points need, declared here so that the library
// builds on a ROCm install without RCCL's development headers (values are NCCL's public ABI: nccl.h's
// ncclResult_t / ncclDataType_t / ncclRedOp_t, a 128-byte unique id passed by value).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[TF_BATCH_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint8 = 1, ncclFloat64 = 8;
static constexpr ncclRedOp_t ncclSum = 0, ncclMax = 2;
static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in every NCCL / RCCL release");

namespace tf {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};
static Rccl g_rccl;

static int rccl_load()
{
    if (g_rccl.so)
        return TF_OK;
    // a librccl the process already mapped (same SONAME) is returned as is, so one HIP runtime is shared
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void *so = nullptr;
    std::string errs;
    for (const char *n : names) {
        so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (so)
            break;
        errs += dlerror();
        errs += "; ";
    }
    if (!so)
        return set_error(TF_ERR_UNSUPPORTED, "tf_batch: cannot load librccl: %s", errs.c_str());
    Rccl r;
    r.so = so;
#define TF_SYM(field, name)                                                                  \
    do {                                                                                     \
        *(void **)(&r.field) = dlsym(so, name);                                              \
        if (!r.field)                                                                        \
            return set_error(TF_ERR_UNSUPPORTED, "tf_batch: librccl lacks %s", name);        \
    } while (0)
    TF_SYM(GetUniqueId, "ncclGetUniqueId");
    TF_SYM(CommInitRank, "ncclCommInitRank");
    TF_SYM(CommDestroy, "ncclCommDestroy");
    TF_SYM(Broadcast, "ncclBroadcast");
    TF_SYM(AllReduce, "ncclAllReduce");
    TF_SYM(Send, "ncclSend");
    TF_SYM(Recv, "ncclRecv");
    TF_SYM(GroupStart, "ncclGroupStart");
    TF_SYM(GroupEnd, "ncclGroupEnd");
    TF_SYM(GetErrorString, "ncclGetErrorString");
    TF_SYM(GetVersion, "ncclGetVersion");
#undef TF_SYM
    g_rccl = r;
    return TF_OK;
}

#define TF_RCCL(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess)                                                                              \
            return tf::set_error(TF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r),      \
                                 __FILE__, __LINE__);                                                       \
    } while (0)

} // namespace tf

using namespace tf;

struct tf_batch {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf scratch; // the reduction's few doubles
    // tf_batch_gather_begin / _end: a gather beside the next pass
    hipStream_t gather_stream = nullptr;
    hipEvent_t gather_from = nullptr, gather_done = nullptr;
    bool gather_pending = false;
    ~tf_batch()
    {
        if (gather_from)
            (void)hipEventDestroy(gather_from);
        if (gather_done)
            (void)hipEventDestroy(gather_done);
        if (gather_stream)
            (void)hipStreamDestroy(gather_stream);
    }
};

TF_API int tf_batch_unique_id(uint8_t *id)
{
    TF_REQUIRE(id, "tf_batch_unique_id: null pointer");
    TF_TRY(ensure_init());
    TF_TRY(rccl_load());
    ncclUniqueId u;
    TF_RCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, TF_BATCH_ID_BYTES);
    return TF_OK;
}

TF_API int tf_batch_init(tf_batch **out, int rank, int world, const uint8_t *id)
{
    TF_REQUIRE(out && id, "tf_batch_init: null pointer");
    TF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tf_batch_init: rank %d of %d", rank, world);
    TF_TRY(ensure_init()); // the communicator binds to the device tf_init chose
    TF_TRY(rccl_load());
    tf_batch *b = new tf_batch;
    b->rank = rank;
    b->world = world;
    ncclUniqueId u;
    memcpy(u.internal, id, TF_BATCH_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&b->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete b;
        return set_error(TF_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    }
    int rc = b->scratch.alloc(64 * sizeof(double));
    if (rc != TF_OK) {
        (void)g_rccl.CommDestroy(b->comm);
        delete b;
        return rc;
    }
    *out = b;
    return TF_OK;
}

TF_API void tf_batch_destroy(tf_batch *b)
{
    if (!b)
        return;
    (void)hipStreamSynchronize(main_stream());
    if (b->gather_stream)
        (void)hipStreamSynchronize(b->gather_stream);
    if (b->comm && g_rccl.CommDestroy)
        (void)g_rccl.CommDestroy(b->comm);
    delete b;
}

TF_API int tf_batch_info(tf_batch *b, int *rank, int *world, int *rccl_version)
{
    TF_REQUIRE(b, "tf_batch_info: null handle");
    if (rank)
        *rank = b->rank;
    if (world)
        *world = b->world;
    if (rccl_version)
        TF_RCCL(g_rccl.GetVersion(rccl_version));
    return TF_OK;
}

TF_API int tf_batch_broadcast(tf_batch *b, void *dev, size_t bytes, int root)
{
    TF_REQUIRE(b && (dev || bytes == 0), "tf_batch_broadcast: null argument");
    TF_REQUIRE(root >= 0 && root < b->world, "tf_batch_broadcast: root %d of %d", root, b->world);
    if (bytes == 0)
        return TF_OK;
    TF_TRY(ensure_init());
    ProfScope ps("batch_broadcast");
    TF_RCCL(g_rccl.Broadcast(dev, dev, bytes, ncclUint8, root, b->comm, main_stream()));
    return TF_OK;
}

// Rank r's `send_bytes` land at recv_dev + sum(recv_bytes[0..r)) on root.  A gather to one root uses
// all of the root's inbound xGMI links at once (point-to-point sends, not a ring: SURVEY.md §8e).
// `recv_offsets` (root only, may be null): where each rank's bytes land instead of the prefix sums.
static int gather_on(tf_batch *b, hipStream_t s, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *recv_bytes,
                     int root, const size_t *recv_offsets = nullptr)
{
    if (b->rank == root) {
        TF_REQUIRE(recv_dev, "tf_batch_gather: root needs a receive buffer");
        if (recv_bytes)
            TF_REQUIRE(recv_bytes[root] == send_bytes, "tf_batch_gather: root's own count %zu != what it sends (%zu)",
                       recv_bytes[root], send_bytes);
    }
    if (b->rank != root) {
        if (send_bytes)
            TF_RCCL(g_rccl.Send(send_dev, send_bytes, ncclUint8, root, b->comm, s));
        return TF_OK;
    }
    size_t off = 0;
    TF_RCCL(g_rccl.GroupStart());
    for (int r = 0; r < b->world; r++) {
        const size_t n = recv_bytes ? recv_bytes[r] : send_bytes;
        char *dst = (char *)recv_dev + (recv_offsets ? recv_offsets[r] : off);
        off += n;
        if (n == 0)
            continue;
        if (r == root) {
            if ((const void *)dst != send_dev) {
                hipError_t e = hipMemcpyAsync(dst, send_dev, n, hipMemcpyDeviceToDevice, s);
                if (e != hipSuccess) {
                    (void)g_rccl.GroupEnd();
                    return set_error(TF_ERR_HIP, "tf_batch_gather: local copy failed: %s", hipGetErrorString(e));
                }
            }
            continue;
        }
        ncclResult_t rr = g_rccl.Recv(dst, n, ncclUint8, r, b->comm, s);
        if (rr != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return set_error(TF_ERR_HIP, "ncclRecv from rank %d failed: %s", r, g_rccl.GetErrorString(rr));
        }
    }
    TF_RCCL(g_rccl.GroupEnd());
    return TF_OK;
}

TF_API int tf_batch_gather(tf_batch *b, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *recv_bytes,
                           int root)
{
    TF_REQUIRE(b, "tf_batch_gather: null handle");
    TF_REQUIRE(root >= 0 && root < b->world, "tf_batch_gather: root %d of %d", root, b->world);
    TF_REQUIRE(send_dev || send_bytes == 0, "tf_batch_gather: null send buffer");
    TF_TRY(ensure_init());
    ProfScope ps("batch_gather");
    return gather_on(b, main_stream(), send_dev, send_bytes, recv_dev, recv_bytes, root);
}

// The gather with a place of its own for every rank's bytes: rank r's send_bytes land at recv_dev + recv_offsets[r].
// What "flows to root" needs (SURVEY.md §8e mode F): the flows of a rank's pass k belong at the clip position of that
// pass's first pair, so that the root's ONE compositor consumes the clip's flows in order, as transflow/pipeline.py:565
// hands them to the reference's one compositor.  The ranges must not overlap (checked on root).
TF_API int tf_batch_gather_at(tf_batch *b, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *recv_bytes,
                              const size_t *recv_offsets, size_t recv_capacity, int root)
{
    TF_REQUIRE(b, "tf_batch_gather_at: null handle");
    TF_REQUIRE(root >= 0 && root < b->world, "tf_batch_gather_at: root %d of %d", root, b->world);
    TF_REQUIRE(send_dev || send_bytes == 0, "tf_batch_gather_at: null send buffer");
    if (b->rank == root) {
        TF_REQUIRE(recv_bytes && recv_offsets, "tf_batch_gather_at: root needs every rank's count and offset");
        for (int r = 0; r < b->world; r++) {
            TF_REQUIRE(recv_bytes[r] <= recv_capacity && recv_offsets[r] <= recv_capacity - recv_bytes[r],
                       "tf_batch_gather_at: rank %d's %zu bytes at %zu leave the %zu-byte buffer", r, recv_bytes[r],
                       recv_offsets[r], recv_capacity);
            for (int q = 0; q < r; q++)
                TF_REQUIRE(recv_bytes[r] == 0 || recv_bytes[q] == 0 || recv_offsets[r] >= recv_offsets[q] + recv_bytes[q] ||
                               recv_offsets[q] >= recv_offsets[r] + recv_bytes[r],
                           "tf_batch_gather_at: the ranges of ranks %d and %d overlap", q, r);
        }
    }
    TF_TRY(ensure_init());
    ProfScope ps("batch_gather");
    return gather_on(b, main_stream(), send_dev, send_bytes, recv_dev, recv_bytes, root, recv_offsets);
}

// The same gather BESIDE what the library stream does next: it starts when the library stream reaches the point of this
// call (the pass whose frames it sends is finished) and runs on a stream of the communicator's own, so the next pass's
// Farneback call and remap need not wait for the frames to cross the links.  tf_batch_gather_end makes the library stream
// wait for it (on the device, not the host): call it before the send buffer is written again.
TF_API int tf_batch_gather_begin(tf_batch *b, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *recv_bytes,
                                 int root)
{
    TF_REQUIRE(b, "tf_batch_gather_begin: null handle");
    TF_REQUIRE(root >= 0 && root < b->world, "tf_batch_gather_begin: root %d of %d", root, b->world);
    TF_REQUIRE(send_dev || send_bytes == 0, "tf_batch_gather_begin: null send buffer");
    TF_REQUIRE(!b->gather_pending, "tf_batch_gather_begin: the previous gather has not been ended (tf_batch_gather_end)");
    TF_TRY(ensure_init());
    if (!b->gather_stream) {
        TF_HIP(hipStreamCreateWithFlags(&b->gather_stream, hipStreamNonBlocking));
        TF_HIP(hipEventCreateWithFlags(&b->gather_from, hipEventDisableTiming));
        TF_HIP(hipEventCreateWithFlags(&b->gather_done, hipEventDisableTiming));
    }
    TF_HIP(hipEventRecord(b->gather_from, main_stream()));
    TF_HIP(hipStreamWaitEvent(b->gather_stream, b->gather_from, 0));
    TF_TRY(gather_on(b, b->gather_stream, send_dev, send_bytes, recv_dev, recv_bytes, root));
    TF_HIP(hipEventRecord(b->gather_done, b->gather_stream));
    b->gather_pending = true;
    return TF_OK;
}

TF_API int tf_batch_gather_end(tf_batch *b)
{
    TF_REQUIRE(b, "tf_batch_gather_end: null handle");
    if (!b->gather_pending)
        return TF_OK;
    TF_HIP(hipStreamWaitEvent(main_stream(), b->gather_done, 0));
    b->gather_pending = false;
    return TF_OK;
}

// values[n] (host) := op over ranks of values[n]; op 0 = sum, 1 = max.  Synchronises the library
// stream: with n = 0 this is the barrier bench.py brackets its timed region with.
TF_API int tf_batch_reduce(tf_batch *b, double *values, int n, int op)
{
    TF_REQUIRE(b, "tf_batch_reduce: null handle");
    TF_REQUIRE(n >= 0 && n <= 63 && (values || n == 0), "tf_batch_reduce: 0..63 values");
    TF_REQUIRE(op == 0 || op == 1, "tf_batch_reduce: op 0 (sum) or 1 (max)");
    TF_TRY(ensure_init());
    hipStream_t s = main_stream();
    double host[64] = {0};
    for (int i = 0; i < n; i++)
        host[i] = values[i];
    const int cnt = n + 1; // one extra element so that n = 0 still synchronises the ranks
    double *d = b->scratch.as<double>();
    TF_HIP(hipMemcpyAsync(d, host, cnt * sizeof(double), hipMemcpyHostToDevice, s));
    TF_RCCL(g_rccl.AllReduce(d, d, cnt, ncclFloat64, op == 0 ? ncclSum : ncclMax, b->comm, s));
    TF_HIP(hipMemcpyAsync(host, d, cnt * sizeof(double), hipMemcpyDeviceToHost, s));
    TF_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; i++)
        values[i] = host[i];
    return TF_OK;
}
