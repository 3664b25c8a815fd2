// The path's one collective from C++: an RCCL all-gather of the token ids on a caller-given HIP stream
// (include/omnitok_comm.h).  librccl.so is resolved with dlopen so that libomnitok.so has no
// link-time dependency on it.  The int64 <-> int32 narrowing / widening around the wire format are two small kernels.
#include "common.h"
#include "../../include/omnitok_comm.h"

#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <string>

namespace {

// The five RCCL entry points used, with the types of rccl.h spelled out (ncclComm_t is an opaque pointer, ncclUniqueId
// is 128 bytes passed by value, ncclDataType_t's ncclInt32 is 2, ncclResult_t's ncclSuccess is 0): <rccl/rccl.h> is not
// included so that the build does not depend on the RCCL development files either.
struct NcclUniqueId { char internal[OMNITOK_COMM_ID_BYTES]; };
typedef void *NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId *);
typedef int (*CommInitRankFn)(NcclComm *, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllGatherFn)(const void *, void *, size_t, int, NcclComm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);
constexpr int kNcclInt32 = 2;

struct Rccl {
    void *handle = nullptr;
    std::string where;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    GetErrorStringFn get_error_string = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
bool g_tried = false;

const Rccl *rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_tried) {
        g_tried = true;
        // the copy already in the process first (PyTorch-ROCm maps its own librccl.so): one RCCL per process
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void *h = nullptr;
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) { g_rccl.where = std::string(n) + " (already mapped)"; break; }
        if (!h)
            for (const char *n : names)
                if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) { g_rccl.where = n; break; }
        if (h) {
            g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
            g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
            g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
            g_rccl.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
            g_rccl.get_error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
            if (g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_gather)
                g_rccl.handle = h;
        }
    }
    return g_rccl.handle ? &g_rccl : nullptr;
}

#define OT_RCCL(r, call)                                                                                    \
    do {                                                                                                    \
        int _e = (call);                                                                                    \
        if (_e != 0) {                                                                                      \
            omnitok::set_error("%s failed: %s (rccl status %d)", #call,                                     \
                               (r)->get_error_string ? (r)->get_error_string(_e) : "?", _e);                \
            return OMNITOK_ERR_HIP;                                                                         \
        }                                                                                                   \
    } while (0)

__global__ void narrow_ids_kernel(const int64_t *__restrict__ in, int32_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}
__global__ void widen_ids_kernel(const int32_t *__restrict__ in, int64_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

}  // namespace

// The communicator is bound to the device it was created on (ncclCommInitRank): staging memory and the narrow / widen
// launches must land there even when the caller's current device is another one (one process driving several GPUs).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

struct omnitok_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1, device = 0;
    int32_t *stage = nullptr;  // [count] send block followed by [world * count] receive block
    int64_t stage_count = 0;
};

extern "C" int omnitok_comm_available(char *where, int where_len) {
    const Rccl *r = rccl();
    if (where && where_len > 0) snprintf(where, where_len, "%s", r ? r->where.c_str() : "librccl.so not found");
    return r ? 1 : 0;
}

extern "C" int omnitok_comm_unique_id(unsigned char *id) {
    OT_CHECK_ARG(id, "comm_unique_id: null pointer");
    const Rccl *r = rccl();
    if (!r) { omnitok::set_error("comm: librccl.so could not be resolved (dlopen)"); return OMNITOK_ERR_UNSUPPORTED; }
    NcclUniqueId u;
    OT_RCCL(r, r->get_unique_id(&u));
    memcpy(id, u.internal, OMNITOK_COMM_ID_BYTES);
    return OMNITOK_OK;
}

extern "C" int omnitok_comm_create(const unsigned char *id, int rank, int world, omnitok_comm **out) {
    OT_CHECK_ARG(id && out, "comm_create: null pointer");
    OT_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "comm_create: rank %d of %d", rank, world);
    const Rccl *r = rccl();
    if (!r) { omnitok::set_error("comm: librccl.so could not be resolved (dlopen)"); return OMNITOK_ERR_UNSUPPORTED; }
    omnitok_comm *c = new omnitok_comm();
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; omnitok::set_error("comm_create: hipGetDevice failed"); return OMNITOK_ERR_HIP; }
    NcclUniqueId u;
    memcpy(u.internal, id, OMNITOK_COMM_ID_BYTES);
    int e = r->comm_init_rank(&c->comm, world, u, rank);
    if (e != 0) {
        omnitok::set_error("ncclCommInitRank(rank %d of %d) failed: %s (rccl status %d)", rank, world,
                           r->get_error_string ? r->get_error_string(e) : "?", e);
        delete c;
        return OMNITOK_ERR_HIP;
    }
    *out = c;
    return OMNITOK_OK;
}

extern "C" void omnitok_comm_destroy(omnitok_comm *c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    const Rccl *r = rccl();
    if (r && c->comm) r->comm_destroy(c->comm);
    if (c->stage) (void)hipFree(c->stage);
    delete c;
}

extern "C" int omnitok_comm_world(omnitok_comm *c) { return c ? c->world : 0; }
extern "C" int omnitok_comm_rank(omnitok_comm *c) { return c ? c->rank : -1; }

extern "C" int omnitok_comm_allgather_i32(omnitok_comm *c, const int32_t *send, int32_t *recv, int64_t count,
                                          omnitok_stream_t stream_) {
    OT_CHECK_ARG(c && c->comm, "comm_allgather: no communicator");
    OT_CHECK_ARG(count >= 0 && (count == 0 || (send && recv)), "comm_allgather: bad arguments");
    if (count == 0) return OMNITOK_OK;
    DeviceGuard guard(c->device);
    const Rccl *r = rccl();
    OT_RCCL(r, r->all_gather(send, recv, (size_t)count, kNcclInt32, c->comm, static_cast<hipStream_t>(stream_)));
    return OMNITOK_OK;
}

extern "C" int omnitok_comm_allgather_ids(omnitok_comm *c, const int64_t *ids_local, int64_t *ids_all, int64_t count,
                                          omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(c && c->comm, "comm_allgather_ids: no communicator");
    OT_CHECK_ARG(count >= 0 && (count == 0 || (ids_local && ids_all)), "comm_allgather_ids: bad arguments");
    if (count == 0) return OMNITOK_OK;
    DeviceGuard guard(c->device);
    if (count > c->stage_count) {
        if (c->stage) OT_HIP(hipFree(c->stage));
        c->stage = nullptr;
        c->stage_count = 0;
        OT_HIP(hipMalloc(reinterpret_cast<void **>(&c->stage), sizeof(int32_t) * (size_t)count * (size_t)(c->world + 1)));
        c->stage_count = count;
    }
    int32_t *send = c->stage, *recv = c->stage + count;
    const unsigned tb = 256;
    hipLaunchKernelGGL(narrow_ids_kernel, dim3((unsigned)((count + tb - 1) / tb)), dim3(tb), 0, stream, ids_local, send, count);
    OT_LAUNCH_CHECK("narrow_ids");
    if (int rc = omnitok_comm_allgather_i32(c, send, recv, count, stream_)) return rc;
    const int64_t total = count * c->world;
    hipLaunchKernelGGL(widen_ids_kernel, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, stream, recv, ids_all, total);
    OT_LAUNCH_CHECK("widen_ids");
    return OMNITOK_OK;
}
