// plba_rccl.cpp — include/plba_rccl.h: the plba_allreduce_fn hook of include/plba.h on RCCL's ncclAllReduce (xGMI), in C++.
//
// The reduced pose normal equations of the landmark shards are all-reduced in place on the stream the
// library's kernels run on; nothing here synchronises the stream or touches the host.  RCCL's entry points are bound at run
// time so that a process which already carries an RCCL (PyTorch ships one) keeps using that single copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "plba_rccl.h"

namespace {

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

thread_local char g_err[384] = "";
std::mutex g_mu;
Api g_api;

template <class F>
bool bind(void* lib, const char* name, F& f) {
    f = reinterpret_cast<F>(dlsym(lib, name));
    return f != nullptr;
}

bool load_api() {
    std::lock_guard<std::mutex> g(g_mu);
    if (g_api.ok) return true;
    // an RCCL that is already mapped into the process first (RTLD_NOLOAD), then the ROCm install
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { snprintf(g_err, sizeof g_err, "librccl could not be loaded: %s", dlerror()); return false; }
    Api a;
    a.lib = lib;
    if (!bind(lib, "ncclGetUniqueId", a.GetUniqueId) || !bind(lib, "ncclCommInitRank", a.CommInitRank) || !bind(lib, "ncclAllReduce", a.AllReduce) ||
        !bind(lib, "ncclCommDestroy", a.CommDestroy) || !bind(lib, "ncclGetErrorString", a.GetErrorString)) {
        snprintf(g_err, sizeof g_err, "librccl lacks an entry point: %s", dlerror());
        return false;
    }
    a.ok = true;
    g_api = a;
    return true;
}

int fail(const char* what, ncclResult_t r) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
    return -1;
}

}  // namespace

struct plba_rccl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

const char* plba_rccl_last_error(void) { return g_err; }

int plba_rccl_unique_id(unsigned char id[PLBA_RCCL_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == PLBA_RCCL_ID_BYTES, "ncclUniqueId size");
    if (!id) { snprintf(g_err, sizeof g_err, "id is NULL"); return -1; }
    if (!load_api()) return -1;
    ncclUniqueId u;
    ncclResult_t r = g_api.GetUniqueId(&u);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    memcpy(id, &u, sizeof u);
    return 0;
}

int plba_rccl_init(plba_rccl_comm** out, int rank, int world, const unsigned char id[PLBA_RCCL_ID_BYTES]) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) { snprintf(g_err, sizeof g_err, "bad argument"); return -1; }
    *out = nullptr;
    if (!load_api()) return -1;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    plba_rccl_comm* c = new plba_rccl_comm;
    c->rank = rank; c->world = world;
    ncclResult_t r = g_api.CommInitRank(&c->comm, world, u, rank);      // collective over all ranks; uses the current HIP device
    if (r != ncclSuccess) { delete c; return fail("ncclCommInitRank", r); }
    *out = c;
    return 0;
}

int plba_rccl_allreduce(void* user, double* device_buf, size_t n, int op, void* stream) {
    plba_rccl_comm* c = static_cast<plba_rccl_comm*>(user);
    if (!c || !c->comm || !device_buf) { snprintf(g_err, sizeof g_err, "bad argument"); return -1; }
    if (n == 0) return 0;
    ncclResult_t r = g_api.AllReduce(device_buf, device_buf, n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return fail("ncclAllReduce", r);
    return 0;
}

void plba_rccl_destroy(plba_rccl_comm* c) {
    if (!c) return;
    if (c->comm && g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
