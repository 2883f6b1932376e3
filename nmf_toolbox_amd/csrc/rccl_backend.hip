// RCCL behind the blocking multi-GPU calls (north_star: "a single RCCL all-reduce over xGMI on the K x K Gram / numerator terms per iteration"; SURVEY 8(e):
// single process, ncclCommInitAll over device_ids, one stream per GPU).  libnmfx does not LINK librccl: it is dlopen'ed on the first multi-GPU call that asks
// for it, so a single-GPU host (a MATLAB workstation) needs nothing installed; the peer-mapped reduce-scatter + all-gather of blocking.hip stays as the
// second backend (nmfx_problem.multi_backend / NMFX_MULTI_BACKEND), and is what runs when device_ids names one device twice (RCCL refuses duplicate GPUs).
#include <dlfcn.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>


#include "api_common.h"

// The handful of RCCL types and prototypes the backend uses, declared here (they are the stable NCCL 2.x ABI: an opaque communicator, int-valued enums) so
// that building libnmfx needs no RCCL development headers -- a single-GPU host has none, and every call goes through the dlsym'ed pointers below anyway
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;                   // (any other value is an error; its text comes from ncclGetErrorString)
typedef enum { ncclFloat32 = 7 } ncclDataType_t;                 // nccl.h: ncclInt8 0, Uint8 1, Int32 2, Uint32 3, Int64 4, Uint64 5, Float16 6, Float32 7, Float64 8
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int *version);
}

namespace nmfx {

namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string path, error;
};

std::mutex g_mu;
RcclApi g_api;
bool g_tried = false;
// one communicator set per device list, kept for the life of the process (creating one costs ~0.1-1 s; a MATLAB session calls nmf() many times).  RCCL does
// not allow two host threads to issue collectives on the same communicator at once, and the rest of this API is per-thread: a set is CHECKED OUT for the whole
// of a blocking call (`busy` under g_mu, waiters on g_cv) -- a second caller on the same devices waits for the first one's factorisation, which it would do
// on the GPUs anyway
struct CommSet {
    std::vector<ncclComm_t> c;
    bool busy = false;
};
std::map<std::vector<int>, CommSet> g_comms;
std::condition_variable g_cv;

// RCCL must sit on the SAME HIP runtime as libnmfx (device pointers and streams are shared): look next to the libamdhip64 this library resolved its HIP symbols
// from first (torch ships its own pair), then by soname
bool load_api() {
    if (g_tried) return g_api.handle != nullptr;
    g_tried = true;
    std::vector<std::string> cand;
    if (const char *env = getenv("NMFX_RCCL_LIB")) cand.push_back(env);
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string d(info.dli_fname);
        const size_t s = d.rfind('/');
        if (s != std::string::npos) cand.push_back(d.substr(0, s + 1) + "librccl.so");
    }
    cand.push_back("librccl.so.1");
    cand.push_back("librccl.so");
    for (const std::string &c : cand) {
        void *h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { const char *e = dlerror(); g_api.error = e ? e : "dlopen failed"; continue; }   // (dlerror() clears the message: ONE call)
        RcclApi a;
        a.handle = h; a.path = c;
#define NMFX_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
        NMFX_SYM(CommInitAll, "ncclCommInitAll"); NMFX_SYM(CommDestroy, "ncclCommDestroy"); NMFX_SYM(AllReduce, "ncclAllReduce");
        NMFX_SYM(GroupStart, "ncclGroupStart"); NMFX_SYM(GroupEnd, "ncclGroupEnd"); NMFX_SYM(GetErrorString, "ncclGetErrorString"); NMFX_SYM(GetVersion, "ncclGetVersion");
#undef NMFX_SYM
        if (a.CommInitAll && a.CommDestroy && a.AllReduce && a.GroupStart && a.GroupEnd && a.GetErrorString) { g_api = a; return true; }
        dlclose(h);
        g_api.error = c + ": not an RCCL library (symbols missing)";
    }
    return false;
}

}  // namespace

// can this device list run on RCCL at all?  (distinct devices, library loadable)
bool rccl_usable(const int *devs, int n, std::string *why) {
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (devs[a] == devs[b]) { if (why) *why = "device_ids names one device more than once (RCCL takes distinct GPUs)"; return false; }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!load_api()) { if (why) *why = "librccl could not be loaded: " + g_api.error; return false; }
    return true;
}

// check the communicators of this device list out for one blocking call (created on first use, cached); comms[g] belongs to devs[g].  Blocks while another
// thread holds the same set; rccl_release hands it back (RcclLease in blocking.hip does that on every return path)
nmfx_status rccl_comms(const int *devs, int n, void **comms_out) {
    std::unique_lock<std::mutex> lk(g_mu);
    if (!load_api()) { set_error("RCCL backend: librccl could not be loaded (%s)", g_api.error.c_str()); return NMFX_ERR_UNSUPPORTED; }
    const std::vector<int> key(devs, devs + n);
    auto it = g_comms.find(key);
    if (it == g_comms.end()) {
        std::vector<ncclComm_t> c(n, nullptr);
        const ncclResult_t rc = g_api.CommInitAll(c.data(), n, devs);
        if (rc != ncclSuccess) { set_error("ncclCommInitAll over %d device(s) failed: %s", n, g_api.GetErrorString(rc)); (void)hipGetLastError(); return NMFX_ERR_HIP; }
        it = g_comms.emplace(key, CommSet{c, false}).first;
    }
    g_cv.wait(lk, [&] { return !it->second.busy; });   // (std::map nodes do not move: `it` stays valid while others insert)
    it->second.busy = true;
    for (int g = 0; g < n; ++g) comms_out[g] = it->second.c[g];
    return NMFX_OK;
}
void rccl_release(const int *devs, int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_comms.find(std::vector<int>(devs, devs + n));
    if (it != g_comms.end()) it->second.busy = false;
    g_cv.notify_all();
}

// in-place sum of `count` floats over the n devices: ONE ncclAllReduce per device inside a group (single host thread), each on its device's stream.  The caller
// holds the communicator set (rccl_comms).  The group is always closed, also when a call inside it fails
nmfx_status rccl_allreduce_f32(void *const *comms, const int *devs, hipStream_t const *streams, float *const *bufs, int n, size_t count) {
    ncclResult_t rc = g_api.GroupStart();
    if (rc != ncclSuccess) { set_error("ncclGroupStart failed: %s", g_api.GetErrorString(rc)); return NMFX_ERR_HIP; }
    hipError_t he = hipSuccess;
    for (int g = 0; g < n && rc == ncclSuccess && he == hipSuccess; ++g) {
        he = hipSetDevice(devs[g]);
        if (he == hipSuccess) rc = g_api.AllReduce(bufs[g], bufs[g], count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comms[g]), streams[g]);
    }
    const ncclResult_t re = g_api.GroupEnd();
    if (he != hipSuccess) { set_error("hipSetDevice inside the RCCL group: %s", hipGetErrorString(he)); (void)hipGetLastError(); return NMFX_ERR_HIP; }
    if (rc == ncclSuccess) rc = re;
    if (rc != ncclSuccess) { set_error("ncclAllReduce (%zu floats, %d devices) failed: %s", count, n, g_api.GetErrorString(rc)); return NMFX_ERR_HIP; }
    return NMFX_OK;
}

}  // namespace nmfx

// which library the RCCL backend runs on ("" when none can be loaded) and its version code (0 if unknown)
extern "C" const char *nmfx_rccl_library(int32_t *version) {
    std::lock_guard<std::mutex> lk(nmfx::g_mu);
    int v = 0;
    const bool ok = nmfx::load_api();
    if (ok && nmfx::g_api.GetVersion) (void)nmfx::g_api.GetVersion(&v);
    if (version) *version = v;
    return ok ? nmfx::g_api.path.c_str() : "";   // (g_api.path is written once, under the lock, before `handle` is: the pointer stays valid for the life of the process)
}
