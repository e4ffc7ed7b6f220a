// ec3d_rccl.cpp — see ec3d_rccl.hpp
#include "ec3d_rccl.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

namespace {
struct Loaded {
    ec3d_rccl_api api{};
    std::string err;
};
std::mutex g_mu;
std::map<std::string, std::unique_ptr<Loaded>> g_libs; // "" = the process's / the system's librccl
} // namespace

const ec3d_rccl_api *ec3d_rccl_load(std::string &why)
{
    // EC3D_RCCL_LIB=<path>: THIS library instead of the process's / the system's librccl (another RCCL build; the tests'
    // loopback transport, tests/support/rccl_loopback.cpp).  Loaded RTLD_LOCAL -- its nccl* symbols must not interpose a
    // real RCCL already in the process -- and announced, so that a job never runs on a substitute silently.  Looked at
    // on every call (a handle keeps the table it was created with): one process may hold handles of both kinds.
    const char *named = getenv("EC3D_RCCL_LIB");
    const std::string key = named ? named : "";
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_libs.find(key);
    if (it != g_libs.end()) {
        if (!it->second->err.empty()) {
            why = it->second->err;
            return nullptr;
        }
        return &it->second->api;
    }
    Loaded &L = *(g_libs[key] = std::make_unique<Loaded>());
    void *h = nullptr;
    if (!key.empty()) {
        h = dlopen(key.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            L.err = "EC3D_RCCL_LIB=" + key + ": " + (dlerror() ? dlerror() : "dlopen failed");
            why = L.err;
            return nullptr;
        }
        fprintf(stderr, "ec3d: RCCL entry points taken from EC3D_RCCL_LIB=%s\n", key.c_str());
    } else {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    }
    if (!h) {
        L.err = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?");
        why = L.err;
        return nullptr;
    }
    auto sym = [&](const char *n, bool required = true) -> void * {
        void *p = dlsym(h, n);
        if (!p && required && L.err.empty()) L.err = std::string("RCCL symbol missing: ") + n;
        return p;
    };
    ec3d_rccl_api &a = L.api;
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount", false);
    a.GetVersion = (decltype(a.GetVersion))sym("ncclGetVersion", false);
    if (!L.err.empty()) {
        why = L.err;
        return nullptr;
    }
    Dl_info di;
    if (dladdr((void *)a.Send, &di) && di.dli_fname) strncpy(a.path, di.dli_fname, sizeof a.path - 1);
    return &a;
}
