// ec3d_rccl.cpp — see ec3d_rccl.hpp
#include "ec3d_rccl.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

const ec3d_rccl_api *ec3d_rccl_load(std::string &why)
{
    static std::mutex mu;
    static ec3d_rccl_api api;
    static bool ok = false;
    static std::string err;
    // tests of the rank driver with several ranks on ONE device (which RCCL refuses): see ec3d_rccl_loopback.cpp
    if (const char *e = getenv("EC3D_RCCL_LOOPBACK"))
        if (!strcmp(e, "1")) return ec3d_rccl_loopback();
    std::lock_guard<std::mutex> lk(mu);
    if (ok) return &api;
    if (!err.empty()) {
        why = err;
        return nullptr;
    }
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        err = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?");
        why = err;
        return nullptr;
    }
    auto sym = [&](const char *n) -> void * {
        void *p = dlsym(h, n);
        if (!p && err.empty()) err = std::string("RCCL symbol missing: ") + n;
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!err.empty()) {
        why = err;
        return nullptr;
    }
    ok = true;
    return &api;
}
