// ec3d_rccl.hpp — RCCL entry points resolved at run time (dlopen; libec3d_hip.so carries no link dependency on it).
//
// north_star: "RCCL halo exchange and all-reduce of the dot products per half-iteration over xGMI ... on a second HIP
// stream".  One process per GPU (eddy_currents_3d_amd/dist.py, the driver's torch.distributed.run launch) drives the
// z-slab plans of csrc/ec3d_multi.hip with these calls: ncclSend / ncclRecv pairs in a group for the halo planes,
// ncclAllGather for the eight partial sums of every rank (added in rank order by the consumer kernels: bit-identical
// decisions on every rank, run to run -- an all-reduce leaves the order to the library).
#pragma once
#include <rccl/rccl.h>

#include <string>

struct ec3d_rccl_api {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    // optional (diagnostics: ec3d_multi_rccl_info): may be null
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*GetVersion)(int *);
    char path[512]; // file the entry points were resolved from (dladdr)
};

// the table, or nullptr with the reason in `why` (library not found / symbol missing).  The process's already loaded
// librccl.so.1 is taken when there is one (under Python that is the copy PyTorch ships beside its own HIP runtime, which
// this library shares then), else the system's (/opt/rocm/lib).
// EC3D_RCCL_LIB=<path> names the library to take them from instead (announced on stderr; tests: the loopback transport of
// tests/support/rccl_loopback.cpp, which is NOT part of this library).
const ec3d_rccl_api *ec3d_rccl_load(std::string &why);
