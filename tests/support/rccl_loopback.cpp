// rccl_loopback.cpp — a LOOPBACK implementation of the RCCL entry points the rank driver uses (csrc/ec3d_rccl.hpp), for
// tests.  TEST INFRASTRUCTURE: built into tests/libec3d_loopback.so (eddy_currents_3d_amd/build.py build_test_support), NOT
// part of libec3d_hip.so.  It exports the entry points under their RCCL names (ncclSend, ncclRecv, ...), so it stands where
// a librccl stands: the product loads it only when EC3D_RCCL_LIB names its path (csrc/ec3d_rccl.cpp announces that on stderr).
//
// Why it exists.  RCCL (like NCCL) refuses two ranks on one device ("invalid usage": tools/rccl_two_ranks_one_gpu_probe.py),
// so on a one-GPU box the one-process-per-GPU driver of csrc/ec3d_multi.hip can only run as a job of ONE rank or as the
// rehearsal of one rank — neither of which exercises what is specific to several ranks: which rows go to which neighbour, in
// which order the send / receive pairs of a group meet, the layout of the all-gathered sums, the facts the ranks exchange at
// set-up.  Here every "rank" is a thread of ONE process (its own handle, its own streams), and the calls have the semantics
// the driver relies on:
//   * ncclSend / ncclRecv inside ncclGroupStart … ncclGroupEnd: the k-th send of rank a to rank b meets the k-th receive of b
//     from a, counts must agree (a mismatch is reported, where real RCCL would hang or corrupt), the copy is ordered behind
//     the sender's stream at the time of the call and ahead of whatever the receiver's stream does next, and the sender's
//     stream does not run on before the copy has left its buffer;
//   * ncclAllGather: rank r's `count` elements land at recvbuff + r * count on every rank, ordered the same way;
//   * ncclCommInitRank returns when all ranks of the id have called it.
// Data moves by hipMemcpyAsync between the ranks' buffers (all on devices of this process); nothing touches the host.
// It is a test double for the TRANSPORT only: plans, stages, kernels and reductions are the product's.
#include <rccl/rccl.h>

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Msg {
    const void *src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr; // recorded on the sender's stream: the data is final
    hipEvent_t done = nullptr;  // recorded on the receiver's stream behind the copy
    bool done_recorded = false;
};
struct World {
    int nranks = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> box; // (from, to) -> messages in order
    uint64_t bar_gen = 0;
    int bar_count = 0;
    // the all-gather in flight
    std::vector<const void *> ag_src;
    std::vector<size_t> ag_bytes;
    std::vector<hipEvent_t> ag_ready, ag_done;
    std::string error;
};
struct Comm {
    World *w;
    int rank;
};
std::mutex g_worlds_m;
std::map<std::string, std::unique_ptr<World>> g_worlds;

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int t_group = 0;
thread_local std::vector<Op> t_ops;
thread_local std::string t_err;

size_t elem_size(ncclDataType_t t)
{
    switch (t) {
    case ncclDouble: case ncclInt64: case ncclUint64: return 8;
    case ncclFloat: case ncclInt32: case ncclUint32: return 4;
    case ncclHalf: return 2;
    default: return 1;
    }
}

ncclResult_t fail(const std::string &why)
{
    t_err = "loopback transport: " + why;
    return ncclInvalidUsage;
}

ncclResult_t flush_ops()
{
    std::vector<Op> ops;
    ops.swap(t_ops);
    std::vector<std::shared_ptr<Msg>> sent;
    // every send of the group is posted before the first receive is waited for (as a group issues its operations together)
    for (const Op &o : ops) {
        if (!o.send) continue;
        auto msg = std::make_shared<Msg>();
        msg->src = o.buf;
        msg->bytes = o.bytes;
        if (hipEventCreateWithFlags(&msg->ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&msg->done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(msg->ready, o.stream) != hipSuccess)
            return fail("event for a send");
        World *w = o.comm->w;
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->box[{o.comm->rank, o.peer}].push_back(msg);
        }
        w->cv.notify_all();
        sent.push_back(msg);
    }
    for (const Op &o : ops) {
        if (o.send) continue;
        World *w = o.comm->w;
        std::shared_ptr<Msg> msg;
        {
            std::unique_lock<std::mutex> lk(w->m);
            auto &q = w->box[{o.peer, o.comm->rank}];
            w->cv.wait(lk, [&] { return !q.empty() || !w->error.empty(); });
            if (!w->error.empty()) return fail(w->error);
            msg = q.front();
            q.pop_front();
            if (msg->bytes != o.bytes) {
                w->error = "a receive of " + std::to_string(o.bytes) + " bytes met a send of " + std::to_string(msg->bytes) +
                           " (rank " + std::to_string(o.comm->rank) + " from rank " + std::to_string(o.peer) + ")";
                w->cv.notify_all();
                return fail(w->error);
            }
        }
        if (hipStreamWaitEvent(o.stream, msg->ready, 0) != hipSuccess ||
            hipMemcpyAsync(o.buf, msg->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
            hipEventRecord(msg->done, o.stream) != hipSuccess)
            return fail("copy of a receive");
        {
            std::lock_guard<std::mutex> lk(w->m);
            msg->done_recorded = true;
        }
        w->cv.notify_all();
    }
    // the sender's stream goes on only behind the copies out of its buffers
    size_t k = 0;
    for (const Op &o : ops) {
        if (!o.send) continue;
        std::shared_ptr<Msg> msg = sent[k++];
        World *w = o.comm->w;
        {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return msg->done_recorded || !w->error.empty(); });
            if (!w->error.empty()) return fail(w->error);
        }
        if (hipStreamWaitEvent(o.stream, msg->done, 0) != hipSuccess) return fail("wait for a send's copy");
        // (the events are small; they are left to the process's end: the receiver's stream may still be using them)
    }
    return ncclSuccess;
}

ncclResult_t lb_GetUniqueId(ncclUniqueId *id)
{
    static std::mutex m;
    static std::mt19937_64 rng(std::random_device{}());
    std::lock_guard<std::mutex> lk(m);
    memset(id, 0, sizeof *id);
    for (int i = 0; i < 4; ++i) {
        const uint64_t v = rng();
        memcpy(id->internal + 8 * i, &v, 8);
    }
    return ncclSuccess;
}

ncclResult_t lb_CommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    const std::string key(id.internal, sizeof id.internal);
    World *w;
    {
        std::lock_guard<std::mutex> lk(g_worlds_m);
        auto &slot = g_worlds[key];
        if (!slot) {
            slot.reset(new World());
            slot->nranks = nranks;
            slot->ag_src.assign((size_t)nranks, nullptr);
            slot->ag_bytes.assign((size_t)nranks, 0);
            slot->ag_ready.assign((size_t)nranks, nullptr);
            slot->ag_done.assign((size_t)nranks, nullptr);
        }
        w = slot.get();
    }
    if (w->nranks != nranks || rank < 0 || rank >= nranks) return fail("ranks disagree on the size of the job");
    {
        std::unique_lock<std::mutex> lk(w->m);
        if (hipEventCreateWithFlags(&w->ag_ready[(size_t)rank], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&w->ag_done[(size_t)rank], hipEventDisableTiming) != hipSuccess)
            return fail("events of a rank");
        ++w->joined;
        w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->joined >= w->nranks; });
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{w, rank});
    return ncclSuccess;
}

ncclResult_t lb_CommDestroy(ncclComm_t c)
{
    delete reinterpret_cast<Comm *>(c);
    return ncclSuccess;
}

ncclResult_t lb_GroupStart()
{
    ++t_group;
    return ncclSuccess;
}

ncclResult_t lb_GroupEnd()
{
    if (t_group <= 0) return fail("ncclGroupEnd without ncclGroupStart");
    if (--t_group > 0) return ncclSuccess;
    return flush_ops();
}

ncclResult_t lb_Send(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    Comm *cm = reinterpret_cast<Comm *>(c);
    if (peer < 0 || peer >= cm->w->nranks) return fail("send to a rank that does not exist");
    t_ops.push_back(Op{true, const_cast<void *>(buf), count * elem_size(t), peer, cm, s});
    return t_group > 0 ? ncclSuccess : flush_ops();
}

ncclResult_t lb_Recv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    Comm *cm = reinterpret_cast<Comm *>(c);
    if (peer < 0 || peer >= cm->w->nranks) return fail("receive from a rank that does not exist");
    t_ops.push_back(Op{false, buf, count * elem_size(t), peer, cm, s});
    return t_group > 0 ? ncclSuccess : flush_ops();
}

// a reusable rendezvous of the ranks of one communicator
bool rendezvous(World *w)
{
    std::unique_lock<std::mutex> lk(w->m);
    const uint64_t gen = w->bar_gen;
    if (++w->bar_count == w->nranks) {
        w->bar_count = 0;
        ++w->bar_gen;
        w->cv.notify_all();
    } else {
        w->cv.wait(lk, [&] { return w->bar_gen != gen || !w->error.empty(); });
    }
    return w->error.empty();
}

ncclResult_t lb_AllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    Comm *cm = reinterpret_cast<Comm *>(c);
    World *w = cm->w;
    const size_t bytes = count * elem_size(t);
    const int me = cm->rank, n = w->nranks;
    // 1. every rank names its part and marks the point of its stream at which the part is final
    w->ag_src[(size_t)me] = send;
    w->ag_bytes[(size_t)me] = bytes;
    if (hipEventRecord(w->ag_ready[(size_t)me], s) != hipSuccess) return fail("all-gather: event");
    if (!rendezvous(w)) return fail(w->error);
    // 2. every rank copies all parts, in rank order, behind those points
    for (int r = 0; r < n; ++r) {
        if (w->ag_bytes[(size_t)r] != bytes) {
            std::lock_guard<std::mutex> lk(w->m);
            w->error = "all-gather: the ranks disagree on the count";
            w->cv.notify_all();
            return fail(w->error);
        }
        if (hipStreamWaitEvent(s, w->ag_ready[(size_t)r], 0) != hipSuccess ||
            hipMemcpyAsync(static_cast<char *>(recv) + (size_t)r * bytes, w->ag_src[(size_t)r], bytes, hipMemcpyDeviceToDevice, s) !=
                hipSuccess)
            return fail("all-gather: copy");
    }
    if (hipEventRecord(w->ag_done[(size_t)me], s) != hipSuccess) return fail("all-gather: event");
    if (!rendezvous(w)) return fail(w->error);
    // 3. nobody's stream runs on (and writes its part again) before every rank has copied it
    for (int r = 0; r < n; ++r)
        if (r != me && hipStreamWaitEvent(s, w->ag_done[(size_t)r], 0) != hipSuccess) return fail("all-gather: wait");
    if (!rendezvous(w)) return fail(w->error); // (the events and the table of parts are free for the next round)
    return ncclSuccess;
}

const char *lb_GetErrorString(ncclResult_t r)
{
    if (r == ncclSuccess) return "no error";
    return t_err.empty() ? "loopback transport: error" : t_err.c_str();
}
} // namespace

// the entry points under their RCCL names (signatures as <rccl/rccl.h> declares them)
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { return lb_GetUniqueId(id); }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) { return lb_CommInitRank(comm, nranks, id, rank); }
ncclResult_t ncclCommDestroy(ncclComm_t c) { return lb_CommDestroy(c); }
ncclResult_t ncclGroupStart() { return lb_GroupStart(); }
ncclResult_t ncclGroupEnd() { return lb_GroupEnd(); }
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return lb_Send(buf, count, t, peer, c, s);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return lb_Recv(buf, count, t, peer, c, s);
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    return lb_AllGather(send, recv, count, t, c, s);
}
const char *ncclGetErrorString(ncclResult_t r) { return lb_GetErrorString(r); }
ncclResult_t ncclCommCount(const ncclComm_t c, int *count)
{
    *count = reinterpret_cast<const Comm *>(c)->w->nranks;
    return ncclSuccess;
}
ncclResult_t ncclGetVersion(int *version)
{
    *version = -1; // not an RCCL: the loopback transport
    return ncclSuccess;
}
}

struct lb_api {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
};
static const lb_api *ec3d_rccl_loopback()
{
    static const lb_api api = {lb_GetUniqueId, lb_CommInitRank, lb_CommDestroy, lb_GroupStart, lb_GroupEnd,
                               lb_Send,        lb_Recv,         lb_AllGather,   lb_GetErrorString};
    return &api;
}

// ---- the transport checked by itself (tests/test_gpu_rank_loopback.py) ----------------------------------------
// Two ranks on device 0, twice over: each sends 3 doubles to the other and receives the other's in ONE group, then
// all-gathers 2 doubles -- results read back and compared; then a send of 2 doubles meets a receive of 1: both ranks must get
// an error.  0 = all as expected; the first failing check otherwise.
extern "C" int ec3d_rccl_loopback_selftest()
{
    const lb_api *api = ec3d_rccl_loopback();
    ncclUniqueId id, id2;
    api->GetUniqueId(&id);
    api->GetUniqueId(&id2);
    int rc[2] = {0, 0};
    auto rank_main = [&](int me) {
        auto bad = [&](int code) { if (!rc[me]) rc[me] = code; };
        if (hipSetDevice(0) != hipSuccess) return bad(1);
        hipStream_t st;
        double *snd, *rcv, *all;
        if (hipStreamCreate(&st) != hipSuccess || hipMalloc(&snd, 3 * 8) != hipSuccess || hipMalloc(&rcv, 3 * 8) != hipSuccess ||
            hipMalloc(&all, 4 * 8) != hipSuccess)
            return bad(2);
        ncclComm_t comm, comm2;
        if (api->CommInitRank(&comm, 2, id, me) != ncclSuccess || api->CommInitRank(&comm2, 2, id2, me) != ncclSuccess) return bad(3);
        for (int round = 0; round < 2 && !rc[me]; ++round) {
            const double mine[3] = {100.0 * me + round, 100.0 * me + round + 0.25, 100.0 * me + round + 0.5};
            if (hipMemcpyAsync(snd, mine, sizeof mine, hipMemcpyHostToDevice, st) != hipSuccess) return bad(4);
            if (api->GroupStart() != ncclSuccess || api->Send(snd, 3, ncclDouble, 1 - me, comm, st) != ncclSuccess ||
                api->Recv(rcv, 3, ncclDouble, 1 - me, comm, st) != ncclSuccess || api->GroupEnd() != ncclSuccess)
                return bad(5);
            if (api->AllGather(snd, all, 2, ncclDouble, comm, st) != ncclSuccess) return bad(6);
            double got[3], gall[4];
            if (hipMemcpyAsync(got, rcv, sizeof got, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(gall, all, sizeof gall, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                return bad(7);
            const int o = 1 - me;
            for (int i = 0; i < 3; ++i)
                if (got[i] != 100.0 * o + round + 0.25 * i) return bad(8);
            for (int r = 0; r < 2; ++r)
                for (int i = 0; i < 2; ++i)
                    if (gall[2 * r + i] != 100.0 * r + round + 0.25 * i) return bad(9);
        }
        // lengths that do not meet: rank 0 sends 2, rank 1 expects 1 (on the second communicator, which is lost with it)
        ncclResult_t e = me == 0 ? api->Send(snd, 2, ncclDouble, 1, comm2, st) : api->Recv(rcv, 1, ncclDouble, 0, comm2, st);
        if (e == ncclSuccess) return bad(10);
        if (std::string(api->GetErrorString(e)).find("met a send") == std::string::npos) return bad(11);
        (void)hipStreamSynchronize(st);
        api->CommDestroy(comm);
        api->CommDestroy(comm2);
        (void)hipFree(snd);
        (void)hipFree(rcv);
        (void)hipFree(all);
        (void)hipStreamDestroy(st);
    };
    std::thread t0(rank_main, 0), t1(rank_main, 1);
    t0.join();
    t1.join();
    return rc[0] ? rc[0] : (rc[1] ? 100 + rc[1] : 0);
}
