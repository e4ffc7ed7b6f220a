// ec3d_multi.hip — z-slab multi-GPU INSIDE the library: one process, N devices, one host thread per slab.
//
// The reference is serial; its caller (src/EC3D.f90:408) must not have to know that the solve
// runs on several GPUs (SURVEY §8b "Threading").  So the decomposition of eddy_currents_3d_amd/dist.py --
// rank g owns the z-planes [k0, k1), halo planes of P and S once per SpMV, three reduction points per
// iteration, rank-ordered sums -- is driven here from C++, behind one handle:
//
//   * every slab is an ordinary ec3d_ctx on its own device (ec3d_assemble_poisson_slab / ec3d_assemble_slab)
//     and runs the stages of ec3d_dist_step; one host thread per slab enqueues them, so the enqueue cost
//     per iteration does not grow with the number of GPUs;
//   * halo planes are PULLED: the consumer's side stream waits for the neighbour's "boundary rows are final"
//     event and copies the planes out of the neighbour's vector (peer access over xGMI) straight into its own
//     ghost rows -- contiguous, in place, no packing -- while the interior launch runs on the compute stream;
//   * dot products are not gathered at all: every rank collapses its partials into its own lsum[8]
//     (k_finalize), records an event, and the consumer kernels of ALL ranks read the N lsum arrays in place
//     through a pointer table (RedSrc::ptrs) and add them in rank order -- identical decisions everywhere,
//     bit-identical to the staged drivers in dist.py; lsum lives in fine-grained memory;
//   * cross-device ordering is hipStreamWaitEvent on events recorded by the other slabs' threads; a thread
//     announces "recorded" through a per-slab sequence counter the others spin on (every thread runs the
//     same plan, and always posts before it waits, so there is no cycle).
// Several slabs may share one device (tests and rehearsals on a one-GPU box): same code, the peer copies
// become local ones.
//
// ONE PROCESS PER GPU (ec3d_multi_create_rank; what torch.distributed.run launches): the same handle, plans and stages with
// ONE local slab -- this process's rank of `world` -- and RCCL as the transport: the halo planes travel as ncclSend /
// ncclRecv pairs in one group on the side stream (communicator of its own, so the transfer runs beside the interior
// launch), the eight sums of every rank are all-gathered (64 B per rank) on the compute stream and added in rank order by
// the consumer kernels, exactly like the gathered copy of eddy_currents_3d_amd/dist.py.  The whole iteration loop is
// enqueued from C++: no Python between two launches.
#include "../../include/ec3d_hip.h"
#include "ec3d_internal.hpp"
#include "ec3d_rccl.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace {
constexpr int RING = 4;
enum { CH_P = 0, CH_S = 1, CH_X = 2, CH_AP = 3, CH_R = 4, NHALO = 5, CH_SUM = 5, CH_HUB = 6, NCH = 7 };
const int kVecOf[NHALO] = {EC3D_VEC_P, EC3D_VEC_S, EC3D_VEC_X, EC3D_VEC_AP, EC3D_VEC_R};

// a run of halo planes inside one work vector: `planes` pieces of `payload` doubles, `pitch` apart
struct Run {
    int64_t start = 0, pitch = 0, payload = 0;
    int planes = 0;
    int64_t lo() const { return start; }
    int64_t hi() const { return start + (int64_t)(planes - 1) * pitch + std::max(payload, planes > 1 ? pitch : payload); }
};
struct Copy {
    int64_t dst, src, cnt;
};
struct Piece {
    int64_t off, cnt;
};

enum { OP_HALO = 0, OP_HALO_START, OP_HALO_WAIT, OP_GATHER, OP_STEP, OP_SKIP_IF_AP };
struct Op {
    int kind, arg;
    int dit = 0; // halo ops: the exchanged vector is that of iteration it + dit (P after K5 is the NEXT iteration's)
};
#define ST(x) Op{OP_STEP, EC3D_STAGE_##x}
const std::vector<Op> kBegin = {{OP_HALO, CH_X}, ST(RESID), {OP_GATHER, 0}, ST(SETUP)};
// (run with it = 0: the first iteration's P)
const std::vector<Op> kBeginVsplit = {{OP_HALO, CH_X}, ST(RESID), {OP_GATHER, 0}, ST(SETUP), {OP_HALO, CH_P, 1}};
// three reduction points per iteration (K3 is launched before ||S|| is known)
const std::vector<Op> kIter = {{OP_HALO, CH_P}, ST(K1), {OP_GATHER, 0}, ST(K2), {OP_HALO, CH_S},
                               ST(K3), {OP_GATHER, 0}, ST(K4), {OP_GATHER, 0}, ST(K5)};
// exchange hidden behind the interior planes of K1/K3 (single-component slabs on a z-marching grid)
const std::vector<Op> kIterOverlap = {{OP_HALO_START, CH_P}, ST(K1_INT), {OP_HALO_WAIT, CH_P}, ST(K1_BND), {OP_GATHER, 0},
                                      ST(K2), {OP_HALO_START, CH_S}, ST(K3_INT), {OP_HALO_WAIT, CH_S}, ST(K3_BND),
                                      {OP_GATHER, 0}, ST(K4), {OP_GATHER, 0}, ST(K5)};
// the same from the producers' side (A-V slabs, any storage): K2/K5 boundary tiles first
const std::vector<Op> kIterVsplit = {{OP_HALO_WAIT, CH_P}, ST(K1), {OP_GATHER, 0}, ST(K2_BND), {OP_HALO_START, CH_S},
                                     ST(K2_INT), {OP_HALO_WAIT, CH_S}, ST(K3), {OP_GATHER, 0}, ST(K4), {OP_GATHER, 0},
                                     ST(K5_BND), {OP_HALO_START, CH_P, 1}, ST(K5_INT)};
// both at once (plan 5, single-component slabs): the producers K2 / K5 run the boundary planes first and start the
// exchange, the consumers K1 / K3 run their interior planes before they wait for it -- the planes have the producer's
// interior launch AND the consumer's to arrive behind, instead of one of the two.  Same exchanges in the same order as
// kIterVsplit (the first P travels from the begin plan on).
const std::vector<Op> kBeginBoth = {{OP_HALO, CH_X}, ST(RESID), {OP_GATHER, 0}, ST(SETUP), {OP_HALO_START, CH_P, 1}};
const std::vector<Op> kIterBoth = {ST(K1_INT), {OP_HALO_WAIT, CH_P}, ST(K1_BND), {OP_GATHER, 0}, ST(K2_BND),
                                   {OP_HALO_START, CH_S}, ST(K2_INT), ST(K3_INT), {OP_HALO_WAIT, CH_S}, ST(K3_BND),
                                   {OP_GATHER, 0}, ST(K4), {OP_GATHER, 0}, ST(K5_BND), {OP_HALO_START, CH_P, 1}, ST(K5_INT)};
// The three-launch iteration on slabs of the single-component operator (every rank >= 32 Mi rows; ec3d_ctx::slab_fused):
// K2 inside K3, K4 as an SpMV kernel, K5 inside the next iteration's K1.  S and P are formed on the halo planes by the
// kernels that read them there (and stored into the ghost rows: Sweep::halo_store), so what travels is AP -- after
// K5-in-K1, for the S = R - alpha*AP of K2-in-K3 and the next P on the halo planes -- and R -- after K4 --: two exchanges
// and three reduction points per iteration, as before.  P and R are exchanged once before the first iteration; the lone K1
// (with its reduction point and the exchange of its AP) runs when AP = A P of the iteration does not exist yet.
const std::vector<Op> kBeginFused = {{OP_HALO, CH_X}, ST(RESID), {OP_GATHER, 0}, ST(SETUP), {OP_HALO, CH_P, 1}, {OP_HALO, CH_R}};
const std::vector<Op> kIterFused = {{OP_SKIP_IF_AP, 3}, ST(K1), {OP_GATHER, 0}, {OP_HALO, CH_AP}, ST(K3), {OP_GATHER, 0}, ST(K4),
                                    {OP_GATHER, 0}, {OP_HALO, CH_R}, ST(K5), {OP_GATHER, 0}, {OP_HALO, CH_AP, 1}};
// the same with the two exchanges hidden: the PRODUCERS of R and AP run planes 0 and np-1 first, the exchange starts, the
// interior planes follow while the planes travel; the consumer's side waits in front of the launch that reads the halo.
// Exchanges and reduction points come in the same order as in kIterFused, so ranks may mix the two.
const std::vector<Op> kIterFusedOverlap = {{OP_SKIP_IF_AP, 3}, ST(K1), {OP_GATHER, 0}, {OP_HALO, CH_AP}, {OP_HALO_WAIT, CH_AP},
                                           ST(K3), {OP_GATHER, 0}, ST(K4F_BND), {OP_HALO_START, CH_R}, ST(K4F_INT), {OP_GATHER, 0},
                                           {OP_HALO_WAIT, CH_R}, ST(K5F_BND), {OP_HALO_START, CH_AP, 1}, ST(K5F_INT),
                                           {OP_GATHER, 0}};
#undef ST

struct Slab {
    ec3d_ctx *c = nullptr;
    int rank = 0, device = 0;
    int32_t k0 = 0, k1 = 0, e0 = 0, e1 = 0; // owned planes [k0,k1), held planes [e0,e1)
    double *lsum = nullptr;
    bool lsum_fine = false;
    uint64_t api_calls = 0, api_iters = 0; // runtime calls / iterations of the last ec3d_multi_iterate (this rank)
    const double **ptr_table = nullptr; // device: every rank's lsum
    hipStream_t side = nullptr;
    hipEvent_t ev_ready[NHALO][RING] = {}, ev_halo[NHALO][RING] = {}, ev_sum[RING] = {}, ev_hub[RING] = {};
    uint64_t seq[NCH] = {0, 0, 0, 0, 0, 0, 0};
    std::atomic<uint64_t> posted[NCH];
    std::vector<Run> send_lo, recv_lo, send_hi, recv_hi; // towards rank-1 / rank+1, same order on both sides
    std::vector<Copy> pull_lo, pull_hi;                  // my ghost rows <- neighbour's rows
    std::vector<Piece> snd_lo, rcv_lo, snd_hi, rcv_hi;   // RCCL: contiguous pieces of my rows to send / my ghost rows to fill
    bool split_ok = false;   // K2 / K5 can run as boundary + interior launch on this slab
    bool overlap_ok = false; // K1 / K3 can run as interior + boundary launch on this slab
    int plan = 0; // 0 plain, 1 overlap (K1/K3 interior + boundary), 2 vsplit (K2/K5 boundary first), 3 three launches
                  // (kIterFused), 4 three launches with the producers of R and AP split around the exchange, 5 = 1 and 2
                  // together (kIterBoth)
    int32_t *stop_pinned = nullptr;
    hipEvent_t ev_stop[2] = {};
    // A-V slab: local reference order [Ax_ext | Ay_ext | Az_ext | U_ext] <-> the global vector
    int64_t nC_ext = 0, nU_ext = 0, n_local = 0;
    std::vector<int32_t> u_glob;        // global U index of every held U unknown
    int64_t own_lo[4] = {0}, own_hi[4] = {0}; // owned rows, local reference numbering
    std::vector<double> io;             // staging for host <-> device copies of an A-V slab
    std::string err;
    // where this slab's thread is (watchdog report): operation, stage / channel, iteration, sequence number
    std::atomic<const char *> at{"idle"};
    std::atomic<int> at_arg{0}, at_it{0};
    std::atomic<uint64_t> at_seq{0};
    Slab() { for (auto &p : posted) p.store(0); }
};

struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    std::function<int(int)> job;
    uint64_t gen = 0;
    int pending = 0;
    std::vector<int> rc;
    bool quit = false;
};
} // namespace

struct ec3d_multi {
    int n = 0;     // LOCAL slabs = host threads of this process (all of them, or one: this process's rank)
    int world = 0; // ranks of the job (== n unless one process per GPU)
    // one process per GPU: RCCL transport (ec3d_multi_create_rank)
    const ec3d_rccl_api *nccl = nullptr;
    ncclComm_t comm_halo = nullptr, comm_sum = nullptr;
    int comm_rank = 0, comm_world = 1; // this process in the communicators
    // Rehearsal of ONE rank of a larger job on one GPU (ec3d_multi_create_rank with as_world > nranks = 1): the slab, plan,
    // launches and RCCL calls of rank `as_rank` of `as_world` -- world and Slab::rank then describe THAT geometry -- with
    // every neighbour mapped to this process itself (send / recv to self) and the sums gathered over the one real rank.
    // Timing and call counts are those of the real rank; the numbers computed are not a solution of anything.
    bool rehearse = false;
    double *gsum = nullptr;    // [world * P_NSLOT] the all-gathered sums
    double *agbuf = nullptr;   // [(world + 1) * kFacts] set-up exchanges between the ranks
    std::vector<std::unique_ptr<Slab>> slab;
    Pool pool;
    std::atomic<bool> abort{false};
    int kind = 0; // 0: no matrix, 1: single component (ec3d_assemble_poisson), 2: A-V system
    int32_t sdx = 0, sdy = 0, sdz = 0;
    int64_t kdz = 0, nC_glob = 0, nU_glob = 0, n_glob = 0;
    int64_t nnz = 0;
    int chunk = 1; // iterations between two looks at the stop flag: ONE value for the job (ranks leave together)
};

namespace {
// only taken under EC3D_MULTI_SERIALIZE_WAITS=1 (see cross_wait)
std::mutex g_cross_wait;
// HIP runtime calls issued by the slab threads (launches are counted in run_plan): what the host pays per iteration
thread_local uint64_t t_api_calls = 0;
#define MHIP(call)                                                                             \
    do {                                                                                       \
        ++t_api_calls;                                                                         \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ec3d_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                 \
            return 100;                                                                        \
        }                                                                                      \
    } while (0)

#define MNCCL(m_, call)                                                                        \
    do {                                                                                       \
        ++t_api_calls;                                                                         \
        ncclResult_t e_ = (call);                                                              \
        if (e_ != ncclSuccess) {                                                               \
            ec3d_set_error(std::string(#call) + ": " + (m_)->nccl->GetErrorString(e_));        \
            return 108;                                                                        \
        }                                                                                      \
    } while (0)

void worker(ec3d_multi *m, int r)
{
    Pool &p = m->pool;
    (void)hipSetDevice(m->slab[(size_t)r]->device);
    uint64_t seen = 0;
    for (;;) {
        std::function<int(int)> job;
        {
            std::unique_lock<std::mutex> lk(p.m);
            p.cv.wait(lk, [&] { return p.quit || p.gen != seen; });
            if (p.quit) return;
            seen = p.gen;
            job = p.job;
        }
        int rc = job(r);
        if (rc) {
            m->slab[(size_t)r]->err = ec3d_last_error();
            m->abort.store(true);
        }
        {
            std::lock_guard<std::mutex> lk(p.m);
            p.rc[(size_t)r] = rc;
            if (--p.pending == 0) p.done_cv.notify_all();
        }
    }
}

// run fn(rank) on every slab's thread; first failure wins (its message becomes ec3d_last_error())
// `watch`: the job makes the slabs wait for each other (halo pulls, reduction points), so a stall is possible and the
// watchdog is armed; jobs that only work on their own slab (assembly, uploads, creation) may take as long as they need.
int run_all(ec3d_multi *m, const std::function<int(int)> &fn, bool watch = false)
{
    Pool &p = m->pool;
    {
        std::unique_lock<std::mutex> lk(p.m);
        p.job = fn;
        p.pending = m->n;
        std::fill(p.rc.begin(), p.rc.end(), 0);
        ++p.gen;
        p.cv.notify_all();
        // watchdog: a job that makes no progress for EC3D_MULTI_WATCHDOG seconds (default 120; 0 = off) is a
        // deadlock between the slabs' streams or threads.  Say where every rank stands, let the ranks that
        // still can give up, and end the process if a thread stays stuck inside the runtime: there is no way
        // to cancel it, and the interfaces this library stands behind have no channel for "hung".
        static const int limit_env = getenv("EC3D_MULTI_WATCHDOG") ? atoi(getenv("EC3D_MULTI_WATCHDOG")) : 120;
        const int limit = watch ? limit_env : 0;
        auto report = [&]() {
            std::string r;
            for (auto &s : m->slab)
                r += "  rank " + std::to_string(s->rank) + ": " + s->at.load() + " arg " + std::to_string(s->at_arg.load()) +
                     " it " + std::to_string(s->at_it.load()) + " seq " + std::to_string(s->at_seq.load()) + " posted [" +
                     std::to_string(s->posted[0].load()) + " " + std::to_string(s->posted[1].load()) + " " +
                     std::to_string(s->posted[2].load()) + " " + std::to_string(s->posted[3].load()) + " " +
                     std::to_string(s->posted[4].load()) + " " + std::to_string(s->posted[5].load()) + " " +
                     std::to_string(s->posted[6].load()) + "]\n";
            return r;
        };
        if (limit <= 0) {
            p.done_cv.wait(lk, [&] { return p.pending == 0; });
        } else {
            std::string last;
            for (;;) {
                if (p.done_cv.wait_for(lk, std::chrono::seconds(limit), [&] { return p.pending == 0; })) break;
                const std::string now = report();
                if (now != last) { // still moving (a long solve): keep waiting
                    last = now;
                    continue;
                }
                fprintf(stderr, "libec3d_hip: multi-GPU job stalled for %d s:\n%s", limit, now.c_str());
                fflush(stderr);
                m->abort.store(true);
                if (p.done_cv.wait_for(lk, std::chrono::seconds(10), [&] { return p.pending == 0; })) break;
                // no signal, no core dump of a 100 GB process, an exit status the caller's shell can read; never a
                // re-exec (the process has touched the GPU)
                fprintf(stderr, "libec3d_hip: a rank is stuck inside the HIP runtime; ending the process (exit status 86)\n");
                fflush(stderr);
                fflush(stdout);
                _Exit(86);
            }
        }
    }
    int rc = 0;
    for (int r = 0; r < m->n; ++r)
        if (p.rc[(size_t)r]) {
            // a rank that only gave up because another one failed reports 90
            if (!rc || rc == 90) {
                rc = p.rc[(size_t)r];
                ec3d_set_error("rank " + std::to_string(r) + ": " + m->slab[(size_t)r]->err);
            }
        }
    if (m->abort.load()) { // leave every stream idle and the channels aligned for the next call
        for (auto &s : m->slab) {
            (void)hipSetDevice(s->device);
            (void)hipDeviceSynchronize();
            for (int ch = 0; ch < NCH; ++ch) {
                s->seq[ch] = 0;
                s->posted[ch].store(0);
            }
        }
        m->abort.store(false);
    }
    return rc;
}

int wait_posted(ec3d_multi *m, Slab &me, Slab &peer, int ch, uint64_t q)
{
    int spins = 0;
    me.at.store("wait_posted");
    me.at_arg.store(ch * 100 + peer.rank);
    me.at_seq.store(q);
    while (peer.posted[ch].load(std::memory_order_acquire) < q) {
        if (m->abort.load(std::memory_order_relaxed)) {
            ec3d_set_error("gave up: another rank failed");
            return 90;
        }
        if (++spins > 256) std::this_thread::yield();
    }
    return 0;
}

int cross_wait(hipStream_t stream, hipEvent_t ev)
{
    // hipStreamWaitEvent is thread safe like every HIP entry point and the threads issue these waits concurrently.
    // Rounds 2 and 3 took a process-wide mutex around every one of them by default -- "a precaution, not a measured
    // need": eight rank threads then queued up behind it at every reduction point and exchange.  It protected against
    // nothing that was ever observed (the suite's 2-8 slab cases and tools/multi_stress.py -- 3, 4 and 8 slabs on one card,
    // 20 x 7 repeated solves each, bit-identical every time: profiles/r04_multi_stress.log -- run with it off), so it is now
    // off; EC3D_MULTI_SERIALIZE_WAITS=1 brings it back for a run that wants to rule the runtime out.
    static const bool serialize = getenv("EC3D_MULTI_SERIALIZE_WAITS") && atoi(getenv("EC3D_MULTI_SERIALIZE_WAITS")) != 0;
    if (serialize) {
        std::lock_guard<std::mutex> lk(g_cross_wait);
        MHIP(hipStreamWaitEvent(stream, ev, 0));
    } else {
        MHIP(hipStreamWaitEvent(stream, ev, 0));
    }
    return 0;
}

// Vector `vec` of iteration `it` in slab `owner`'s memory.  The ring position is worked out from MY handle's state
// (every rank of a job runs the same plan with the same depths, and my state is current when I get here), the pointer
// taken from the owner's tables, which do not change while a job runs: another rank's thread may be iterations ahead or
// behind with its own host-side bookkeeping.
constexpr int kFacts = 15;         // doubles every rank tells the others at set-up (finish_setup)
constexpr int kPlainVec = INT_MIN; // `it` of a caller that means the plain work vector (uploads, probes, the time loop's X)
double *vec_of(const ec3d_ctx *me, const ec3d_ctx *owner, int vec, int it)
{
    if (it == kPlainVec) return owner->vec[vec];
    const bool f51 = ec3d_fused51(me);
    const int D = ec3d_xdefer(me), pd = me->pdepth;
    switch (vec) {
    case EC3D_VEC_P: return (f51 || D > 1) ? owner->pbuf[((it + me->p_off) % pd + pd) % pd] : owner->vec[EC3D_VEC_P];
    case EC3D_VEC_AP: return f51 ? owner->apbuf[it & 1] : owner->vec[EC3D_VEC_AP];
    case EC3D_VEC_S: return D > 1 ? owner->sbuf[((it % me->sdepth) + me->sdepth) % me->sdepth] : owner->vec[EC3D_VEC_S];
    default: return owner->vec[vec];
    }
}

int halo_start(ec3d_multi *m, Slab &s, int v, int it = kPlainVec)
{
    const uint64_t q = ++s.seq[v];
    const int i = (int)(q % RING), vi = kVecOf[v];
    s.at.store("halo_start:record");
    s.at_arg.store(v);
    s.at_seq.store(q);
    MHIP(hipEventRecord(s.ev_ready[v][i], s.c->stream));
    s.posted[v].store(q, std::memory_order_release);
    // my own earlier readers of the ghost rows are behind this point of my compute stream
    MHIP(hipStreamWaitEvent(s.side, s.ev_ready[v][i], 0));
    if (m->nccl) {
        // one process per GPU: my boundary rows go out and my ghost rows come in as send / recv pairs of ONE group on the
        // side stream (the neighbour's matching pair is in its group; lower neighbour first on every rank)
        // (the k-th send to a neighbour meets its k-th receive from me: both sides list the blocks in the same order)
        double *mine = vec_of(s.c, s.c, vi, it);
        const bool lo = s.rank > 0 && !(s.snd_lo.empty() && s.rcv_lo.empty());
        const bool hi = s.rank + 1 < m->world && !(s.snd_hi.empty() && s.rcv_hi.empty());
        if (lo || hi) {
            s.at.store("halo_start:rccl group");
            MNCCL(m, m->nccl->GroupStart());
            if (lo) {
                const int pr = m->rehearse ? m->comm_rank : s.rank - 1;
                for (const Piece &c : s.snd_lo) MNCCL(m, m->nccl->Send(mine + c.off, (size_t)c.cnt, ncclDouble, pr, m->comm_halo, s.side));
                for (const Piece &c : s.rcv_lo) MNCCL(m, m->nccl->Recv(mine + c.off, (size_t)c.cnt, ncclDouble, pr, m->comm_halo, s.side));
            }
            if (hi) {
                const int pr = m->rehearse ? m->comm_rank : s.rank + 1;
                for (const Piece &c : s.snd_hi) MNCCL(m, m->nccl->Send(mine + c.off, (size_t)c.cnt, ncclDouble, pr, m->comm_halo, s.side));
                for (const Piece &c : s.rcv_hi) MNCCL(m, m->nccl->Recv(mine + c.off, (size_t)c.cnt, ncclDouble, pr, m->comm_halo, s.side));
            }
            MNCCL(m, m->nccl->GroupEnd());
        }
        MHIP(hipEventRecord(s.ev_halo[v][i], s.side));
        return 0;
    }
    for (int dir = -1; dir <= 1; dir += 2) {
        const std::vector<Copy> &cp = dir < 0 ? s.pull_lo : s.pull_hi;
        const int pr = s.rank + dir;
        if (pr < 0 || pr >= m->n || cp.empty()) continue;
        Slab &peer = *m->slab[(size_t)pr];
        int rc = wait_posted(m, s, peer, v, q);
        if (rc) return rc;
        s.at.store("halo_start:cross_wait");
        if ((rc = cross_wait(s.side, peer.ev_ready[v][i]))) return rc;
        s.at.store("halo_start:copy");
        double *mine = vec_of(s.c, s.c, vi, it);
        const double *theirs = vec_of(s.c, peer.c, vi, it);
        // EC3D_MULTI_FORCE_PEER_API=1: the peer-copy call also between slabs of ONE device (tests on a one-GPU box)
        static const bool force_peer = getenv("EC3D_MULTI_FORCE_PEER_API") && atoi(getenv("EC3D_MULTI_FORCE_PEER_API")) != 0;
        for (const Copy &c : cp) {
            if (peer.device == s.device && !force_peer)
                MHIP(hipMemcpyAsync(mine + c.dst, theirs + c.src, (size_t)c.cnt * 8, hipMemcpyDeviceToDevice, s.side));
            else
                MHIP(hipMemcpyPeerAsync(mine + c.dst, s.device, theirs + c.src, peer.device, (size_t)c.cnt * 8, s.side));
        }
    }
    MHIP(hipEventRecord(s.ev_halo[v][i], s.side));
    return 0;
}

int halo_wait(Slab &s, int v)
{
    if (s.seq[v] == 0) return 0;
    MHIP(hipStreamWaitEvent(s.c->stream, s.ev_halo[v][(int)(s.seq[v] % RING)], 0));
    return 0;
}

// "gather": nothing moves -- my lsum is final behind my event; my next kernel may read everybody's once all
// events are behind it.  All-to-all waits would cost N-1 cross-stream waits per rank and point (measured on
// one card: 74 / 157 / 373 / 1022 us of enqueue time per iteration at 1 / 2 / 4 / 8 slabs).  Instead the "final"
// events are folded up a binomial tree: rank r waits for its children r + 2^j (every j with r % 2^(j+1) == 0 and
// r + 2^j < N; a child's event is recorded behind the child's own waits, so it covers the child's whole subtree),
// records its own event, and rank 0 -- behind its log2(N) children -- records ONE event every other rank waits
// for.  Waits per reduction point: N - 1 going up (at most ceil(log2 N) on any one rank: 3 on rank 0 at N = 8, where
// a flat hub issued 7) plus N - 1 going down.  EC3D_MULTI_FLAT_HUB=1 restores the flat hub.
int gather(ec3d_multi *m, Slab &s)
{
    const uint64_t q = ++s.seq[CH_SUM];
    const int i = (int)(q % RING);
    s.at.store("gather:record");
    s.at_seq.store(q);
    if (m->nccl) {
        // one process per GPU: every rank's eight sums to every rank (64 B each), on the compute stream, where the
        // producer's collapse launch wrote them and the consumer kernel reads the gathered copy
        s.at.store("gather:rccl all_gather");
        MNCCL(m, m->nccl->AllGather(s.lsum, m->gsum, P_NSLOT, ncclDouble, m->comm_sum, s.c->stream));
        return 0;
    }
    if (m->n == 1) return 0;
    static const bool flat = getenv("EC3D_MULTI_FLAT_HUB") && atoi(getenv("EC3D_MULTI_FLAT_HUB")) != 0;
    Slab &hub = *m->slab[0];
    int rc = 0;
    // children of this rank in the tree (flat hub: every other rank is a child of rank 0)
    for (int step = 1; step < m->n; step <<= 1) {
        if (flat ? s.rank != 0 : (s.rank & (2 * step - 1)) != 0) break;
        const int lo = flat ? 1 : s.rank + step, hi = flat ? m->n : std::min(m->n, lo + 1);
        for (int h = lo; h < hi; ++h) {
            Slab &peer = *m->slab[(size_t)h];
            if ((rc = wait_posted(m, s, peer, CH_SUM, q))) return rc;
            s.at.store("gather:cross_wait");
            if ((rc = cross_wait(s.c->stream, peer.ev_sum[i]))) return rc;
        }
        if (flat) break;
    }
    if (s.rank != 0) {
        MHIP(hipEventRecord(s.ev_sum[i], s.c->stream)); // behind my sums and my subtree's
        s.posted[CH_SUM].store(q, std::memory_order_release);
        if ((rc = wait_posted(m, s, hub, CH_HUB, q))) return rc;
        s.at.store("gather:cross_wait hub");
        return cross_wait(s.c->stream, hub.ev_hub[i]);
    }
    MHIP(hipEventRecord(s.ev_hub[i], s.c->stream)); // behind rank 0's own sums and everybody else's
    s.posted[CH_HUB].store(q, std::memory_order_release);
    return 0;
}

// a slab too thin for interior + boundary launches of K1 / K3 inside plan 5: the whole kernel where the boundary launch stands
int unsplit_spmv_stage(int st)
{
    switch (st) {
    case EC3D_STAGE_K1_BND: return EC3D_STAGE_K1;
    case EC3D_STAGE_K3_BND: return EC3D_STAGE_K3;
    case EC3D_STAGE_K1_INT:
    case EC3D_STAGE_K3_INT: return -1;
    default: return st;
    }
}

int unsplit_stage(int st)
{
    switch (st) {
    case EC3D_STAGE_K2_BND: return EC3D_STAGE_K2;
    case EC3D_STAGE_K5_BND: return EC3D_STAGE_K5;
    case EC3D_STAGE_K2_INT:
    case EC3D_STAGE_K5_INT: return -1;
    default: return st;
    }
}

// kernel (0..4 = K1..K5) a stage's time is booked under; -1: none
int kernel_of_stage(int st)
{
    switch (st) {
    case EC3D_STAGE_K1: case EC3D_STAGE_K1_INT: case EC3D_STAGE_K1_BND: return 0;
    case EC3D_STAGE_K2: case EC3D_STAGE_K2_INT: case EC3D_STAGE_K2_BND: return 1;
    case EC3D_STAGE_K3: case EC3D_STAGE_K3_INT: case EC3D_STAGE_K3_BND: return 2;
    case EC3D_STAGE_K4: case EC3D_STAGE_K4F_BND: case EC3D_STAGE_K4F_INT: return 3;
    case EC3D_STAGE_K5: case EC3D_STAGE_K5_INT: case EC3D_STAGE_K5_BND: case EC3D_STAGE_K5F_BND: case EC3D_STAGE_K5F_INT: return 4;
    default: return -1;
    }
}

struct StageTimer {
    std::vector<hipEvent_t> ev; // pairs
    std::vector<int> kern;      // 0 .. 4: the stage's kernel; T_GATHER / T_HALO_WAIT: a reduction point / a wait for halo planes
    bool sync_points = false;   // bracket the reduction points and the halo waits on the compute stream as well
};
enum { T_GATHER = 5, T_HALO_WAIT = 6 };
// instrumented pass: how long the compute stream stands at a reduction point / a wait for halo planes (an event in front, one
// behind: with nothing to wait for, the pair measures what two events cost)
template <class F> int timed_sync(StageTimer *tm, Slab &s, int kind, F &&f)
{
    if (!tm || !tm->sync_points) return f();
    hipEvent_t a, b;
    MHIP(hipEventCreate(&a));
    MHIP(hipEventCreate(&b));
    MHIP(hipEventRecord(a, s.c->stream));
    int rc = f();
    if (rc) return rc;
    MHIP(hipEventRecord(b, s.c->stream));
    tm->ev.push_back(a);
    tm->ev.push_back(b);
    tm->kern.push_back(kind);
    return 0;
}

int run_plan(ec3d_multi *m, Slab &s, const std::vector<Op> &plan, int it, double tol, StageTimer *tm)
{
    int rc = 0;
    s.at_it.store(it);
    for (size_t oi = 0; oi < plan.size(); ++oi) {
        const Op &op = plan[oi];
        switch (op.kind) {
        case OP_HALO:
            if ((rc = halo_start(m, s, op.arg, it + op.dit))) return rc;
            if ((rc = timed_sync(tm, s, T_HALO_WAIT, [&] { return halo_wait(s, op.arg); }))) return rc;
            break;
        case OP_HALO_START: if ((rc = halo_start(m, s, op.arg, it + op.dit))) return rc; break;
        case OP_HALO_WAIT: if ((rc = timed_sync(tm, s, T_HALO_WAIT, [&] { return halo_wait(s, op.arg); }))) return rc; break;
        case OP_GATHER: if ((rc = timed_sync(tm, s, T_GATHER, [&] { return gather(m, s); }))) return rc; break;
        case OP_SKIP_IF_AP: // AP = A P of this iteration came out of the last K5-in-K1: no lone K1, no sum, no exchange
            if (it != 1 && s.c->ap_valid_for == it) oi += (size_t)op.arg;
            break;
        default: {
            int st = op.arg;
            if ((s.plan == 2 || s.plan == 5) && !s.split_ok) st = unsplit_stage(st);
            if (s.plan == 5 && !s.overlap_ok) st = unsplit_spmv_stage(st);
            if (st < 0) break;
            s.at.store("stage");
            s.at_arg.store(st);
            const int k = tm ? kernel_of_stage(st) : -1;
            t_api_calls += (uint64_t)ec3d_dist_launches(s.c, st, it); // the kernel launches of the stage
            if (k >= 0) {
                hipEvent_t a, b;
                MHIP(hipEventCreate(&a));
                MHIP(hipEventCreate(&b));
                MHIP(hipEventRecord(a, s.c->stream));
                rc = ec3d_dist_step(s.c, st, it, tol);
                MHIP(hipEventRecord(b, s.c->stream));
                tm->ev.push_back(a);
                tm->ev.push_back(b);
                tm->kern.push_back(k);
            } else {
                rc = ec3d_dist_step(s.c, st, it, tol);
            }
            if (rc) return rc;
        }
        }
    }
    return 0;
}

const std::vector<Op> &begin_plan(const Slab &s)
{
    return s.plan == 5 ? kBeginBoth : s.plan >= 3 ? kBeginFused : s.plan == 2 ? kBeginVsplit : kBegin;
}
const std::vector<Op> &iter_plan(const Slab &s)
{
    return s.plan == 5 ? kIterBoth : s.plan == 4 ? kIterFusedOverlap : s.plan == 3 ? kIterFused : s.plan == 2 ? kIterVsplit
           : s.plan == 1 ? kIterOverlap : kIter;
}

int drain(Slab &s)
{
    s.at.store("drain:side stream");
    MHIP(hipStreamSynchronize(s.side));
    s.at.store("drain:compute stream");
    MHIP(hipStreamSynchronize(s.c->stream));
    s.at.store("idle");
    return 0;
}

void slab_bounds(int sdz, int rank, int world, int32_t &k0, int32_t &k1)
{
    const int base = sdz / world, rem = sdz % world;
    k0 = rank * base + std::min(rank, rem);
    k1 = k0 + base + (rank < rem ? 1 : 0);
}

// Peer access from this slab's device to every other device that holds a slab.  Runs on every slab's thread BEFORE
// anything is allocated (ec3d_multi_create), so that every allocation a peer will touch -- the work vectors the
// halo planes are pulled from, lsum, which the peers' kernels dereference -- is made with the mappings in place.
int enable_peers(ec3d_multi *m, Slab &s)
{
    MHIP(hipSetDevice(s.device));
    for (auto &o : m->slab) {
        if (o->device == s.device) continue;
        int can = 0;
        MHIP(hipDeviceCanAccessPeer(&can, s.device, o->device));
        if (!can) {
            ec3d_set_error("ec3d_multi: device " + std::to_string(s.device) + " cannot access device " +
                           std::to_string(o->device) + " (no peer path)");
            return 104;
        }
        hipError_t e = hipDeviceEnablePeerAccess(o->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            ec3d_set_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            return 100;
        }
        (void)hipGetLastError();
    }
    return 0;
}

// (re)create the per-slab context and the multi-rank plumbing around it
int slab_reset(ec3d_multi *m, Slab &s)
{
    MHIP(hipSetDevice(s.device));
    if (!s.c) {
        int rc = ec3d_create(&s.c, s.device);
        if (rc) return rc;
        MHIP(hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking));
        for (int v = 0; v < NHALO; ++v)
            for (int i = 0; i < RING; ++i) {
                MHIP(hipEventCreateWithFlags(&s.ev_ready[v][i], hipEventDisableTiming));
                MHIP(hipEventCreateWithFlags(&s.ev_halo[v][i], hipEventDisableTiming));
            }
        for (int i = 0; i < RING; ++i) MHIP(hipEventCreateWithFlags(&s.ev_sum[i], hipEventDisableTiming));
        for (int i = 0; i < RING; ++i) MHIP(hipEventCreateWithFlags(&s.ev_hub[i], hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) MHIP(hipEventCreateWithFlags(&s.ev_stop[i], hipEventDisableTiming));
        MHIP(hipHostMalloc(&s.stop_pinned, 2 * sizeof(int32_t), hipHostMallocDefault));
        // The 8 sums every other GPU's kernels read in place, launch after launch: fine-grained (coherent across
        // devices) memory.  Peer access between all the devices involved was enabled BEFORE this allocation
        // (enable_peers, ec3d_multi_create).  Plain hipMalloc memory is only good enough when every slab sits
        // on this same device (rehearsals on a one-GPU box); across devices there is no silent fallback.
        bool several_devices = false;
        for (auto &o : m->slab) several_devices |= o->device != s.device;
        if (hipExtMallocWithFlags((void **)&s.lsum, P_NSLOT * sizeof(double), hipDeviceMallocFinegrained) == hipSuccess) {
            s.lsum_fine = true;
        } else {
            (void)hipGetLastError();
            if (several_devices) {
                ec3d_set_error("ec3d_multi: device " + std::to_string(s.device) + " cannot allocate fine-grained memory "
                               "(hipExtMallocWithFlags, hipDeviceMallocFinegrained) for the partial sums the other GPUs "
                               "read in place; coarse-grained memory is not coherent across devices -- refusing to run "
                               "on several devices without it");
                return 107;
            }
            MHIP(hipMalloc(&s.lsum, P_NSLOT * sizeof(double)));
        }
        MHIP(hipMemset(s.lsum, 0, P_NSLOT * sizeof(double)));
        MHIP(hipMalloc(&s.ptr_table, (size_t)m->n * sizeof(double *)));
    }
    s.send_lo.clear(); s.recv_lo.clear(); s.send_hi.clear(); s.recv_hi.clear();
    s.pull_lo.clear(); s.pull_hi.clear();
    s.u_glob.clear();
    s.split_ok = false;
    s.plan = 0;
    return 0;
}

// after every slab has its matrix: pointer tables, dist mode, halo copy lists, launch plans
int finish_setup(ec3d_multi *m)
{
    // copy lists: my recv run i towards rank+1 <- rank+1's send run i towards me (and the mirror image)
    auto pair_up = [&](const std::vector<Run> &recv, const std::vector<Run> &send, std::vector<Copy> &out) -> int {
        out.clear();
        if (recv.size() != send.size()) {
            ec3d_set_error("ec3d_multi: neighbouring slabs disagree on the halo layout");
            return 105;
        }
        for (size_t i = 0; i < recv.size(); ++i) {
            const Run &r = recv[i], &t = send[i];
            if (r.planes != t.planes || r.payload != t.payload) {
                ec3d_set_error("ec3d_multi: neighbouring slabs disagree on the halo size");
                return 105;
            }
            if (r.payload == 0) continue;
            if (r.pitch == t.pitch || r.planes == 1) { // one contiguous piece (plane padding included)
                const int64_t cnt = r.planes == 1 ? r.payload : (int64_t)r.planes * r.pitch;
                out.push_back(Copy{r.start, t.start, cnt});
            } else {
                for (int p = 0; p < r.planes; ++p) out.push_back(Copy{r.start + p * r.pitch, t.start + p * t.pitch, r.payload});
            }
        }
        return 0;
    };
    int rc = 0;
    // one process per GPU: contiguous pieces of my own runs (the neighbour's matching lists come out of the same rules on
    // the same grid and storage format, which the ranks compare below)
    auto pieces_of = [](const std::vector<Run> &runs, std::vector<Piece> &out) {
        out.clear();
        for (const Run &r : runs) {
            if (r.payload == 0) continue;
            out.push_back(Piece{r.start, r.planes == 1 ? r.payload : (int64_t)r.planes * r.pitch});
        }
    };
    if (m->nccl) {
        Slab &s = *m->slab[0];
        pieces_of(s.send_lo, s.snd_lo);
        pieces_of(s.recv_lo, s.rcv_lo);
        pieces_of(s.send_hi, s.snd_hi);
        pieces_of(s.recv_hi, s.rcv_hi);
    } else {
        for (int g = 0; g < m->n; ++g) {
            Slab &s = *m->slab[(size_t)g];
            if (g + 1 < m->n && (rc = pair_up(s.recv_hi, m->slab[(size_t)g + 1]->send_lo, s.pull_hi))) return rc;
            if (g > 0 && (rc = pair_up(s.recv_lo, m->slab[(size_t)g - 1]->send_hi, s.pull_lo))) return rc;
        }
    }
    std::vector<const double *> tab((size_t)m->n);
    for (int g = 0; g < m->n; ++g) tab[(size_t)g] = m->slab[(size_t)g]->lsum;
    // What every rank has to know of every other: can it run the three-launch iteration, the depth of its rings, its
    // storage format, its size, what it sends to its neighbours.  One process: read off the slabs; one process per GPU:
    // eight doubles per rank, all-gathered once.
    // (h_*: a digest of the SEQUENCE of piece lengths of that direction -- one process per GPU: the k-th ncclSend of a
    // rank must meet the k-th ncclRecv of its neighbour with the same count, and a mismatch there is a hang, not an error)
    struct RankFacts { double fused_ok, xd, sav, n_pad, snd_lo, rcv_lo, snd_hi, rcv_hi, xasync, both_splits, fsplit,
                       h_snd_lo, h_rcv_lo, h_snd_hi, h_rcv_hi; };
    static_assert(sizeof(RankFacts) == kFacts * sizeof(double), "kFacts doubles");
    std::vector<RankFacts> facts((size_t)(m->nccl ? m->comm_world : m->n));
    auto facts_of = [&](const Slab &sl) {
        const ec3d_ctx *c = sl.c;
        auto total = [](const std::vector<Run> &rs) { double t = 0; for (const Run &r : rs) t += (double)r.payload * r.planes; return t; };
        RankFacts f{};
        f.fused_ok = (c->fuse23_ok && c->fuse51_ok && c->k4s_ok && c->pp_base && c->own_vectors) ? 1.0 : 0.0;
        f.xd = (c->pp_base && c->own_vectors) ? (double)c->xdefer : 1.0;
        f.xasync = (c->pp_base && c->own_vectors && c->xasync_cap) ? 1.0 : 0.0;
        // (an A-V slab splits K2 / K5 by tile lists made from its halo runs: always possible where there is a neighbour)
        f.both_splits = (c->have_matrix && c->can_overlap && (c->A.sav || ec3d_dist_can_split_planes(c))) ? 1.0 : 0.0;
        f.sav = c->A.sav ? 1.0 : 0.0;
        f.n_pad = (double)c->A.n_pad;
        f.snd_lo = total(sl.send_lo); f.rcv_lo = total(sl.recv_lo); f.snd_hi = total(sl.send_hi); f.rcv_hi = total(sl.recv_hi);
        f.fsplit = c->can_fsplit ? 1.0 : 0.0;
        // FNV-1a over (count, then every piece's length in order), folded to 52 bits: exact in a double
        auto digest = [](const std::vector<Piece> &ps) {
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](uint64_t v) { for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xFFu; h *= 1099511628211ull; } };
            mix((uint64_t)ps.size());
            for (const Piece &q : ps) mix((uint64_t)q.cnt);
            return (double)((h ^ (h >> 52)) & ((1ull << 52) - 1));
        };
        f.h_snd_lo = digest(sl.snd_lo); f.h_rcv_lo = digest(sl.rcv_lo); f.h_snd_hi = digest(sl.snd_hi); f.h_rcv_hi = digest(sl.rcv_hi);
        return f;
    };
    if (m->nccl) {
        const RankFacts mine = facts_of(*m->slab[0]);
        rc = run_all(m, [&](int) -> int {
            Slab &s = *m->slab[0];
            double *snd = m->agbuf + (size_t)m->comm_world * kFacts;
            MHIP(hipMemcpyAsync(snd, &mine, sizeof mine, hipMemcpyHostToDevice, s.c->stream));
            MNCCL(m, m->nccl->AllGather(snd, m->agbuf, kFacts, ncclDouble, m->comm_sum, s.c->stream));
            MHIP(hipMemcpyAsync(facts.data(), m->agbuf, facts.size() * sizeof(RankFacts), hipMemcpyDeviceToHost, s.c->stream));
            MHIP(hipStreamSynchronize(s.c->stream));
            return 0;
        }, true);
        if (rc) return rc;
    } else {
        for (int g = 0; g < m->n; ++g) facts[(size_t)g] = facts_of(*m->slab[(size_t)g]);
    }
    for (size_t g = 0; g + 1 < facts.size(); ++g) {
        if (facts[(size_t)g].sav != facts[(size_t)g + 1].sav) {
            ec3d_set_error("ec3d_multi: slabs chose different storage formats; call ec3d_multi_set_format(h, -1, 0) to use "
                           "bands + tail everywhere");
            return 105;
        }
        if (facts[(size_t)g].snd_hi != facts[(size_t)g + 1].rcv_lo || facts[(size_t)g].rcv_hi != facts[(size_t)g + 1].snd_lo) {
            ec3d_set_error("ec3d_multi: neighbouring slabs disagree on the halo size");
            return 105;
        }
        if (m->nccl && (facts[(size_t)g].h_snd_hi != facts[(size_t)g + 1].h_rcv_lo ||
                        facts[(size_t)g].h_rcv_hi != facts[(size_t)g + 1].h_snd_lo)) {
            ec3d_set_error("ec3d_multi: ranks " + std::to_string(g) + " and " + std::to_string(g + 1) +
                           " cut their halo into different pieces (same total): the send / receive pairs would not meet");
            return 105;
        }
    }
    {   // about 0.4 ms of device work between two looks at the stop flag (as ec3d_solve.hip does)
        double big = 0;
        for (const RankFacts &f : facts) big = std::max(big, f.n_pad);
        const double est_us = big * 264.0 / 4.0e6 + 12.0 + 20.0 * (m->world > 1);
        m->chunk = (int)std::min<double>(32.0, std::max<double>(1.0, 400.0 / est_us));
    }
    // What the whole job can do (the plan is a property of the job: the exchanges of the three-launch iteration differ
    // from those of the five-launch one, and the rings of the deferred X update decide where an exchanged P or S lives).
    //   fused: the single-component operator, EVERY slab on 2-D tiles with both fusions, the SpMV-form K4 and the spare
    //          buffers in place (choose_sweep's size rule per slab: from 32 Mi rows; EC3D_FUSE23 / EC3D_FUSE51 / EC3D_K4S
    //          force it on small grids).  EC3D_SLAB_FUSE=0 keeps five launches.
    //   xd:    the smallest depth any slab allocated rings for (ec3d_spare_pair: 4 from 4.5 Mi streamed rows).
    //          EC3D_SLAB_XDEFER=1 switches it off, 2 .. 4 caps it.
    //   xasync: the groups of X updates on a stream of their own (ec3d_xasync) when every slab holds rings of two groups
    //          (ec3d_spare_pair: a z-slab with the X update deferred does; EC3D_XASYNC=0 never) -- the ring depth decides
    //          where an exchanged P or S lives, so it is the job's, like xd.
    //   both:  five launches with the exchange behind TWO of them (plan 5: K2 / K5 boundary planes first and K1 / K3 interior
    //          planes first) when every slab can split both ways and the largest holds fewer than 10 Mi rows.  Measured
    //          through the rank rehearsal against plan 1 (profiles/r05_plan5_both_splits.log): 2 Mi rows per rank 0.158 ->
    //          0.149 ms per iteration, 6.75 Mi 0.268 -> 0.261, 16 Mi 0.480 -> 0.493 (there the four extra launches, each
    //          beside a send / recv kernel, cost more than the waiting they remove).  The START of the P exchange differs
    //          from plans 0 / 1 (behind K5's boundary launch, an iteration ahead), so the choice is the job's.
    //          A-V slabs of the structured form (five planes per neighbour travel, one interior launch of a vector kernel
    //          does not cover them): plan 5 against plan 2 -- config 5 on 8 / 4 / 2 ranks 0.306 -> 0.287 / 0.383 -> 0.343 /
    //          0.504 -> 0.484 ms per iteration, config 3 on 2 / 4 ranks 0.277 -> 0.259 / 0.255 -> 0.249
    //          (profiles/r05_plan5_av_slabs.log): at every size.
    bool fused = m->kind == 1 && m->world > 1;
    bool xasync = m->world > 1;
    bool both = m->world > 1;
    bool fsplit = true; // plans 3 and 4 issue the all-gather and the send / recv group in opposite host order: the job's choice
    double job_rows = 0;
    int xd = EC3D_XD_MAX;
    for (const RankFacts &f : facts) {
        fused = fused && f.fused_ok != 0.0;
        xd = std::min(xd, (int)f.xd);
        xasync = xasync && f.xasync != 0.0;
        both = both && f.both_splits != 0.0;
        fsplit = fsplit && f.fsplit != 0.0;
        job_rows = std::max(job_rows, f.n_pad);
    }
    both = both && (m->kind == 2 || job_rows < 10.0 * 1048576.0); // (A-V slabs: measured a gain at every size, below)
    if (const char *e = getenv("EC3D_SLAB_FUSE")) fused = fused && atoi(e) != 0;
    if (const char *e = getenv("EC3D_SLAB_XDEFER")) xd = std::min(xd, std::max(1, atoi(e)));
    // (a one-slab job keeps its slab's own depth: five launches, X every D-th iteration from 4.5 Mi rows, like any slab)
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        MHIP(hipMemcpy(s.ptr_table, tab.data(), tab.size() * sizeof(double *), hipMemcpyHostToDevice));
        ec3d_ctx *c = s.c;
        c->dist = true;
        c->nranks = m->nccl ? m->comm_world : m->world;
        c->lsum = s.lsum;
        c->gsum = m->nccl ? m->gsum : nullptr;          // one process per GPU: the all-gathered copy
        c->lsum_ptrs = m->nccl ? nullptr : s.ptr_table; // one process: every rank's sums read in place
        c->slab_fused = fused;
        c->slab_xd = xd;
        // (three-launch slabs keep the applying K4: measured, the second launch costs them more than it fills)
        const bool xa_forced = getenv("EC3D_XASYNC") && atoi(getenv("EC3D_XASYNC")) == 2;
        c->slab_xasync = xasync && xd > 1 && (!fused || xa_forced);
        c->sweep_s.halo_store = fused ? ((s.rank > 0 ? 1 : 0) | (s.rank + 1 < m->world ? 2 : 0)) : 0;
        c->sweep_fb.halo_store = c->sweep_fi.halo_store = c->sweep_s.halo_store;
        if (c->pp_base) { // every rank cycles P and S through the same number of buffers
            const int D = ec3d_xdefer(c), depth = ec3d_xasync(c) ? 2 * D : D;
            c->pdepth = std::max(2, depth);
            c->sdepth = std::max(1, depth);
        }
        s.plan = 0;
        s.split_ok = false;
        // EC3D_SLAB_PLAN (the SAME value on every rank: plan 2 orders its exchanges differently) picks the five-launch plan of
        // a single-component job: 0 exchange in front of K1 / K3, 1 K1 / K3 split around it, 2 K2 / K5 boundary tiles first
        int want_plan = getenv("EC3D_SLAB_PLAN") ? atoi(getenv("EC3D_SLAB_PLAN")) : -1;
        // Plan 5 is OPT-IN (EC3D_SLAB_PLAN=5, the same on every rank) until a job of two real devices has verified it
        // bit for bit: everything measured for it (-3 ... -5 % below 10 Mi rows per rank, -10 % on A-V slabs) was measured
        // with the "neighbour" on the same card (`both` says where the one-card measurements would have picked it).
        (void)both;
        s.overlap_ok = false;
        if (m->kind == 1 && !((want_plan == 2 || want_plan == 5) && !fused && m->world > 1)) {
            const bool no_fsplit = getenv("EC3D_SLAB_FSPLIT") && atoi(getenv("EC3D_SLAB_FSPLIT")) == 0;
            s.plan = fused ? ((fsplit && !no_fsplit) ? 4 : 3) : (ec3d_can_overlap(c) && want_plan != 0) ? 1 : 0;
        } else if (m->kind == 2 && want_plan == 0) {
            s.plan = 0; // (measurement: the A-V job with the exchange in front of K1 / K3, no split launches)
        } else if (m->world > 1) {
            // the ORDER of exchanges is a property of the job: every A-V rank uses the producer-side
            // plan; a rank whose slab is all boundary runs the whole kernels in that order
            s.plan = want_plan == 5 ? 5 : 2;
            s.overlap_ok = s.plan == 5 && ec3d_can_overlap(c);
            std::vector<int64_t> lo, hi;
            for (const std::vector<Run> *rs : {&s.send_lo, &s.recv_lo, &s.send_hi, &s.recv_hi})
                for (const Run &r : *rs)
                    if (r.payload > 0) {
                        lo.push_back(r.lo());
                        hi.push_back(r.hi());
                    }
            int32_t en = 0;
            if (m->kind == 1) { // the single-component operator: whole planes, window sweeps
                int rc2 = ec3d_dist_set_boundary_planes(c, &en);
                if (rc2) return rc2;
            } else if (!lo.empty()) {
                int rc2 = ec3d_dist_set_boundary_rows(c, (int32_t)lo.size(), lo.data(), hi.data(), &en);
                if (rc2) return rc2;
            }
            s.split_ok = en != 0;
        }
        return 0;
    });
}

// owned ranges (local reference numbering) and halo runs (DEVICE rows) of an A-V slab whose matrix is in place.
// upl[p] = held conducting cells before held plane p (np + 1 entries).
void av_layout(ec3d_multi *m, Slab &s, const std::vector<int64_t> &upl)
{
    const int H = 2;
    ec3d_ctx *c = s.c;
    const int64_t kdz = m->kdz, nC = s.nC_ext, mloc = s.nU_ext;
    const int64_t p0 = s.k0 - s.e0, p1 = s.k1 - s.e0;
    for (int d = 0; d < 3; ++d) {
        s.own_lo[d] = d * nC + p0 * kdz;
        s.own_hi[d] = d * nC + p1 * kdz;
    }
    s.own_lo[3] = 3 * nC + upl[(size_t)p0];
    s.own_hi[3] = 3 * nC + upl[(size_t)p1];
    // The 7-point stencil and the U rows read A one plane away; only the one-sided A-U stencils
    // (src/EC3D.f90:697-706) reach two planes, and they read U
    struct Blk { int64_t base, pitch, payload; int h; };
    std::vector<Blk> blocks;
    const bool structured = c->A.sav != 0;
    if (structured) {
        for (int d = 0; d < 4; ++d) blocks.push_back(Blk{d * c->nCd, c->pitch, kdz, d < 3 ? 1 : H});
    } else {
        for (int d = 0; d < 3; ++d) blocks.push_back(Blk{d * nC, kdz, kdz, 1});
    }
    // Structured form: the U block is grid-shaped, but only cells of a conductor hold an unknown -- every other row of it
    // is inert and stays exactly zero in every vector.  Two planes of U that hold no conductor cell therefore need not
    // travel: the ghost rows they would fill are zero already (most cuts of a real model lie in air: BASELINE config 5 on
    // 8 ranks has the conductor at ONE of its seven cuts).  Both sides of a cut decide from the same planes' cell counts.
    // OPT-IN (EC3D_AV_SEND_EMPTY_U=0) until a job of two real devices has run once: by default every U plane travels.
    const bool u_always = !(getenv("EC3D_AV_SEND_EMPTY_U") && atoi(getenv("EC3D_AV_SEND_EMPTY_U")) == 0);
    auto u_cells = [&](int64_t a, int64_t b) { return upl[(size_t)std::min<int64_t>(b, (int64_t)upl.size() - 1)] - upl[(size_t)std::max<int64_t>(a, 0)]; };
    // (a rehearsal sends to itself: what it sends towards a cut must pair with what it receives from there, so a side's U
    // planes travel when EITHER of the two pairs of planes around the cut holds a conductor cell)
    auto travels = [&](const Blk &b, int64_t first_plane, int64_t cut) {
        if (!structured || b.h != H || u_always) return true;
        return m->rehearse ? u_cells(cut - H, cut + H) > 0 : u_cells(first_plane, first_plane + H) > 0;
    };
    if (s.e0 < s.k0) {
        for (const Blk &b : blocks) {
            if (travels(b, p0, p0)) s.send_lo.push_back(Run{b.base + p0 * b.pitch, b.pitch, b.payload, b.h});
            if (travels(b, p0 - b.h, p0)) s.recv_lo.push_back(Run{b.base + (p0 - b.h) * b.pitch, b.pitch, b.payload, b.h});
        }
        if (!structured) { // compact U block: the U cells of my first two owned planes / my lower halo planes
            const int64_t ulo = upl[(size_t)p0], cnt_s = upl[(size_t)(p0 + H)] - ulo;
            s.send_lo.push_back(Run{3 * nC + ulo, cnt_s, cnt_s, 1});
            s.recv_lo.push_back(Run{3 * nC, ulo, ulo, 1});
        }
    }
    if (s.k1 < s.e1) {
        for (const Blk &b : blocks) {
            if (travels(b, p1 - b.h, p1)) s.send_hi.push_back(Run{b.base + (p1 - b.h) * b.pitch, b.pitch, b.payload, b.h});
            if (travels(b, p1, p1)) s.recv_hi.push_back(Run{b.base + p1 * b.pitch, b.pitch, b.payload, b.h});
        }
        if (!structured) {
            const int64_t uend = upl[(size_t)p1], cnt_s = uend - upl[(size_t)(p1 - H)], cnt_r = mloc - uend;
            s.send_hi.push_back(Run{3 * nC + uend - cnt_s, cnt_s, cnt_s, 1});
            s.recv_hi.push_back(Run{3 * nC + uend, cnt_r, cnt_r, 1});
        }
    }
}

__global__ void k_fill(double *p, int64_t n, double v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// NaN into every row of work vector `vi` that the halo exchange is about to fill (the recv runs), on the slab's
// compute stream: a pull that comes late, copies the wrong rows or never happens then shows in the result instead
// of leaving behind whatever was uploaded there.  Only rows the exchange WRITES are touched: halo rows no owned row
// reads (the outer halo plane of the A blocks) keep their finite values, since a band coefficient of zero still
// multiplies its operand.
int poison_recv_rows(Slab &s, int vi)
{
    double *v = s.c->vec[vi];
    const double nan = std::nan("");
    for (const std::vector<Run> *rs : {&s.recv_lo, &s.recv_hi})
        for (const Run &r : *rs)
            for (int p = 0; p < r.planes; ++p) {
                if (r.payload <= 0) continue;
                k_fill<<<(unsigned)((r.payload + 255) / 256), 256, 0, s.c->stream>>>(v + r.start + (int64_t)p * r.pitch,
                                                                                        r.payload, nan);
                MHIP(hipGetLastError());
            }
    return 0;
}

int need(ec3d_multi *m, const char *who)
{
    if (!m || m->kind == 0) {
        ec3d_set_error(std::string(who) + ": no matrix (call ec3d_multi_assemble* first)");
        return 3;
    }
    return 0;
}

// host vector in the reference's global numbering -> this slab's work vector (owned AND halo entries)
int slab_upload(ec3d_multi *m, Slab &s, int which, const double *glob)
{
    ec3d_ctx *c = s.c;
    int rc = 0;
    if (m->kind == 1) {
        rc = ec3d_vec_h2d(c, c->vec[which], glob + (int64_t)s.k0 * m->kdz);
    } else {
        s.io.resize((size_t)s.n_local);
        for (int d = 0; d < 3; ++d)
            memcpy(s.io.data() + (size_t)d * s.nC_ext, glob + d * m->nC_glob + (int64_t)s.e0 * m->kdz,
                   (size_t)s.nC_ext * sizeof(double));
        double *u = s.io.data() + (size_t)3 * s.nC_ext;
        const double *gu = glob + 3 * m->nC_glob;
        for (int64_t q = 0; q < s.nU_ext; ++q) u[q] = gu[s.u_glob[(size_t)q]];
        rc = ec3d_vec_h2d(c, c->vec[which], s.io.data());
    }
    if (rc) return rc;
    MHIP(hipStreamSynchronize(c->stream));
    return 0;
}

// the OWNED entries of this slab's work vector -> the global host vector
int slab_download(ec3d_multi *m, Slab &s, int which, double *glob)
{
    ec3d_ctx *c = s.c;
    int rc = 0;
    if (m->kind == 1) {
        if ((rc = ec3d_vec_d2h(c, glob + (int64_t)s.k0 * m->kdz, c->vec[which]))) return rc;
        MHIP(hipStreamSynchronize(c->stream));
        return 0;
    }
    s.io.resize((size_t)s.n_local);
    if ((rc = ec3d_vec_d2h(c, s.io.data(), c->vec[which]))) return rc;
    MHIP(hipStreamSynchronize(c->stream));
    for (int d = 0; d < 3; ++d)
        memcpy(glob + d * m->nC_glob + (int64_t)s.k0 * m->kdz, s.io.data() + (size_t)s.own_lo[d],
               (size_t)(s.own_hi[d] - s.own_lo[d]) * sizeof(double));
    double *gu = glob + 3 * m->nC_glob;
    for (int64_t q = s.own_lo[3]; q < s.own_hi[3]; ++q) gu[s.u_glob[(size_t)(q - 3 * s.nC_ext)]] = s.io[(size_t)q];
    return 0;
}

// one reference solve on the resident slabs (src/solvers.f90:3-50); every rank returns the same iter
int slab_solve(ec3d_multi *m, Slab &s, double tol, int32_t itmax, int32_t *iter_out, int *hit_itmax)
{
    ec3d_ctx *c = s.c;
    const int64_t total = std::max<int64_t>(0, (int64_t)itmax + 1); // src/solvers.f90:25-29
    int rc = run_plan(m, s, begin_plan(s), 0, tol, nullptr);
    if (rc) return rc;
    c->xd_last = (int)std::min<int64_t>(total, INT_MAX); // the itmax exit: the last iteration applies the pending X updates
    // every rank must look at the flag after the same iterations, so the chunk is a property of the job
    // (finish_setup: from the largest slab), not of this slab
    const int chunk = m->chunk;
    int64_t launched = 0;
    int ci = 0;
    bool stopped = false;
    // every rank derives the same decisions from the same sums, so all see the flag at the same chunk and
    // leave together; iterations enqueued past the exit return at once and touch nothing
    while (launched < total && !stopped) {
        const int64_t n = std::min<int64_t>(chunk, total - launched);
        for (int64_t i = 0; i < n; ++i)
            if ((rc = run_plan(m, s, iter_plan(s), (int)(++launched), 0.0, nullptr))) return rc;
        if ((rc = ec3d_read_state_async(c, &s.stop_pinned[ci & 1]))) return rc;
        MHIP(hipEventRecord(s.ev_stop[ci & 1], c->stream));
        if (ci > 0) {
            s.at.store("solve:wait for the previous chunk's stop flag");
            MHIP(hipEventSynchronize(s.ev_stop[(ci - 1) & 1]));
            if (s.stop_pinned[(ci - 1) & 1] != INT_MAX) stopped = true;
        }
        ++ci;
    }
    if ((rc = drain(s))) return rc;
    int32_t si = 0;
    if ((rc = ec3d_read_state(c, &si, nullptr, nullptr))) return rc;
    if (si >= 0) { // an exit with X updates pending (deferred X update, K4 in SpMV form): applied now, on the owned rows
        if ((rc = ec3d_flush_x(c, si))) return rc;
        MHIP(hipStreamSynchronize(c->stream));
    }
    *iter_out = si >= 0 ? si : (int32_t)total;
    *hit_itmax = si < 0;
    return 0;
}
} // namespace

// ---------------------------------------------------------------------------------------------
extern "C" int ec3d_multi_create(ec3d_multi_handle *mh, int32_t nranks, const int32_t *devices)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ec3d_set_error("ec3d_multi_create: no HIP device available (this library has no CPU path)");
        return 101;
    }
    if (nranks < 1 || nranks > 64) {
        ec3d_set_error("ec3d_multi_create: nranks must be 1..64");
        return 2;
    }
    if (!devices && nranks > ndev) {
        ec3d_set_error("ec3d_multi_create: needs " + std::to_string(nranks) + " devices, this machine has " +
                       std::to_string(ndev));
        return 103;
    }
    for (int r = 0; devices && r < nranks; ++r)
        if (devices[r] < 0 || devices[r] >= ndev) {
            ec3d_set_error("ec3d_multi_create: device ordinal " + std::to_string(devices[r]) + " out of range (" +
                           std::to_string(ndev) + " devices)");
            return 102;
        }
    ec3d_multi *m = new ec3d_multi();
    m->n = nranks;
    m->world = nranks;
    for (int r = 0; r < nranks; ++r) {
        m->slab.emplace_back(new Slab());
        m->slab.back()->rank = r;
        m->slab.back()->device = devices ? devices[r] : r;
    }
    m->pool.rc.assign((size_t)nranks, 0);
    for (int r = 0; r < nranks; ++r) m->pool.th.emplace_back(worker, m, r);
    int rc = run_all(m, [&](int r) { return enable_peers(m, *m->slab[(size_t)r]); }); // all of them, then allocate
    if (!rc) rc = run_all(m, [&](int r) { return slab_reset(m, *m->slab[(size_t)r]); });
    if (rc) {
        std::string keep = ec3d_last_error();
        ec3d_multi_destroy(m);
        ec3d_set_error(keep);
        return rc;
    }
    *mh = m;
    return 0;
}

// ---- one process per GPU ------------------------------------------------------------------------------------
extern "C" int ec3d_rccl_unique_id(void *id128)
{
    std::string why;
    const ec3d_rccl_api *api = ec3d_rccl_load(why);
    if (!api || !id128) {
        ec3d_set_error("ec3d_rccl_unique_id: " + (api ? std::string("no buffer") : why));
        return api ? 2 : 109;
    }
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id around as 128 bytes");
    ncclUniqueId id;
    ncclResult_t e = api->GetUniqueId(&id);
    if (e != ncclSuccess) {
        ec3d_set_error(std::string("ncclGetUniqueId: ") + api->GetErrorString(e));
        return 108;
    }
    memcpy(id128, &id, sizeof id);
    return 0;
}

extern "C" int ec3d_multi_create_rank(ec3d_multi_handle *mh, int32_t rank, int32_t nranks, int32_t device,
                                      const void *id_halo, const void *id_sum, int32_t as_rank, int32_t as_world)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ec3d_set_error("ec3d_multi_create_rank: no HIP device available (this library has no CPU path)");
        return 101;
    }
    if (nranks < 1 || rank < 0 || rank >= nranks || device < 0 || device >= ndev || !id_halo || !id_sum) {
        ec3d_set_error("ec3d_multi_create_rank: need 0 <= rank < nranks, a device of this machine and two unique ids");
        return 2;
    }
    const bool rehearse = as_world > 0;
    if (rehearse && (nranks != 1 || as_rank < 0 || as_rank >= as_world)) {
        ec3d_set_error("ec3d_multi_create_rank: a rehearsal (as_world > 0) runs in a one-rank job, 0 <= as_rank < as_world");
        return 2;
    }
    std::string why;
    const ec3d_rccl_api *api = ec3d_rccl_load(why);
    if (!api) {
        ec3d_set_error("ec3d_multi_create_rank: " + why);
        return 109;
    }
    ec3d_multi *m = new ec3d_multi();
    m->n = 1;
    m->nccl = api;
    m->comm_rank = rank;
    m->comm_world = nranks;
    m->rehearse = rehearse;
    m->world = rehearse ? as_world : nranks;
    m->slab.emplace_back(new Slab());
    m->slab.back()->rank = rehearse ? as_rank : rank;
    m->slab.back()->device = device;
    m->pool.rc.assign(1, 0);
    m->pool.th.emplace_back(worker, m, 0);
    ncclUniqueId ih, is;
    memcpy(&ih, id_halo, sizeof ih);
    memcpy(&is, id_sum, sizeof is);
    int rc = run_all(m, [&](int) -> int {
        Slab &s = *m->slab[0];
        int rc2 = slab_reset(m, s);
        if (rc2) return rc2;
        // two communicators: the halo planes travel on the side stream while the compute stream -- which carries the
        // all-gathers of the sums -- runs the interior launch; every rank issues the calls of each in the same order
        MNCCL(m, api->CommInitRank(&m->comm_halo, nranks, ih, rank));
        MNCCL(m, api->CommInitRank(&m->comm_sum, nranks, is, rank));
        MHIP(hipMalloc(&m->gsum, (size_t)nranks * P_NSLOT * sizeof(double)));
        MHIP(hipMemset(m->gsum, 0, (size_t)nranks * P_NSLOT * sizeof(double)));
        MHIP(hipMalloc(&m->agbuf, (size_t)(nranks + 1) * kFacts * sizeof(double)));
        return 0;
    });
    if (rc) {
        std::string keep = ec3d_last_error();
        ec3d_multi_destroy(m);
        ec3d_set_error(keep);
        return rc;
    }
    *mh = m;
    return 0;
}

extern "C" int ec3d_multi_destroy(ec3d_multi_handle m)
{
    if (!m) return 0;
    (void)run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        (void)hipSetDevice(s.device);
        (void)hipDeviceSynchronize();
        if (s.c) (void)ec3d_destroy(s.c);
        s.c = nullptr;
        if (s.side) (void)hipStreamDestroy(s.side);
        for (int v = 0; v < NHALO; ++v)
            for (int i = 0; i < RING; ++i) {
                if (s.ev_ready[v][i]) (void)hipEventDestroy(s.ev_ready[v][i]);
                if (s.ev_halo[v][i]) (void)hipEventDestroy(s.ev_halo[v][i]);
            }
        for (int i = 0; i < RING; ++i) {
            if (s.ev_sum[i]) (void)hipEventDestroy(s.ev_sum[i]);
            if (s.ev_hub[i]) (void)hipEventDestroy(s.ev_hub[i]);
        }
        for (int i = 0; i < 2; ++i)
            if (s.ev_stop[i]) (void)hipEventDestroy(s.ev_stop[i]);
        if (s.stop_pinned) (void)hipHostFree(s.stop_pinned);
        if (s.lsum) (void)hipFree(s.lsum);
        if (s.ptr_table) (void)hipFree(s.ptr_table);
        if (m->nccl) {
            if (m->comm_halo) (void)m->nccl->CommDestroy(m->comm_halo);
            if (m->comm_sum) (void)m->nccl->CommDestroy(m->comm_sum);
            if (m->gsum) (void)hipFree(m->gsum);
            if (m->agbuf) (void)hipFree(m->agbuf);
        }
        return 0;
    });
    {
        std::lock_guard<std::mutex> lk(m->pool.m);
        m->pool.quit = true;
        m->pool.cv.notify_all();
    }
    for (auto &t : m->pool.th) t.join();
    delete m;
    return 0;
}

extern "C" int ec3d_multi_ranks(ec3d_multi_handle m) { return m ? m->world : 0; }

extern "C" int ec3d_multi_slab(ec3d_multi_handle m, int32_t rank, ec3d_handle *h, int32_t *k0, int32_t *k1)
{
    if (!m || rank < 0 || rank >= m->n) return 2;
    Slab &s = *m->slab[(size_t)rank];
    if (h) *h = s.c;
    if (k0) *k0 = s.k0;
    if (k1) *k1 = s.k1;
    return 0;
}

extern "C" int ec3d_multi_set_format(ec3d_multi_handle m, int dictionary, int structured)
{
    if (!m) return 2;
    for (auto &s : m->slab) {
        if (dictionary >= 0) s->c->use_dict = dictionary != 0;
        if (structured >= 0) s->c->use_sav = structured != 0;
    }
    return 0;
}

// ---- the single-component operator of BASELINE configs 2 and 4 (ec3d_assemble_poisson) on z-slabs -------
extern "C" int ec3d_multi_assemble_poisson(ec3d_multi_handle m, int32_t sdx, int32_t sdy, int32_t sdz,
                                           const double *BND, const double *delta)
{
    if (!m) return 2;
    if (sdz < m->world) {
        ec3d_set_error("ec3d_multi_assemble_poisson: fewer z-planes than ranks");
        return 2;
    }
    m->kind = 0;
    m->sdx = sdx; m->sdy = sdy; m->sdz = sdz;
    m->kdz = (int64_t)sdx * sdy;
    m->nC_glob = m->kdz * sdz;
    m->nU_glob = 0;
    m->n_glob = m->nC_glob;
    int rc = run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = slab_reset(m, s);
        if (rc2) return rc2;
        slab_bounds(sdz, s.rank, m->world, s.k0, s.k1);
        s.e0 = s.k0;
        s.e1 = s.k1;
        rc2 = m->world == 1 ? ec3d_assemble_poisson(s.c, sdx, sdy, sdz, BND, delta)
                            : ec3d_assemble_poisson_slab(s.c, sdx, sdy, sdz, s.k0, s.k1, BND, delta);
        if (rc2) return rc2;
        const int64_t kdz = m->kdz, n = (int64_t)(s.k1 - s.k0) * kdz;
        s.n_local = n;
        if (s.c->ghost < kdz && m->world > 1) {
            ec3d_set_error("ec3d_multi: ghost zone smaller than a plane");
            return 105;
        }
        if (s.rank > 0) {
            s.send_lo.push_back(Run{0, kdz, kdz, 1});
            s.recv_lo.push_back(Run{-kdz, kdz, kdz, 1});
        }
        if (s.rank + 1 < m->world) {
            s.send_hi.push_back(Run{n - kdz, kdz, kdz, 1});
            s.recv_hi.push_back(Run{n, kdz, kdz, 1});
        }
        return 0;
    });
    if (rc) return rc;
    m->kind = 1;
    m->nnz = 0;
    for (auto &s : m->slab) m->nnz += s->c->A.nnz;
    return finish_setup(m);
}

// ---- the full A-V system (ec3d_assemble, src/EC3D.f90:465-1049) on z-slabs: global tables in, the library
//      cuts the extended slabs (two halo planes per interior side) and renumbers the U unknowns per slab
extern "C" int ec3d_multi_assemble(ec3d_multi_handle m, int32_t sdx, int32_t sdy, int32_t sdz, const int8_t *geoPHYS,
                                   const int32_t *geoPHYS_C, const double *valPHYS, int32_t nsub_glob,
                                   const double *BND, const double *delta, double dt)
{
    if (!m) return 2;
    const int H = 2;
    if (m->world > 1 && sdz < H * m->world) {
        ec3d_set_error("ec3d_multi_assemble: every rank needs at least two z-planes");
        return 2;
    }
    m->kind = 0;
    m->sdx = sdx; m->sdy = sdy; m->sdz = sdz;
    m->kdz = (int64_t)sdx * sdy;
    m->nC_glob = m->kdz * sdz;
    int64_t nU = 0;
    for (int64_t q = 0; q < m->nC_glob; ++q) nU += geoPHYS_C[q] != 0;
    m->nU_glob = nU;
    m->n_glob = 3 * m->nC_glob + nU;
    for (int64_t q = 0; q < m->nC_glob; ++q)
        if (geoPHYS_C[q] != 0 && (geoPHYS_C[q] <= 3 * m->nC_glob || geoPHYS_C[q] > m->n_glob)) {
            ec3d_set_error("ec3d_multi_assemble: geoPHYS_C holds a U column id outside 3*nCells+1 .. n");
            return 2;
        }
    int rc = run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = slab_reset(m, s);
        if (rc2) return rc2;
        slab_bounds(sdz, s.rank, m->world, s.k0, s.k1);
        s.e0 = std::max(0, s.k0 - H);
        s.e1 = std::min(sdz, s.k1 + H);
        const int64_t kdz = m->kdz, np = s.e1 - s.e0, nC = np * kdz;
        const int8_t *geo = geoPHYS + (int64_t)s.e0 * kdz;
        const int32_t *gc = geoPHYS_C + (int64_t)s.e0 * kdz;
        std::vector<int32_t> gc_ext((size_t)nC, 0);
        std::vector<int64_t> upl((size_t)np + 1, 0); // conducting cells before held plane p
        int64_t mloc = 0;
        for (int64_t p = 0; p < np; ++p) {
            upl[(size_t)p] = mloc;
            for (int64_t q = p * kdz; q < (p + 1) * kdz; ++q)
                if (gc[q] != 0) {
                    gc_ext[(size_t)q] = (int32_t)(3 * nC + 1 + mloc++);
                    s.u_glob.push_back((int32_t)(gc[q] - 3 * m->nC_glob - 1));
                }
        }
        upl[(size_t)np] = mloc;
        s.nC_ext = nC;
        s.nU_ext = mloc;
        s.n_local = 3 * nC + mloc;
        if (m->world == 1)
            rc2 = ec3d_assemble(s.c, sdx, sdy, sdz, geo, gc_ext.data(), valPHYS, nsub_glob, BND, delta, dt);
        else
            rc2 = ec3d_assemble_slab(s.c, sdx, sdy, sdz, s.e0, s.e1, s.k0, s.k1, geo, gc_ext.data(), valPHYS, nsub_glob,
                                     BND, delta, dt);
        if (rc2) return rc2;
        av_layout(m, s, upl);
        return 0;
    });
    if (rc) return rc;
    // (neighbours must agree on the storage -- a slab that fell back to bands + tail next to a structured one would
    // exchange differently shaped blocks: finish_setup compares)
    m->kind = 2;
    m->nnz = 0; // per-slab counts include inert halo rows: not summed
    return finish_setup(m);
}

// ---- a single-component 7-point operator that arrives as CSR (BASELINE configs 2 and 4 through the drop-in symbol):
//      seven bands at (-kdz, -sdx, -1, 0, 1, sdx, kdz) and nothing else, n a whole number of planes.  Every rank takes
//      the band rows of its planes (same coefficients, same order), the ghost zones carry the neighbours' planes -- the
//      layout of ec3d_multi_assemble_poisson.  Returns -1 when the matrix is not of that kind.
static int multi_set_cube_csr(ec3d_multi *m, int32_t n, const double *valA, const int32_t *irow, const int32_t *jcol)
{
    HostMatrix M;
    if (ec3d_csr_to_host_matrix(n, valA, irow, jcol, M) != 0) return -1;
    int64_t sdx = 0, kdz = 0;
    if (!ec3d_host_matrix_is_cube(M, sdx, kdz)) return -1;
    const int64_t sdz = (int64_t)n / kdz;
    if (sdz < m->world) {
        ec3d_set_error("ec3d_multi_set_matrix_csr: fewer z-planes than ranks");
        return 2;
    }
    m->sdx = (int32_t)sdx; m->sdy = (int32_t)(kdz / sdx); m->sdz = (int32_t)sdz;
    m->kdz = kdz;
    m->nC_glob = (int64_t)n;
    m->nU_glob = 0;
    m->n_glob = (int64_t)n;
    int rc = run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = slab_reset(m, s);
        if (rc2) return rc2;
        slab_bounds((int)sdz, s.rank, m->world, s.k0, s.k1);
        s.e0 = s.k0;
        s.e1 = s.k1;
        const int64_t rows = (int64_t)(s.k1 - s.k0) * kdz, r0 = (int64_t)s.k0 * kdz;
        HostMatrix S;
        S.n = rows;
        S.n_pad = (rows + EC3D_TILE - 1) / EC3D_TILE * EC3D_TILE;
        S.nb = 7;
        for (int b = 0; b < 7; ++b) S.off[b] = M.off[b];
        S.bands.assign((size_t)7 * S.n_pad, 0.0);
        for (int b = 0; b < 7; ++b)
            memcpy(&S.bands[(size_t)b * S.n_pad], &M.bands[(size_t)b * M.n_pad + r0], (size_t)rows * sizeof(double));
        S.tail_id.assign((size_t)S.n_pad, -1);
        S.tile_flag.assign((size_t)(S.n_pad / EC3D_TILE), 0);
        S.chunk_ptr.assign(1, 0);
        S.nnz = 0;
        for (int64_t q = r0; q < r0 + rows; ++q) S.nnz += irow[q + 1] - irow[q];
        if (s.c->use_dict) ec3d_build_dictionary_host(S);
        if ((rc2 = ec3d_upload_matrix(s.c, S, m->world > 1 ? kdz : 0))) return rc2;
        s.n_local = rows;
        if (s.c->ghost < kdz && m->world > 1) {
            ec3d_set_error("ec3d_multi: ghost zone smaller than a plane");
            return 105;
        }
        if (s.rank > 0) {
            s.send_lo.push_back(Run{0, kdz, kdz, 1});
            s.recv_lo.push_back(Run{-kdz, kdz, kdz, 1});
        }
        if (s.rank + 1 < m->world) {
            s.send_hi.push_back(Run{rows - kdz, kdz, kdz, 1});
            s.recv_hi.push_back(Run{rows, kdz, kdz, 1});
        }
        return 0;
    });
    if (rc) return rc;
    m->kind = 1;
    m->nnz = M.nnz;
    return finish_setup(m);
}

// ---- the reference's CSR triple (what sprsbcgstabwr_ receives): recognised as the A-V system on a grid
//      (ec3d_sav_csr.cpp), then cut into slabs of the structured form -----------------------------------------
extern "C" int ec3d_multi_set_matrix_csr(ec3d_multi_handle m, int32_t n, const double *valA, const int32_t *irow,
                                         const int32_t *jcol)
{
    if (!m) return 2;
    const int H = 2;
    m->kind = 0;
    SavHost G;
    if (ec3d_csr_to_sav_host(n, valA, irow, jcol, G) != 0) {
        const int rc1 = multi_set_cube_csr(m, n, valA, irow, jcol); // a single-component 7-point operator?
        if (rc1 >= 0) return rc1;
        ec3d_set_error("ec3d_multi_set_matrix_csr: the matrix is not recognised as the reference's A-V system on a "
                       "grid (ec3d_probe_csr) nor as a single-component 7-point operator on one, so there are no "
                       "z-planes to cut it along; use one GPU");
        return 7;
    }
    const int64_t sdz = G.nCd / G.pitch;
    {
        std::string why;
        const int rc0 = ec3d_sav_cuttable(G, m->world, why);
        if (rc0) { // e.g. a cube whose plane count is a multiple of 3, read as three blocks of planes / 3 (too few planes
                   // per rank for that reading, or coupled across the blocks' faces): as ec3d_probe_csr_multi, any
                   // refusal of the A-V reading is followed by the single-component one
            const int rc1 = multi_set_cube_csr(m, n, valA, irow, jcol);
            if (rc1 >= 0) return rc1;
        }
        if (rc0) {
            ec3d_set_error("ec3d_multi_set_matrix_csr: " + why);
            return rc0;
        }
    }
    m->sdx = (int32_t)G.sdx; m->sdy = (int32_t)(G.plane / G.sdx); m->sdz = (int32_t)sdz;
    m->kdz = G.plane;
    m->nC_glob = G.plane * sdz;
    m->nU_glob = (int64_t)G.cond_cell.size();
    m->n_glob = n;
    int rc = run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = slab_reset(m, s);
        if (rc2) return rc2;
        slab_bounds((int)sdz, s.rank, m->world, s.k0, s.k1);
        s.e0 = std::max(0, s.k0 - H);
        s.e1 = std::min((int32_t)sdz, s.k1 + H);
        const int64_t np = s.e1 - s.e0;
        if (m->world == 1) {
            if ((rc2 = ec3d_upload_sav(s.c, G))) return rc2;
        } else {
            SavHost L;
            ec3d_sav_slice(G, s.e0, s.e1, s.k0, s.k1, L);
            if ((rc2 = ec3d_upload_sav(s.c, L))) return rc2;
        }
        // held conducting cells: one contiguous run of the global scan-order list
        std::vector<int64_t> upl((size_t)np + 1, 0);
        int64_t first = -1, cnt = 0;
        for (size_t q = 0; q < G.cond_cell.size(); ++q) {
            const int64_t pl = G.cond_cell[q] / G.pitch;
            if (pl < s.e0 || pl >= s.e1) continue;
            if (first < 0) first = (int64_t)q;
            ++cnt;
            ++upl[(size_t)(pl - s.e0) + 1];
        }
        for (int64_t p = 0; p < np; ++p) upl[(size_t)p + 1] += upl[(size_t)p];
        s.u_glob.resize((size_t)cnt);
        for (int64_t q = 0; q < cnt; ++q) s.u_glob[(size_t)q] = (int32_t)(first + q);
        s.nC_ext = np * G.plane;
        s.nU_ext = cnt;
        s.n_local = 3 * s.nC_ext + cnt;
        av_layout(m, s, upl);
        return 0;
    });
    if (rc) return rc;
    m->kind = 2;
    m->nnz = G.nnz;
    return finish_setup(m);
}

extern "C" int ec3d_multi_upload(ec3d_multi_handle m, int which, const double *host)
{
    int rc = need(m, "ec3d_multi_upload");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    return run_all(m, [&](int r) { return slab_upload(m, *m->slab[(size_t)r], which, host); });
}

extern "C" int ec3d_multi_download(ec3d_multi_handle m, int which, double *host)
{
    int rc = need(m, "ec3d_multi_download");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    return run_all(m, [&](int r) { return slab_download(m, *m->slab[(size_t)r], which, host); });
}

extern "C" int ec3d_multi_size(ec3d_multi_handle m, int64_t *n)
{
    int rc = need(m, "ec3d_multi_size");
    if (rc) return rc;
    *n = m->n_glob;
    return 0;
}

extern "C" int ec3d_multi_solve_resident(ec3d_multi_handle m, double tolerance, int32_t itmax, int32_t *iter)
{
    int rc = need(m, "ec3d_multi_solve_resident");
    if (rc) return rc;
    std::vector<int32_t> its((size_t)m->n, -1);
    std::vector<int> hit((size_t)m->n, 0);
    rc = run_all(m, [&](int r) {
        return slab_solve(m, *m->slab[(size_t)r], tolerance, itmax, &its[(size_t)r], &hit[(size_t)r]);
    }, true);
    if (rc) return rc;
    for (int r = 1; r < m->n; ++r)
        if (its[(size_t)r] != its[0]) {
            ec3d_set_error("ec3d_multi_solve: ranks disagree on the iteration count (" + std::to_string(its[0]) + " vs " +
                           std::to_string(its[(size_t)r]) + ")");
            return 106;
        }
    *iter = its[0];
    if (hit[0]) {
        // itmax exit: the reference prints norm2(R) and returns (src/solvers.f90:25-28); ||R||^2 = the ranks'
        // last R.R sums added in rank order
        double s = 0.0;
        if (m->nccl) { // one process per GPU: every rank's last sums are in the gathered copy
            std::vector<double> all((size_t)m->comm_world * P_NSLOT);
            (void)hipSetDevice(m->slab[0]->device);
            EC3D_HIP(hipMemcpy(all.data(), m->gsum, all.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int g = 0; g < m->comm_world; ++g) s += all[(size_t)g * P_NSLOT + P_RR];
        }
        for (auto &sl : m->slab) {
            if (m->nccl) break;
            double v = 0.0;
            (void)hipSetDevice(sl->device);
            EC3D_HIP(hipMemcpy(&v, sl->lsum + P_RR, sizeof v, hipMemcpyDeviceToHost));
            s += v;
        }
        if (ec3d_itmax_print_hold) {
            *ec3d_itmax_print_hold = std::sqrt(s);
        } else if (!m->nccl || m->comm_rank == 0) { // one process per GPU: the line appears once, as the reference's does
            ec3d_print_rnorm(std::sqrt(s));
            fflush(stdout);
        }
    }
    return 0;
}

extern "C" int ec3d_multi_solve(ec3d_multi_handle m, const double *b, double *x, double tolerance, int32_t itmax,
                                int32_t *iter)
{
    int rc = need(m, "ec3d_multi_solve");
    if (rc) return rc;
    if ((rc = run_all(m, [&](int r) -> int {
             Slab &s = *m->slab[(size_t)r];
             int rc2 = slab_upload(m, s, EC3D_VEC_B, b);
             return rc2 ? rc2 : slab_upload(m, s, EC3D_VEC_X, x);
         })))
        return rc;
    if ((rc = ec3d_multi_solve_resident(m, tolerance, itmax, iter))) return rc;
    return run_all(m, [&](int r) { return slab_download(m, *m->slab[(size_t)r], EC3D_VEC_X, x); });
}

// ---- the time loop around the solve on slabs (src/EC3D.f90:275-404, :412-433; SlabSolver.rhs_step) ------
extern "C" int ec3d_multi_rhs_step(ec3d_multi_handle m, int32_t moving, int32_t nsrc, const int32_t *src_index,
                                   const double *src_value)
{
    int rc = need(m, "ec3d_multi_rhs_step");
    if (rc) return rc;
    if (m->kind != 2) {
        ec3d_set_error("ec3d_multi_rhs_step: needs a matrix from ec3d_multi_assemble");
        return 3;
    }
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        // the U-row right-hand sides read A one plane away: refresh the X halo first
        int rc2 = halo_start(m, s, CH_X);
        if (rc2) return rc2;
        if ((rc2 = halo_wait(s, CH_X))) return rc2;
        // sources outside the held planes are dropped, the rest renumbered locally (1-based ids)
        std::vector<int32_t> idx;
        std::vector<double> val;
        for (int32_t q = 0; q < nsrc; ++q) {
            const int64_t g0 = (int64_t)src_index[q] - 1, d = g0 / m->nC_glob, cell = g0 % m->nC_glob;
            const int64_t plane = cell / m->kdz;
            if (d > 2 || plane < s.e0 || plane >= s.e1) continue;
            idx.push_back((int32_t)(d * s.nC_ext + (cell - (int64_t)s.e0 * m->kdz) + 1));
            val.push_back(src_value[q]);
        }
        int32_t dummy_i = 0;
        double dummy_v = 0.0;
        if ((rc2 = ec3d_rhs_step(s.c, moving, (int32_t)idx.size(), idx.empty() ? &dummy_i : idx.data(),
                                 val.empty() ? &dummy_v : val.data())))
            return rc2;
        return drain(s);
    }, true);
}

extern "C" int ec3d_multi_post_update(ec3d_multi_handle m)
{
    int rc = need(m, "ec3d_multi_post_update");
    if (rc) return rc;
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = ec3d_post_update(s.c);
        return rc2 ? rc2 : drain(s);
    });
}

extern "C" int ec3d_multi_vtk_fields(ec3d_multi_handle m, const double *delta, float *field_A, float *field_eddy,
                                     float *field_source, float *field_B)
{
    int rc = need(m, "ec3d_multi_vtk_fields");
    if (rc) return rc;
    if (m->kind != 2) {
        ec3d_set_error("ec3d_multi_vtk_fields: needs a matrix from ec3d_multi_assemble");
        return 3;
    }
    if (field_eddy) memset(field_eddy, 0, (size_t)3 * m->nC_glob * sizeof(float)); // slabs without conductor skip it
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = halo_start(m, s, CH_X); // the curl reads the neighbours' planes
        if (rc2) return rc2;
        if ((rc2 = halo_wait(s, CH_X))) return rc2;
        const size_t off = (size_t)3 * s.k0 * m->kdz;
        if ((rc2 = ec3d_vtk_fields(s.c, delta, field_A + off, field_eddy ? field_eddy + off : nullptr, field_source + off,
                                   field_B + off)))
            return rc2;
        return drain(s);
    }, true);
}

// Field output overlapped with the next time step, over the slabs (ec3d_vtk_fields_begin / _wait on every slab's own
// handle: its field kernel behind its post-update, its copy into its own pinned buffers on its own side stream).  The
// slabs take their slots in step, so one slot number names the same output step on all of them; _wait hands out ONE
// slab's part -- cells k0*sdx*sdy .. k1*sdx*sdy of every vector, which are consecutive in field_N.vtk
// (src/utilites.f90:222-289 writes cell by cell, z outermost).
extern "C" int ec3d_multi_vtk_fields_begin(ec3d_multi_handle m, const double *delta, int32_t big_endian, int32_t *slot)
{
    int rc = need(m, "ec3d_multi_vtk_fields_begin");
    if (rc) return rc;
    if (m->kind != 2 || !slot) {
        ec3d_set_error("ec3d_multi_vtk_fields_begin: needs a matrix from ec3d_multi_assemble and a slot to return");
        return m->kind != 2 ? 3 : 2;
    }
    std::vector<int32_t> got((size_t)m->n, -1);
    rc = run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = halo_start(m, s, CH_X); // the curl reads the neighbours' planes
        if (rc2) return rc2;
        if ((rc2 = halo_wait(s, CH_X))) return rc2;
        return ec3d_vtk_fields_begin(s.c, delta, big_endian, &got[(size_t)r]); // enqueued only: nothing waits here
    }, true);
    if (rc) return rc;
    for (int r = 1; r < m->n; ++r)
        if (got[(size_t)r] != got[0]) {
            ec3d_set_error("ec3d_multi_vtk_fields_begin: the slabs' output slots are out of step");
            return 4;
        }
    *slot = got[0];
    return 0;
}

extern "C" int ec3d_multi_vtk_fields_wait(ec3d_multi_handle m, int32_t slot, int32_t rank, const float **field_A,
                                          const float **field_eddy, const float **field_source, const float **field_B,
                                          int64_t *cell0, int64_t *ncells)
{
    int rc = need(m, "ec3d_multi_vtk_fields_wait");
    if (rc) return rc;
    if (rank < 0 || rank >= m->n) {
        ec3d_set_error("ec3d_multi_vtk_fields_wait: no such slab");
        return 2;
    }
    Slab &s = *m->slab[(size_t)rank];
    if (cell0) *cell0 = (int64_t)s.k0 * m->kdz;
    return ec3d_vtk_fields_wait(s.c, slot, field_A, field_eddy, field_source, field_B, ncells);
}

// ---- bench "steps": exits disabled, launches only ---------------------------------------------------------
extern "C" int ec3d_multi_iterate_begin(ec3d_multi_handle m)
{
    int rc = need(m, "ec3d_multi_iterate_begin");
    if (rc) return rc;
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = run_plan(m, s, begin_plan(s), 0, -1.0, nullptr);
        return rc2 ? rc2 : drain(s);
    }, true);
}

namespace {
// kernel_ms[5]: stage averages of local slab `timed_rank`; sync_ms[2] / sync_n[2]: per iteration, the time the compute stream of
// that slab stood at the reduction points / at the waits for halo planes, and how many of each an iteration has
int multi_iterate(ec3d_multi *m, int32_t first_iter, int32_t count, int timed_rank, double *kernel_ms, double *sync_ms,
                  int32_t *sync_n)
{
    int rc = need(m, "ec3d_multi_iterate");
    if (rc) return rc;
    if (first_iter != m->slab[0]->c->it_next) { // (as ec3d_iterate: the device state is addressed by the iteration number)
        ec3d_set_error("ec3d_multi_iterate: first_iter = " + std::to_string(first_iter) + " does not continue the iterations "
                       "of this handle (next: " + std::to_string(m->slab[0]->c->it_next) + "; ec3d_multi_iterate_begin "
                       "starts again from 1)");
        return 6;
    }
    for (auto &sp : m->slab) { // as ec3d_iterate: the groups of the deferred X update are counted from this call's first
        ec3d_ctx *c = sp->c;   // iteration, its last one applies what is pending
        c->xd_base = first_iter;
        c->xd_last = first_iter + count - 1;
    }
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        StageTimer tm;
        tm.sync_points = sync_ms != nullptr;
        StageTimer *tp = (kernel_ms && r == timed_rank) ? &tm : nullptr;
        int rc2 = 0;
        const uint64_t calls0 = t_api_calls;
        for (int it = first_iter; it < first_iter + count; ++it)
            if ((rc2 = run_plan(m, s, iter_plan(s), it, 0.0, tp))) return rc2;
        s.api_calls = t_api_calls - calls0;
        s.api_iters = (uint64_t)std::max(0, count);
        if (!kernel_ms) return 0; // asynchronous: ec3d_multi_synchronize() joins
        if ((rc2 = drain(s))) return rc2;
        if (tp) {
            for (int k = 0; k < 5; ++k) kernel_ms[k] = 0.0;
            if (sync_ms) sync_ms[0] = sync_ms[1] = 0.0;
            if (sync_n) sync_n[0] = sync_n[1] = 0;
            for (size_t i = 0; i < tm.kern.size(); ++i) {
                float ms = 0.f;
                MHIP(hipEventElapsedTime(&ms, tm.ev[2 * i], tm.ev[2 * i + 1]));
                const int k = tm.kern[i];
                if (k < 5) {
                    kernel_ms[k] += (double)ms / std::max(1, count);
                } else if (sync_ms) {
                    sync_ms[k - T_GATHER] += (double)ms / std::max(1, count);
                    if (sync_n) ++sync_n[k - T_GATHER];
                }
            }
            if (sync_n)
                for (int q = 0; q < 2; ++q) sync_n[q] /= std::max(1, count);
            for (hipEvent_t e : tm.ev) (void)hipEventDestroy(e);
        }
        return 0;
    }, true);
}
} // namespace

extern "C" int ec3d_multi_iterate(ec3d_multi_handle m, int32_t first_iter, int32_t count, double *kernel_ms)
{
    return multi_iterate(m, first_iter, count, 0, kernel_ms, nullptr, nullptr);
}

extern "C" int ec3d_multi_iterate_timed(ec3d_multi_handle m, int32_t first_iter, int32_t count, int32_t rank, double *kernel_ms,
                                        double *sync_ms, int32_t *sync_n)
{
    if (!m || rank < 0 || rank >= m->n || !kernel_ms || !sync_ms || !sync_n) return 2;
    return multi_iterate(m, first_iter, count, rank, kernel_ms, sync_ms, sync_n);
}

// which RCCL this one-process-per-GPU handle talks through, and what IT says the job is: ranks of the communicator
// (ncclCommCount on the communicator the sums travel on), library version (ncclGetVersion), file the entry points came from
extern "C" int ec3d_multi_rccl_info(ec3d_multi_handle m, int32_t *nranks, int32_t *version, char *path, int32_t path_cap)
{
    if (!m || !m->nccl) {
        ec3d_set_error("ec3d_multi_rccl_info: not a one-process-per-GPU handle (ec3d_multi_create_rank)");
        return 3;
    }
    int n = -1, v = 0;
    if (m->nccl->CommCount && m->comm_sum) MNCCL(m, m->nccl->CommCount(m->comm_sum, &n));
    if (m->nccl->GetVersion) MNCCL(m, m->nccl->GetVersion(&v));
    if (nranks) *nranks = n;
    if (version) *version = v;
    if (path && path_cap > 0) {
        strncpy(path, m->nccl->path, (size_t)path_cap - 1);
        path[path_cap - 1] = 0;
    }
    return 0;
}

// HIP runtime calls (launches, event records / waits, copies) rank `rank`'s thread issued per iteration in the last
// ec3d_multi_iterate: the host-side price of an iteration, which has to stay below its device time
extern "C" int ec3d_multi_api_calls(ec3d_multi_handle m, int32_t rank, double *per_iteration)
{
    if (!m || rank < 0 || rank >= m->n || !per_iteration) return 2;
    const Slab &s = *m->slab[(size_t)rank];
    *per_iteration = s.api_iters ? (double)s.api_calls / (double)s.api_iters : 0.0;
    return 0;
}

extern "C" int ec3d_multi_plan(ec3d_multi_handle m, int32_t *plan, int32_t *x_every)
{
    int rc = need(m, "ec3d_multi_plan");
    if (rc) return rc;
    const Slab &s = *m->slab[0];
    if (plan) *plan = s.plan;
    if (x_every) *x_every = ec3d_xdefer(s.c);
    return 0;
}

extern "C" int ec3d_multi_halo_rows(ec3d_multi_handle m, int32_t rank, int64_t *sent, int64_t *received)
{
    int rc = need(m, "ec3d_multi_halo_rows");
    if (rc) return rc;
    if (rank < 0 || rank >= m->n) return 2;
    const Slab &s = *m->slab[(size_t)rank];
    auto total = [](const std::vector<Run> &rs) { int64_t t = 0; for (const Run &r : rs) t += r.payload * (int64_t)r.planes; return t; };
    if (sent) *sent = total(s.send_lo) + total(s.send_hi);
    if (received) *received = total(s.recv_lo) + total(s.recv_hi);
    return 0;
}

extern "C" int ec3d_multi_synchronize(ec3d_multi_handle m)
{
    if (!m) return 2;
    // watched: this is where the asynchronous ec3d_multi_iterate (bench.py's timed region) is joined -- every rank sits
    // in hipStreamSynchronize behind cross-slab event waits, the very stall the watchdog exists for
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        if (!s.c) return 0;
        MHIP(hipSetDevice(s.device));
        return drain(s);
    }, true);
}

// ||B - A X|| / ||B|| over all slabs (as ec3d_true_residual): X halo refreshed, the residual stage on every
// slab, the ranks' two sums added on the host in rank order.
extern "C" int ec3d_multi_true_residual(ec3d_multi_handle m, double *rel, double *bnorm)
{
    int rc = need(m, "ec3d_multi_true_residual");
    if (rc) return rc;
    std::vector<double> bb((size_t)std::max(m->world, m->comm_world), 0.0), rr((size_t)std::max(m->world, m->comm_world), 0.0);
    rc = run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = halo_start(m, s, CH_X);
        if (rc2) return rc2;
        if ((rc2 = halo_wait(s, CH_X))) return rc2;
        if ((rc2 = ec3d_dist_step(s.c, EC3D_STAGE_RESID, 0, 0.0))) return rc2;
        if (m->nccl && (rc2 = gather(m, s))) return rc2;
        if ((rc2 = drain(s))) return rc2;
        if (m->nccl) { // every rank's two sums, rank order
            std::vector<double> all((size_t)m->comm_world * P_NSLOT);
            MHIP(hipMemcpy(all.data(), m->gsum, all.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int g = 0; g < m->comm_world; ++g) {
                bb[(size_t)g] = all[(size_t)g * P_NSLOT + P_BB];
                rr[(size_t)g] = all[(size_t)g * P_NSLOT + P_RR_INIT];
            }
            return 0;
        }
        double v[P_NSLOT];
        MHIP(hipMemcpy(v, s.lsum, sizeof v, hipMemcpyDeviceToHost));
        bb[(size_t)r] = v[P_BB];
        rr[(size_t)r] = v[P_RR_INIT];
        return 0;
    }, true);
    if (rc) return rc;
    double sb = 0.0, sr = 0.0;
    for (size_t r = 0; r < bb.size(); ++r) {
        sb += bb[r];
        sr += rr[r];
    }
    if (bnorm) *bnorm = std::sqrt(sb);
    *rel = sb > 0.0 ? std::sqrt(sr / sb) : std::sqrt(sr);
    return 0;
}

// y = A*x over the slabs (src/solvers.f90:54-61), host vectors in the global numbering: x goes to every slab's P,
// the rows the halo exchange fills are set to NaN, the P halo planes are exchanged the way an iteration does it,
// every slab multiplies, the owned parts of AP come back.  Parity probe of the slab operators AND of the exchange:
// a result equal to the undivided operator's means every halo row arrived, in time, from the right place.
extern "C" int ec3d_multi_spmv(ec3d_multi_handle m, const double *x, double *y)
{
    int rc = need(m, "ec3d_multi_spmv");
    if (rc) return rc;
    return run_all(m, [&](int r) -> int {
        Slab &s = *m->slab[(size_t)r];
        int rc2 = slab_upload(m, s, EC3D_VEC_P, x);
        if (rc2) return rc2;
        // the probe must depend on the transport: what the exchange is to deliver is NaN until it does
        if (m->world > 1 && (rc2 = poison_recv_rows(s, EC3D_VEC_P))) return rc2;
        if ((rc2 = halo_start(m, s, CH_P))) return rc2;
        if ((rc2 = halo_wait(s, CH_P))) return rc2;
        ec3d_launch_spmv(s.c->A.view(), s.c->sweep_s, s.c->vec[EC3D_VEC_P], s.c->vec[EC3D_VEC_AP], s.c->stream);
        MHIP(hipGetLastError());
        if ((rc2 = drain(s))) return rc2;
        return slab_download(m, s, EC3D_VEC_AP, y);
    }, true);
}
