// ec3d_solve.hip — the solve loop, host side of src/solvers.f90:3-50.
//
// The loop body is five asynchronous launches per iteration (ec3d_kernels.hip); all scalars and the
// convergence decision stay on the device.  The host runs ahead by up to two chunks of iterations and
// learns about an exit from an asynchronous copy of the SolverState; launches issued past the exit are
// no-ops, so the result is exactly the reference's.
#include "../../include/ec3d_hip.h"
#include "ec3d_internal.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

// ---------------------------------------------------------------------------------------------
// Where a consumer finds the sums it needs.  Single GPU: the producer's per-workgroup partials; the
// producers of slots BB, RR_INIT, D1, D2, D3 are SpMV-type kernels (sweep_s), those of SS, RR, RR0N
// vector kernels (sweep) -- every consumer reads slots of one producer class only.
static int parts_of(const ec3d_ctx *c, int producer, bool split)
{
    // (a split launch of the three-launch iteration -- K4 in SpMV form or K5-in-K1 as boundary + interior launch -- leaves
    // the boundary launch's partials followed by the interior launch's)
    const int fsplit = c->sweep_fb.nblk + c->sweep_fi.nblk;
    if (producer == EC3D_BY_K2) return ec3d_fused23(c) ? c->sweep_s.nblk : c->sweep_k2.nblk; // fused: S.S comes from K23
    if (producer == EC3D_BY_K4) return ec3d_k4s(c) ? (split ? fsplit : c->sweep_s.nblk) : c->sweep.nblk; // K4 in SpMV form sums on that grid
    if (split && ec3d_fused51(c) && c->slab_fused) return fsplit;
    return split ? c->sweep_int.nblk + c->sweep_bnd.nblk : c->sweep_s.nblk;
}
RedSrc ec3d_src_of(const ec3d_ctx *c, int producer)
{
    if (c->dist && c->lsum_ptrs) return RedSrc{nullptr, c->nranks, 0, 0, c->lsum_ptrs};
    if (c->dist) return RedSrc{c->gsum, c->nranks, P_NSLOT, 1, nullptr};
    return RedSrc{c->partials, parts_of(c, producer, false), 1, c->sweep.pstride, nullptr};
}
RedSrc ec3d_part_of(const ec3d_ctx *c, int producer, bool split)
{
    return RedSrc{c->partials, parts_of(c, producer, split), 1, c->sweep.pstride, nullptr};
}

// Where vector `vec` of iteration `it` lives: P in the ring pbuf (K5-in-K1 alternates two buffers, the deferred X update
// keeps D), AP in apbuf (K5-in-K1), S in sbuf (deferred X update); otherwise the plain work vector.  The multi-rank
// drivers address the halo exchange of an iteration's vector through the same function (ec3d_multi.hip).
double *ec3d_vec_at(const ec3d_ctx *c, int vec, int it)
{
    const bool f51 = ec3d_fused51(c);
    const int D = ec3d_xdefer(c), pd = c->pdepth;
    switch (vec) {
    case EC3D_VEC_P: return (f51 || D > 1) ? c->pbuf[((it + c->p_off) % pd + pd) % pd] : c->vec[EC3D_VEC_P];
    case EC3D_VEC_AP: return f51 ? c->apbuf[it & 1] : c->vec[EC3D_VEC_AP];
    case EC3D_VEC_S: return D > 1 ? c->sbuf[((it % c->sdepth) + c->sdepth) % c->sdepth] : c->vec[EC3D_VEC_S];
    default: return c->vec[vec];
    }
}

// the five launches of one iteration; `k` selects one of them (1..5) or all (0).
// With the fusions of the 2-D-tile kernels (single rank, vectors beyond the caches) an iteration is THREE launches:
//   stage 1: K1 -- only in iteration 1 (afterwards AP = A P was produced by the previous iteration's stage 5)
//   stage 2: empty, stage 3: K23 (S = R - alpha*AP inside AS = A S)
//   stage 4: K4 -- as an SpMV kernel that computes A S again (ec3d_k4s: K23 then does not store AS), or the vector kernel
//   stage 5: K51 (the exits and the P update of K5, then the NEXT iteration's K1 on the new P)
// P(it) and AP(it) then live in pbuf[it % pdepth] / apbuf[it & 1] (ec3d_ctx).
// With the X update deferred (ec3d_xdefer = D > 1; three launches or five) P(it) and S(it) live in rings of D buffers, K4
// leaves X alone except in the last iteration of a group of D (counted from xd_base) or of the call (xd_last), where it
// applies what is pending; an exit in between is completed by ec3d_flush_x.
void ec3d_launch_stage(ec3d_ctx *c, const MatView &A, int it, int k, int part)
{
    double **v = c->vec;
    const Sweep &sw = c->sweep, &ss = part == 1 ? c->sweep_fb : part == 2 ? c->sweep_fi : c->sweep_s;
    hipStream_t s = c->stream;
    const bool fused = ec3d_fused23(c); // K2 inside K3 (2-D tiles, single rank): stage 2 is empty, stage 3 is K23
    const bool f51 = ec3d_fused51(c);
    const int D = ec3d_xdefer(c), pd = c->pdepth;
    const bool ring = D > 1; // P(it) in the ring also on the five-launch iteration (K5 then writes the next buffer)
    double *P = ec3d_vec_at(c, EC3D_VEC_P, it), *AP = ec3d_vec_at(c, EC3D_VEC_AP, it);
    double *S = ec3d_vec_at(c, EC3D_VEC_S, it);
    const auto pidx = [&](int i) { return ((i + c->p_off) % pd + pd) % pd; };
    // fused: AP(it) was produced by the previous iteration's K51 -- unless this call does not continue that
    // iteration (iteration 1, ec3d_iterate from another first_iter, ec3d_time_kernel): then K1 runs on its own
    if ((k == 0 || k == 1) && (!f51 || it == 1 || c->ap_valid_for != it))
        ec3d_launch_k1(A, ss, c->state, it, P, v[EC3D_VEC_R0], AP, c->partials, s);
    if ((k == 0 || k == 2) && !fused) {
        ec3d_launch_k2(c->sweep_k2, ec3d_src_of(c, EC3D_BY_SPMV), c->state, it, v[EC3D_VEC_R], AP, S, c->partials, s);
        c->scur = D > 1 ? it % c->sdepth : 1;
    }
    if ((k == 0 || k == 3) && fused) {
        ec3d_launch_k23(A, ss, ec3d_src_of(c, EC3D_BY_SPMV), c->state, it, v[EC3D_VEC_R], AP, S,
                        ec3d_k4s(c) ? nullptr : v[EC3D_VEC_AS], c->partials, s); // (K4 in SpMV form computes A S again)
        c->scur = D > 1 ? it % c->sdepth : 1;
    }
    if ((k == 0 || k == 3) && !fused)
        ec3d_launch_k3(A, ss, c->state, it, S, v[EC3D_VEC_AS], c->partials, s);
    if (k == 0 || k == 4) {
        // position of this iteration in its group of D, and how many updates an applying launch finds pending
        const int xm = D > 1 ? (it - c->xd_base) % D : 0;
        // (ec3d_xasync: no K4 touches X; alpha / omega of iteration it wait in entry it % 2D, and the group's own launch --
        // below, behind the K4 of its last iteration -- applies them)
        const bool xa = ec3d_xasync(c);
        const bool group_end = D > 1 && (xm == D - 1 || it >= c->xd_last);
        const bool apply = !xa && (D <= 1 || group_end);
        const int xe = xa ? it % (2 * D) : xm; // the entry this K4 leaves its alpha / omega in
        if (ec3d_k4s(c)) {
            const double *pp[EC3D_XD_MAX] = {nullptr}, *sp[EC3D_XD_MAX] = {nullptr};
            const int ne = apply ? xm + 1 : 0;
            for (int j = 0; j < ne; ++j) { // iterations it - xm .. it, oldest first
                pp[j] = c->pbuf[pidx(it - xm + j)];
                sp[j] = D > 1 ? c->sbuf[(it - xm + j) % c->sdepth] : S;
            }
            if (ne == 0) sp[0] = S;
            ec3d_launch_k4s(A, ss, ec3d_src_of(c, EC3D_BY_K2), ec3d_src_of(c, EC3D_BY_SPMV), c->state, it, ne, xe, pp, sp,
                            v[EC3D_VEC_R0], v[EC3D_VEC_X], v[EC3D_VEC_R], c->partials, c->hist, c->hist_cap, s);
        } else if (D <= 1 || (apply && xm == 0)) {
            ec3d_launch_k4(sw, ec3d_src_of(c, EC3D_BY_K2), ec3d_src_of(c, EC3D_BY_SPMV), c->state, it, P, S, v[EC3D_VEC_AS],
                           v[EC3D_VEC_R0], v[EC3D_VEC_X], v[EC3D_VEC_R], c->partials, c->hist, c->hist_cap, s);
        } else {
            const double *pp[EC3D_XD_MAX] = {nullptr}, *sp[EC3D_XD_MAX] = {nullptr};
            const int ne = apply ? xm + 1 : 0;
            for (int j = 0; j < ne; ++j) { // iterations it - xm .. it, oldest first
                pp[j] = c->pbuf[pidx(it - xm + j)];
                sp[j] = c->sbuf[(it - xm + j) % c->sdepth];
            }
            if (ne == 0) sp[0] = S;
            ec3d_launch_k4d(sw, ec3d_src_of(c, EC3D_BY_K2), ec3d_src_of(c, EC3D_BY_SPMV), c->state, it, ne, xe, pp, sp,
                            v[EC3D_VEC_AS], v[EC3D_VEC_R0], v[EC3D_VEC_X], v[EC3D_VEC_R], c->partials, c->hist, c->hist_cap,
                            s);
        }
        if (xa && group_end && part != 1) ec3d_launch_x_group_of(c, it - xm, xm + 1, it >= c->xd_last);
    }
    if ((k == 0 || k == 5) && part != 1) c->it_next = it + 1;
    if ((k == 0 || k == 5) && !f51) {
        ec3d_launch_k5(c->sweep_k5, ec3d_src_of(c, EC3D_BY_K4), c->state, it, v[EC3D_VEC_R], AP, P,
                       ring ? c->pbuf[pidx(it + 1)] : P, v[EC3D_VEC_R0], c->hist, c->hist_cap, s);
        if (ring) c->pcur = pidx(it + 1);
    }
    if ((k == 0 || k == 5) && f51) {
        ec3d_launch_k51(A, ss, ec3d_src_of(c, EC3D_BY_K4), c->state, it, v[EC3D_VEC_R], P, AP, c->pbuf[pidx(it + 1)],
                        c->apbuf[(it + 1) & 1], v[EC3D_VEC_R0], c->partials, c->hist, c->hist_cap, s);
        c->ap_valid_for = it + 1;
        c->pcur = pidx(it + 1);
        c->apcur = (it + 1) & 1;
    }
}

// a new run of iterations on this handle: whatever the second stream still holds belongs to the last one and is waited for
void ec3d_xgroups_reset(ec3d_ctx *c)
{
    if (c->xstream && c->xg_n > 0) (void)hipStreamWaitEvent(c->stream, c->ev_xdone[(c->xg_n - 1) & 1], 0);
    c->xg_n = 0;
    c->xg_done_upto = 0;
}

// ec3d_xasync: the X updates of iterations first .. first + count - 1 as a launch of their own on the second stream,
// ordered behind what the main stream holds now (the K4 of the group's last iteration).  Before that, the main stream
// is made to wait for the PREVIOUS group's launch: the kernels that follow write the ring buffers and SolverState
// entries two groups back, i.e. that group's.  join: the main stream also waits for this launch (the last group of a
// call: whoever synchronises the main stream then has X).
void ec3d_launch_x_group_of(ec3d_ctx *c, int first, int count, bool join)
{
    const int D = ec3d_xdefer(c), d2 = 2 * D, pd = c->pdepth;
    const double *pp[EC3D_XD_MAX], *sp[EC3D_XD_MAX];
    for (int j = 0; j < EC3D_XD_MAX; ++j) { // entries past the count are never dereferenced: any valid pointer
        const int itj = first + std::min(j, count - 1);
        pp[j] = c->pbuf[((itj + c->p_off) % pd + pd) % pd];
        sp[j] = c->sbuf[itj % c->sdepth];
    }
    if (c->xinline) { // on the iteration's own stream, the vector kernels' grid: ordered by the stream itself
        // (256 ... 768 workgroups instead: the iteration within 0.3 % of the full grid's, profiles/r06_x_groups_own_launch.log)
        ec3d_launch_x_group(c->sweep, c->state, pp, sp, first, count, d2, c->vec[EC3D_VEC_X], 0, c->stream);
        ++c->xg_n;
        c->xg_done_upto = first + count - 1;
        return;
    }
    // 128 workgroups (half a workgroup per CU): beside the iteration's kernels the launch takes a small share of the
    // bandwidth, in the gaps between them -- halo planes under way, sums being gathered -- it has the card to itself.
    // 64 ... 256 measured within 2 % of each other on 7 - 16 Mi-row slabs, the vector kernels' own grid (512 and more) 3 % worse
    // (profiles/r05_x_groups_on_a_second_stream.log).  EC3D_XASYNC_WGS overrides (0: the vector kernels' grid).
    const int wgs = getenv("EC3D_XASYNC_WGS") ? atoi(getenv("EC3D_XASYNC_WGS")) : 128;
    EC3D_NOTE(c, hipEventRecord(c->ev_xready, c->stream));
    if (c->xg_n > 0) EC3D_NOTE(c, hipStreamWaitEvent(c->stream, c->ev_xdone[(c->xg_n - 1) & 1], 0));
    EC3D_NOTE(c, hipStreamWaitEvent(c->xstream, c->ev_xready, 0));
    ec3d_launch_x_group(c->sweep, c->state, pp, sp, first, count, d2, c->vec[EC3D_VEC_X], wgs, c->xstream);
    EC3D_NOTE(c, hipEventRecord(c->ev_xdone[c->xg_n & 1], c->xstream));
    if (join) EC3D_NOTE(c, hipStreamWaitEvent(c->stream, c->ev_xdone[c->xg_n & 1], 0));
    ++c->xg_n;
    c->xg_done_upto = first + count - 1;
}

// The X updates an exit at iteration stop_iter left pending (deferred X update): enqueued behind everything else.
int ec3d_flush_x(ec3d_ctx *c, int stop_iter)
{
    const int D = ec3d_xdefer(c);
    if (ec3d_xasync(c)) {
        // the groups already enqueued end themselves at the exit (k_x_group); the group the exit lies in may not have
        // been enqueued yet (the host stopped before its last iteration): now, cut at the exit by the kernel itself
        if (stop_iter > c->xg_done_upto) ec3d_launch_x_group_of(c, c->xg_done_upto + 1, D, true);
        else if (c->xg_n > 0 && !c->xinline) EC3D_HIP(hipStreamWaitEvent(c->stream, c->ev_xdone[(c->xg_n - 1) & 1], 0));
        EC3D_HIP(hipGetLastError());
        EC3D_ASYNC_CHECK(c);
        return 0;
    }
    if ((D <= 1 && !ec3d_k4s(c)) || stop_iter < c->xd_base) return 0;
    const int xm = (stop_iter - c->xd_base) % D;
    const double *pp[EC3D_XD_MAX], *sp[EC3D_XD_MAX];
    for (int j = 0; j < EC3D_XD_MAX; ++j) { // entries past the pending count are never dereferenced: any valid pointer
        const int itj = stop_iter - xm + std::min(j, xm);
        pp[j] = c->pbuf[((itj + c->p_off) % c->pdepth + c->pdepth) % c->pdepth];
        sp[j] = D > 1 ? c->sbuf[itj % c->sdepth] : c->vec[EC3D_VEC_S];
    }
    ec3d_launch_x_flush(c->sweep, c->state, pp, sp, c->vec[EC3D_VEC_X], c->stream);
    EC3D_HIP(hipGetLastError());
    return 0;
}

void ec3d_launch_iteration(ec3d_ctx *c, const MatView &A, int it) { ec3d_launch_stage(c, A, it, 0); }

int ec3d_launch_begin(ec3d_ctx *c, const MatView &A, double tol)
{
    double **v = c->vec;
    ec3d_launch_residual(A, c->sweep_s, v[EC3D_VEC_X], v[EC3D_VEC_B], v[EC3D_VEC_R], v[EC3D_VEC_R0], v[EC3D_VEC_P],
                         c->partials, c->stream);
    ec3d_launch_setup(c->state, ec3d_src_of(c, EC3D_BY_SPMV), tol, c->stream);
    c->pcur = c->apcur = c->scur = 1; // P = R went to vec[P] = pbuf[1]
    c->ap_valid_for = 0;
    c->p_off = 0;
    c->it_next = 1;
    c->xd_base = 1;
    c->xd_last = INT_MAX;
    ec3d_xgroups_reset(c);
    EC3D_HIP(hipGetLastError());
    return 0;
}

int ec3d_single_rank_only(ec3d_ctx *c, const char *who)
{
    // c->dist alone (a full-grid handle configured with nranks = 1, the 1-rank rehearsal layout) must be
    // refused as well: its kernels take their sums from gsum, which only the staged driver fills
    if (c->halo > 0 || c->nranks > 1 || c->dist) {
        ec3d_set_error(std::string(who) + ": this handle holds one z-slab of a multi-rank problem; drive it "
                                          "with ec3d_dist_step (eddy_currents_3d_amd/dist.py)");
        return 4;
    }
    return 0;
}

static int ensure_hist(ec3d_ctx *c, int64_t cap)
{
    if (cap <= 0) {
        c->hist_cap = 0;
        return 0;
    }
    if (c->hist) (void)hipFree(c->hist);
    c->hist = nullptr;
    EC3D_HIP(hipMalloc(&c->hist, (size_t)cap * 2 * sizeof(double)));
    EC3D_HIP(hipMemsetAsync(c->hist, 0xFF, (size_t)cap * 2 * sizeof(double), c->stream)); // NaN = "not reached"
    c->hist_cap = cap;
    return 0;
}

static int solve_core(ec3d_ctx *c, double tol, int32_t itmax, int32_t *iter, double *hist_host, int32_t hist_cap,
                      bool print_on_itmax)
{
    const MatView A = c->A.view();
    const int64_t total = std::max<int64_t>(0, (int64_t)itmax + 1); // src/solvers.f90:25-29
    int rc = ensure_hist(c, hist_host ? std::min<int64_t>(hist_cap, total) : 0);
    if (rc) return rc;
    if ((rc = ec3d_launch_begin(c, A, tol))) return rc;
    c->xd_last = (int)std::min<int64_t>(total, INT_MAX); // the itmax exit: the last iteration applies what is pending

    // iterations per poll: about 0.4 ms of device work, so an exit is noticed within ~1 ms
    const double est_us = (double)c->A.n_pad * 264.0 / 4.0e6 + 12.0;
    const int chunk = (int)std::min<double>(32.0, std::max<double>(1.0, 400.0 / est_us));
    int64_t launched = 0;
    int ci = 0;
    bool stopped = false;
    while (launched < total && !stopped) {
        const int64_t m = std::min<int64_t>(chunk, total - launched);
        for (int64_t i = 0; i < m; ++i) ec3d_launch_iteration(c, A, (int)(++launched));
        EC3D_HIP(hipGetLastError());
        EC3D_ASYNC_CHECK(c);
        EC3D_HIP(hipMemcpyAsync(&c->state_pinned[ci & 1], c->state, sizeof(SolverState), hipMemcpyDeviceToHost,
                                c->stream));
        EC3D_HIP(hipEventRecord(c->ev[ci & 1], c->stream));
        if (ci > 0) {
            EC3D_HIP(hipEventSynchronize(c->ev[(ci - 1) & 1]));
            if (c->state_pinned[(ci - 1) & 1].stop_iter != INT_MAX) stopped = true;
        }
        ++ci;
    }
    EC3D_HIP(hipStreamSynchronize(c->stream));
    SolverState fin;
    EC3D_HIP(hipMemcpy(&fin, c->state, sizeof fin, hipMemcpyDeviceToHost));
    if (fin.stop_iter != INT_MAX && fin.npend > 0) { // an exit with X updates pending (deferred X update): apply them
        if ((rc = ec3d_flush_x(c, fin.stop_iter))) return rc;
        EC3D_HIP(hipStreamSynchronize(c->stream));
    }
    if (fin.stop_iter != INT_MAX) {
        *iter = fin.stop_iter;
    } else {
        *iter = (int32_t)total; // itmax exit: the reference prints norm2(R) and returns (:25-28)
        if (print_on_itmax) {
            // ||R|| as the device summed it: the last K5's value (K5 of iteration itmax + 1 ran without an exit), or the
            // setup's when no iteration ran at all (itmax < 0) -- the same number the residual history holds
            if (ec3d_itmax_print_hold) *ec3d_itmax_print_hold = fin.rnorm;
            else ec3d_print_rnorm(fin.rnorm);
        }
    }
    if (hist_host && c->hist_cap > 0)
        EC3D_HIP(hipMemcpy(hist_host, c->hist, (size_t)c->hist_cap * 2 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ec3d_solve_resident(ec3d_handle c, double tolerance, int32_t itmax, int32_t *iter,
                                   double *resid_hist, int32_t hist_cap)
{
    int rc = ec3d_need_matrix(c, "ec3d_solve_resident");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_solve_resident"))) return rc;
    return solve_core(c, tolerance, itmax, iter, resid_hist, hist_cap, true);
}

extern "C" int ec3d_solve(ec3d_handle c, const double *b, double *x, double tolerance, int32_t itmax,
                          int32_t *iter, double *resid_hist, int32_t hist_cap)
{
    int rc = ec3d_need_matrix(c, "ec3d_solve");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_solve"))) return rc;
    if ((rc = ec3d_vec_h2d(c, c->vec[EC3D_VEC_B], b))) return rc;
    if ((rc = ec3d_vec_h2d(c, c->vec[EC3D_VEC_X], x))) return rc;
    if ((rc = solve_core(c, tolerance, itmax, iter, resid_hist, hist_cap, true))) return rc;
    if ((rc = ec3d_vec_d2h(c, x, c->vec[EC3D_VEC_X]))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

