// ec3d_measure.hip — timing entry points used by bench.py and the tools (hipEvents on the handle's stream).
#include "../../include/ec3d_hip.h"
#include "ec3d_internal.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

// ---------------------------------------------------------------------------------------------
// measurement
extern "C" int ec3d_time_iterations(ec3d_handle c, int32_t iters, double *ms_total)
{
    int rc = ec3d_need_matrix(c, "ec3d_time_iterations");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_time_iterations"))) return rc;
    const MatView A = c->A.view();
    c->hist_cap = 0;
    if ((rc = ec3d_launch_begin(c, A, -1.0))) return rc; // tol < 0: no exit, no restart
    EC3D_HIP(hipEventRecord(c->t0, c->stream));
    c->xd_last = iters;
    for (int it = 1; it <= iters; ++it) ec3d_launch_iteration(c, A, it);
    EC3D_HIP(hipEventRecord(c->t1, c->stream));
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipEventSynchronize(c->t1));
    float ms = 0.f;
    EC3D_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    *ms_total = ms;
    return 0;
}

// Bench "steps": exits disabled (tol < 0), launches only, no host synchronisation.
extern "C" int ec3d_iterate_begin(ec3d_handle c)
{
    int rc = ec3d_need_matrix(c, "ec3d_iterate_begin");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_iterate_begin"))) return rc;
    c->hist_cap = 0;
    return ec3d_launch_begin(c, c->A.view(), -1.0);
}

// the fusions in force on this handle: what stages 2 and 1 of ec3d_iterate's kernel_ms mean
extern "C" int ec3d_get_fusion(ec3d_handle c, int32_t *k2_in_k3, int32_t *k5_in_k1)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_fusion");
    if (rc) return rc;
    if (k2_in_k3) *k2_in_k3 = ec3d_fused23(c) ? 1 : 0;
    if (k5_in_k1) *k5_in_k1 = ec3d_fused51(c) ? 1 : 0;
    return 0;
}

extern "C" int ec3d_get_x_interval(ec3d_handle c, int32_t *iterations)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_x_interval");
    if (rc) return rc;
    if (iterations) *iterations = ec3d_xdefer(c);
    return 0;
}

extern "C" int ec3d_get_x_groups(ec3d_handle c, int32_t *second_stream, int32_t *groups_launched)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_x_groups");
    if (rc) return rc;
    if (second_stream) *second_stream = !ec3d_xasync(c) ? 0 : (c->xinline && !c->dist && c->halo == 0) ? 2 : 1;
    if (groups_launched) *groups_launched = c->xg_n;
    return 0;
}

extern "C" int ec3d_get_k4_form(ec3d_handle c, int32_t *spmv_form)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_k4_form");
    if (rc) return rc;
    if (spmv_form) *spmv_form = ec3d_k4s(c) ? 1 : 0;
    return 0;
}

extern "C" int ec3d_iterate(ec3d_handle c, int32_t first_iter, int32_t count, double *kernel_ms)
{
    int rc = ec3d_need_matrix(c, "ec3d_iterate");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_iterate"))) return rc;
    const MatView A = c->A.view();
    // deferred X update: groups counted from this call's first iteration, its last one applies what is pending --
    // every call leaves X complete
    // The device state is addressed by the iteration number -- rr0[it & 1], AP in apbuf[it & 1], P and S in their rings --
    // so a call has to continue where the last one ended (ec3d_iterate_begin starts again from 1): iterate(1, n) twice
    // would read an older P and the other rr0.
    if (first_iter != c->it_next) {
        ec3d_set_error("ec3d_iterate: first_iter = " + std::to_string(first_iter) + " does not continue the iterations of this "
                       "handle (next: " + std::to_string(c->it_next) + "; ec3d_iterate_begin starts again from 1)");
        return 6;
    }
    c->xd_base = first_iter;
    c->xd_last = first_iter + count - 1;
    if (!kernel_ms) {
        for (int it = first_iter; it < first_iter + count; ++it) ec3d_launch_iteration(c, A, it);
        EC3D_HIP(hipGetLastError());
        return 0;
    }
    // per-kernel durations: an event at every kernel boundary of every iteration, on our stream
    std::vector<hipEvent_t> ev((size_t)count * 6);
    for (auto &e : ev) EC3D_HIP(hipEventCreate(&e));
    hipStream_t s = c->stream;
    for (int i = 0; i < count; ++i) {
        const int it = first_iter + i;
        hipEvent_t *e = &ev[(size_t)i * 6];
        EC3D_HIP(hipEventRecord(e[0], s));
        for (int k = 1; k <= 5; ++k) {
            ec3d_launch_stage(c, A, it, k);
            EC3D_HIP(hipEventRecord(e[k], s));
        }
    }
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < 5; ++k) kernel_ms[k] = 0.0;
    for (int i = 0; i < count; ++i)
        for (int k = 0; k < 5; ++k) {
            float ms = 0.f;
            EC3D_HIP(hipEventElapsedTime(&ms, ev[(size_t)i * 6 + k], ev[(size_t)i * 6 + k + 1]));
            kernel_ms[k] += (double)ms / count;
        }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return 0;
}

extern "C" int ec3d_time_kernel(ec3d_handle c, int kernel, int32_t reps, double *ms_per_launch)
{
    int rc = ec3d_need_matrix(c, "ec3d_time_kernel");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_time_kernel"))) return rc;
    const MatView A = c->A.view();
    double **v = c->vec;
    hipStream_t s = c->stream;
    c->hist_cap = 0;
    if (kernel < EC3D_K_SPMV || kernel > EC3D_K5) {
        ec3d_set_error("ec3d_time_kernel: unknown kernel");
        return 2;
    }
    if (kernel == EC3D_K2 && ec3d_fused23(c)) { // stage 2 launches nothing on this handle: no time to report
        ec3d_set_error("ec3d_time_kernel: K2 runs inside K3 on this handle (ec3d_get_fusion); time EC3D_K3 instead");
        return 5;
    }
    if ((rc = ec3d_launch_begin(c, A, -1.0))) return rc;
    c->xd_last = 1;
    ec3d_launch_iteration(c, A, 1); // populate every partial slot and the scalars
    c->xd_base = c->xd_last = 2;     // (the single stages below: the classic K4, nothing pending)
    auto one = [&]() {
        if (kernel == EC3D_K_SPMV)
            ec3d_launch_spmv(A, c->sweep_s, v[EC3D_VEC_P], v[EC3D_VEC_AP], s);
        else {
            // K5-in-K1 handles: EC3D_K1 times the plain K1 of iteration 2 (on the AP buffer the fused launch filled),
            // EC3D_K5 the fused K5 + K1 launch
            if (kernel == EC3D_K1) c->ap_valid_for = 0;
            ec3d_launch_stage(c, A, 2, kernel);
        }
    };
    one(); // warm
    EC3D_HIP(hipEventRecord(c->t0, s));
    for (int i = 0; i < reps; ++i) one();
    EC3D_HIP(hipEventRecord(c->t1, s));
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipEventSynchronize(c->t1));
    float ms = 0.f;
    EC3D_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    *ms_per_launch = (double)ms / std::max(1, reps);
    return 0;
}


// ||B - A X|| / ||B|| of the resident vectors, computed on the device by the setup kernel of the solve
// (src/solvers.f90:14-21: R = B - A X, partial sums of B.B and R.R), the workgroup partials added on the
// host in workgroup order.  Overwrites the work vectors R, R0, P -- which the next solve rebuilds anyway.
// What a caller uses to check a returned x independently of the iteration's own recurrence.
extern "C" int ec3d_true_residual(ec3d_handle c, double *rel, double *bnorm)
{
    int rc = ec3d_need_matrix(c, "ec3d_true_residual");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_true_residual"))) return rc;
    double **v = c->vec;
    ec3d_launch_residual(c->A.view(), c->sweep_s, v[EC3D_VEC_X], v[EC3D_VEC_B], v[EC3D_VEC_R], v[EC3D_VEC_R0],
                         v[EC3D_VEC_P], c->partials, c->stream);
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipStreamSynchronize(c->stream));
    const int nb = c->sweep_s.nblk;
    std::vector<double> part((size_t)nb);
    double s[2] = {0.0, 0.0};
    const int slot[2] = {P_BB, P_RR_INIT};
    for (int k = 0; k < 2; ++k) {
        EC3D_HIP(hipMemcpy(part.data(), c->partials + (size_t)slot[k] * c->sweep.pstride, part.size() * sizeof(double),
                           hipMemcpyDeviceToHost));
        for (int q = 0; q < nb; ++q) s[k] += part[(size_t)q];
    }
    if (bnorm) *bnorm = std::sqrt(s[0]);
    *rel = s[0] > 0.0 ? std::sqrt(s[1] / s[0]) : std::sqrt(s[1]);
    return 0;
}
