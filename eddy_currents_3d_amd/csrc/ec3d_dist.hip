// ec3d_dist.hip — per-stage entry points for the multi-rank (z-slab) schedule driven by
// eddy_currents_3d_amd/dist.py: the library launches, the host moves halos and sums between the stages.
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
// multi-rank (z-slab) building blocks: one process per GPU drives these from
// eddy_currents_3d_amd/dist.py with torch.distributed (RCCL) between the stages
extern "C" int ec3d_vector_layout(ec3d_handle c, int64_t *ghost, int64_t *n, int64_t *n_pad, int64_t *halo)
{
    int rc = ec3d_need_matrix(c, "ec3d_vector_layout");
    if (rc) return rc;
    *ghost = c->ghost;
    *n = c->A.n;
    *n_pad = c->A.n_pad;
    *halo = c->halo;
    return 0;
}

// Use caller-owned device memory for the 8 work vectors: EC3D_NVEC * (ghost + n_pad + ghost) doubles,
// zero filled by the caller.  Vector v's element 0 is at base[v*len + ghost].
extern "C" int ec3d_adopt_vectors(ec3d_handle c, double *base)
{
    int rc = ec3d_need_matrix(c, "ec3d_adopt_vectors");
    if (rc) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    if (c->vec_base && c->own_vectors) (void)hipFree(c->vec_base);
    const int64_t len = c->ghost + c->A.n_pad + c->ghost;
    c->vec_base = base;
    c->own_vectors = false;
    for (int v = 0; v < EC3D_NVEC; ++v) c->vec[v] = base + (size_t)v * len + c->ghost;
    return ec3d_spare_pair(c); // adopted vectors: no spare pair, the kernels of an iteration stay unfused
}

extern "C" int ec3d_dist_configure(ec3d_handle c, int32_t nranks, double *lsum_device, double *gsum_device)
{
    if (nranks == 1 && !lsum_device && !gsum_device) { // back to the single-rank loop (ec3d_solve & co.)
        c->nranks = 1;
        c->lsum = c->gsum = nullptr;
        c->lsum_ptrs = nullptr;
        c->dist = false;
        return 0;
    }
    if (nranks < 1 || !lsum_device || !gsum_device) {
        ec3d_set_error("ec3d_dist_configure: need nranks >= 1 and two device buffers");
        return 2;
    }
    c->nranks = nranks;
    c->lsum = lsum_device;
    c->gsum = gsum_device;
    c->dist = true;
    return 0;
}

extern "C" int ec3d_dist_set_boundary_rows(ec3d_handle c, int32_t nranges, const int64_t *lo, const int64_t *hi,
                                           int32_t *enabled)
{
    int rc = ec3d_need_matrix(c, "ec3d_dist_set_boundary_rows");
    if (rc) return rc;
    if (nranges < 0 || (nranges > 0 && (!lo || !hi))) return 2;
    if (enabled) *enabled = 0;
    // tiles the vector kernels visit: the front sweep and the occupied U tiles of the structured form
    const Sweep &sw = c->sweep;
    std::vector<int32_t> visit((size_t)sw.ntiles);
    for (int64_t t = 0; t < sw.ntiles; ++t) visit[(size_t)t] = (int32_t)ec3d_phys_tile(sw, t);
    if (sw.ulist_n) {
        std::vector<int32_t> ul((size_t)sw.ulist_n);
        EC3D_HIP(hipMemcpy(ul.data(), sw.ulist, ul.size() * 4, hipMemcpyDeviceToHost));
        visit.insert(visit.end(), ul.begin(), ul.end());
    }
    std::vector<int32_t> vb, vi;
    for (int32_t t : visit) {
        const int64_t r0 = (int64_t)t * EC3D_TILE, r1 = r0 + EC3D_TILE;
        bool bnd = false;
        for (int32_t q = 0; q < nranges && !bnd; ++q) bnd = lo[q] < r1 && hi[q] > r0;
        (bnd ? vb : vi).push_back(t);
    }
    if (c->vb_list) (void)hipFree(c->vb_list);
    if (c->vi_list) (void)hipFree(c->vi_list);
    c->vb_list = c->vi_list = nullptr;
    c->can_vsplit = false;
    if (vb.empty() || vi.empty()) return 0; // nothing to split (single rank, or a slab that is all boundary)
    EC3D_HIP(hipMalloc(&c->vb_list, vb.size() * 4));
    EC3D_HIP(hipMalloc(&c->vi_list, vi.size() * 4));
    EC3D_HIP(hipMemcpy(c->vb_list, vb.data(), vb.size() * 4, hipMemcpyHostToDevice));
    EC3D_HIP(hipMemcpy(c->vi_list, vi.data(), vi.size() * 4, hipMemcpyHostToDevice));
    auto list_sweep = [&](const int32_t *list, size_t len, int max_blk, int part_off) {
        Sweep s = sw;
        s.ntiles = 0; // list only
        s.ulist = list;
        s.ulist_n = (int)len;
        s.nblk = (int)std::min<size_t>(len, (size_t)max_blk);
        s.S = 0;
        s.part_off = part_off;
        return s;
    };
    c->sweep_vb = list_sweep(c->vb_list, vb.size(), 256, 0);
    c->sweep_vi = list_sweep(c->vi_list, vi.size(), sw.nblk, c->sweep_vb.nblk);
    c->can_vsplit = true;
    if (enabled) *enabled = 1;
    return 0;
}

// The same split for a z-slab of the SINGLE-COMPONENT operator whose planes are whole tiles, without tile lists: the boundary
// launch takes the first and the last owned plane, the interior launch the planes between, both as windows of the ordinary
// strided sweep (Sweep::win_*: logical tile t of the boundary launch is tile t of plane 0 for t < tpp, of plane np-1 after;
// the interior launch starts at plane 1) -- a list-driven K2 / K5 costs 15-30 us more at 16 Mi rows than the strided one.
// can K2 / K5 of this slab run as a launch over planes 0 and np-1 followed by one over the planes between?
bool ec3d_dist_can_split_planes(const ec3d_ctx *c)
{
    const Sweep &ss = c->sweep_s, &k2 = c->sweep_k2;
    if (c->A.sav || c->halo <= 0 || c->nown != 0 || k2.ulist_n != 0 || ss.zm_tpp <= 0 || k2.win_nt != 0) return false;
    const int64_t tpp = ss.zm_tpp, total = k2.ntiles, np = total / tpp;
    return np * tpp == total && np >= 3;
}

int ec3d_dist_set_boundary_planes(ec3d_ctx *c, int32_t *enabled)
{
    if (enabled) *enabled = 0;
    const Sweep &ss = c->sweep_s, &k2 = c->sweep_k2;
    if (c->vb_list) (void)hipFree(c->vb_list);
    if (c->vi_list) (void)hipFree(c->vi_list);
    c->vb_list = c->vi_list = nullptr;
    c->can_vsplit = false;
    if (!ec3d_dist_can_split_planes(c)) return 0;
    const int64_t tpp = ss.zm_tpp, total = k2.ntiles;
    Sweep vb = k2, vi = k2;
    vb.ntiles = 2 * tpp;
    vb.win_nt = tpp;
    vb.win_blk = total - tpp;
    vb.win_t0 = 0;
    vb.S = 0;
    vb.vec_depth = 1;
    vb.nblk = (int)std::max<int64_t>(8, std::min<int64_t>(2 * tpp, 256) / 8 * 8);
    vb.part_off = 0;
    vi.ntiles = total - 2 * tpp;
    vi.win_nt = vi.ntiles;
    vi.win_blk = total;
    vi.win_t0 = tpp;
    vi.nblk = (int)std::min<int64_t>(k2.nblk, std::max<int64_t>(8, vi.ntiles / 8 * 8));
    vi.S = k2.S > 0 ? vi.nblk / 8 : 0;
    vi.part_off = vb.nblk;
    c->sweep_vb = vb;
    c->sweep_vi = vi;
    c->can_vsplit = true;
    if (enabled) *enabled = 1;
    return 0;
}

extern "C" int ec3d_dist_step(ec3d_handle c, int32_t stage, int32_t it, double tolerance)
{
    int rc = ec3d_need_matrix(c, "ec3d_dist_step");
    if (rc) return rc;
    if (!c->dist) {
        ec3d_set_error("ec3d_dist_step: call ec3d_dist_configure first");
        return 3;
    }
    const MatView A = c->A.view();
    double **v = c->vec;
    auto fin = [&](int producer, unsigned mask, bool split = false) {
        ec3d_launch_finalize(ec3d_part_of(c, producer, split), c->lsum, mask, c->stream);
    };
    // K2's S.S is wanted behind the SAME gather as K3's AS.S and AS.AS (K3 is launched before ||S|| is known: the S exit is
    // K4's), so K2 leaves its workgroups' partials where they are and K3's collapse launch folds both: one launch fewer per
    // iteration on the five-launch plans, the same sums in the same order
    auto fin_k3 = [&](bool split) {
        const RedSrc k3 = ec3d_part_of(c, EC3D_BY_SPMV, split);
        const unsigned m3 = 1u << P_D2 | 1u << P_D3 | (ec3d_fused23(c) ? 1u << P_SS : 0u);
        if (c->ss_parts > 0 && !ec3d_fused23(c))
            ec3d_launch_finalize2(RedSrc{c->partials, c->ss_parts, 1, c->sweep.pstride, nullptr}, 1u << P_SS, k3, m3, c->lsum, c->stream);
        else
            ec3d_launch_finalize(k3, c->lsum, m3, c->stream);
        c->ss_parts = 0;
    };
    auto need_split = [&]() {
        if (!c->can_overlap) ec3d_set_error("ec3d_dist_step: this slab cannot split K1/K3 (see ec3d_can_overlap)");
        return c->can_overlap;
    };
    switch (stage) {
    case EC3D_STAGE_RESID:
        c->hist_cap = 0;
        c->pcur = c->apcur = c->scur = 1; // (as ec3d_launch_begin: P = R goes to vec[P] = pbuf[1])
        c->ap_valid_for = 0;
        c->p_off = 0;
        c->it_next = 1;
        c->xd_base = 1;
        c->xd_last = INT_MAX;
        c->ss_parts = 0;
        ec3d_xgroups_reset(c);
        ec3d_launch_residual(A, c->sweep_s, v[EC3D_VEC_X], v[EC3D_VEC_B], v[EC3D_VEC_R], v[EC3D_VEC_R0],
                             v[EC3D_VEC_P], c->partials, c->stream);
        fin(EC3D_BY_SPMV, 1u << P_BB | 1u << P_RR_INIT);
        break;
    case EC3D_STAGE_SETUP: ec3d_launch_setup(c->state, ec3d_src_of(c, EC3D_BY_SPMV), tolerance, c->stream); break;
    // On a slab that runs the three-launch iteration (ec3d_ctx::slab_fused) stage 1 launches K1 only when AP = A P of this
    // iteration does not exist yet (iteration 1, or a call that does not continue the last one), stage 2 nothing, stage 3
    // K2-in-K3 (sums S.S, AS.S, AS.AS), stage 4 K4 in SpMV form, stage 5 K5-in-K1, which sums AP.R0 of the NEXT iteration.
    case EC3D_STAGE_K1:
        if (ec3d_fused51(c) && it != 1 && c->ap_valid_for == it) break;
        ec3d_launch_stage(c, A, it, 1);
        fin(EC3D_BY_SPMV, 1u << P_D1);
        break;
    case EC3D_STAGE_K2:
        if (ec3d_fused23(c)) break;
        ec3d_launch_stage(c, A, it, 2);
        c->ss_parts = ec3d_part_of(c, EC3D_BY_K2, false).count;
        break;
    case EC3D_STAGE_K3:
        ec3d_launch_stage(c, A, it, 3);
        fin_k3(false);
        break;
    case EC3D_STAGE_K4: ec3d_launch_stage(c, A, it, 4); fin(EC3D_BY_K4, 1u << P_RR | 1u << P_RR0N); break;
    case EC3D_STAGE_K5:
        ec3d_launch_stage(c, A, it, 5);
        if (ec3d_fused51(c)) fin(EC3D_BY_SPMV, 1u << P_D1);
        break;
    case EC3D_STAGE_K1_INT:
        if (!need_split()) return 3;
        ec3d_launch_k1(A, c->sweep_int, c->state, it, ec3d_vec_at(c, EC3D_VEC_P, it), v[EC3D_VEC_R0], v[EC3D_VEC_AP],
                       c->partials, c->stream);
        break;
    case EC3D_STAGE_K1_BND:
        if (!need_split()) return 3;
        ec3d_launch_k1(A, c->sweep_bnd, c->state, it, ec3d_vec_at(c, EC3D_VEC_P, it), v[EC3D_VEC_R0], v[EC3D_VEC_AP],
                       c->partials, c->stream);
        fin(EC3D_BY_SPMV, 1u << P_D1, true);
        break;
    case EC3D_STAGE_K3_INT:
        if (!need_split()) return 3;
        ec3d_launch_k3(A, c->sweep_int, c->state, it, ec3d_vec_at(c, EC3D_VEC_S, it), v[EC3D_VEC_AS], c->partials, c->stream);
        break;
    case EC3D_STAGE_K3_BND:
        if (!need_split()) return 3;
        ec3d_launch_k3(A, c->sweep_bnd, c->state, it, ec3d_vec_at(c, EC3D_VEC_S, it), v[EC3D_VEC_AS], c->partials, c->stream);
        fin_k3(true);
        break;
    case EC3D_STAGE_K2_BND:
    case EC3D_STAGE_K2_INT: {
        if (!c->can_vsplit) {
            ec3d_set_error("ec3d_dist_step: call ec3d_dist_set_boundary_rows first");
            return 3;
        }
        const bool bnd = stage == EC3D_STAGE_K2_BND;
        ec3d_launch_k2(bnd ? c->sweep_vb : c->sweep_vi, ec3d_src_of(c, EC3D_BY_SPMV), c->state, it, v[EC3D_VEC_R], v[EC3D_VEC_AP],
                       ec3d_vec_at(c, EC3D_VEC_S, it), c->partials, c->stream);
        c->scur = ec3d_xdefer(c) > 1 ? it % c->sdepth : 1;
        if (!bnd) c->ss_parts = c->sweep_vb.nblk + c->sweep_vi.nblk; // (both launches' partials, folded by K3's collapse launch)
        break;
    }
    case EC3D_STAGE_K5_BND:
    case EC3D_STAGE_K5_INT:
        if (!c->can_vsplit) {
            ec3d_set_error("ec3d_dist_step: call ec3d_dist_set_boundary_rows first");
            return 3;
        }
        // (deferred X update: the new P goes to the next buffer of the ring, as in ec3d_launch_stage)
        ec3d_launch_k5(stage == EC3D_STAGE_K5_BND ? c->sweep_vb : c->sweep_vi, ec3d_src_of(c, EC3D_BY_K4), c->state, it,
                       v[EC3D_VEC_R], v[EC3D_VEC_AP], ec3d_vec_at(c, EC3D_VEC_P, it), ec3d_vec_at(c, EC3D_VEC_P, it + 1),
                       v[EC3D_VEC_R0], c->hist, c->hist_cap, c->stream);
        if (ec3d_xdefer(c) > 1) c->pcur = ((it + 1 + c->p_off) % c->pdepth + c->pdepth) % c->pdepth;
        c->it_next = it + 1;
        break;
    case EC3D_STAGE_K4F_BND:
    case EC3D_STAGE_K4F_INT:
    case EC3D_STAGE_K5F_BND:
    case EC3D_STAGE_K5F_INT: {
        if (!c->can_fsplit || !c->slab_fused || !ec3d_k4s(c)) {
            ec3d_set_error("ec3d_dist_step: this slab does not run the three-launch iteration in split launches");
            return 3;
        }
        const bool k4 = stage == EC3D_STAGE_K4F_BND || stage == EC3D_STAGE_K4F_INT;
        const bool second = stage == EC3D_STAGE_K4F_INT || stage == EC3D_STAGE_K5F_INT;
        ec3d_launch_stage(c, A, it, k4 ? 4 : 5, second ? 2 : 1);
        if (second) {
            if (k4) fin(EC3D_BY_K4, 1u << P_RR | 1u << P_RR0N, true);
            else fin(EC3D_BY_SPMV, 1u << P_D1, true);
        }
        break;
    }
    default: ec3d_set_error("ec3d_dist_step: unknown stage"); return 2;
    }
    EC3D_HIP(hipGetLastError());
    EC3D_ASYNC_CHECK(c);
    return 0;
}

// kernel launches ec3d_dist_step issues for a stage (the stage's kernel and, where it produces sums, k_finalize); on a slab
// that runs the three-launch iteration stages 1 and 2 are empty and stage 5 produces a sum
int ec3d_dist_launches(const ec3d_ctx *c, int stage, int it)
{
    const int fin = 1; // the collapse launch behind a producer of sums
    switch (stage) {
    case EC3D_STAGE_K1: return (ec3d_fused51(c) && it != 1 && c->ap_valid_for == it) ? 0 : 1 + fin;
    case EC3D_STAGE_K2: return ec3d_fused23(c) ? 0 : 1; // (its S.S partials are folded by K3's collapse launch)
    case EC3D_STAGE_K5: return ec3d_fused51(c) ? 1 + fin : 1;
    case EC3D_STAGE_SETUP: case EC3D_STAGE_K1_INT: case EC3D_STAGE_K3_INT: case EC3D_STAGE_K2_BND: case EC3D_STAGE_K2_INT:
    case EC3D_STAGE_K5_BND: case EC3D_STAGE_K5_INT: case EC3D_STAGE_K4F_BND: case EC3D_STAGE_K5F_BND: return 1;
    default: return 1 + fin;
    }
}

extern "C" int ec3d_can_overlap(ec3d_handle c) { return c && c->have_matrix && c->can_overlap ? 1 : 0; }

extern "C" int ec3d_read_state_async(ec3d_handle c, int32_t *stop_iter_pinned)
{
    if (!c || !c->state || !stop_iter_pinned) return 2;
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipMemcpyAsync(stop_iter_pinned, &c->state->stop_iter, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    return 0;
}

// synchronous read of the device-resident solver state (stream is drained first)
extern "C" int ec3d_read_state(ec3d_handle c, int32_t *stop_iter, int32_t *stop_kind, double *bnorm)
{
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    SolverState st;
    EC3D_HIP(hipMemcpy(&st, c->state, sizeof st, hipMemcpyDeviceToHost));
    if (stop_iter) *stop_iter = st.stop_iter == INT_MAX ? -1 : st.stop_iter;
    if (stop_kind) *stop_kind = st.stop_kind;
    if (bnorm) *bnorm = st.bnorm;
    return 0;
}

// how often the restart rule of src/solvers.f90:47-49 fired in the last solve (counted by K5's lead thread)
extern "C" int ec3d_get_restart_count(ec3d_handle c, int32_t *count)
{
    if (!c || !c->state || !count) return 2;
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    SolverState st;
    EC3D_HIP(hipMemcpy(&st, c->state, sizeof st, hipMemcpyDeviceToHost));
    *count = st.restarts;
    return 0;
}

