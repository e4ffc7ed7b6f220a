// ec3d_internal.hpp — shared declarations of libec3d_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ec3d_hip.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>

#define EC3D_THREADS 256
#define EC3D_TILE 512 /* rows per tile: every thread owns 2 consecutive rows (one 16-B access) */
#define EC3D_MAXB 16  /* DIA bands */
#define EC3D_CHUNK 64 /* sliced-ELL slice height = one wavefront */
#define EC3D_OUT_SLOTS 3 /* pinned host buffers of the overlapped field output (ec3d_vtk_fields_begin) */
#ifndef EC3D_PX
#define EC3D_PX 128   /* 2-D tile (patch) of the z-marching kernels: cells along x (one wave = one patch row) ... */
#endif
#define EC3D_PY (EC3D_TILE / EC3D_PX) /* ... and along y */

// ---------------------------------------------------------------------------------------------
// Device format.  Rows [0, n_pad).  Row r of A is
//     sum_b band[b][r] * x[r + off[b]]   (bands in ascending offset order)
//   + sum_j tval[...] * x[tcol[...]]     (tail entries of row r, in stored order)
// which is the reference's ascending-column summation order (src/solvers.f90:59 after full_sort,
// src/EC3D.f90:715) whenever every tail column of a row exceeds its band columns; rows that
// violate that are stored entirely in the tail (their band coefficients are zero).
struct MatView {
    const double *band[EC3D_MAXB];
    int64_t off[EC3D_MAXB];
    int nb;
    int has_tail;
    const int32_t *tail_id;   // [n_pad]  -1 or index of the row's tail slot
    const uint8_t *tile_flag; // [ntiles] 1 when any row of the tile has a tail
    const int64_t *chunk_ptr; // [nchunk+1] entry offsets of the 64-row slices
    const int32_t *tcol;      // 0-based column
    const double *tval;
    // dictionary form of the bands (ncls > 0): band[b][r] == table[cls[r] * nb + b]
    const uint8_t *cls; // [n_pad]
    const double *table;
    int ncls;
    int pm1; // bands 2 and 4 of a 7-band operator are the offsets -1 and +1
    // Structured A-V form (sav != 0): unknowns [Ax | Ay | Az | U] with U EMBEDDED in the grid (one U
    // slot per cell, 4*nC rows), so the A<->U couplings are fixed-offset stencil slots too and there is
    // no tail.  A class = 7 band coefficients + 9 coupling coefficients (table stride 16):
    //   row of block d < 3, classes [sav_a0, sav_u0): + sum_{m=-2..2} t[7+m+2] * x[r + (3-d)*nC + m*step_d]
    //   row of block 3,     classes [sav_u0, sav_zero): sum_{d,j} t[7+3d+j] * x[r - (3-d)*nC + (j-1)*step_d]
    //                       added BEFORE the bands (the A columns are the lower ones)
    int sav;
    int sav_a0, sav_u0, sav_zero;
    int64_t sav_nC, sav_step[3];
};

// where a kernel finds the partial sums it has to finish: value i of slot s is
// base[s * slot_mul + i * stride], i < count.  Single GPU: the producer's per-workgroup partials
// (count = nblk, stride = 1, slot_mul = nblk).  Multi rank: the all-gathered per-rank sums
// (count = nranks, stride = P_NSLOT, slot_mul = 1).  In-library multi-GPU (ec3d_multi.hip): no gathered
// copy at all -- ptrs[i] is rank i's own lsum[P_NSLOT], read in place through the peer mapping
// (value i of slot s = ptrs[i][s]).
struct RedSrc {
    const double *base;
    int count, stride, slot_mul;
    const double *const *ptrs;
};

// blockIdx -> tile map (XCD aware when S > 0): see ec3d_tile_of() in ec3d_kernels.hip
struct Sweep {
    int64_t ntiles; // LOGICAL tiles of the front sweep (see win_* below); the U-block list comes behind them
    int64_t n; // rows owned; rows in [n, ntiles*EC3D_TILE) are padding
    int nblk;
    int S;
    int nt;      // nontemporal policy for once-touched streams (large problems)
    int vec_depth; // vector kernels: tiles whose operands are requested together (1, 2, 4)
    int pstride; // doubles between two slots of the partial-sum buffer
    // z-marching map of the SpMV kernels (zm_tpp > 0): a workgroup owns one 512-row position of the
    // xy-plane ("column") and walks zm_pps consecutive planes, so x[r-kdz], x[r] stay in registers
    int zm_tpp;  // tiles per plane = kdz / 512
    int zm_pps;  // planes per z segment
    // 2-D tiles of the z-marching single-component kernels (patch_npx > 0): a tile is a patch of EC3D_PX x EC3D_PY
    // cells of the xy plane instead of 512 consecutive cells; tile q of a plane is patch (q % patch_npx, q / patch_npx),
    // thread t owns cells 2 (t % (EC3D_PX/2)) and the next one of the patch's row t / (EC3D_PX/2).  See patch_pair in
    // ec3d_kernels.hip.
    int patch_npx;  // patches per grid row = sdx / EC3D_PX
    int64_t patch_sdx;
    // Runtime-shaped 2-D tiles of the z-marching STRUCTURED A-V kernels (rp_px > 0; sav_patch_step in
    // ec3d_kernels.hip): a tile is a patch of rp_px x rp_py cells (rp_px even, a divisor of the grid's sdx,
    // rp_px * rp_py <= 512) of one xy plane of one block; zm_tpp = rp_npx * ceil(rp_sdy / rp_py) patches per plane,
    // patch q of a plane is (q % rp_npx, q / rp_npx); thread t owns cells 2t, 2t + 1 of the patch in row-major order
    // (threads with 2t >= rp_px * rp_py, and rows beyond rp_sdy in the last patch row, idle).  A plane of a block
    // starts every rp_pitch device rows (the structured form's plane pitch), so tile T = P * zm_tpp + q lies in
    // plane P of the four stacked blocks.  rp_flag[T]: some row of the A-block patch T is coupled to U.
    int rp_px, rp_py, rp_npx, rp_sdy;
    int64_t rp_sdx, rp_pitch;
    const uint8_t *rp_flag;
    // rows that count in the dot products when this handle holds an A-V slab on an extended grid whose planes
    // are NOT tile aligned (nown > 0; bands + tail, or the structured form on a small grid): [Ax | Ay | Az | U]
    // each contribute one owned index range.  Tile-aligned structured slabs use the window below instead.
    int nown;
    int64_t own_lo[4], own_hi[4];
    // Window (structured A-V slab with tile-aligned planes, win_nt > 0): a block of the device layout holds
    // win_blk tiles (all held planes), of which the win_nt tiles from win_t0 on are OWNED; the halo planes'
    // rows are read (neighbours' values) but never swept: no kernel computes or stores there and every swept
    // row counts in the dot products.  Logical tile L of the front sweep is physical tile
    // (L / win_nt) * win_blk + win_t0 + L % win_nt; the same in planes (tiles / zm_tpp) for the z-march.
    int64_t win_nt, win_blk, win_t0;
    // split launches of the SpMV kernels in a z-slab, so the halo exchange overlaps the interior:
    // zm_pl0 > 0: z-march over planes [zm_pl0, zm_pl0 + zm_npl) only (interior launch);
    // bnd_last >= 0: the launch covers planes 0 and bnd_last only (boundary launch, plain tile order)
    int zm_pl0, zm_npl;
    // zm_plstep > 1: the launch covers the logical planes zm_pl0, zm_pl0 + zm_plstep, ... (zm_npl of them), one plane per
    // workgroup step sequence (zm_pps = 1): the boundary launch of the 2-D-tile kernels in a z-slab -- planes 0 and np-1
    int zm_plstep;
    int bnd_last;  // -1: not a boundary launch
    // z-slab of the single-component operator running the three-launch iteration (K2 inside K3, K5 inside the next K1):
    // S and P are FORMED on the halo planes by the kernels that need them there, and stored into the vectors' ghost rows
    // for the next kernel's stencil (K4 in SpMV form reads S one plane away; the next K5-in-K1 reads the old P there).
    // bit 0: this slab has a lower z-neighbour (store the plane below plane 0), bit 1: an upper one (above the last plane)
    int halo_store;
    int part_off;  // first partial-sum index this launch writes within a slot
    // structured A-V form: tiles [0, ntiles) are swept as usual (the three A blocks); of the tiles
    // behind them (the grid-shaped U block) only those holding an unknown are visited, from a list --
    // everything else there is identically zero in every vector and stays so
    const int32_t *ulist;
    int ulist_n;
    // INTERLEAVED z-march of the structured A-V form (il_planes > 0; single-rank handle, tile-aligned planes; walk_zm_il in
    // ec3d_kernels.hip): a workgroup owns one column and a range of xy planes k of ONE block's extent, and at every plane
    // visits the tiles of A_x, A_y, A_z there and -- when it holds an unknown -- the U tile, before it moves to plane
    // k + 1.  The coupling operands of a row (U next to an A row's cell, A_x / A_y / A_z next to a U row's cell,
    // src/EC3D.f90:656-711, :766-959) are then lines the same workgroup (or its XCD neighbour) fetched within the last
    // step or two: they come out of the L2 instead of HBM.  il_planes = planes per block, il_umask = one bit per
    // (column, plane): word col * il_nw + plane / 32; the trailing U list is empty (ulist_n = 0).
    int il_planes = 0, il_nw = 0;
    const uint32_t *il_umask = nullptr;
    // the march's work list: workgroup b takes planes [il_seg[4 b + 1], il_seg[4 b + 2]) of column il_seg[4 b] (an empty range:
    // nothing).  Cut by WEIGHT (a plane with a U tile costs more than one without), per column, the columns of XCD label x =
    // b % 8 being its cpx adjacent ones, so that every workgroup of the launch ends at about the same time.
    const int32_t *il_seg = nullptr;
};

// first of the two consecutive rows thread t of a workgroup owns in `tile`
template <class SW>
__host__ __device__ inline int64_t ec3d_row_of(const SW &sw, int64_t tile, int t)
{
    if (sw.patch_npx <= 0) return tile * EC3D_TILE + 2 * (int64_t)t;
    const int64_t plane = tile / sw.zm_tpp, q = tile % sw.zm_tpp;
    const int64_t py = q / sw.patch_npx, px = q % sw.patch_npx;
    return plane * (int64_t)sw.zm_tpp * EC3D_TILE + (py * EC3D_PY + t / (EC3D_PX / 2)) * sw.patch_sdx + px * EC3D_PX +
           2 * (int64_t)(t % (EC3D_PX / 2));
}

// logical -> physical tile of the front sweep (identity without a window)
template <class SW>
__host__ __device__ inline int64_t ec3d_phys_tile(const SW &sw, int64_t t)
{
    if (sw.win_nt <= 0) return t;
    return (t / sw.win_nt) * sw.win_blk + sw.win_t0 + t % sw.win_nt;
}

// blockIdx -> tile map of a sweep (host and device: ec3d_get_visit_order enumerates with the same function).
// Returns the PHYSICAL tile, or -1 when workgroup b has no i-th tile in the front sweep.
// MODE: -1 = decide from the sweep's fields (host enumeration, vector kernels); 1 = the z-marching map is known to
// apply; 0 = it is known not to.  (The z-marching SpMV kernels walk the same sequence incrementally,
// EC3D_ZSWEEP in ec3d_kernels.hip: plane after plane of one column, restarting at every block of a windowed slab.)
template <int MODE = -1, class SW = Sweep>
__host__ __device__ inline int64_t ec3d_tile_of(const SW &sw, int b, int64_t i)
{
    if (MODE != 1 && sw.bnd_last >= 0) {
        // boundary launch of a z-slab: the first and the last owned plane, plain tile order
        const int64_t t = i * (int64_t)sw.nblk + b;
        if (t >= 2 * (int64_t)sw.zm_tpp) return -1;
        return (t < sw.zm_tpp ? 0 : (int64_t)sw.bnd_last * sw.zm_tpp) + t % sw.zm_tpp;
    }
    if (MODE == 1 || (MODE == -1 && sw.zm_tpp > 0)) {
        // XCD label c owns zm_tpp/8 adjacent columns, so the +-sdx lines a column needs were fetched
        // by a neighbour on the same XCD one step earlier (L2 hit); logical plane k = pl0 + seg*pps + i
        const int cpx = (sw.zm_tpp + 7) >> 3, c = b & 7, s = b >> 3;
        const int64_t col = c * cpx + s % cpx, seg = s / cpx;
        const int64_t pl = seg * sw.zm_pps + i;
        if (col >= sw.zm_tpp || i >= sw.zm_pps || (sw.zm_npl > 0 && pl >= sw.zm_npl)) return -1;
        const int64_t t = (sw.zm_pl0 + (sw.zm_plstep > 1 ? pl * sw.zm_plstep : pl)) * sw.zm_tpp + col;
        return t < sw.ntiles ? ec3d_phys_tile(sw, t) : -1;
    }
    int64_t t;
    if (sw.S > 0) {
        int64_t c = b & 7, s = b >> 3;
        t = (i * 8 + c) * sw.S + s;
    } else {
        t = i * (int64_t)sw.nblk + b;
    }
    return t < sw.ntiles ? ec3d_phys_tile(sw, t) : -1;
}
#define EC3D_XD_MAX 4 // deepest deferral of the X update (iterations whose P and S are kept)
struct SolverState {
    double rr0[2]; // R·R0 entering iteration it is rr0[it & 1]
    double alpha, omega;
    double bnorm, tol;
    double rnorm;  // ||R|| of the last iteration that got as far as its K5 (what the itmax exit prints, solvers.f90:27)
    int stop_iter; // INT_MAX while running; iteration at which an exit was taken
    int stop_kind; // 1: ‖S‖ exit (solvers.f90:34-38), 2: ‖R‖ exit (:43), 0: none / ‖b‖ = 0
    int restarts;  // times the restart R0 = R, P = R (solvers.f90:47-49) fired in this solve (ec3d_get_restart_count)
    int pad_;
    // X updates not applied yet (k4d_x_r_update): alpha, omega of the pending iterations, oldest first
    // (two groups' worth: with the groups applied by a kernel of their own beside the iteration -- ec3d_xasync -- the
    // entry of iteration it is it % (2 D), and group g + 1 fills its half while group g's kernel still reads the other)
    double pend_alpha[2 * EC3D_XD_MAX], pend_omega[2 * EC3D_XD_MAX];
    int npend;     // how many
    int pend_half; // the newest one is the ||S|| exit's X = X + alpha*P (no omega*S term)
};

// host-side image of the format (CSR conversion / export)
struct HostMatrix {
    int64_t n = 0, n_pad = 0, nnz = 0;
    int nb = 0;
    int64_t off[EC3D_MAXB] = {0};
    std::vector<double> bands;      // nb * n_pad
    std::vector<int32_t> tail_id;   // n_pad
    std::vector<uint8_t> tile_flag; // n_pad / EC3D_TILE
    int64_t ntail = 0;
    std::vector<int64_t> chunk_ptr;
    std::vector<int32_t> tcol;
    std::vector<double> tval;
    // dictionary form (ncls > 0): `bands` may then be empty
    std::vector<uint8_t> cls;
    std::vector<double> table;
    int ncls = 0;
};

// host-side image of the structured A-V form recognised in a CSR matrix (ec3d_sav_csr.cpp)
struct SavHost {
    int64_t n_ref = 0, n_dev = 0, n_pad = 0, nnz = 0;
    int64_t sdx = 0, plane = 0, pitch = 0, nCd = 0;
    int a0 = 0, u0 = 0, zero = 0, ncls = 0;
    std::vector<uint8_t> cls, tile_flag;
    std::vector<double> table; // ncls * 16
    std::vector<int32_t> ulist, cond_cell;
    int64_t ntiles_front = 0;
    // a z-slab cut out of a recognised system (ec3d_sav_slice): rows of the halo planes are inert and only the
    // owned planes of each block count in the dot products (as ec3d_assemble_slab sets it up natively)
    int nown = 0;
    int64_t own_lo[4] = {0, 0, 0, 0}, own_hi[4] = {0, 0, 0, 0};
    int64_t halo = 0;
    int64_t u_first = 0; // index, among the whole system's U unknowns, of the slab's first held one
};

struct DevMatrix {
    int64_t n = 0, n_pad = 0, nnz = 0;
    int nb = 0;
    int64_t off[EC3D_MAXB] = {0};
    double *bands = nullptr;
    int32_t *tail_id = nullptr;
    uint8_t *tile_flag = nullptr;
    int64_t ntail = 0, nchunk = 0, tail_entries = 0;
    int64_t *chunk_ptr = nullptr;
    int32_t *tcol = nullptr;
    double *tval = nullptr;
    uint8_t *cls = nullptr; // dictionary form (ncls > 0): bands == nullptr
    double *table = nullptr;
    int ncls = 0;
    // structured A-V form (see MatView)
    int sav = 0, sav_a0 = 0, sav_u0 = 0, sav_zero = 0;
    int32_t *ulist = nullptr; // tiles of the U block that hold at least one unknown
    int ulist_n = 0;
    std::vector<int32_t> ulist_host; // host copy for the visit-order export
    // the same for runtime-shaped 2-D tiles (Sweep::rp_*), made by choose_sweep for the shape it picked
    int rp_px = 0, rp_py = 0;
    uint8_t *rp_flag = nullptr;            // per patch tile of the three A blocks: coupled
    std::vector<int32_t> rp_ulist_host;    // patch tiles of the U block that hold an unknown, ascending
    int64_t ntiles_front = 0; // tiles swept unconditionally
    int64_t sav_nC = 0, sav_step[3] = {0, 0, 0};
    int64_t bytes = 0;
    MatView view() const;
};

struct ec3d_ctx {
    int device = 0;
    hipStream_t stream = nullptr;         // the stream every launch goes to
    hipStream_t own_stream_obj = nullptr; // created by ec3d_create; `stream` may point elsewhere
    DevMatrix A;
    bool have_matrix = false;
    int64_t ghost = 0;     // zero halo (doubles) on both sides of every vector
    double *vec_base = nullptr;
    double *vec[8] = {nullptr};
    Sweep sweep{};   // vector kernels: K4's grid (and the geometry every other sweep is derived from)
    Sweep sweep_k2{}, sweep_k5{}; // K2 (2 reads + 1 write) and K5 (3 + 1) like other workgroup counts than K4 (5 + 2)
    Sweep sweep_s{}; // SpMV kernels (K1, K3, residual, spmv)
    Sweep sweep_int{}, sweep_bnd{}; // z-slab: interior / boundary-plane launches of K1 and K3
    bool can_overlap = false;
    // z-slab on 2-D tiles: planes {0, np-1} / 1 .. np-2 as two launches of the 2-D-tile kernels (K4 in SpMV form and
    // K5-in-K1 of the three-launch iteration: the producers of the exchanged R and AP); can_fsplit: np >= 3
    Sweep sweep_fb{}, sweep_fi{};
    bool can_fsplit = false;
    bool fuse23_ok = false; // 2-D tiles: K2 may run inside K3 (single rank only, see ec3d_fused23)
    bool fuse51_ok = false; // 2-D tiles: K5 may run inside the next iteration's K1 (ec3d_fused51)
    bool k4s_ok = false;    // dictionary cube on 2-D tiles: K4 may run as an SpMV kernel that computes A S again (ec3d_k4s)
    // z-slab of a multi-rank job (set by the job's driver once EVERY rank can do it: the exchanges differ, so the plan is a
    // property of the job): the three-launch iteration on this slab (AP and R are exchanged instead of P and S, see
    // Sweep::halo_store); the X update deferred over slab_xd iterations (0: not on this slab)
    bool slab_fused = false;
    int slab_xd = 0;
    // P(it) lives in pbuf[(it + p_off) % pdepth] (0: iterations are numbered from the last ec3d_launch_begin)
    int p_off = 0;
    // The iteration the next launch has to be: the device state is addressed by the iteration number (rr0[it & 1], AP in
    // apbuf[it & 1], P and S in their rings), so ec3d_iterate / ec3d_multi_iterate must continue where the last call ended
    int it_next = 1;
    // K5-in-K1 reads the previous iteration's P and AP while it writes the new ones (neighbouring workgroups read the
    // old values of cells this one owns), so both vectors alternate between two buffers: P(it) lives in
    // pbuf[it & 1], AP(it) in apbuf[it & 1]; index 1 is vec[EC3D_VEC_P] / vec[EC3D_VEC_AP], index 0 the spare pair
    // With the X update deferred over D iterations (k4d_x_r_update; only together with both fusions) P(it) lives in
    // pbuf[it % D] and S(it) in sbuf[it % D] (index 1 = vec[EC3D_VEC_P] / vec[EC3D_VEC_S]); AP keeps its two buffers.
    double *pp_base = nullptr;
    int64_t pp_len = 0;    // doubles allocated at pp_base
    double *pbuf[2 * EC3D_XD_MAX] = {nullptr}, *sbuf[2 * EC3D_XD_MAX] = {nullptr}, *apbuf[2] = {nullptr, nullptr};
    int pdepth = 2;        // buffers P cycles through (2, or D; 2 D with the X groups on a stream of their own)
    int sdepth = 1;        // buffers S cycles through (D; 2 D with the X groups on a stream of their own)
    int ring_cap = 0;      // ring buffers allocated for each of P and S (ec3d_spare_pair): D, or 2 D
    // The groups of D pending X updates applied by a kernel of their own (k_x_group) on a second, low-priority stream,
    // beside the following iterations, instead of by every D-th K4: nothing in the loop reads X, so the work fills what
    // the iteration leaves idle -- on a z-slab the waits for halo planes and reduced sums (ec3d_xasync)
    bool xasync_cap = false;  // the rings hold two groups (2 D buffers each)
    bool xasync_forced = false; // EC3D_XASYNC=2: also on a handle that is no slab (tests)
    bool slab_xasync = false; // a z-slab: the job's driver said so (every rank the same ring depth)
    // ... or as launches of their own on the iteration's OWN stream, behind the K4 of each group's last iteration (rings of one
    // group suffice): the undivided handle's three-launch iteration, where K4 in SpMV form with ten more operand streams in
    // the applying launch ran 18 % over what its bytes allow and a light K4 + one streaming launch per group do not
    bool xinline = false;
    hipStream_t xstream = nullptr;
    hipEvent_t ev_xready = nullptr, ev_xdone[2] = {nullptr, nullptr};
    // first failed runtime call of a launcher that cannot return a status (ec3d_launch_x_group_of sits inside the void stage
    // launchers): noted there, turned into an error code by whoever checks the stage's launches (EC3D_ASYNC_CHECK)
    hipError_t async_err = hipSuccess;
    const char *async_what = nullptr;
    int ss_parts = 0;      // z-slab: workgroup partials of S.S that K2 left for K3's collapse launch to fold (0: none pending)
    int xg_n = 0;          // groups launched since the last ec3d_launch_begin
    int xg_done_upto = 0;  // the last iteration whose X update an enqueued k_x_group covers
    int xdefer = 1;        // D: iterations between two X updates on this handle (1: every iteration, the classic K4)
    int xd_base = 1;       // the iteration the groups of D are counted from (1 in a solve; ec3d_iterate: its first_iter)
    int xd_last = 0x7fffffff; // the last iteration the present call is going to launch: it applies whatever is pending
    int pcur = 1;          // which P buffer holds the CURRENT P (what ec3d_download and ec3d_device_vector hand out)
    int apcur = 1, scur = 1; // the same for AP and S
    int ap_valid_for = 0;  // K5-in-K1: the iteration whose AP = A P the last K51 launch already produced (0: none)
    // K2/K5 as boundary + interior launches (ec3d_dist_set_boundary_rows): tile lists on the device
    Sweep sweep_vb{}, sweep_vi{};
    int32_t *vb_list = nullptr, *vi_list = nullptr;
    int32_t *us_list = nullptr; // structured form: the U tiles in the order the z-marching SpMV kernels take them (choose_sweep)
    int32_t *ii_list = nullptr, *ib_list = nullptr; // structured z-slab, K1 / K3 split: the interior launch's U tiles, the boundary launch's tiles
    std::vector<int32_t> us_host; // host copy (visit-order export)
    uint32_t *il_umask = nullptr;      // interleaved z-march of the structured form (Sweep::il_*): one bit per (column, plane)
    std::vector<uint32_t> il_umask_host;
    int32_t *il_seg = nullptr;         // ... and its work list (four int32 per workgroup)
    std::vector<int32_t> il_seg_host;
    bool can_vsplit = false;
    int nown = 0;    // ownership ranges of an A-V slab (see Sweep)
    int64_t own_lo[4] = {0}, own_hi[4] = {0};
    bool own_vectors = true;
    bool dist = false;
    // multi-rank (z-slab) mode: reductions come from the all-gathered per-rank sums
    int nranks = 1;
    double *lsum = nullptr, *gsum = nullptr; // caller-owned device buffers (P_NSLOT, nranks*P_NSLOT)
    const double *const *lsum_ptrs = nullptr; // device array of nranks pointers: every rank's lsum (ec3d_multi.hip)
    int64_t halo = 0;                        // doubles per halo plane (kdz), 0 when not a slab
    bool use_dict = true;
    bool use_sav = true;   // structured A-V form for ec3d_assemble when the problem allows it (EC3D_SAV)
    int64_t n_ref = 0;     // unknowns in the reference's numbering (what host vectors hold); = A.n unless sav
    // sav: every xy plane starts on a tile boundary (pitch >= plane cells, multiple of EC3D_TILE) so the
    // z-marching SpMV map always applies; plane == pitch == 0 otherwise.  Device cell of reference cell q:
    // (q / plane) * pitch + q % plane; a component block holds nCd = sdz * pitch device rows.
    int64_t plane = 0, pitch = 0, nCd = 0;
    int64_t dev_cell(int64_t q) const { return pitch == plane ? q : (q / plane) * pitch + q % plane; }
    int64_t ref_cell(int64_t p) const { return pitch == plane ? p : (p / pitch) * plane + p % pitch; }
    int64_t planes() const { return pitch ? nCd / pitch : 0; } // xy planes per component block
    double *io_tmp = nullptr; // sav: staging for the U part of host<->device vector copies
    int nblk_request = 0;
    int nt_request = -1; // -1 auto, 0/1 forced (EC3D_NT)
    int zm_request = 1;  // z-marching SpMV map when the grid allows it (EC3D_ZMARCH)
    int shuffle_request = 1; // +-1 neighbours by lane shuffle (EC3D_SHUFFLE)
    double *partials = nullptr; // 8 * nblk doubles
    SolverState *state = nullptr;
    SolverState *state_pinned = nullptr; // 2 slots
    double *hist = nullptr;
    int64_t hist_cap = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // assembly by-products (1-based ids, reference order)
    std::vector<int32_t> cel_bnd[6];
    // grid of the last native assembly (0 when the matrix came from CSR)
    int32_t sdx = 0, sdy = 0, sdz = 0;
    int64_t n_cells = 0; // cells this handle holds per component (a z-slab: its extended planes only)
    int32_t slab_e0 = 0, slab_k0 = 0, slab_k1 = 0; // global planes: first held, owned range [k0, k1)
    // per-step RHS build / post-update on the device (src/EC3D.f90:370-404, :412-433)
    int64_t n_cond = 0;            // conducting cells (U unknowns), scan order
    int n_cond_domains = 0;
    int32_t *cond_cell = nullptr;  // [n_cond] 0-based DEVICE cell index (dev_cell)
    double *cond_a = nullptr;      // [n_cond] 2*C/dt of the cell's domain (PHYS_C%valdom)
    int32_t *bnd_list = nullptr;   // the six cel_bnd* lists, 0-based unknown ids, concatenated
    int64_t bnd_off[7] = {0};
    double *rhs_tmp = nullptr;     // [3*n_cond] scratch for the moving-source reset
    int32_t *src_idx = nullptr;    // per-step source scatter staging
    double *src_val = nullptr;
    int64_t src_cap = 0;
    // plain band streams whose placement was probed (place_bands): kept across a change of matrix of the same size, so
    // the probe runs once per handle and size, and what it found (candidate times in us, the one kept)
    double *placed_bands = nullptr;
    size_t placed_bytes = 0;
    bool bands_placed = false;
    std::vector<float> place_us;
    int place_kept = -1;
    // the work vectors' placement (place_vectors): doubles per vector the probe last ran for (0: never), the candidates'
    // iteration times, which one was kept, what the search cost
    int64_t vplace_len = 0;
    double *parked_vec = nullptr, *parked_pp = nullptr; // the chosen allocation between two matrices (ec3d_free_matrix)
    int64_t parked_pp_len = 0;
    std::vector<float> vplace_us;
    int vplace_kept = -1;
    float vplace_ms = 0.f;
    std::vector<uint64_t> src_seen; // host: one bit per A unknown, all zero between calls (repeat check of ec3d_rhs_step)
    // field output (ec3d_output.hip): device scratch for the four float32 vectors, the conductor mask, and -- for
    // output overlapped with the next time step -- a side stream with two pinned host buffers
    float *out_dev = nullptr;
    int64_t out_cells = 0;
    int32_t *out_mask = nullptr;
    float *out_pinned[EC3D_OUT_SLOTS] = {nullptr, nullptr, nullptr};
    hipStream_t out_stream = nullptr;
    hipEvent_t out_ev_fields = nullptr, out_ev_free = nullptr, out_ev_copied[EC3D_OUT_SLOTS] = {nullptr, nullptr, nullptr};
    int out_next = 0;
    bool out_busy = false;
    unsigned out_started = 0;  // bit i: slot i has had an ec3d_vtk_fields_begin (its event is worth waiting for)
};

// partial-sum slots inside ctx->partials (each nblk doubles)
enum { P_BB = 0, P_RR_INIT = 1, P_D1 = 2, P_SS = 3, P_D2 = 4, P_D3 = 5, P_RR = 6, P_RR0N = 7, P_NSLOT = 8 };

void ec3d_set_error(const std::string &msg);
// a REAL(8) as the reference's `print*` writes it (src/solvers.f90:27, flang's list-directed output); buf >= 40
void ec3d_format_list_directed(double v, char *buf);
// the same value as gfortran's list-directed output writes it (one blank + G25.17E3); buf >= 40
void ec3d_format_list_directed_gfortran(double v, char *buf);
// the itmax exit's line: norm2(R) on stdout, as the reference prints it (src/solvers.f90:27, `print*`).  List-directed
// output is compiler specific: flang's form by default (the toolchain of this image, which built oracle/_ref),
// EC3D_PRINT_STYLE=gfortran for the form of the reference's own Makefile (src/Makefile:1-28)
void ec3d_print_rnorm(double rnorm);
// itmax exit: the reference prints norm2(R) (src/solvers.f90:25-28).  When this points somewhere, the solve entry
// points store the value there instead of printing it (the drop-in prints it once its result is accepted).
extern thread_local double *ec3d_itmax_print_hold;
#define EC3D_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ec3d_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                 \
            return 100;                                                                        \
        }                                                                                      \
    } while (0)
// a runtime call inside a launcher without a status of its own: the first failure is kept in the context ...
#define EC3D_NOTE(c_, call)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess && (c_)->async_err == hipSuccess) {                               \
            (c_)->async_err = e_;                                                              \
            (c_)->async_what = #call;                                                          \
        }                                                                                      \
    } while (0)
// ... and reported where the stage's launches are checked
#define EC3D_ASYNC_CHECK(c_)                                                                   \
    do {                                                                                       \
        if ((c_)->async_err != hipSuccess) {                                                   \
            ec3d_set_error(std::string((c_)->async_what) + ": " + hipGetErrorString((c_)->async_err)); \
            (c_)->async_err = hipSuccess;                                                      \
            return 100;                                                                        \
        }                                                                                      \
    } while (0)

// scratch device memory released at scope exit, error returns included
template <class T> struct DevTmp {
    T *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, count * sizeof(T)); }
    operator T *() const { return p; }
};

// ec3d_format.cpp
int ec3d_csr_to_host_matrix(int64_t n, const double *valA, const int32_t *irow, const int32_t *jcol,
                            HostMatrix &M);
void ec3d_host_matrix_to_csr(const HostMatrix &M, std::vector<int32_t> &irow, std::vector<int32_t> &jcol,
                             std::vector<double> &valA);

// ec3d_sav_csr.cpp: 0, or -1 when the matrix does not have the structure
int ec3d_csr_to_sav_host(int64_t n, const double *valA, const int32_t *irow, const int32_t *jcol, SavHost &S);
// Can a recognised system be cut into `nranks` z-slabs?  0, or 2 (fewer than two planes per rank) / 7 (something
// couples across the z faces of a component: not the reference's system on a box) with the reason in `why`.
// seven bands at (-kdz, -sdx, -1, 0, 1, sdx, kdz), no tail, n a whole number of planes of kdz rows (kdz a whole
// number of sdx lines): a single-component 7-point operator on a grid, which can be cut along z plane by plane
bool ec3d_host_matrix_is_cube(const HostMatrix &M, int64_t &sdx, int64_t &kdz);
int ec3d_sav_cuttable(const SavHost &G, int nranks, std::string &why);
// planes [e0, e1) of a recognised system as a slab that owns [k0, k1) (same pitch, classes and table)
void ec3d_sav_slice(const SavHost &G, int64_t e0, int64_t e1, int64_t k0, int64_t k1, SavHost &L);

// ec3d_context.hip
int ec3d_upload_sav(ec3d_ctx *c, const SavHost &S);
// halo > 0: the matrix is a z-slab of a single-component operator whose planes hold `halo` rows (the ghost zones of
// the vectors then carry the neighbours' planes, ec3d_assemble_poisson_slab does the same natively)
int ec3d_upload_matrix(ec3d_ctx *c, const HostMatrix &M, int64_t halo = 0);
int ec3d_download_matrix(ec3d_ctx *c, HostMatrix &M);
void ec3d_free_matrix(ec3d_ctx *c);
int ec3d_prepare_vectors(ec3d_ctx *c);
// device memory for the plain band streams: the allocation a placement probe chose earlier, when the size fits
int ec3d_alloc_bands(ec3d_ctx *c, double **bands, size_t bytes);
int ec3d_spare_pair(ec3d_ctx *c);
// host vector (reference numbering, n_ref entries) <-> device vector (device numbering)
int ec3d_vec_h2d(ec3d_ctx *c, double *dev, const double *host);
int ec3d_vec_d2h(ec3d_ctx *c, double *host, const double *dev);

// ec3d_context.hip
int ec3d_need_matrix(ec3d_ctx *c, const char *who); // 0, or 3 + error text when the handle has no matrix
// ec3d_solve.hip: where a consumer kernel finds its sums, and the launches of one iteration
// who produced the partial sums a consumer needs: an SpMV-type stage (slots BB, RR_INIT, D1, D2, D3), K2 (SS)
// or K4 (RR, RR0N)
enum { EC3D_BY_K4 = 0, EC3D_BY_SPMV = 1, EC3D_BY_K2 = 2 };
RedSrc ec3d_src_of(const ec3d_ctx *c, int producer);
RedSrc ec3d_part_of(const ec3d_ctx *c, int producer, bool split = false);
// k = 1..5, 0 = all five; part: 0 the whole launch, 1 / 2 the boundary / interior launch of stage 4 or 5 on a slab that runs
// the three-launch iteration (sweep_fb / sweep_fi)
void ec3d_launch_stage(ec3d_ctx *c, const MatView &A, int it, int k, int part = 0);
inline bool ec3d_fused23(const ec3d_ctx *c) { return c->fuse23_ok && ((!c->dist && c->halo == 0) || c->slab_fused); }
inline bool ec3d_fused51(const ec3d_ctx *c)
{
    return c->fuse51_ok && ((!c->dist && c->halo == 0) || c->slab_fused) && c->pp_base != nullptr;
}
// K4 in SpMV form (k4s_x_r_spmv): only inside the three-launch iteration
inline bool ec3d_k4s(const ec3d_ctx *c) { return c->k4s_ok && ec3d_fused23(c) && ec3d_fused51(c); }
// deferred X update: single rank, own vectors (the rings exist), and the iteration either fully fused or not at all
inline int ec3d_xdefer(const ec3d_ctx *c)
{
    if (c->xdefer <= 1 || c->pp_base == nullptr) return 1;
    if (c->dist || c->halo != 0) { // a z-slab: only when the job's driver said so (the exchanged P / S then live in the rings)
        if (c->slab_xd <= 1) return 1;
        return (ec3d_fused23(c) == ec3d_fused51(c)) ? std::min(c->xdefer, c->slab_xd) : 1;
    }
    return (ec3d_fused23(c) == ec3d_fused51(c)) ? c->xdefer : 1;
}
// the pending X updates applied group by group on a stream of their own (only with the X update deferred)
inline bool ec3d_xasync(const ec3d_ctx *c)
{
    if (ec3d_xdefer(c) <= 1) return false;
    if (c->xinline && !c->dist && c->halo == 0) return true;
    if (!c->xasync_cap) return false;
    return (c->dist || c->halo != 0) ? c->slab_xasync : c->xasync_forced;
}
// where vector `vec` (EC3D_VEC_P / _AP / _S; anything else: the plain work vector) of iteration `it` lives on this handle
double *ec3d_vec_at(const ec3d_ctx *c, int vec, int it);
int ec3d_flush_x(ec3d_ctx *c, int stop_iter); // the pending X updates after an exit at stop_iter (enqueued)
bool ec3d_dist_can_split_planes(const ec3d_ctx *c);
void ec3d_launch_x_group_of(ec3d_ctx *c, int first, int count, bool join);
void ec3d_xgroups_reset(ec3d_ctx *c);
void ec3d_launch_iteration(ec3d_ctx *c, const MatView &A, int it);
int ec3d_launch_begin(ec3d_ctx *c, const MatView &A, double tol);
int ec3d_single_rank_only(ec3d_ctx *c, const char *who);
// K2 / K5 of a single-component z-slab as boundary-plane + interior launch (window sweeps, no tile lists); ec3d_dist.hip
int ec3d_dist_set_boundary_planes(ec3d_ctx *c, int32_t *enabled);
int ec3d_dist_launches(const ec3d_ctx *c, int stage, int it); // ec3d_dist.hip (call BEFORE the stage is launched)

// ec3d_kernels.hip — launchers (all asynchronous on `s`)
void ec3d_launch_spmv(const MatView &A, const Sweep &sw, const double *x, double *y, hipStream_t s);
void ec3d_launch_residual(const MatView &A, const Sweep &sw, const double *x, const double *b, double *r,
                          double *r0, double *p, double *part, hipStream_t s);
void ec3d_launch_finalize(const RedSrc &src, double *lsum, unsigned mask, hipStream_t s);
void ec3d_launch_finalize2(const RedSrc &a, unsigned mask_a, const RedSrc &b, unsigned mask_b, double *lsum, hipStream_t s);
void ec3d_launch_setup(SolverState *st, const RedSrc &src, double tol, hipStream_t s);
void ec3d_launch_k1(const MatView &A, const Sweep &sw, const SolverState *st, int it, const double *p,
                    const double *r0, double *ap, double *part, hipStream_t s);
void ec3d_launch_k2(const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r, const double *ap,
                    double *sv, double *part, hipStream_t s);
void ec3d_launch_k3(const MatView &A, const Sweep &sw, SolverState *st, int it, const double *sv, double *as,
                    double *part, hipStream_t s);
void ec3d_launch_k23(const MatView &A, const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r,
                     const double *ap, double *sv, double *as, double *part, hipStream_t s);
void ec3d_launch_k51(const MatView &A, const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r,
                     const double *p_old, const double *ap_old, double *p_new, double *ap_new, double *r0, double *part,
                     double *hist, int64_t hist_cap, hipStream_t s);
void ec3d_launch_k4d(const Sweep &sw, const RedSrc &src_ss, const RedSrc &src, SolverState *st, int it, int ne, int xm,
                     const double *const *p, const double *const *sv, const double *as, const double *r0, double *x,
                     double *r, double *part, double *hist, int64_t hist_cap, hipStream_t s);
void ec3d_launch_k4s(const MatView &A, const Sweep &sw, const RedSrc &src_ss, const RedSrc &src, SolverState *st, int it,
                     int ne, int xm, const double *const *p, const double *const *sv, const double *r0, double *x, double *r,
                     double *part, double *hist, int64_t hist_cap, hipStream_t s);
void ec3d_launch_x_flush(const Sweep &sw, const SolverState *st, const double *const *p, const double *const *sv,
                         double *x, hipStream_t s);
void ec3d_launch_x_group(const Sweep &sw, const SolverState *st, const double *const *p, const double *const *sv, int first,
                         int count, int d2, double *x, int nblk, hipStream_t s);
void ec3d_launch_k4(const Sweep &sw, const RedSrc &src_ss, const RedSrc &src, SolverState *st, int it,
                    const double *p, const double *sv, const double *as, const double *r0, double *x, double *r,
                    double *part, double *hist, int64_t hist_cap, hipStream_t s);
void ec3d_launch_k5(const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r, const double *ap,
                    const double *p_old, double *p, double *r0, double *hist, int64_t hist_cap, hipStream_t s);

// ec3d_assemble.hip
// planes [e0, e1) of the global grid are held (e0 = 0, e1 = sdz: everything); rows of planes outside
// [k0, k1) are inert (zero coefficients): they only carry the neighbours' values
int ec3d_assemble_device(ec3d_ctx *c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t e0, int32_t e1, int32_t k0,
                         int32_t k1, const int8_t *geoPHYS, const int32_t *geoPHYS_C, const double *valPHYS,
                         int32_t nsub_glob, const double *BND, const double *delta, double dt);
// same plane arguments as ec3d_assemble_device; returns -1 when the structured form does not apply
int ec3d_assemble_sav_device(ec3d_ctx *c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t e0, int32_t e1, int32_t k0,
                             int32_t k1, const int8_t *geoPHYS, const int32_t *geoPHYS_C, const double *valPHYS,
                             int32_t nsub_glob, const double *BND, const double *delta, double dt);
int ec3d_assemble_poisson_device(ec3d_ctx *c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t k0, int32_t k1,
                                 const double *BND, const double *delta);
// ec3d_format.cpp / ec3d_context.hip: dictionary compression of the bands
int ec3d_build_dictionary_host(HostMatrix &M);
// ec3d_output.hip
void ec3d_free_output(ec3d_ctx *c);
// ec3d_rhs.hip
void ec3d_free_rhs(ec3d_ctx *c);
int ec3d_setup_rhs(ec3d_ctx *c, int64_t nCells, const int8_t *geoPHYS, const int32_t *geoPHYS_C,
                   const double *valPHYS, int32_t nsub_glob, double dt);
