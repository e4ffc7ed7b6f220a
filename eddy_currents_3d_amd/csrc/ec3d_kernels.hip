// ec3d_kernels.hip — gfx950 kernels of the BiCGSTAB-with-restart hot path.
//
// Reference arithmetic restated (src/solvers.f90:3-61, see the per-kernel notes) with these
// MI355X-first choices:
//   * matrix = one class byte per row + a coefficient table in LDS (dictionary / structured A-V form), or DIA bands
//     (SoA, one fp64 stream per band) + sliced-ELL tail; no column indices on the band part, every stream is read
//     with 16-byte accesses, one tile = 512 rows -- consecutive ones, or a 128 x 4 patch of the xy plane whose
//     +-sdx neighbours travel through LDS (patch_pair);
//   * one iteration = 5 launches (K1..K5), every vector op fused into the kernel that produces its operand, every
//     dot product into the kernel that produces its vector -- or 3 launches on large single-rank problems, where
//     S and P are formed inside the SpMV kernels that read them (k23_s_spmv_dots, k51_p_spmv_dot);
//   * reductions are deterministic: per-thread sequential over its tiles, 64-lane shuffle tree,
//     4 wave sums left to right, one partial per workgroup; the NEXT kernel's workgroups each
//     re-reduce the partials in the same order (a few KB from L2), so there is no atomics, no
//     inter-workgroup handshake and no host round trip; scalars (alpha, omega, beta, rr0) live
//     in a device-resident SolverState;
//   * convergence is decided on the device: an exit writes stop_iter, later launches become
//     no-ops, the host polls asynchronously (ec3d_solve.hip);
//   * the SpMV kernels march in z (one xy position per workgroup, the planes below and at the row in registers);
//     their columns are dealt to the 8 XCD labels (blockIdx % 8) in contiguous runs, so what a neighbour column
//     fetched is in the same XCD's L2; the vector kernels take the grid, tile order and batching depth measured
//     best for each (choose_sweep in ec3d_context.hip).
// Built with -ffp-contract=off: products and sums are rounded separately, exactly as the
// reference's x86-64 object code does; the oracle's "GPU order" twin reproduces every bit.
#include "ec3d_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef int i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
    return v; // lane 0
}

// sums NV values over the workgroup; result valid in every thread.  lds: NV*4 doubles.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *lds)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double t = wave_sum(v[k]);
        if (lane == 0) lds[k * 4 + w] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = ((lds[k * 4 + 0] + lds[k * 4 + 1]) + lds[k * 4 + 2]) + lds[k * 4 + 3];
    __syncthreads();
}

// every workgroup re-reduces the producer's partial sums, same order everywhere.  Single GPU:
// the previous kernel's per-workgroup partials; multi rank: the all-gathered per-rank sums.
template <int NV>
__device__ __forceinline__ void reduce_partials(const RedSrc &src, const int (&slot)[NV], double (&out)[NV],
                                                double *lds)
{
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double a = 0.0;
        if (src.ptrs) { // one value per rank, each in that rank's own memory
            for (int i = threadIdx.x; i < src.count; i += EC3D_THREADS) a = a + src.ptrs[i][slot[k]];
        } else {
            const double *p = src.base + (int64_t)slot[k] * src.slot_mul;
            for (int i = threadIdx.x; i < src.count; i += EC3D_THREADS) a = a + p[(int64_t)i * src.stride];
        }
        out[k] = a;
    }
    block_sum<NV>(out, lds);
}

// The same in two halves, so that a kernel can REQUEST the producer's partials together with everything else it
// needs before its first tile -- the exit word, the scalars of SolverState, the class table -- and wait once: each
// of these used to be a dependent memory round trip of its own (exit word -> partials -> scalars -> table -> first
// tile), which is what a launch costs on a problem that fits the caches (the reference's shipped inputs: 0.4-0.8 M
// unknowns, where a kernel's whole sweep is one or two more round trips).  partials_request issues the loads of
// the first EC3D_PMAX values a thread adds (1536 partials cover every default grid) and uses none of them;
// partials_finish adds them in the order of reduce_partials -- the same sums -- and loads what is left, if anything.
#define EC3D_PMAX 6
template <int NV> struct PartialsEarly { double v[NV][EC3D_PMAX]; };
template <int NV>
__device__ __forceinline__ void partials_request(const RedSrc &src, const int (&slot)[NV], PartialsEarly<NV> &e)
{
    // branch free: an index beyond the count reads entry 0 (always there) and its value is dropped -- a load behind a
    // branch of its own is waited for at the end of that branch, and the requests would go out one by one again
    if (src.ptrs) { // one value per rank, each in that rank's own memory
#pragma unroll
        for (int j = 0; j < EC3D_PMAX; ++j) {
            const int i = (int)threadIdx.x + j * EC3D_THREADS;
            const double *q = src.ptrs[i < src.count ? i : 0];
#pragma unroll
            for (int k = 0; k < NV; ++k) e.v[k][j] = q[slot[k]];
        }
    } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double *p = src.base + (int64_t)slot[k] * src.slot_mul;
#pragma unroll
            for (int j = 0; j < EC3D_PMAX; ++j) {
                const int i = (int)threadIdx.x + j * EC3D_THREADS;
                e.v[k][j] = p[(int64_t)(i < src.count ? i : 0) * src.stride];
            }
        }
    }
}
template <int NV>
__device__ __forceinline__ void partials_finish(const RedSrc &src, const int (&slot)[NV], const PartialsEarly<NV> &e,
                                                double (&out)[NV], double *lds)
{
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < EC3D_PMAX; ++j) // (a value beyond the count enters as +0.0, which changes no sum)
            a = a + ((int)threadIdx.x + j * EC3D_THREADS < src.count ? e.v[k][j] : 0.0);
        for (int i = (int)threadIdx.x + EC3D_PMAX * EC3D_THREADS; i < src.count; i += EC3D_THREADS) {
            if (src.ptrs) a = a + src.ptrs[i][slot[k]];
            else a = a + src.base[(int64_t)slot[k] * src.slot_mul + (int64_t)i * src.stride];
        }
        out[k] = a;
    }
    block_sum<NV>(out, lds);
}

// A producer's workgroup leaves its partial sums: one per slot, at the launch's offset within the slot.  The consumer kernel
// re-reduces them (single rank), or a one-workgroup k_finalize launch collapses them into the rank's eight sums (multi rank).
// (Round 5 built the collapse INTO the producers -- the last workgroup to arrive, found through per-blockIdx % 8 arrival
// counters, folding the partials in k_finalize's order; write-through partials and agent-scope loads, no cache flushed --
// bit-identical and no faster than the four small launches it replaced: the serial tail behind the last workgroup costs what a
// launch costs.  Removed; profiles/r05_partial_sum_collapse_in_kernel_vs_launch.log, commit "Experiment: partial sums
// collapsed by the producers' last workgroup".)
template <int NV, class SW>
__device__ __forceinline__ void publish_partials(const SW &sw, double *part, const int (&slot)[NV], const double (&acc)[NV],
                                                 double *)
{
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) part[slot[k] * sw.pstride + sw.part_off + blockIdx.x] = acc[k];
    }
}

// Streams that are touched once per launch use the nontemporal (streaming) cache policy when the
// vectors are far larger than the 256 MiB Infinity Cache (NT = true): on a 512^3 grid that is worth
// +10..25 % on the pure vector stages (tools/stream_bench.hip).  Small problems keep the default
// policy so consecutive kernels find their operands in L2 / Infinity Cache.
template <bool NT>
__device__ __forceinline__ d2 load2(const double *__restrict__ p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
    return *reinterpret_cast<const d2 *>(p);
}


// ---------------------------------------------------------------------------------------------
// What a kernel receives.  MatView / Sweep (ec3d_internal.hpp) describe a matrix and a launch for every format
// and every map; passed by value they kept ~60 scalar registers busy in kernels that read a dozen of them, and the
// SpMV kernels spilled 15-34 of them to vector lanes.  Each template instance now gets a view holding only the
// fields it reads (made from MatView / Sweep by the launchers at the end of this file).
struct TailDev {
    const int32_t *tail_id;   // [n_pad]  -1 or index of the row's tail slot
    const uint8_t *tile_flag; // [ntiles] 1 when any row of the tile has a tail
    const int64_t *chunk_ptr; // [nchunk+1] entry offsets of the 64-row slices
    const int32_t *tcol;
    const double *tval;
};
// Matrix formats the row kernel is specialised for (template parameter FMT):
//   FMT_GENERIC  any number of bands, one fp64 stream per band
//   FMT_DIA7     7 bands, unrolled (72 B/row: 56 coefficients + x + y)
//   FMT_DICT7    7 bands whose coefficient 7-tuples take <= 256 distinct values ("stencil classes"):
//                one class byte per row + a table staged in LDS (17 B/row: 1 + x + y).  The values
//                multiplied are the same doubles, so results are bit-identical to FMT_DIA7.
//   FMT_SAV      the structured A-V form (MatView::sav): class byte per row, U on the grid, no tail
enum { FMT_GENERIC = 0, FMT_DIA7 = 7, FMT_DICT7 = 107, FMT_SAV = 207 };
#define EC3D_SAV_STRIDE 16

template <int FMT> struct MatDev;
template <> struct MatDev<FMT_GENERIC> {
    const double *band[EC3D_MAXB];
    int64_t off[EC3D_MAXB];
    int nb;
    TailDev t;
    __device__ __forceinline__ int64_t boff(int b) const { return off[b]; }
};
template <> struct MatDev<FMT_DIA7> {
    const double *band[7];
    int64_t off[7];
    int pm1;
    TailDev t;
    __device__ __forceinline__ int64_t boff(int b) const { return off[b]; }
};
template <> struct MatDev<FMT_DICT7> {
    const uint8_t *cls;
    const double *table;
    int64_t off[7];
    int ncls, pm1;
    TailDev t;
    __device__ __forceinline__ int64_t boff(int b) const { return off[b]; }
};
template <> struct MatDev<FMT_SAV> {
    const uint8_t *cls;
    const uint8_t *tile_flag; // 1: some row of the tile is coupled (uniform per tile)
    const double *table;
    int64_t nC, sdx, pitch; // rows per block; band offsets are (-pitch, -sdx, -1, 0, 1, sdx, pitch)
    int64_t planes;         // xy planes per block = nC / pitch
    int ncls, pm1;
    int a0, u0, zero; // class ranges (MatView); only the kernels for grids without tile-aligned planes read them
    __device__ __forceinline__ int64_t boff(int b) const
    {
        return b == 0 ? -pitch : b == 1 ? -sdx : b == 2 ? -1 : b == 3 ? 0 : b == 4 ? 1 : b == 5 ? sdx : pitch;
    }
    __device__ __forceinline__ int64_t step(int d) const { return d == 0 ? 1 : d == 1 ? sdx : pitch; }
};

// the z-marching launch as its kernels see it
struct SweepZ {
    int64_t ntiles, n; // logical front tiles; rows >= n are padding
    int tpp, pps, npl, pl0;
    int plstep;           // > 1: logical plane lp of the launch is physical plane pl0 + lp * plstep (pps = 1; Sweep::zm_plstep)
    int hs_mask, hs_last; // z-slab, K2-in-K3 / K5-in-K1: store the formed vector on the halo planes (Sweep::halo_store);
                          // hs_last = the slab's last owned plane
    int pstride, part_off;
    int ulist_n;
    const int32_t *ulist;
    int64_t win_npo, win_npb, win_p0; // window in planes (owned per block, held per block, first owned); npo = 0: none
    int patch_npx;                    // 2-D tiles (Sweep::patch_npx); zm_tpp doubles as tpp for ec3d_row_of
    int64_t patch_sdx;
    int zm_tpp;
    int keep; // producers whose output stays cacheable (EC3D_KEEP_*)
    int rp_px, rp_py, rp_npx, rp_sdy; // runtime-shaped 2-D tiles of the structured kernels (Sweep::rp_*)
    const uint8_t *rp_flag;
    int il_planes, il_nw; // interleaved z-march of the structured form (Sweep::il_*)
    const uint32_t *il_umask;
    const int32_t *il_seg;
};
// the launch of a vector kernel (K2, K4, K5) as it sees it: logical tiles t0, t0 + stride, ... of the front sweep
// (the XCD-aware map of ec3d_tile_of: t0 = (b % 8) * S + b / 8, stride = 8 S; or t0 = b, stride = nblk), then its
// share of the tile list.  Passing the whole Sweep cost the loops a kernel-argument reload per tile.
struct SweepV {
    int64_t ntiles, n;
    int S, nblk;
    int pstride, part_off;
    int ulist_n, two;
    const int32_t *ulist;
    int64_t win_nt, win_blk, win_t0;
    int nown;
    int64_t own_lo[4], own_hi[4];
    int keep; // producers whose output stays cacheable (EC3D_KEEP_*)
};
// the policy bits travel in Sweep::nt above bit 0 (bit 0: nontemporal streams)
__device__ __forceinline__ int keep_of(const Sweep &sw) { return sw.nt >> 1; }
__device__ __forceinline__ int keep_of(const SweepZ &sw) { return sw.keep; }
__device__ __forceinline__ int keep_of(const SweepV &sw) { return sw.keep; }
template <bool ZM> struct SweepSel { typedef Sweep type; };
template <> struct SweepSel<true> { typedef SweepZ type; };

// rows >= n (padding up to the tile; in a z-slab they overlap the upper halo plane) are never
// stored and contribute +0 to every dot product
template <bool NT>
__device__ __forceinline__ void store2(double *__restrict__ v, int64_t r, int64_t n, double a, double b)
{
    if (r + 1 < n) {
        if (NT)
            __builtin_nontemporal_store(d2{a, b}, reinterpret_cast<d2 *>(v + r));
        else
            *reinterpret_cast<d2 *>(v + r) = d2{a, b};
    } else if (r < n) {
        v[r] = a;
    }
}
// A producer's output at sizes where the NEXT kernel can find it in the 256 MiB Infinity Cache: `keep` (uniform, from
// the launch's policy bits, see choose_sweep) stores it cacheable although the launch's streams are nontemporal.
#define EC3D_KEEP_AP 1  /* K1's AP  -> K2 */
#define EC3D_KEEP_S 2   /* K2's S   -> K3 */
#define EC3D_KEEP_R 8   /* K4's R   -> K5, K1's dot */
#define EC3D_KEEP_P 32  /* K5's P   -> K1 */
template <bool NT>
__device__ __forceinline__ void store2k(double *__restrict__ v, int64_t r, int64_t n, double a, double b, bool keep)
{
    if (NT && keep) store2<false>(v, r, n, a, b);
    else store2<NT>(v, r, n, a, b);
}
// rows that take part in the dot products: all rows < n, or -- for one z-slab of the A-V system held
// on an extended grid whose planes are not tile aligned -- the owned index ranges only.  (Tile-aligned slabs
// sweep owned tiles only: Sweep::win_*.)
__device__ __forceinline__ bool row_owned(const Sweep &sw, int64_t r)
{
    if (sw.nown == 0) return r < sw.n;
    bool o = false;
    for (int q = 0; q < sw.nown; ++q) o |= (r >= sw.own_lo[q]) & (r < sw.own_hi[q]);
    return o;
}
__device__ __forceinline__ bool row_owned(const SweepZ &sw, int64_t r) { return r < sw.n; }
__device__ __forceinline__ bool row_owned(const SweepV &sw, int64_t r)
{
    if (sw.nown == 0) return r < sw.n;
    bool o = false;
    for (int q = 0; q < sw.nown; ++q) o |= (r >= sw.own_lo[q]) & (r < sw.own_hi[q]);
    return o;
}
#define EC3D_MASK2(r, sw_, a, b)                                                               \
    do {                                                                                       \
        if (!row_owned(sw_, (r))) (a) = 0.0;                                                   \
        if (!row_owned(sw_, (r) + 1)) (b) = 0.0;                                               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// The tiles a workgroup visits, in order: the front sweep (ec3d_tile_of), then its share of the U-block list.
// body(tile, first): `first` (a compile-time constant at every call site, so the loop bodies are specialised and
// branch free) says that nothing is carried over from the previous tile -- the z-march registers must be loaded.
typedef std::integral_constant<bool, true> TrueC;
typedef std::integral_constant<bool, false> FalseC;

// plain maps (vector kernels: MODE -1; SpMV kernels on grids without a z-march and boundary launches: MODE 0)
template <int MODE, class BODY>
__device__ __forceinline__ void walk_plain(const Sweep &sw, BODY &&body)
{
    for (int64_t it = 0;; ++it) {
        const int64_t tile = ec3d_tile_of<MODE>(sw, blockIdx.x, it);
        if (tile < 0) break;
        body(tile, TrueC());
    }
    for (int64_t lst = blockIdx.x; lst < sw.ulist_n; lst += sw.nblk) body((int64_t)sw.ulist[lst], TrueC());
}
// vector kernels: the same sequence as ec3d_tile_of<-1> gives for their sweeps, walked by a stride.
// load(tile) requests a tile's operands and returns them; fin(tile, operands) computes, stores and accumulates.
// With sw.two the operands of TWO tiles are requested before the first is finished (twice the bytes in flight
// per wave); tiles are finished in the order of the one-tile walk, so the sums are the same sums.
template <class LOAD, class FIN>
__device__ __forceinline__ void walk_vec(const SweepV &sw, LOAD &&load, FIN &&fin)
{
    const int b = blockIdx.x;
    const int64_t t0 = sw.S > 0 ? (int64_t)(b & 7) * sw.S + (b >> 3) : b;
    const int64_t stride = sw.S > 0 ? (int64_t)8 * sw.S : sw.nblk;
    if (sw.win_nt > 0) {
        for (int64_t t = t0; t < sw.ntiles; t += stride) {
            const int64_t p = (t / sw.win_nt) * sw.win_blk + sw.win_t0 + t % sw.win_nt;
            auto a = load(p);
            fin(p, a);
        }
    } else if (sw.two == 4) {
        int64_t t = t0;
        for (; t + 3 * stride < sw.ntiles; t += 4 * stride) {
            auto a = load(t);
            auto c = load(t + stride);
            auto d = load(t + 2 * stride);
            auto e = load(t + 3 * stride);
            fin(t, a);
            fin(t + stride, c);
            fin(t + 2 * stride, d);
            fin(t + 3 * stride, e);
        }
        for (; t < sw.ntiles; t += stride) {
            auto a = load(t);
            fin(t, a);
        }
    } else if (sw.two == 2) {
        int64_t t = t0;
        for (; t + stride < sw.ntiles; t += 2 * stride) {
            auto a = load(t);
            auto c = load(t + stride);
            fin(t, a);
            fin(t + stride, c);
        }
        if (t < sw.ntiles) {
            auto a = load(t);
            fin(t, a);
        }
    } else {
        for (int64_t t = t0; t < sw.ntiles; t += stride) {
            auto a = load(t);
            fin(t, a);
        }
    }
    for (int64_t lst = b; lst < sw.ulist_n; lst += sw.nblk) {
        const int64_t p = sw.ulist[lst];
        auto a = load(p);
        fin(p, a);
    }
}
// ec3d_tile_of<1> walked incrementally: one column, runs of consecutive planes.  A run ends where the segment
// ends or -- in a windowed slab -- where the owned planes of a block end: the physical plane then jumps over the
// halo planes and the march starts afresh in the next block.
template <bool SPEC, class BODY>
__device__ __forceinline__ void walk_zm(const SweepZ &sw, BODY &&body)
{
    const int cpx = (sw.tpp + 7) >> 3, c = blockIdx.x & 7, sg = blockIdx.x >> 3;
    const int col = c * cpx + sg % cpx;
    int64_t lp = (int64_t)(sg / cpx) * sw.pps; // logical plane, relative to pl0
    int64_t lend = lp + sw.pps;
    if (sw.npl > 0 && lend > sw.npl) lend = sw.npl;
    if (sw.plstep <= 1) { // logical planes whose tile of this column exists: (pl0 + lp) * tpp + col < ntiles
        const int64_t nlp = (sw.ntiles - col + sw.tpp - 1) / sw.tpp - sw.pl0;
        if (lend > nlp) lend = nlp;
    }
    if (col >= sw.tpp) lend = lp;
    const bool win = sw.win_npo > 0;
    int64_t rem = 0, pl = sw.pl0 + (sw.plstep > 1 ? lp * sw.plstep : lp); // (plstep > 1: one plane per workgroup, pps = 1)
    if (win && lp < lend) {
        rem = pl % sw.win_npo;
        pl = (pl / sw.win_npo) * sw.win_npb + sw.win_p0 + rem;
    }
    if constexpr (SPEC) { // two copies of the body in the front sweep, each without the test
        while (lp < lend) {
            int64_t run = lend - lp;
            if (win && run > sw.win_npo - rem) run = sw.win_npo - rem;
            int64_t tile = pl * sw.tpp + col;
            body(tile, TrueC());
            for (int64_t q = 1; q < run; ++q) {
                tile += sw.tpp;
                body(tile, FalseC());
            }
            lp += run;
            pl += run + (sw.win_npb - sw.win_npo); // next block's first owned plane (windowed slab only)
            rem = 0;
        }
        for (int64_t lst = blockIdx.x; lst < sw.ulist_n; lst += gridDim.x) body((int64_t)sw.ulist[lst], TrueC());
    } else {
        // a body too large to have three times (structured form: two copies already cost 25-45 spilled
        // registers): ONE loop over the front sweep and the list, `first` a run-time flag
        bool fresh = true;
        int64_t lst = -1;
        for (;;) {
            int64_t tile;
            bool first;
            if (lst < 0 && lp < lend) {
                tile = pl * sw.tpp + col;
                first = fresh || (win && rem == 0);
                fresh = false;
                ++lp;
                ++pl;
                if (win && ++rem == sw.win_npo) {
                    rem = 0;
                    pl += sw.win_npb - sw.win_npo;
                }
            } else {
                if (lst < 0) lst = blockIdx.x;
                if (lst >= sw.ulist_n) break;
                tile = sw.ulist[lst];
                if (tile < 0) break; // a hole of the XCD-local list (choose_sweep) ends this workgroup's share
                lst += gridDim.x;
                first = true;
            }
            body(tile, first);
        }
    }
}
// Where a thread of a runtime-shaped patch sweep (structured A-V kernels, Sweep::rp_*) stands: set by the walker
// before every step.  An idle thread (its cells lie beyond the patch, or beyond the grid's last row in a ragged patch
// row) goes through the step like everybody else -- the workgroup meets at a barrier every step -- on the plane's
// first row, whose loads are all valid; it stores nothing and adds +0.0 to every sum.
struct PatchPos {
    int64_t P;  // plane over the four stacked blocks
    int q;      // patch within the plane
    int64_t r;  // first of the thread's two rows
    bool live;
    int tx, ty; // the thread's cells (tx, tx + 1) of patch row ty: fixed for the launch
};
template <class BODY>
__device__ __forceinline__ void walk_zm_rt(const SweepZ &sw, int64_t pitch, int64_t sdx, PatchPos &pp, BODY &&body)
{
    const int t2 = 2 * (int)threadIdx.x;
    pp.ty = t2 / sw.rp_px;
    pp.tx = t2 - pp.ty * sw.rp_px;
    const bool inpatch = pp.ty < sw.rp_py;
    int64_t po = 0; // the thread's first cell within a plane
    auto place = [&](int q) {
        const int pyi = q / sw.rp_npx, pxi = q - pyi * sw.rp_npx;
        const int gy = pyi * sw.rp_py + pp.ty;
        pp.q = q;
        pp.live = inpatch && gy < sw.rp_sdy;
        po = pp.live ? (int64_t)gy * sdx + (int64_t)pxi * sw.rp_px + pp.tx : 0;
    };
    // the front sweep: one column (patch position), consecutive planes of the three A blocks (walk_zm without windows)
    const int cpx = (sw.tpp + 7) >> 3, c = blockIdx.x & 7, sg = blockIdx.x >> 3;
    const int col = c * cpx + sg % cpx;
    int64_t lp = (int64_t)(sg / cpx) * sw.pps;
    int64_t lend = lp + sw.pps;
    if (sw.npl > 0 && lend > sw.npl) lend = sw.npl;
    {
        const int64_t nlp = (sw.ntiles - col + sw.tpp - 1) / sw.tpp - sw.pl0;
        if (lend > nlp) lend = nlp;
    }
    if (col >= sw.tpp) lend = lp;
    if (lp < lend) place(col);
    bool fresh = true;
    int64_t lst = -1;
    for (;;) {
        bool first;
        if (lst < 0 && lp < lend) {
            pp.P = sw.pl0 + lp;
            first = fresh;
            fresh = false;
            ++lp;
        } else {
            if (lst < 0) lst = blockIdx.x;
            if (lst >= sw.ulist_n) break;
            const int T = sw.ulist[lst];
            if (T < 0) break; // a hole of the XCD-local list ends this workgroup's share
            lst += gridDim.x;
            pp.P = T / sw.tpp;
            place(T - (int)pp.P * sw.tpp);
            first = true;
        }
        pp.r = pp.P * pitch + po;
        body(pp.P * sw.tpp + pp.q, first);
    }
}
template <int FMT, bool ZM, bool SPEC, bool PATCH, class SW, class AD, class BODY>
__device__ __forceinline__ void walk_spmv(const AD &A, const SW &sw, PatchPos &pp, BODY &&body)
{
    if constexpr (FMT == 207 /* FMT_SAV */ && ZM && PATCH) walk_zm_rt(sw, A.pitch, A.sdx, pp, body);
    else if constexpr (ZM) walk_zm<SPEC>(sw, body);
    else walk_plain<0>(sw, body);
}
#define EC3D_ROW const int64_t r = tile * EC3D_TILE + 2 * (int64_t)threadIdx.x
// SpMV kernels: the thread's first row in `tile`, whether it owns rows there at all (runtime-shaped patches have idle
// threads), and the row count its stores compare with (0 for an idle thread: nothing is stored)
#define EC3D_ROW_S                                                                                                     \
    const int64_t r = (FMT == FMT_SAV && PATCH) ? pp.r                                                                 \
                      : PATCH ? ec3d_row_of(sw, tile, (int)threadIdx.x) : tile * EC3D_TILE + 2 * (int64_t)threadIdx.x; \
    const bool live_ = (FMT == FMT_SAV && PATCH) ? pp.live : true;                                                     \
    const int64_t nst = live_ ? sw.n : 0
#define EC3D_IDLE2(a, b)                                                                                               \
    do {                                                                                                               \
        if (!live_) { (a) = 0.0; (b) = 0.0; }                                                                          \
    } while (0)

// The vector a row kernel multiplies by, behind a small accessor (pair = two consecutive entries from
// an 8-byte-aligned address, at = one gathered entry).
// (rpair / rat + form: the same values in two halves -- the loads, and what is computed from them -- for a caller that
// wants every request of a step out before the first one is waited for: a fused vector's pair() under `if` keeps its
// arithmetic, and with it a wait for everything in flight, inside the branch)
struct VecPlain {
    const double *__restrict__ x;
    struct Raw2 { d2 v; };
    struct Raw1 { double v; };
    __device__ __forceinline__ d2 pair(int64_t i) const { return *reinterpret_cast<const d2u *>(x + i); }
    __device__ __forceinline__ double at(int64_t i) const { return x[i]; }
    __device__ __forceinline__ Raw2 rpair(int64_t i) const { return Raw2{pair(i)}; }
    __device__ __forceinline__ Raw1 rat(int64_t i) const { return Raw1{x[i]}; }
    __device__ __forceinline__ d2 form(const Raw2 &w) const { return w.v; }
    __device__ __forceinline__ double form(const Raw1 &w) const { return w.v; }
    static __device__ __forceinline__ void to_carry(const Raw2 &w, d2 (&c)[3]) { c[0] = w.v; }
    static __device__ __forceinline__ Raw2 from_carry(const d2 (&c)[3]) { return Raw2{c[0]}; }
};
// S = R - alpha*AP formed where it is read (K2 fused into K3, src/solvers.f90:33 inside :39): every value is the same
// expression, rounded the same way, whichever thread forms it -- the owner that stores it or a neighbour that needs it
struct VecFused {
    const double *__restrict__ rv;
    const double *__restrict__ ap;
    double alpha;
    __device__ __forceinline__ d2 pair(int64_t i) const
    {
        const d2 q = *reinterpret_cast<const d2u *>(rv + i), a = *reinterpret_cast<const d2u *>(ap + i);
        return d2{q.x - alpha * a.x, q.y - alpha * a.y};
    }
    __device__ __forceinline__ double at(int64_t i) const { return rv[i] - alpha * ap[i]; }
    struct Raw2 { d2 q, a; };
    struct Raw1 { double q, a; };
    __device__ __forceinline__ Raw2 rpair(int64_t i) const
    {
        return Raw2{*reinterpret_cast<const d2u *>(rv + i), *reinterpret_cast<const d2u *>(ap + i)};
    }
    __device__ __forceinline__ Raw1 rat(int64_t i) const { return Raw1{rv[i], ap[i]}; }
    __device__ __forceinline__ d2 form(const Raw2 &w) const { return d2{w.q.x - alpha * w.a.x, w.q.y - alpha * w.a.y}; }
    __device__ __forceinline__ double form(const Raw1 &w) const { return w.q - alpha * w.a; }
    static __device__ __forceinline__ void to_carry(const Raw2 &w, d2 (&c)[3]) { c[0] = w.q; c[1] = w.a; }
    static __device__ __forceinline__ Raw2 from_carry(const d2 (&c)[3]) { return Raw2{c[0], c[1]}; }
};
// P = R + beta*(P - omega*AP) formed where it is read (K5 fused into the next iteration's K1, src/solvers.f90:46 inside
// :30), or P = R after a restart (:47-49); reads the PREVIOUS iteration's P and AP, which live in other buffers than
// the ones this launch writes (neighbouring workgroups read them too)
struct VecFusedP {
    const double *__restrict__ rv;
    const double *__restrict__ p;
    const double *__restrict__ ap;
    double beta, omega;
    bool restart;
    __device__ __forceinline__ d2 pair(int64_t i) const
    {
        const d2 q = *reinterpret_cast<const d2u *>(rv + i);
        if (restart) return q;
        const d2 pv = *reinterpret_cast<const d2u *>(p + i), a = *reinterpret_cast<const d2u *>(ap + i);
        return d2{q.x + beta * (pv.x - omega * a.x), q.y + beta * (pv.y - omega * a.y)};
    }
    __device__ __forceinline__ double at(int64_t i) const
    {
        if (restart) return rv[i];
        return rv[i] + beta * (p[i] - omega * ap[i]);
    }
    struct Raw2 { d2 q, pv, a; };
    struct Raw1 { double q, pv, a; };
    __device__ __forceinline__ Raw2 rpair(int64_t i) const
    {
        Raw2 w{};
        w.q = *reinterpret_cast<const d2u *>(rv + i);
        if (!restart) {
            w.pv = *reinterpret_cast<const d2u *>(p + i);
            w.a = *reinterpret_cast<const d2u *>(ap + i);
        }
        return w;
    }
    __device__ __forceinline__ Raw1 rat(int64_t i) const
    {
        Raw1 w{};
        w.q = rv[i];
        if (!restart) {
            w.pv = p[i];
            w.a = ap[i];
        }
        return w;
    }
    __device__ __forceinline__ d2 form(const Raw2 &w) const
    {
        if (restart) return w.q;
        return d2{w.q.x + beta * (w.pv.x - omega * w.a.x), w.q.y + beta * (w.pv.y - omega * w.a.y)};
    }
    __device__ __forceinline__ double form(const Raw1 &w) const
    {
        if (restart) return w.q;
        return w.q + beta * (w.pv - omega * w.a);
    }
    static __device__ __forceinline__ void to_carry(const Raw2 &w, d2 (&c)[3]) { c[0] = w.q; c[1] = w.pv; c[2] = w.a; }
    static __device__ __forceinline__ Raw2 from_carry(const d2 (&c)[3]) { return Raw2{c[0], c[1], c[2]}; }
};
// Tail of one row: s += tval[e] * x[tcol[e]] over the row's slots of its 64-row slice, in stored order.
// The loads are issued in batches (all values/columns of a batch, then all gathers, then the adds in
// order): a plain loop is a chain of two dependent global loads per entry, ~1-2 us each, and a 13-entry
// U row then costs more than the whole banded part of the tile.
template <int B, class V>
__device__ __forceinline__ double tail_batch(const TailDev &A, const V &x, int64_t e0, int cnt, double s)
{
    double tv[B], xv[B];
    int tc[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
        const bool on = j < cnt;
        tv[j] = on ? A.tval[e0 + (int64_t)j * EC3D_CHUNK] : 0.0;
        tc[j] = on ? A.tcol[e0 + (int64_t)j * EC3D_CHUNK] : 0;
    }
#pragma unroll
    for (int j = 0; j < B; ++j) xv[j] = j < cnt ? x.at(tc[j]) : 0.0;
#pragma unroll
    for (int j = 0; j < B; ++j)
        if (j < cnt) s = s + tv[j] * xv[j];
    return s;
}

template <class V>
__device__ __forceinline__ double tail_add(const TailDev &A, const V &x, int t, double s)
{
    const int64_t base = A.chunk_ptr[t >> 6], end = A.chunk_ptr[(t >> 6) + 1];
    int w = (int)((end - base) >> 6); // slots per row in this slice
    int64_t e = base + (t & 63);
    if (w <= 4) return tail_batch<4, V>(A, x, e, w, s);
    while (w > 0) {
        const int c = w < 8 ? w : 8;
        s = tail_batch<8, V>(A, x, e, c, s);
        e += (int64_t)8 * EC3D_CHUNK;
        w -= 8;
    }
    return s;
}

template <int FMT>
__device__ __forceinline__ void stage_table(const MatDev<FMT> &A, double *tbl)
{
    if constexpr (FMT == FMT_DICT7 || FMT == FMT_SAV) {
        const int cnt = A.ncls * (FMT == FMT_SAV ? EC3D_SAV_STRIDE : 7);
        for (int i = threadIdx.x; i < cnt; i += EC3D_THREADS) tbl[i] = A.table[i];
        __syncthreads();
    }
}
// the same in two halves (see partials_request): the first four entries a thread copies -- 1024 doubles, the 64
// classes of a natively assembled structured system -- are requested early, stored behind the exit test
struct TableEarly { double v[4]; };
template <int FMT>
__device__ __forceinline__ void table_request(const MatDev<FMT> &A, TableEarly &e)
{
    if constexpr (FMT == FMT_DICT7 || FMT == FMT_SAV) {
        const int cnt = A.ncls * (FMT == FMT_SAV ? EC3D_SAV_STRIDE : 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = (int)threadIdx.x + j * EC3D_THREADS;
            e.v[j] = A.table[i < cnt ? i : 0]; // branch free (see partials_request)
        }
    }
}
template <int FMT>
__device__ __forceinline__ void table_store(const MatDev<FMT> &A, const TableEarly &e, double *tbl)
{
    if constexpr (FMT == FMT_DICT7 || FMT == FMT_SAV) {
        const int cnt = A.ncls * (FMT == FMT_SAV ? EC3D_SAV_STRIDE : 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = (int)threadIdx.x + j * EC3D_THREADS;
            if (i < cnt) tbl[i] = e.v[j];
        }
        for (int i = (int)threadIdx.x + 4 * EC3D_THREADS; i < cnt; i += EC3D_THREADS) tbl[i] = A.table[i];
        __syncthreads();
    }
}

// Structured A-V form, grids WITHOUT tile-aligned planes (small problems): the coupling slots of rows r, r+1
// (see MatView), added in slot order = ascending column order = the reference's row-sum order.  A slot whose
// coefficient is zero is not an entry of the reference's row: it is neither loaded nor added.  Both rows take
// their operands from one 16-byte load per slot.
// the two running sums are final here and no load moves across (register pressure, see the callers)
#define EC3D_PIN(a, b) asm volatile("" : "+v"(a), "+v"(b)::"memory")
template <class V>
__device__ __forceinline__ void sav_u_pre(const MatDev<FMT_SAV> &A, const double *t0, const double *t1, const V &x,
                                          int64_t r, double &s0, double &s1)
{
    // per component: the (up to) three loads first, then the adds in slot order -- a load inside the
    // add chain would cost one memory round trip per slot
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int64_t base = r - (3 - d) * A.nC;
        double v0[3], v1[3];
        d2 xx[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            v0[j] = t0[7 + 3 * d + j];
            v1[j] = t1[7 + 3 * d + j];
            xx[j] = d2{0.0, 0.0};
            if (v0[j] != 0.0 || v1[j] != 0.0) xx[j] = x.pair(base + (j - 1) * A.step(d));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (v0[j] != 0.0) s0 = s0 + v0[j] * xx[j].x;
            if (v1[j] != 0.0) s1 = s1 + v1[j] * xx[j].y;
        }
        EC3D_PIN(s0, s1); // keep the batches apart: nine operand pairs in flight would cost 36 registers
    }
}
template <class V>
__device__ __forceinline__ void sav_a_post(const MatDev<FMT_SAV> &A, const double *t0, const double *t1, const V &x,
                                           int64_t r, int d, double &s0, double &s1)
{
    const int64_t base = r + (3 - d) * A.nC, step = A.step(d);
    double v0[5], v1[5];
    d2 xx[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        v0[m] = t0[7 + m];
        v1[m] = t1[7 + m];
        xx[m] = d2{0.0, 0.0};
        if (v0[m] != 0.0 || v1[m] != 0.0) xx[m] = x.pair(base + (m - 2) * step);
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        if (v0[m] != 0.0) s0 = s0 + v0[m] * xx[m].x;
        if (v1[m] != 0.0) s1 = s1 + v1[m] * xx[m].y;
    }
}

// x values of the planes below / at the current row, carried across the steps of a z-march
struct ZRegs {
    d2 xm, xc;
    d2 rim; // 2-D tiles: the outer neighbour row of the NEXT step's centre plane (first / last patch row only)
    d2 cr[3]; // ... or its operands as requested at the end of a step, still on their way (patch_pair, request mode 4)
    double edge; // 2-D tiles, request mode 5: the cell beside the patch row's end in the NEXT step's centre plane (edge lanes)
};

// ---------------------------------------------------------------------------------------------
// LDS staging of operands that would not fit the register budget while in flight: an LDS-DMA load
// (global_load_lds_dwordx4) has no register destination.  Every thread fetches ITS OWN 16 bytes into its own
// slot and reads nothing else back, so no barrier is involved -- LDS serves as spill space for loads in
// flight.  One wave instruction writes 1 KiB at (wave-uniform base) + lane * 16.
#define EC3D_NSTAGE 4 /* 16-byte slots per thread: 4 x 4 KiB per workgroup */
__device__ __forceinline__ void stage_issue(const double *gp, double *stg, int k)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                     (__attribute__((address_space(3))) void *)(stg + k * EC3D_TILE +
                                                                                (threadIdx.x >> 6) * 128),
                                     16, 0, 0);
}
__device__ __forceinline__ d2 stage_read(const double *stg, int k)
{
    return *reinterpret_cast<const d2 *>(stg + k * EC3D_TILE + 2 * threadIdx.x);
}
// every vector-memory operation of this wave has landed (LDS-DMA included: it counts in vmcnt)
// (the instruction through the builtin, so that the compiler's own wait-count bookkeeping sees it -- behind an inline
// asm it kept every load of the step for outstanding and put a vmcnt(0) of its own in front of the next step's first
// write to one of their registers; the empty asm keeps LDS reads from moving above)
#define EC3D_VM_DRAIN                                                                                                  \
    do {                                                                                                               \
        __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0), expcnt and lgkmcnt left alone (gfx9 encoding) */              \
        asm volatile("" ::: "memory");                                                                                 \
    } while (0)

// Structured A-V form on a grid with tile-aligned planes (every grid worth timing): rows r, r+1 of A*x in a
// tile that lies entirely in one block.  The operands of the coupling slots are requested TOGETHER with the band
// operands -- one memory round trip per tile -- through the LDS staging slots, where the earlier version made
// one dependent round trip per batch of slots (class byte -> table -> coefficient != 0 -> load; 2 round trips
// in a coupled A tile, 4 in a U tile) because their operands had no registers to wait in:
//   coupled A_d tile: U(cell + m*step_d), m = -1, 0, +1 (d = x: the aligned pairs at -2, 0, +2, which hold
//                     every slot m = -2..2 of both rows); the outer slots m = -2, +2 of d = y, z occur on
//                     conductor faces only and are fetched afterwards by the waves that meet one
//   U tile:           A_x(cell - 1 .. cell + 2) from one register pair and two lane shuffles, A_y(cell -+ sdx),
//                     A_z(cell -+ pitch) staged; the own-cell slots of y and z (conductor faces) afterwards
// Same products, same order as before: bands ascending then U slots (A rows), A slots then bands (U rows) =
// ascending columns = the reference's row sum (src/solvers.f90:59 after src/EC3D.f90:715); a slot whose
// coefficient is zero is not an entry of the reference's row and is not added.
// the 7-point part of a z-march step: operands of rows r, r+1 for the seven bands (as the single-component
// kernels take them: plane below / centre from the registers, +-1 by lane shuffle).  Named registers, no
// arrays: an operand array shared by the two tile kinds below ended up as one 32-register tuple in scratch.
struct SavBand {
    d2 zm, ym, c, yp, zp; // x[r - pitch], x[r - sdx], x[r], x[r + sdx], x[r + pitch]   (pairs r, r+1)
    double left, right;   // x[r - 1], x[r + 2] on the wave's edge lanes
};
template <class V>
__device__ __forceinline__ void sav_band_loads(const MatDev<FMT_SAV> &A, const V &x, int64_t r, bool first, ZRegs &z,
                                               SavBand &o)
{
    const int lane = threadIdx.x & 63;
    // the wave's two outer neighbours in ONE predicated load: two separate ifs compiled to an if/else whose else
    // branch (every wave has lanes 1..62) began with s_waitcnt vmcnt(0) -- a memory round trip before the band loads
    double edge = 0.0;
    if (lane == 0 || lane == 63) edge = x.at(lane == 0 ? r - 1 : r + 2);
    o.left = edge;
    o.right = edge;
    o.ym = x.pair(r - A.sdx);
    o.yp = x.pair(r + A.sdx);
    o.zp = x.pair(r + A.pitch);
    if (first) {
        o.zm = x.pair(r - A.pitch);
        o.c = x.pair(r);
    } else {
        o.zm = z.xm;
        o.c = z.xc;
    }
    z.xm = o.c;
    z.xc = o.zp;
}
__device__ __forceinline__ void sav_band_sum(const double *t0, const double *t1, const SavBand &o, double &s0,
                                             double &s1)
{
    const int lane = threadIdx.x & 63;
    double left = o.left, right = o.right;
    const double l = __shfl_up(o.c.y, 1, 64), rr = __shfl_down(o.c.x, 1, 64);
    if (lane != 0) left = l;
    if (lane != 63) right = rr;
    s0 = s0 + t0[0] * o.zm.x;
    s1 = s1 + t1[0] * o.zm.y;
    s0 = s0 + t0[1] * o.ym.x;
    s1 = s1 + t1[1] * o.ym.y;
    s0 = s0 + t0[2] * left;
    s1 = s1 + t1[2] * o.c.x;
    s0 = s0 + t0[3] * o.c.x;
    s1 = s1 + t1[3] * o.c.y;
    s0 = s0 + t0[4] * o.c.y;
    s1 = s1 + t1[4] * right;
    s0 = s0 + t0[5] * o.yp.x;
    s1 = s1 + t1[5] * o.yp.y;
    s0 = s0 + t0[6] * o.zp.x;
    s1 = s1 + t1[6] * o.zp.y;
}

// tile_flag[tile] through the scalar unit: the dword that holds the byte (gfx9 has no scalar byte load; the array is
// allocated with 4 bytes to spare)
__device__ __forceinline__ bool sav_tile_coupled(const MatDev<FMT_SAV> &A, int64_t tile)
{
    const int t = __builtin_amdgcn_readfirstlane((int)tile);
    // constant address space: the flags are written by no kernel of this library's solve, and only a load the
    // compiler knows to be unclobbered goes through the scalar unit (s_load_dword, its own counter: no vmcnt wait)
    typedef const __attribute__((address_space(4))) unsigned *cptr;
    const unsigned w = ((cptr)(uintptr_t)A.tile_flag)[t >> 2];
    return ((w >> ((t & 3) * 8)) & 0xFFu) != 0;
}
template <class V>
__device__ __forceinline__ void sav_pair_zm(const MatDev<FMT_SAV> &A, const double *tbl, double *stg, const V &x,
                                            int64_t r, int64_t tile, bool first, ZRegs &z, double &s0, double &s1,
                                            d2 &ctr)
{
    const int lane = threadIdx.x & 63;
    // which block the tile lies in, decided on the scalar unit from the tile number (blocks are whole numbers of tiles
    // here): taken from the lanes' rows the tests are vector compares, the branches below run under exec masks and
    // the compiler protects registers both sides write with waits for every load in flight
    const int64_t trow = (int64_t)__builtin_amdgcn_readfirstlane((int)tile) * EC3D_TILE;
    const bool urow = trow >= 3 * A.nC;
    const unsigned short cc = *reinterpret_cast<const unsigned short *>(A.cls + r);
    // coupled? (uniform)  A U tile that is visited holds an unknown, so it is; an A tile's flag comes through the
    // scalar unit (read as a byte it was a vector load, and a wait on vmcnt, in front of the step's loads)
    const bool cpl = urow ? true : sav_tile_coupled(A, tile);
    SavBand bo;
    s0 = 0.0;
    s1 = 0.0;
    // EVERY load of the step is requested before the first value is looked at (the class bytes included: with the
    // two tile kinds requesting their operands in branches of their own, the compiler hoisted the classes' first use
    // above the branch and the step began with a wait for them)
    const int d = (trow >= A.nC) + (trow >= 2 * A.nC); // component of an A tile
    d2 xa;      // U tile only (left unset elsewhere: a value to set is a register to protect with a wait)
    double xae;
    if (urow) {
        xae = 0.0;
        const double *ay = x.x + r - 2 * A.nC, *az = x.x + r - A.nC;
        stage_issue(ay - A.sdx, stg, 0);
        stage_issue(ay + A.sdx, stg, 1);
        stage_issue(az - A.pitch, stg, 2);
        stage_issue(az + A.pitch, stg, 3);
        const int64_t rx = r - 3 * A.nC;
        xa = x.pair(rx);
        if (lane == 0 || lane == 63) xae = x.at(lane == 0 ? rx - 1 : rx + 2); // one predicated load (see sav_band_loads)
    } else if (cpl) {
        const double *u = x.x + r + (3 - d) * A.nC;
        const int64_t st = d == 0 ? 2 : A.step(d);
        stage_issue(u - st, stg, 0);
        stage_issue(u, stg, 1);
        stage_issue(u + st, stg, 2);
    }
    sav_band_loads(A, x, r, first, z, bo);
    ctr = bo.c;
    const double *t0 = tbl + (cc & 0xFF) * EC3D_SAV_STRIDE, *t1 = tbl + (cc >> 8) * EC3D_SAV_STRIDE;
    if (urow) {
        // ---- a tile of the U block ----
        double xal = xae, xar = xae;
        EC3D_VM_DRAIN;
        {   // A_x(cell - 1), A_x(cell), A_x(cell + 1)
            const double l = __shfl_up(xa.y, 1, 64), rr = __shfl_down(xa.x, 1, 64);
            if (lane != 0) xal = l;
            if (lane != 63) xar = rr;
            const double o0[3] = {xal, xa.x, xa.y}, o1[3] = {xa.x, xa.y, xar};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double v0 = t0[7 + j], v1 = t1[7 + j];
                if (v0 != 0.0) s0 = s0 + v0 * o0[j];
                if (v1 != 0.0) s1 = s1 + v1 * o1[j];
            }
        }
        EC3D_PIN(s0, s1); // one group of coefficients and operands in registers at a time
#pragma unroll
        for (int dd = 1; dd < 3; ++dd) { // A_y, A_z
            const double vm0 = t0[7 + 3 * dd], vm1 = t1[7 + 3 * dd], vc0 = t0[8 + 3 * dd], vc1 = t1[8 + 3 * dd],
                         vp0 = t0[9 + 3 * dd], vp1 = t1[9 + 3 * dd];
            d2 oc = d2{0.0, 0.0};
            const bool want = vc0 != 0.0 || vc1 != 0.0; // own-cell slot: rows on a conductor face only
            if (__any(want)) {
                if (want) oc = x.pair(r - (3 - dd) * A.nC);
            }
            const d2 om = stage_read(stg, 2 * dd - 2), op = stage_read(stg, 2 * dd - 1);
            if (vm0 != 0.0) s0 = s0 + vm0 * om.x;
            if (vm1 != 0.0) s1 = s1 + vm1 * om.y;
            if (vc0 != 0.0) s0 = s0 + vc0 * oc.x;
            if (vc1 != 0.0) s1 = s1 + vc1 * oc.y;
            if (vp0 != 0.0) s0 = s0 + vp0 * op.x;
            if (vp1 != 0.0) s1 = s1 + vp1 * op.y;
            EC3D_PIN(s0, s1);
        }
        sav_band_sum(t0, t1, bo, s0, s1);
        return;
    }
    // ---- a tile of an A block ----
    sav_band_sum(t0, t1, bo, s0, s1);
    if (cpl) {
        EC3D_PIN(s0, s1); // the band operands are dead from here
        EC3D_VM_DRAIN;
        const d2 q0 = stage_read(stg, 0), q1 = stage_read(stg, 1), q2 = stage_read(stg, 2);
        double o0[5], o1[5];
        if (d == 0) { // cells r-2 .. r+3 lie in the three aligned pairs
            o0[0] = q0.x; o0[1] = q0.y; o0[2] = q1.x; o0[3] = q1.y; o0[4] = q2.x;
            o1[0] = q0.y; o1[1] = q1.x; o1[2] = q1.y; o1[3] = q2.x; o1[4] = q2.y;
        } else {
            o0[1] = q0.x; o0[2] = q1.x; o0[3] = q2.x;
            o1[1] = q0.y; o1[2] = q1.y; o1[3] = q2.y;
            // outer slots: one-sided stencils at a conductor face (src/EC3D.f90:667-676) only
            const bool wlo = t0[7] != 0.0 || t1[7] != 0.0, whi = t0[11] != 0.0 || t1[11] != 0.0;
            d2 qlo = d2{0.0, 0.0}, qhi = d2{0.0, 0.0};
            if (__any(wlo || whi)) {
                const int64_t base = r + (3 - d) * A.nC, st = A.step(d);
                if (wlo) qlo = x.pair(base - 2 * st);
                if (whi) qhi = x.pair(base + 2 * st);
            }
            o0[0] = qlo.x; o1[0] = qlo.y;
            o0[4] = qhi.x; o1[4] = qhi.y;
        }
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const double v0 = t0[7 + m], v1 = t1[7 + m];
            if (v0 != 0.0) s0 = s0 + v0 * o0[m];
            if (v1 != 0.0) s1 = s1 + v1 * o1[m];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The INTERLEAVED z-march of the structured A-V form (round 6; Sweep::il_*, single-rank handles with tile-aligned planes).
// A workgroup owns one column (a 512-cell position of the xy plane) and a range of planes, and at every plane k takes the
// tiles of A_x, A_y, A_z there TOGETHER -- and, where the cell range holds a conductor cell, the U tile as well (one bit per
// (column, plane), il_umask; a coupled A row implies that bit: its cell carries a U unknown).  Every block's planes below /
// at the row are carried in registers of its own, the band operands of all (three or four) tiles of a step are requested
// before the first value is looked at, and then EVERY coupling operand of the reference's rows is a band operand of another
// block that is already there:
//   A_x row (src/EC3D.f90:656-711), U(cell + m), m = -2 .. 2:   U's centre pair and its two neighbour lanes' pairs
//   A_y row, U(cell + m sdx):    m = -1, 0, +1 = U's ym, c, yp;  m = -2, +2 (conductor faces only) loaded by the waves that meet one
//   A_z row, U(cell + m pitch):  m = -1, 0, +1 = U's zm, c, zp;  m = -2, +2 likewise
//   U row (:766-959), A_x(cell - 1 .. cell + 1) = A_x's left, c, right;  A_y(cell -+ sdx, cell) = A_y's ym, yp, c;
//                     A_z(cell -+ pitch, cell) = A_z's zm, zp, c
// -- no staged loads, no second look at memory.  With the U tiles in a list behind the front sweep (sav_pair_zm) these
// operands were fetched again: measured at BASELINE config 3's stated size (256^3, 53.2 M unknowns: HBM-bound, each vector 426
// MB), the SpMV kernels read 781 MB per launch where the same system WITHOUT its conductor reads 488 MB and the U rows add 34 MB
// of their own -- 260 MB of re-fetched coupling operands (profiles/r06_av256_*).  Same products, same order as sav_pair_zm:
// bands ascending then U slots (A rows), A slots then bands (U rows) = ascending columns = the reference's row sum
// (src/solvers.f90:59 after src/EC3D.f90:715); a zero slot is neither loaded nor added.  The dot products are summed in the
// visit order A_x, A_y, A_z, (U) per plane, which ec3d_get_visit_order reports to the oracle's twin.
// What a block carries from step to step: the planes below / at the row, and -- requested a step AHEAD, as the last loads a
// step issues -- everything of the NEXT step that comes from HBM: the plane above (the +-sdx rows and the edge lanes are lines
// a neighbour fetched a step earlier: L2), the class bytes and the kernel's own operand (K1: R0).  The memory counter retires
// loads in order -- a wait for a later load is a wait for every earlier one -- so behind the step's own loads these requests
// delay nothing of the step; they are in flight while it computes (the copies of the carried registers at the loop's boundary
// are what finally waits for them).  Two things measured and not kept (profiles/r06_av256_*): the same requests by LDS-DMA
// into slots (no registers, three workgroups per CU) -- the compiler puts vmcnt(0) in front of the first LDS read behind an
// LDS-DMA, the class table's included, so the step waited for its requests at once; and under `if (more)` -- the carried
// register is then defined on two paths and the copy that merges them is a use of the loaded value.
struct IlRegs {
    d2 xm, xc, xn, qn;
    unsigned short ccn;
};
struct IlOps {
    SavBand b;
    unsigned short cc;
};
// the step's loads of one block: U = the U block (outer neighbours as PAIRS: the A_x rows read U two cells away)
template <bool U, class V, class PRE>
__device__ __forceinline__ void il_block_loads(const MatDev<FMT_SAV> &A, const V &x, PRE &&pre, int64_t r, bool first, IlRegs &z,
                                               IlOps &o, d2 &q, d2 &epair)
{
    const int lane = threadIdx.x & 63;
    if constexpr (U) {
        epair = d2{0.0, 0.0};
        if (lane == 0 || lane == 63) epair = x.pair(lane == 0 ? r - 2 : r + 2); // one predicated load (see sav_band_loads)
        o.b.left = epair.y;  // lane 0: U(r - 1)
        o.b.right = epair.x; // lane 63: U(r + 2)
    } else {
        double edge = 0.0;
        if (lane == 0 || lane == 63) edge = x.at(lane == 0 ? r - 1 : r + 2);
        o.b.left = edge;
        o.b.right = edge;
    }
    o.b.ym = x.pair(r - A.sdx);
    o.b.yp = x.pair(r + A.sdx);
    if (first) { // nothing carried, nothing requested ahead
        o.b.zm = x.pair(r - A.pitch);
        o.b.c = x.pair(r);
        o.b.zp = x.pair(r + A.pitch);
        o.cc = *reinterpret_cast<const unsigned short *>(A.cls + r);
        q = pre(r);
    } else {
        o.b.zm = z.xm;
        o.b.c = z.xc;
        o.b.zp = z.xn;
        o.cc = z.ccn;
        q = z.qn;
    }
    z.xm = o.b.c;
    z.xc = o.b.zp;
}
// ... and, last of the step, what the block's NEXT step needs from HBM (the workgroup's last step asks for its own plane
// again and drops it: no branch)
template <class V, class PRE>
__device__ __forceinline__ void il_block_ahead(const MatDev<FMT_SAV> &A, const V &x, PRE &&pre, int64_t r, bool more, IlRegs &z)
{
    const int64_t rn = r + (more ? A.pitch : 0);
    z.xn = x.pair(rn + A.pitch);
    z.ccn = *reinterpret_cast<const unsigned short *>(A.cls + rn);
    z.qn = pre(rn);
}
// the U slots m = -2 .. 2 of rows r, r + 1 of an A block, operands given
__device__ __forceinline__ void il_a_slots(const double *t0, const double *t1, const double (&o0)[5], const double (&o1)[5],
                                           double &s0, double &s1)
{
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const double v0 = t0[7 + m], v1 = t1[7 + m];
        if (v0 != 0.0) s0 = s0 + v0 * o0[m];
        if (v1 != 0.0) s1 = s1 + v1 * o1[m];
    }
}
//   pre(r) -> d2   the kernel's own operand of rows r, r + 1 (K1: R0; the residual kernel: b; or nothing)
//   emit(r, s0, s1, ctr, q)   the row sums of rows r, r + 1, the centre pair of x there, and what pre returned for r
template <class V, class PRE, class EMIT>
__device__ __forceinline__ void walk_zm_il(const MatDev<FMT_SAV> &A, const SweepZ &sw, const double *tbl, const V &x, PRE &&pre,
                                           EMIT &&emit)
{
    const int lane = threadIdx.x & 63;
    typedef const __attribute__((address_space(4))) uint32_t *cptr; // through the scalar unit (see sav_tile_coupled)
    // this workgroup's entry of the work list (Sweep::il_seg): column, first plane, end
    const cptr sgp = (cptr)(uintptr_t)sw.il_seg + 4 * (int64_t)blockIdx.x;
    const int col = (int)sgp[0];
    int k = (int)sgp[1];
    const int kend = (int)sgp[2];
    const cptr um = (cptr)(uintptr_t)sw.il_umask + (int64_t)col * sw.il_nw;
    auto ubit = [&](int kk) -> bool { return (um[kk >> 5] >> (kk & 31)) & 1u; };
    IlRegs z0, z1, z2, z3;
    bool first = true, ufirst = true;
    bool cpl = k < kend ? ubit(k) : false;
    for (; k < kend; ++k) {
        const bool more = k + 1 < kend;
        const bool ncpl = more ? ubit(k + 1) : false; // (a scalar load of its own counter; the word stays in the scalar cache)
        // (the tile number through an opaque scalar register: left to itself the compiler strength-reduces every stream's
        // address of every block into an induction variable of its own -- some forty 64-bit pointers, and spills)
        int64_t t0 = (int64_t)k * sw.tpp + col;
        asm volatile("" : "+s"(t0));
        const int64_t r0 = t0 * EC3D_TILE + 2 * (int64_t)threadIdx.x, r1 = r0 + A.nC, r2 = r1 + A.nC, r3 = r2 + A.nC;
        // two copies of the step, each without the other's loads
        auto step = [&](auto cplc) {
            constexpr bool CPL = decltype(cplc)::value;
            // ---- the step's own loads (L2), then the next step's (HBM) ----
            IlOps a0, a1, a2, au;
            d2 ue, none, q0, q1, q2, q3;
            il_block_loads<false>(A, x, pre, r0, first, z0, a0, q0, none);
            il_block_loads<false>(A, x, pre, r1, first, z1, a1, q1, none);
            il_block_loads<false>(A, x, pre, r2, first, z2, a2, q2, none);
            if constexpr (CPL) il_block_loads<true>(A, x, pre, r3, ufirst, z3, au, q3, ue);
            il_block_ahead(A, x, pre, r0, more, z0);
            il_block_ahead(A, x, pre, r1, more, z1);
            il_block_ahead(A, x, pre, r2, more, z2);
            if constexpr (CPL) il_block_ahead(A, x, pre, r3, more, z3); // (used when the next plane has a U tile as well)
            const SavBand &b0 = a0.b, &b1 = a1.b, &b2 = a2.b, &bu = au.b;
            // ---- A_x rows ----
            double s0 = 0.0, s1 = 0.0;
            {
                const double *t0p = tbl + (a0.cc & 0xFF) * EC3D_SAV_STRIDE, *t1p = tbl + (a0.cc >> 8) * EC3D_SAV_STRIDE;
                sav_band_sum(t0p, t1p, b0, s0, s1);
                if constexpr (CPL) { // U(cell - 2 .. cell + 3): the centre pairs of lanes l - 1, l, l + 1
                    d2 lo, hi;
                    lo.x = __shfl_up(bu.c.x, 1, 64);
                    lo.y = __shfl_up(bu.c.y, 1, 64);
                    hi.x = __shfl_down(bu.c.x, 1, 64);
                    hi.y = __shfl_down(bu.c.y, 1, 64);
                    if (lane == 0) lo = ue;
                    if (lane == 63) hi = ue;
                    const double o0[5] = {lo.x, lo.y, bu.c.x, bu.c.y, hi.x}, o1[5] = {lo.y, bu.c.x, bu.c.y, hi.x, hi.y};
                    il_a_slots(t0p, t1p, o0, o1, s0, s1);
                }
                emit(r0, s0, s1, b0.c, q0);
            }
            // ---- A_y, A_z rows ----
            auto a_rows = [&](const IlOps &a, const d2 q, const int64_t r, const d2 um1, const d2 up1, const int64_t st) {
                const double *t0p = tbl + (a.cc & 0xFF) * EC3D_SAV_STRIDE, *t1p = tbl + (a.cc >> 8) * EC3D_SAV_STRIDE;
                s0 = 0.0;
                s1 = 0.0;
                sav_band_sum(t0p, t1p, a.b, s0, s1);
                if constexpr (CPL) {
                    // outer slots: one-sided stencils at a conductor face (src/EC3D.f90:667-676) only
                    const bool wlo = t0p[7] != 0.0 || t1p[7] != 0.0, whi = t0p[11] != 0.0 || t1p[11] != 0.0;
                    d2 qlo = d2{0.0, 0.0}, qhi = d2{0.0, 0.0};
                    if (__any(wlo || whi)) {
                        if (wlo) qlo = x.pair(r3 - 2 * st);
                        if (whi) qhi = x.pair(r3 + 2 * st);
                    }
                    const double o0[5] = {qlo.x, um1.x, bu.c.x, up1.x, qhi.x}, o1[5] = {qlo.y, um1.y, bu.c.y, up1.y, qhi.y};
                    il_a_slots(t0p, t1p, o0, o1, s0, s1);
                }
                emit(r, s0, s1, a.b.c, q);
            };
            a_rows(a1, q1, r1, bu.ym, bu.yp, A.sdx);
            a_rows(a2, q2, r2, bu.zm, bu.zp, A.pitch);
            // ---- U rows: the A slots (A_x, A_y, A_z: the lower columns), then the bands ----
            if constexpr (CPL) {
                const double *t0p = tbl + (au.cc & 0xFF) * EC3D_SAV_STRIDE, *t1p = tbl + (au.cc >> 8) * EC3D_SAV_STRIDE;
                s0 = 0.0;
                s1 = 0.0;
                {
                    double xal = b0.left, xar = b0.right;
                    const double l = __shfl_up(b0.c.y, 1, 64), rr = __shfl_down(b0.c.x, 1, 64);
                    if (lane != 0) xal = l;
                    if (lane != 63) xar = rr;
                    const double o0[3] = {xal, b0.c.x, b0.c.y}, o1[3] = {b0.c.x, b0.c.y, xar};
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double v0 = t0p[7 + j], v1 = t1p[7 + j];
                        if (v0 != 0.0) s0 = s0 + v0 * o0[j];
                        if (v1 != 0.0) s1 = s1 + v1 * o1[j];
                    }
                }
                auto a_slots = [&](const int dd, const d2 om, const d2 oc, const d2 op) {
                    const double vm0 = t0p[7 + 3 * dd], vm1 = t1p[7 + 3 * dd], vc0 = t0p[8 + 3 * dd], vc1 = t1p[8 + 3 * dd],
                                 vp0 = t0p[9 + 3 * dd], vp1 = t1p[9 + 3 * dd];
                    if (vm0 != 0.0) s0 = s0 + vm0 * om.x;
                    if (vm1 != 0.0) s1 = s1 + vm1 * om.y;
                    if (vc0 != 0.0) s0 = s0 + vc0 * oc.x;
                    if (vc1 != 0.0) s1 = s1 + vc1 * oc.y;
                    if (vp0 != 0.0) s0 = s0 + vp0 * op.x;
                    if (vp1 != 0.0) s1 = s1 + vp1 * op.y;
                };
                a_slots(1, b1.ym, b1.c, b1.yp);
                a_slots(2, b2.zm, b2.c, b2.zp);
                sav_band_sum(t0p, t1p, bu, s0, s1);
                emit(r3, s0, s1, bu.c, q3);
            }
        };
        if (cpl) {
            step(TrueC());
            ufirst = false;
        } else {
            step(FalseC());
            ufirst = true;
        }
        first = false;
        cpl = ncpl;
    }
}

// ---------------------------------------------------------------------------------------------
// Structured A-V form on runtime-shaped 2-D tiles (round 4).  What patch_pair below does for the single-component
// cube -- a workgroup owns a patch of the xy plane and marches in z; the plane above is ONE 16-byte global load per
// thread, the plane below and the centre are carried in registers, and the in-plane neighbours are the centre values
// the workgroup's other threads hold, exchanged through LDS (two 4 KiB buffers, one barrier per step) -- for the
// reference's own system [Ax | Ay | Az | U] on ANY grid with an even sdx: the patch is rp_px x rp_py cells with rp_px
// a divisor of sdx chosen when the matrix is set (choose_sweep: 102 x 5 on the 306-wide refinement of the shipped
// geometry, 128 x 4 on 256- and 384-wide grids), threads own cells 2t, 2t + 1 of the patch in row-major order, so a
// wave is no longer a patch row and the +-1 neighbours come out of the LDS buffer as well (two 8-byte reads where
// patch_pair shuffles lanes); only the patch's rim (first / last row, requested a plane ahead like patch_pair's, and
// the first / last cell of every row) goes to memory.  Global loads per step: 1 full-wave load + rim, where the linear
// tile (sav_pair_zm) issues 3.  The coupling operands of a coupled A tile / a U tile are requested with the band
// operands as in sav_pair_zm: two through LDS-DMA staging slots (16 KiB of slots would cost the sixth workgroup per
// CU: table 8 + exchange 8 + slots 8 = 24 KiB), the others into registers; when the operand vector is FORMED where it
// is read (K2 inside K3, K5 inside K1: V is not VecPlain) all of them go through registers.  Same products, same
// order as sav_pair_zm (bands ascending then U slots for A rows, A slots then bands for U rows = ascending columns =
// the reference's row sum, src/solvers.f90:59 after src/EC3D.f90:715): A*x is bit-identical; the dot products are
// summed in this thread -> cell assignment, which ec3d_geom::patch_* tells the oracle's twin.
#define EC3D_NSTAGE_RT 2
template <class V>
__device__ __forceinline__ void sav_patch_step(const MatDev<FMT_SAV> &A, const SweepZ &sw, const double *tbl,
                                               double *lds, int step, const V &x, const PatchPos &pp, bool first,
                                               ZRegs &z, double &s0, double &s1, d2 &ctr)
{
    constexpr bool STAGED = std::is_same<V, VecPlain>::value;
    const int t = threadIdx.x, tx = pp.tx, ty = pp.ty;
    const int px = sw.rp_px, py = sw.rp_py, hx = px >> 1;
    const int64_t r = pp.r, sdx = A.sdx, pitch = A.pitch;
    double *cur = lds + (step & 1) * EC3D_TILE, *nxt = lds + ((step + 1) & 1) * EC3D_TILE;
    double *stg = lds + 2 * EC3D_TILE; // the staging slots behind the two exchange buffers
    // which block the plane lies in: uniform, from the plane number
    const int64_t P = pp.P;
    const bool urow = P >= 3 * A.planes;
    const int d = (P >= A.planes) + (P >= 2 * A.planes);
    bool cpl = true; // a visited U tile holds an unknown; an A tile's flag comes through the scalar unit
    if (!urow) {
        typedef const __attribute__((address_space(4))) unsigned *cptr;
        const int T = __builtin_amdgcn_readfirstlane((int)(P * sw.tpp + pp.q));
        const unsigned w = ((cptr)(uintptr_t)sw.rp_flag)[T >> 2];
        cpl = ((w >> ((T & 3) * 8)) & 0xFFu) != 0;
    }
    s0 = 0.0;
    s1 = 0.0;
    // ---- every load of the step is requested before the first value is looked at ----
    const unsigned short cc = *reinterpret_cast<const unsigned short *>(A.cls + r);
    d2 c0v, c1v, c2v, c3v; // coupling operands in registers (which ones depends on the tile kind and on STAGED)
    d2 xlo, xhi;           // U tile: A_x(cell - 1, cell), A_x(cell + 1, cell + 2)
    if (urow) {
        const int64_t rx = r - 3 * A.nC, ry = r - 2 * A.nC, rz = r - A.nC;
        if constexpr (STAGED) {
            stage_issue(x.x + ry - sdx, stg, 0);
            stage_issue(x.x + ry + sdx, stg, 1);
        } else {
            c0v = x.pair(ry - sdx);
            c1v = x.pair(ry + sdx);
        }
        c2v = x.pair(rz - pitch);
        c3v = x.pair(rz + pitch);
        xlo = x.pair(rx - 1);
        xhi = x.pair(rx + 1);
    } else if (cpl) {
        const int64_t u = r + (3 - d) * A.nC, st = d == 0 ? 2 : A.step(d);
        if constexpr (STAGED) {
            stage_issue(x.x + u - st, stg, 0);
            stage_issue(x.x + u + st, stg, 1);
        } else {
            c0v = x.pair(u - st);
            c1v = x.pair(u + st);
        }
        c2v = x.pair(u);
    }
    // the band operands: plane above, the row's two outer cells, the patch's outer rows one plane ahead
    const d2 zp = x.pair(r + pitch);
    // the row's outer neighbour of its first / last thread in ONE predicated load (px >= 4: never both)
    double edge = 0.0;
    if (tx == 0 || tx == px - 2) edge = x.at(tx == 0 ? r - 1 : r + 2);
    const bool rimrow = ty == 0 || ty == py - 1;
    const int64_t roff = ty == 0 ? -sdx : sdx;
    d2 rimc = d2{0.0, 0.0};
    if constexpr (STAGED) {
        // beside the centre plane, where the stencil reads it: the unfused kernels' traffic did not change when
        // patch_pair's rim moved a plane ahead (PMC 25.5 / 17.4 B per row either way), and the carried pair is four
        // registers these instances do not have
        if (rimrow) rimc = x.pair(r + roff);
    } else if (rimrow) { // K2-in-K3 / K5-in-K1 (two or three operand vectors per step): a plane ahead, see patch_pair
        rimc = first ? x.pair(r + roff) : z.rim;
        z.rim = x.pair(r + pitch + roff);
    }
    d2 zm;
    if (first) { // nothing carried over: plane below and centre from memory, centre into this step's buffer
        zm = x.pair(r - pitch);
        *reinterpret_cast<d2 *>(cur + 2 * t) = x.pair(r);
    } else {
        zm = z.xm;
    }
    // this step's buffer is complete (written at the end of the previous step, or just now): a raw barrier behind a
    // wait for the LDS writes only, so the loads above stay in flight across it (see patch_pair)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const double *t0 = tbl + (cc & 0xFF) * EC3D_SAV_STRIDE, *t1 = tbl + (cc >> 8) * EC3D_SAV_STRIDE;
    // The in-plane band operands (and the centre pair itself, the thread's own slot: it is not carried in registers)
    // are read from the exchange buffer only where the row sum reaches the bands -- in a U tile behind the A slots,
    // whose operands then do not share the register file with them (these kernels sit at the budget of six
    // workgroups per CU).
    d2 ym, yp, cx;
    double left, right;
    auto bands = [&]() {
        asm volatile("" ::: "memory"); // no LDS read moves above this point
        ym = rimc;
        yp = rimc;
        if (ty > 0) ym = *reinterpret_cast<const d2 *>(cur + 2 * (t - hx));
        if (ty < py - 1) yp = *reinterpret_cast<const d2 *>(cur + 2 * (t + hx));
        left = edge;
        right = edge;
        if (tx > 0) left = cur[2 * t - 1];
        if (tx < px - 2 && ty < py) right = cur[2 * t + 2];
        cx = *reinterpret_cast<const d2 *>(cur + 2 * t);
        s0 = s0 + t0[0] * zm.x;
        s1 = s1 + t1[0] * zm.y;
        s0 = s0 + t0[1] * ym.x;
        s1 = s1 + t1[1] * ym.y;
        s0 = s0 + t0[2] * left;
        s1 = s1 + t1[2] * cx.x;
        s0 = s0 + t0[3] * cx.x;
        s1 = s1 + t1[3] * cx.y;
        s0 = s0 + t0[4] * cx.y;
        s1 = s1 + t1[4] * right;
        s0 = s0 + t0[5] * yp.x;
        s1 = s1 + t1[5] * yp.y;
        s0 = s0 + t0[6] * zp.x;
        s1 = s1 + t1[6] * zp.y;
    };
    if (urow) {
        // ---- a tile of the U block: the A slots first (A_x, A_y, A_z: the lower columns), then the bands ----
        if constexpr (STAGED) EC3D_VM_DRAIN;
        {
            const double o0[3] = {xlo.x, xlo.y, xhi.x}, o1[3] = {xlo.y, xhi.x, xhi.y};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double v0 = t0[7 + j], v1 = t1[7 + j];
                if (v0 != 0.0) s0 = s0 + v0 * o0[j];
                if (v1 != 0.0) s1 = s1 + v1 * o1[j];
            }
        }
        EC3D_PIN(s0, s1);
#pragma unroll
        for (int dd = 1; dd < 3; ++dd) { // A_y, A_z
            const double vm0 = t0[7 + 3 * dd], vm1 = t1[7 + 3 * dd], vc0 = t0[8 + 3 * dd], vc1 = t1[8 + 3 * dd],
                         vp0 = t0[9 + 3 * dd], vp1 = t1[9 + 3 * dd];
            d2 oc = d2{0.0, 0.0};
            const bool want = vc0 != 0.0 || vc1 != 0.0; // own-cell slot: rows on a conductor face only
            if (__any(want)) {
                if (want) oc = x.pair(r - (3 - dd) * A.nC);
            }
            d2 om, op;
            if (dd == 1) {
                if constexpr (STAGED) {
                    om = stage_read(stg, 0);
                    op = stage_read(stg, 1);
                } else {
                    om = c0v;
                    op = c1v;
                }
            } else {
                om = c2v;
                op = c3v;
            }
            if (vm0 != 0.0) s0 = s0 + vm0 * om.x;
            if (vm1 != 0.0) s1 = s1 + vm1 * om.y;
            if (vc0 != 0.0) s0 = s0 + vc0 * oc.x;
            if (vc1 != 0.0) s1 = s1 + vc1 * oc.y;
            if (vp0 != 0.0) s0 = s0 + vp0 * op.x;
            if (vp1 != 0.0) s1 = s1 + vp1 * op.y;
            EC3D_PIN(s0, s1);
        }
        bands();
    } else {
        // ---- a tile of an A block: the bands, then the U slots m = -2 .. 2 ----
        bands();
        if (cpl) {
            EC3D_PIN(s0, s1);
            d2 q0, q2;
            if constexpr (STAGED) {
                EC3D_VM_DRAIN;
                q0 = stage_read(stg, 0);
                q2 = stage_read(stg, 1);
            } else {
                q0 = c0v;
                q2 = c1v;
            }
            const d2 q1 = c2v;
            double o0[5], o1[5];
            if (d == 0) { // cells r-2 .. r+3 lie in the three aligned pairs
                o0[0] = q0.x; o0[1] = q0.y; o0[2] = q1.x; o0[3] = q1.y; o0[4] = q2.x;
                o1[0] = q0.y; o1[1] = q1.x; o1[2] = q1.y; o1[3] = q2.x; o1[4] = q2.y;
            } else {
                o0[1] = q0.x; o0[2] = q1.x; o0[3] = q2.x;
                o1[1] = q0.y; o1[2] = q1.y; o1[3] = q2.y;
                // outer slots: one-sided stencils at a conductor face (src/EC3D.f90:667-676) only
                const bool wlo = t0[7] != 0.0 || t1[7] != 0.0, whi = t0[11] != 0.0 || t1[11] != 0.0;
                d2 qlo = d2{0.0, 0.0}, qhi = d2{0.0, 0.0};
                if (__any(wlo || whi)) {
                    const int64_t base = r + (3 - d) * A.nC, st = A.step(d);
                    if (wlo) qlo = x.pair(base - 2 * st);
                    if (whi) qhi = x.pair(base + 2 * st);
                }
                o0[0] = qlo.x; o1[0] = qlo.y;
                o0[4] = qhi.x; o1[4] = qhi.y;
            }
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const double v0 = t0[7 + m], v1 = t1[7 + m];
                if (v0 != 0.0) s0 = s0 + v0 * o0[m];
                if (v1 != 0.0) s1 = s1 + v1 * o1[m];
            }
        }
    }
    // the plane above is the next step's centre: into the other buffer (nobody reads that one before the next barrier)
    *reinterpret_cast<d2 *>(nxt + 2 * t) = zp;
    ctr = cx;
    z.xm = cx;
}

// ---------------------------------------------------------------------------------------------
// 2-D tiles for the single-component 7-point operator (north_star: "LDS-staged neighbour stencils").  A workgroup
// owns a patch of EC3D_PX x EC3D_PY = 128 x 4 cells of the xy plane and marches in z: thread t holds cells
// (2q, 2q+1), q = t % 64, of patch row y = t / 64 (a wave = one patch row).  Per step and thread ONE 16-byte global
// load brings the plane above; the plane below and the centre are the z-march registers; +-1 are lane shuffles inside
// the row (its two end lanes read the neighbouring patch); and +-sdx -- the patch rows above and below, which the
// 512-consecutive-cells tile had to fetch with two more global loads per thread -- are the centre values the
// workgroup's other threads hold, passed through LDS: when a thread's plane-above pair arrives it is stored to the
// buffer the NEXT step reads (two buffers, one barrier per step).  Only the patch's first and last row go to memory
// for their outer neighbour.  Global loads per wave and step: 1 full + the rim (two of the four waves) + 2 lanes of edge,
// where the linear tile issues 3 full + 2 lanes.  Same products in the same order: A*x is bit-identical; the dot
// products are summed in this thread -> cell assignment, which ec3d_geom::patch_x tells the oracle's twin.
// How patch_pair makes a step's requests, per operand vector (A/B switches of round 6; the defaults are what was measured
// best at 512^3): 0 = every request of the step together, forms behind the LDS exchange; 1 = the rim row's requests behind
// the barrier (the neighbouring patch's requests for the same lines are then out already); 2 = the round-5 step (edge
// lanes and rim row each formed inside their branch, behind a wait for everything in flight); 3 = the rim row behind the
// arrival of the plane above; 4 = the rim row at the END of the step, its operands carried into the next step unformed;
// 5 = as 3, and the edge cells a plane ahead with the rim row (a step's first requests are the plane above and nothing else)
#ifndef EC3D_PP_PLAIN
#define EC3D_PP_PLAIN 2
#endif
#ifndef EC3D_PP_FUSED
#define EC3D_PP_FUSED 4
#endif
#ifndef EC3D_PP_FUSEDP
#define EC3D_PP_FUSEDP 5
#endif
template <class V> struct PatchReq { static constexpr int mode = EC3D_PP_PLAIN; };
template <> struct PatchReq<VecFused> { static constexpr int mode = EC3D_PP_FUSED; };
template <> struct PatchReq<VecFusedP> { static constexpr int mode = EC3D_PP_FUSEDP; };
template <int FMT, bool NTB, class V>
__device__ __forceinline__ void patch_pair_branchy(const MatDev<FMT> &A, const double *tbl, double *pbuf, int step, const V &x,
                                           int64_t r, bool first, ZRegs &z, double &s0, double &s1, d2 &ctr)
{
    // (Round 6 measured the plane above requested a step AHEAD here, as the interleaved march of the structured form does:
    // K2-in-K3 680 -> 818 us, K5-in-K1 1227 -> 1275 us at 512^3 -- behind the barrier of the LDS exchange the four waves of a
    // workgroup wait for the slowest one's request at the end of EVERY step; profiles/r06_patch_plane_ahead_512.log.)
    constexpr int HX = EC3D_PX / 2; // lanes per patch row
    const int t = threadIdx.x, y = t / HX, q = t % HX;
    const int64_t sdx = A.off[5], kdz = A.off[6];
    double *cur = pbuf + (step & 1) * EC3D_TILE, *nxt = pbuf + ((step + 1) & 1) * EC3D_TILE;
    unsigned short cc = 0;
    if constexpr (FMT == FMT_DICT7) cc = *reinterpret_cast<const unsigned short *>(A.cls + r);
    d2 c[7];
    if constexpr (FMT == FMT_DIA7) {
#pragma unroll
        for (int b = 0; b < 7; ++b) c[b] = load2<NTB>(A.band[b] + r);
    }
    // every global load of the step first
    const d2 zp = x.pair(r + kdz);
    double left = 0.0, right = 0.0;
    if (q == 0) left = x.at(r - 1);
    if (q == HX - 1) right = x.at(r + 2);
    // The patch's first and last row take their outer neighbour from memory -- one plane AHEAD: the row asked for now
    // is the one beside the plane above, i.e. the lines the neighbouring patch is asking for at this very step as ITS
    // plane above, so the two requests meet in the L2.  Asked for a step later (beside the centre plane, as the stencil
    // reads it) the lines had to survive a whole step of every workgroup of the XCD: with the three operand vectors of
    // K5-in-K1 they did not (PMC 56.3 B/row against 49; 52.3 now).  The vectors' ghost zones cover the row beside the plane
    // above the last one (ec3d_prepare_vectors); it is never used.
    const bool rimrow = y == 0 || y == EC3D_PY - 1;
    const int64_t roff = y == 0 ? -sdx : sdx;
    d2 rimc = d2{0.0, 0.0};
    if (rimrow) {
        rimc = first ? x.pair(r + roff) : z.rim;
        z.rim = x.pair(r + kdz + roff);
    }
    d2 ym = rimc, yp = rimc;
    d2 zm, cx;
    if (first) { // nothing carried over: plane below and centre from memory, centre into this step's buffer
        zm = x.pair(r - kdz);
        cx = x.pair(r);
        *reinterpret_cast<d2 *>(cur + 2 * t) = cx;
    } else {
        zm = z.xm;
        cx = z.xc;
    }
    // this step's buffer is complete (written at the end of the previous step, or just now).  A raw barrier behind a
    // wait for the LDS writes only: __syncthreads() also drains every global load in flight (vmcnt(0)) before the
    // barrier, which would put the LDS exchange BEHIND the arrival of the plane above instead of beside it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (y > 0) ym = *reinterpret_cast<const d2 *>(cur + 2 * (t - HX));
    if (y < EC3D_PY - 1) yp = *reinterpret_cast<const d2 *>(cur + 2 * (t + HX));
    ctr = cx;
    {
        const double l = __shfl_up(cx.y, 1, 64), rr = __shfl_down(cx.x, 1, 64);
        if (q != 0) left = l;
        if (q != HX - 1) right = rr;
    }
    if constexpr (FMT == FMT_DIA7) {
        s0 = c[0].x * zm.x;
        s1 = c[0].y * zm.y;
        s0 = s0 + c[1].x * ym.x;
        s1 = s1 + c[1].y * ym.y;
        s0 = s0 + c[2].x * left;
        s1 = s1 + c[2].y * cx.x;
        s0 = s0 + c[3].x * cx.x;
        s1 = s1 + c[3].y * cx.y;
        s0 = s0 + c[4].x * cx.y;
        s1 = s1 + c[4].y * right;
        s0 = s0 + c[5].x * yp.x;
        s1 = s1 + c[5].y * yp.y;
        s0 = s0 + c[6].x * zp.x;
        s1 = s1 + c[6].y * zp.y;
    } else {
        const double *t0 = tbl + (cc & 0xFF) * 7, *t1 = tbl + (cc >> 8) * 7;
        s0 = t0[0] * zm.x;
        s1 = t1[0] * zm.y;
        s0 = s0 + t0[1] * ym.x;
        s1 = s1 + t1[1] * ym.y;
        s0 = s0 + t0[2] * left;
        s1 = s1 + t1[2] * cx.x;
        s0 = s0 + t0[3] * cx.x;
        s1 = s1 + t1[3] * cx.y;
        s0 = s0 + t0[4] * cx.y;
        s1 = s1 + t1[4] * right;
        s0 = s0 + t0[5] * yp.x;
        s1 = s1 + t1[5] * yp.y;
        s0 = s0 + t0[6] * zp.x;
        s1 = s1 + t1[6] * zp.y;
    }
    // the plane above is the next step's centre: into the other buffer (nobody reads that one before the next barrier)
    *reinterpret_cast<d2 *>(nxt + 2 * t) = zp;
    z.xm = cx;
    z.xc = zp;
}

template <int FMT, bool NTB, class V>
__device__ __forceinline__ void patch_pair(const MatDev<FMT> &A, const double *tbl, double *pbuf, int step, const V &x,
                                           int64_t r, bool first, ZRegs &z, double &s0, double &s1, d2 &ctr)
{
    // (Round 6 measured the plane above requested a step AHEAD here, as the interleaved march of the structured form does:
    // K2-in-K3 680 -> 818 us, K5-in-K1 1227 -> 1275 us at 512^3 -- behind the barrier of the LDS exchange the four waves of a
    // workgroup wait for the slowest one's request at the end of EVERY step; profiles/r06_patch_plane_ahead_512.log.)
    constexpr int RQ = PatchReq<V>::mode;
    if constexpr (RQ == 2) {
        patch_pair_branchy<FMT, NTB>(A, tbl, pbuf, step, x, r, first, z, s0, s1, ctr);
        return;
    }
    constexpr int HX = EC3D_PX / 2; // lanes per patch row
    const int t = threadIdx.x, y = t / HX, q = t % HX;
    const int64_t sdx = A.off[5], kdz = A.off[6];
    double *cur = pbuf + (step & 1) * EC3D_TILE, *nxt = pbuf + ((step + 1) & 1) * EC3D_TILE;
    unsigned short cc = 0;
    if constexpr (FMT == FMT_DICT7) cc = *reinterpret_cast<const unsigned short *>(A.cls + r);
    d2 c[7];
    if constexpr (FMT == FMT_DIA7) {
#pragma unroll
        for (int b = 0; b < 7; ++b) c[b] = load2<NTB>(A.band[b] + r);
    }
    // Every request of the step first, and nothing formed from any of them yet (round 6): with a fused vector
    // (S = R - alpha*AP, P = R + beta*(P - omega*AP)) an x.at() / x.pair() under `if` kept its arithmetic inside the branch,
    // and the branch then began with a wait for EVERYTHING in flight -- the plane above included: the edge lanes (every
    // wave has them), then the rim row, each cost the wave a memory round trip of its own before the barrier.  Now: the
    // raw operands of plane above, edge cell (ONE predicated request for both edge lanes, as sav_band_loads) and rim
    // row go out together, the LDS exchange runs beside them, and the forms follow behind it, by every lane (a lane
    // that requested nothing forms zeros).  Same expressions on the same operands: the same bits.
    const typename V::Raw2 zp_r = x.rpair(r + kdz);
    const bool edge = q == 0 || q == HX - 1;
    typename V::Raw1 e_r{};
    if (edge && (RQ != 5 || first)) e_r = x.rat(q == 0 ? r - 1 : r + 2);
    // The patch's first and last row take their outer neighbour from memory -- one plane AHEAD: the row asked for now
    // is the one beside the plane above, i.e. the lines the neighbouring patch is asking for at this very step as ITS
    // plane above, so the two requests meet in the L2.  Asked for a step later (beside the centre plane, as the stencil
    // reads it) the lines had to survive a whole step of every workgroup of the XCD: with the three operand vectors of
    // K5-in-K1 they did not (PMC 56.3 B/row against 49; 52.3 now).  The vectors' ghost zones cover the row beside the plane
    // above the last one (ec3d_prepare_vectors); it is never used.
    const bool rimrow = y == 0 || y == EC3D_PY - 1;
    const int64_t roff = y == 0 ? -sdx : sdx;
    typename V::Raw2 rimc_r{}, rimn_r{};
    if (rimrow) {
        if (first) rimc_r = x.rpair(r + roff);
        if (RQ == 0 || (RQ == 1 && first)) rimn_r = x.rpair(r + kdz + roff);
    }
    d2 zm, cx;
    if (first) { // nothing carried over: plane below and centre from memory, centre into this step's buffer
        const typename V::Raw2 zm_r = x.rpair(r - kdz), cx_r = x.rpair(r);
        zm = x.form(zm_r);
        cx = x.form(cx_r);
        *reinterpret_cast<d2 *>(cur + 2 * t) = cx;
    } else {
        zm = z.xm;
        cx = z.xc;
    }
    // this step's buffer is complete (written at the end of the previous step, or just now).  A raw barrier behind a
    // wait for the LDS writes only: __syncthreads() also drains every global load in flight (vmcnt(0)) before the
    // barrier, which would put the LDS exchange BEHIND the arrival of the plane above instead of beside it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (RQ == 1) {
        if (rimrow && !first) rimn_r = x.rpair(r + kdz + roff);
    }
    d2 rimc;
    if constexpr (RQ == 4) rimc = first ? x.form(rimc_r) : x.form(V::from_carry(z.cr));
    else rimc = first ? x.form(rimc_r) : z.rim;
    d2 ym = rimc, yp = rimc;
    if (y > 0) ym = *reinterpret_cast<const d2 *>(cur + 2 * (t - HX));
    if (y < EC3D_PY - 1) yp = *reinterpret_cast<const d2 *>(cur + 2 * (t + HX));
    ctr = cx;
    double left, right;
    {
        double ev = x.form(e_r);
        if constexpr (RQ == 5) ev = first ? ev : z.edge;
        const double l = __shfl_up(cx.y, 1, 64), rr = __shfl_down(cx.x, 1, 64);
        left = q != 0 ? l : ev;
        right = q != HX - 1 ? rr : ev;
    }
    const d2 zp = x.form(zp_r);
    typename V::Raw1 en_r{};
    if constexpr (RQ == 3 || RQ == 5) { // the rim row behind the arrival of the plane above: one more round trip, beside the sums
        if (rimrow) rimn_r = x.rpair(r + kdz + roff);
        if constexpr (RQ == 5) { // ... and the edge cells of the plane above, for the next step
            if (edge) en_r = x.rat(q == 0 ? r + kdz - 1 : r + kdz + 2);
        }
    }
    if constexpr (FMT == FMT_DIA7) {
        s0 = c[0].x * zm.x;
        s1 = c[0].y * zm.y;
        s0 = s0 + c[1].x * ym.x;
        s1 = s1 + c[1].y * ym.y;
        s0 = s0 + c[2].x * left;
        s1 = s1 + c[2].y * cx.x;
        s0 = s0 + c[3].x * cx.x;
        s1 = s1 + c[3].y * cx.y;
        s0 = s0 + c[4].x * cx.y;
        s1 = s1 + c[4].y * right;
        s0 = s0 + c[5].x * yp.x;
        s1 = s1 + c[5].y * yp.y;
        s0 = s0 + c[6].x * zp.x;
        s1 = s1 + c[6].y * zp.y;
    } else {
        const double *t0 = tbl + (cc & 0xFF) * 7, *t1 = tbl + (cc >> 8) * 7;
        s0 = t0[0] * zm.x;
        s1 = t1[0] * zm.y;
        s0 = s0 + t0[1] * ym.x;
        s1 = s1 + t1[1] * ym.y;
        s0 = s0 + t0[2] * left;
        s1 = s1 + t1[2] * cx.x;
        s0 = s0 + t0[3] * cx.x;
        s1 = s1 + t1[3] * cx.y;
        s0 = s0 + t0[4] * cx.y;
        s1 = s1 + t1[4] * right;
        s0 = s0 + t0[5] * yp.x;
        s1 = s1 + t1[5] * yp.y;
        s0 = s0 + t0[6] * zp.x;
        s1 = s1 + t1[6] * zp.y;
    }
    // the plane above is the next step's centre: into the other buffer (nobody reads that one before the next barrier)
    *reinterpret_cast<d2 *>(nxt + 2 * t) = zp;
    z.xm = cx;
    z.xc = zp;
    if constexpr (RQ == 4) { // the rim row requested now, waited for behind the next step's barrier
        if (rimrow) rimn_r = x.rpair(r + kdz + roff);
        V::to_carry(rimn_r, z.cr);
    } else {
        z.rim = x.form(rimn_r);
        if constexpr (RQ == 5) z.edge = x.form(en_r);
    }
}

// rows r, r+1 of A*x (src/solvers.f90:58-59): bands in ascending column order, then the tail.
// ZM: band 0 / 3 / 6 (offsets -kdz, 0, +kdz) come from / go to the registers `z`.
// `ctr` returns x[r], x[r+1] (the centre band's operand).
template <int FMT, bool ZM, bool TAIL, bool NTB, bool PATCH, class V, class SW>
__device__ __forceinline__ void spmv_pair(const MatDev<FMT> &A, const SW &sw, const PatchPos &pp, const double *tbl,
                                          double *stg, int &step, const V &x, int64_t r, int64_t tile, bool first,
                                          ZRegs &z, double &s0, double &s1, d2 &ctr)
{
    if constexpr (FMT == FMT_SAV && ZM && PATCH) {
        sav_patch_step(A, sw, tbl, stg, step++, x, pp, first, z, s0, s1, ctr);
        return;
    }
    if constexpr (FMT == FMT_SAV && ZM && !PATCH) {
        if constexpr (std::is_same<V, VecPlain>::value) sav_pair_zm(A, tbl, stg, x, r, tile, first, z, s0, s1, ctr);
        return;
    }
    if constexpr (PATCH && (FMT == FMT_DIA7 || FMT == FMT_DICT7)) {
        patch_pair<FMT, NTB>(A, tbl, stg, step++, x, r, first, z, s0, s1, ctr);
        return;
    }
    uint8_t tflag = 0; // per tile: any tail row (bands + tail) / any coupled row (structured form)
    if constexpr (FMT == FMT_DIA7 || FMT == FMT_DICT7 || FMT == FMT_SAV) {
        d2 xv[7];
        // the +-1 neighbours (bands 2 and 4 of the 7-point operator) are the centre pairs of the
        // adjacent lanes: take them by lane shuffle instead of two unaligned 16-byte loads; only the
        // first/last lane of a wave reads its outer neighbour from memory.  The SpMV kernels are
        // bound by L1/TA load issue, not by HBM, so 3 full-wave loads per step instead of 5 matter.
        const bool pm1 = ZM ? true : A.pm1 != 0; // a z-marching grid is the 7-point one (choose_sweep)
        // every load of the step is issued before the first use: class bytes, the two edge-lane
        // neighbours, then the band operands (one round trip per step instead of three)
        unsigned short cc = 0;
        if constexpr (FMT == FMT_DICT7 || FMT == FMT_SAV) cc = *reinterpret_cast<const unsigned short *>(A.cls + r);
        if constexpr (FMT == FMT_SAV) tflag = A.tile_flag[tile];
        else if constexpr (TAIL) tflag = A.t.tile_flag[tile];
        const int lane = threadIdx.x & 63;
        double left = 0.0, right = 0.0;
        if (pm1) {
            if (lane == 0) left = x.at(r - 1);
            if (lane == 63) right = x.at(r + 2);
        }
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            if (ZM && (b == 0 || b == 3) && !first) continue;
            if (pm1 && (b == 2 || b == 4)) continue;
            xv[b] = x.pair(r + A.boff(b));
        }
        if (ZM) {
            if (!first) { xv[0] = z.xm; xv[3] = z.xc; }
            z.xm = xv[3];
            z.xc = xv[6];
        }
        ctr = xv[3];
        if (pm1) {
            const double l = __shfl_up(ctr.y, 1, 64), rr = __shfl_down(ctr.x, 1, 64);
            if (lane != 0) left = l;
            if (lane != 63) right = rr;
            xv[2] = d2{left, ctr.x};
            xv[4] = d2{ctr.y, right};
        }
        if constexpr (FMT == FMT_DIA7) {
            d2 c[7];
#pragma unroll
            for (int b = 0; b < 7; ++b) c[b] = load2<NTB>(A.band[b] + r); // coefficients: touched once per launch
            s0 = c[0].x * xv[0].x;
            s1 = c[0].y * xv[0].y;
#pragma unroll
            for (int b = 1; b < 7; ++b) {
                s0 = s0 + c[b].x * xv[b].x;
                s1 = s1 + c[b].y * xv[b].y;
            }
        } else if constexpr (FMT == FMT_DICT7) {
            const double *t0 = tbl + (cc & 0xFF) * 7, *t1 = tbl + (cc >> 8) * 7;
            s0 = t0[0] * xv[0].x;
            s1 = t1[0] * xv[0].y;
#pragma unroll
            for (int b = 1; b < 7; ++b) {
                s0 = s0 + t0[b] * xv[b].x;
                s1 = s1 + t1[b] * xv[b].y;
            }
        } else { // FMT_SAV without tile-aligned planes: U rows take their A couplings first, A rows their U couplings last
            const int c0 = cc & 0xFF, c1 = cc >> 8;
            const double *t0 = tbl + c0 * EC3D_SAV_STRIDE, *t1 = tbl + c1 * EC3D_SAV_STRIDE;
            const bool cpl = tflag != 0; // any coupled row in this tile (uniform)
            // the coupling slots of an A class mean U columns, those of a U class A columns: a row of the
            // other kind goes through the all-zero class
            const double *zt = tbl + A.zero * EC3D_SAV_STRIDE;
            const int64_t nA = 3 * A.nC;
            s0 = 0.0;
            s1 = 0.0;
            if (cpl && r + 1 >= nA) sav_u_pre(A, r >= nA ? t0 : zt, t1, x, r, s0, s1);
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                s0 = s0 + t0[b] * xv[b].x;
                s1 = s1 + t1[b] * xv[b].y;
            }
            if (cpl && r < nA) {
                EC3D_PIN(s0, s1); // the band operands are dead from here: their registers take the slots'
                const int d0 = (r >= A.nC) + (r >= 2 * A.nC);
                const int d1 = (r + 1 >= A.nC) + (r + 1 >= 2 * A.nC);
                const bool in0 = c0 >= A.a0 && c0 < A.u0, in1 = c1 >= A.a0 && c1 < A.u0 && r + 1 < nA;
                if (d0 == d1) {
                    sav_a_post(A, in0 ? t0 : zt, in1 ? t1 : zt, x, r, d0, s0, s1);
                } else { // the pair straddles two component blocks (odd block length)
                    double dummy = 0.0;
                    if (in0) sav_a_post(A, t0, zt, x, r, d0, s0, dummy);
                    if (in1) sav_a_post(A, zt, t1, x, r, d1, dummy, s1);
                }
            }
        }
    } else {
        s0 = 0.0;
        s1 = 0.0;
        for (int b = 0; b < A.nb; ++b) {
            const d2 c = *reinterpret_cast<const d2 *>(A.band[b] + r);
            const d2 xv = x.pair(r + A.off[b]);
            s0 = s0 + c.x * xv.x;
            s1 = s1 + c.y * xv.y;
        }
        ctr = x.pair(r);
        if constexpr (TAIL) tflag = A.t.tile_flag[tile];
    }
    if constexpr (TAIL && FMT != FMT_SAV) {
        if (tflag) {
            i2 t = *reinterpret_cast<const i2 *>(A.t.tail_id + r);
            if (t.x >= 0) s0 = tail_add(A.t, x, t.x, s0);
            if (t.y >= 0) s1 = tail_add(A.t, x, t.y, s1);
        }
    }
}

// the SpMV grids are sized for 6 workgroups per CU (choose_sweep): keep the register count within that
// (the structured form without z-marching only runs on grids too small for plane-aligned tiles: it takes the
// registers it needs -- 5 per CU -- instead of spilling)
// (the interleaved z-march of the structured form -- TAIL stands for it there, EC3D_IL -- holds the band operands of
// four blocks at once: two workgroups per CU)
#define EC3D_IL (FMT == FMT_SAV && ZM && TAIL && !PATCH)
#define EC3D_SPMV_OCC __attribute__((amdgpu_waves_per_eu(EC3D_IL ? 2 : (FMT == FMT_SAV && !ZM) ? 5 : 6)))
#define EC3D_SPMV_T template <int FMT, bool NT, bool ZM, bool TAIL, bool PATCH>
#define EC3D_SWEEP_OF(ZM_) typename SweepSel<ZM_>::type
// classes in the LDS table (0 for the formats without one)
template <int FMT> __device__ __forceinline__ int ncls_of(const MatDev<FMT> &A)
{
    if constexpr (FMT == FMT_SAV || FMT == FMT_DICT7) return A.ncls;
    return 0;
}
#define EC3D_TBL_DECL                                                                          \
    extern __shared__ double tbl[]; /* the class table, sized at launch (EC3D_TBL_BYTES) */    \
    double *stg = tbl + (FMT == FMT_SAV ? ncls_of<FMT>(A) * EC3D_SAV_STRIDE : ((ncls_of<FMT>(A) * 7 + 1) & ~1)) /* staging slots (sav_pair_zm) / centre-plane buffers (patch_pair) behind the table */

// ---------------------------------------------------------------------------------------------
// plain y = A x  (src/solvers.f90:54-61)
EC3D_SPMV_T __global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k_spmv(MatDev<FMT> A, EC3D_SWEEP_OF(ZM) sw,
                                                                                   const double *__restrict__ x,
                                                                                   double *__restrict__ y)
{
    EC3D_TBL_DECL;
    stage_table<FMT>(A, tbl);
    ZRegs zr;
    PatchPos pp{};
    int pstep = 0; // steps taken (2-D tiles: which of the two LDS buffers holds the centre plane)
    if constexpr (EC3D_IL) {
        walk_zm_il(A, sw, tbl, VecPlain{x}, [](int64_t) { return d2{0.0, 0.0}; },
                   [&](int64_t r, double s0, double s1, d2, d2) { store2<NT>(y, r, INT64_MAX, s0, s1); });
    } else {
        walk_spmv<FMT, ZM, FMT != FMT_SAV, PATCH>(A, sw, pp, [&](int64_t tile, auto fc) {
            EC3D_ROW_S;
            double s0, s1;
            d2 ctr;
            spmv_pair<FMT, ZM, TAIL, NT, PATCH>(A, sw, pp, tbl, stg, pstep, VecPlain{x}, r, tile, (bool)fc, zr, s0, s1, ctr);
            store2<NT>(y, r, nst, s0, s1);
        });
    }
}

// setup: R = B - A X ; R0 = R ; P = R ; partials of B·B and R·R   (src/solvers.f90:14-21)
EC3D_SPMV_T __global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k_residual(
    MatDev<FMT> A, EC3D_SWEEP_OF(ZM) sw, const double *__restrict__ x, const double *__restrict__ b,
    double *__restrict__ rv, double *__restrict__ r0, double *__restrict__ p, double *__restrict__ part)
{
    __shared__ double lds[8];
    EC3D_TBL_DECL;
    stage_table<FMT>(A, tbl);
    ZRegs zr;
    PatchPos pp{};
    int pstep = 0; // steps taken (2-D tiles: which of the two LDS buffers holds the centre plane)
    double acc[2] = {0.0, 0.0};
    if constexpr (EC3D_IL) {
        walk_zm_il(A, sw, tbl, VecPlain{x}, [&](int64_t r) { return load2<NT>(b + r); },
                   [&](int64_t r, double s0, double s1, d2, d2 bv) {
                       double e0 = bv.x - s0, e1 = bv.y - s1, b0 = bv.x, b1 = bv.y;
                       store2<NT>(rv, r, INT64_MAX, e0, e1); // (every row of the interleaved march is a row of the system)
                       store2<NT>(r0, r, INT64_MAX, e0, e1);
                       store2<NT>(p, r, INT64_MAX, e0, e1);
                       acc[0] = acc[0] + b0 * b0;
                       acc[0] = acc[0] + b1 * b1;
                       acc[1] = acc[1] + e0 * e0;
                       acc[1] = acc[1] + e1 * e1;
                   });
    } else {
        walk_spmv<FMT, ZM, FMT != FMT_SAV, PATCH>(A, sw, pp, [&](int64_t tile, auto fc) {
            EC3D_ROW_S;
            double s0, s1;
            d2 ctr;
            spmv_pair<FMT, ZM, TAIL, NT, PATCH>(A, sw, pp, tbl, stg, pstep, VecPlain{x}, r, tile, (bool)fc, zr, s0, s1, ctr);
            d2 bv = load2<NT>(b + r);
            double e0 = bv.x - s0, e1 = bv.y - s1, b0 = bv.x, b1 = bv.y;
            store2<NT>(rv, r, nst, e0, e1);
            store2<NT>(r0, r, nst, e0, e1);
            store2<NT>(p, r, nst, e0, e1);
            EC3D_MASK2(r, sw, e0, e1);
            EC3D_IDLE2(e0, e1);
            EC3D_MASK2(r, sw, b0, b1);
            EC3D_IDLE2(b0, b1);
            acc[0] = acc[0] + b0 * b0;
            acc[0] = acc[0] + b1 * b1;
            acc[1] = acc[1] + e0 * e0;
            acc[1] = acc[1] + e1 * e1;
        });
    }
    block_sum<2>(acc, lds);
    {
        const int pslot_[2] = {P_BB, P_RR_INIT};
        publish_partials<2>(sw, part, pslot_, acc, lds);
    }
}

// multi-rank only: collapse this rank's per-workgroup partials of the slots in `mask` into lsum[slot]
// (same tree as reduce_partials), ready for the all_gather
__global__ __launch_bounds__(EC3D_THREADS) void k_finalize(RedSrc src, double *lsum, unsigned mask)
{
    __shared__ double lds[4];
    for (int sl = 0; sl < P_NSLOT; ++sl) {
        if (!(mask & (1u << sl))) continue;
        const int slot[1] = {sl};
        double v[1];
        reduce_partials<1>(src, slot, v, lds);
        if (threadIdx.x == 0) lsum[sl] = v[0];
    }
}

__global__ __launch_bounds__(EC3D_THREADS) void k_finalize2(RedSrc a, unsigned mask_a, RedSrc b, unsigned mask_b, double *lsum)
{
    __shared__ double lds[4];
    for (int pass = 0; pass < 2; ++pass) {
        const RedSrc &src = pass ? b : a;
        const unsigned mask = pass ? mask_b : mask_a;
        for (int sl = 0; sl < P_NSLOT; ++sl) {
            if (!(mask & (1u << sl))) continue;
            const int slot[1] = {sl};
            double v[1];
            reduce_partials<1>(src, slot, v, lds);
            if (threadIdx.x == 0) lsum[sl] = v[0];
        }
    }
}

// The exit state travels as ONE 64-bit word: stop_iter in the low half, stop_kind in the high half
// (SolverState keeps them adjacent and 8-byte aligned).  A reader never sees the iteration of one exit with
// the kind of another, whoever wrote it and whenever: K5's entry test reads a pair that its own launch's lead
// thread may be writing at that moment.
__device__ __forceinline__ unsigned long long *stop_word(const SolverState *st)
{
    return reinterpret_cast<unsigned long long *>(const_cast<int *>(&st->stop_iter));
}
__device__ __forceinline__ void stop_publish(SolverState *st, int it, int kind)
{
    __hip_atomic_store(stop_word(st), (unsigned long long)(unsigned)it | ((unsigned long long)(unsigned)kind << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stop_read(const SolverState *st, int &it, int &kind)
{
    // one indivisible 64-bit load with the ordinary cache policy: agent scope (sc1) sends every wave of the launch
    // to the same L2 line past its L1 -- measured +35..80 us per kernel at 512^3.  A value from the L1 is at worst
    // an OLDER pair, which the entry tests treat like "not yet published".
    const unsigned long long w = __hip_atomic_load(stop_word(st), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    it = (int)(unsigned)(w & 0xFFFFFFFFull);
    kind = (int)(unsigned)(w >> 32);
}
// everything requested above this line stays above it (the compiler would otherwise sink a load to its first use,
// behind the exit test -- and the requests would go out one after the other again)
#define EC3D_REQUESTS_OUT asm volatile("" ::: "memory")
__device__ __forceinline__ int stop_iter_of(const SolverState *st)
{
    int it, kind;
    stop_read(st, it, kind);
    return it;
}

// Bnorm, rr0, "‖b‖ = 0 -> return" (src/solvers.f90:21-23); one workgroup
__global__ __launch_bounds__(EC3D_THREADS) void k_setup(SolverState *st, RedSrc src, double tol)
{
    __shared__ double lds[8];
    const int slot[2] = {P_BB, P_RR_INIT};
    double v[2];
    reduce_partials<2>(src, slot, v, lds);
    if (threadIdx.x == 0) {
        const double bnorm = sqrt(v[0]);
        st->bnorm = bnorm;
        st->tol = tol;
        st->rr0[1] = v[1]; // iteration 1 reads rr0[1 & 1]
        st->rr0[0] = 0.0;
        st->alpha = 0.0;
        st->omega = 0.0;
        st->restarts = 0;
        st->npend = 0;
        st->pend_half = 0;
        st->rnorm = sqrt(v[1]);
        stop_publish(st, (bnorm == 0.0) ? 0 : INT_MAX, 0);
    }
}

// K1: AP = A P ; partial AP·R0    (src/solvers.f90:30, :32 denominator)
EC3D_SPMV_T __global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k1_spmv_dot(
    MatDev<FMT> A, EC3D_SWEEP_OF(ZM) sw, const SolverState *st, int it, const double *__restrict__ p,
    const double *__restrict__ r0, double *__restrict__ ap, double *__restrict__ part)
{
    __shared__ double lds[4];
    EC3D_TBL_DECL;
    TableEarly te;
    table_request<FMT>(A, te); // with the exit word: one round trip before the first tile instead of two
    const int stop_it = stop_iter_of(st);
    EC3D_REQUESTS_OUT;
    if (stop_it < it) return;
    table_store<FMT>(A, te, tbl);
    ZRegs zr;
    PatchPos pp{};
    int pstep = 0; // steps taken (2-D tiles: which of the two LDS buffers holds the centre plane)
    double acc[1] = {0.0};
    if constexpr (EC3D_IL) {
        walk_zm_il(A, sw, tbl, VecPlain{p}, [&](int64_t r) { return load2<NT>(r0 + r); },
                   [&](int64_t r, double s0, double s1, d2, d2 q) {
                       store2k<NT>(ap, r, INT64_MAX, s0, s1, keep_of(sw) & EC3D_KEEP_AP);
                       acc[0] = acc[0] + s0 * q.x;
                       acc[0] = acc[0] + s1 * q.y;
                   });
    } else {
        walk_spmv<FMT, ZM, FMT != FMT_SAV, PATCH>(A, sw, pp, [&](int64_t tile, auto fc) {
            EC3D_ROW_S;
            double s0, s1;
            d2 ctr;
            // R0's pair: requested before the step's own loads in the structured kernels (A-V K1 128 -> 124 us; their
            // steps wait on LDS-DMA slots), after them elsewhere (2-D tiles: 593 vs 603 us at 512^3, 76 vs 80 at 256^3)
            d2 q;
            if constexpr (FMT == FMT_SAV && !PATCH) q = load2<NT>(r0 + r);
            spmv_pair<FMT, ZM, TAIL, NT, PATCH>(A, sw, pp, tbl, stg, pstep, VecPlain{p}, r, tile, (bool)fc, zr, s0, s1, ctr);
            if constexpr (FMT != FMT_SAV || PATCH) q = load2<NT>(r0 + r);
            store2k<NT>(ap, r, nst, s0, s1, keep_of(sw) & EC3D_KEEP_AP);
            EC3D_MASK2(r, sw, s0, s1);
            EC3D_IDLE2(s0, s1);
            acc[0] = acc[0] + s0 * q.x;
            acc[0] = acc[0] + s1 * q.y;
        });
    }
    block_sum<1>(acc, lds);
    {
        const int pslot_[1] = {P_D1};
        publish_partials<1>(sw, part, pslot_, acc, lds);
    }
}

// K2: alpha = rr0 / (AP·R0) ; S = R - alpha*AP ; partial S·S   (src/solvers.f90:31-34)
template <bool NT>
__global__ __launch_bounds__(EC3D_THREADS) void k2_s_update(SweepV sw, RedSrc src, SolverState *st, int it,
                                                            const double *__restrict__ rv,
                                                            const double *__restrict__ ap, double *__restrict__ sv,
                                                            double *__restrict__ part)
{
    __shared__ double lds[4];
    const int slot[1] = {P_D1};
    PartialsEarly<1> pe;
    partials_request<1>(src, slot, pe); // exit word, partials and rr0 in one round trip (see partials_request)
    const double rr0 = st->rr0[it & 1];
    const int stop_it = stop_iter_of(st);
    EC3D_REQUESTS_OUT;
    if (stop_it < it) return;
    double d[1];
    partials_finish<1>(src, slot, pe, d, lds);
    const double alpha = rr0 / d[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) st->alpha = alpha;
    double acc[1] = {0.0};
    struct Ops { d2 a, q; };
    walk_vec(sw, [&](int64_t tile) {
        EC3D_ROW;
        return Ops{load2<NT>(ap + r), load2<NT>(rv + r)};
    }, [&](int64_t tile, const Ops &o) {
        EC3D_ROW;
        const d2 a = o.a, q = o.q;
        double s0 = q.x - alpha * a.x, s1 = q.y - alpha * a.y;
        store2k<NT>(sv, r, sw.n, s0, s1, keep_of(sw) & EC3D_KEEP_S);
        EC3D_MASK2(r, sw, s0, s1);
        acc[0] = acc[0] + s0 * s0;
        acc[0] = acc[0] + s1 * s1;
    });
    block_sum<1>(acc, lds);
    {
        const int pslot_[1] = {P_SS};
        publish_partials<1>(sw, part, pslot_, acc, lds);
    }
}

// K3: AS = A S ; partials AS·S and AS·AS (src/solvers.f90:39-40).  Launched before ‖S‖ is known
// (one global reduction point less per iteration): when the ‖S‖ exit of :34-38 is then
// taken by K4, AS is simply never used -- results are unchanged.
EC3D_SPMV_T __global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k3_spmv_dots(
    MatDev<FMT> A, EC3D_SWEEP_OF(ZM) sw, SolverState *st, int it, const double *__restrict__ sv,
    double *__restrict__ as, double *__restrict__ part)
{
    __shared__ double lds[8];
    EC3D_TBL_DECL;
    TableEarly te;
    table_request<FMT>(A, te);
    const int stop_it = stop_iter_of(st);
    EC3D_REQUESTS_OUT;
    if (stop_it < it) return;
    table_store<FMT>(A, te, tbl);
    ZRegs zr;
    PatchPos pp{};
    int pstep = 0; // steps taken (2-D tiles: which of the two LDS buffers holds the centre plane)
    double acc[2] = {0.0, 0.0};
    if constexpr (EC3D_IL) {
        walk_zm_il(A, sw, tbl, VecPlain{sv}, [](int64_t) { return d2{0.0, 0.0}; },
                   [&](int64_t r, double s0, double s1, d2 q, d2) {
                       store2<NT>(as, r, INT64_MAX, s0, s1);
                       acc[0] = acc[0] + s0 * q.x;
                       acc[0] = acc[0] + s1 * q.y;
                       acc[1] = acc[1] + s0 * s0;
                       acc[1] = acc[1] + s1 * s1;
                   });
    } else {
        walk_spmv<FMT, ZM, FMT != FMT_SAV, PATCH>(A, sw, pp, [&](int64_t tile, auto fc) {
            EC3D_ROW_S;
            double s0, s1;
            d2 q;
            spmv_pair<FMT, ZM, TAIL, NT, PATCH>(A, sw, pp, tbl, stg, pstep, VecPlain{sv}, r, tile, (bool)fc, zr, s0, s1, q);
            store2<NT>(as, r, nst, s0, s1);
            EC3D_MASK2(r, sw, s0, s1);
            EC3D_IDLE2(s0, s1);
            acc[0] = acc[0] + s0 * q.x;
            acc[0] = acc[0] + s1 * q.y;
            acc[1] = acc[1] + s0 * s0;
            acc[1] = acc[1] + s1 * s1;
        });
    }
    block_sum<2>(acc, lds);
    {
        const int pslot_[2] = {P_D2, P_D3};
        publish_partials<2>(sw, part, pslot_, acc, lds);
    }
}

// K2 + K3 in one launch (2-D tiles only, single rank): alpha = rr0 / (AP.R0); S = R - alpha*AP formed per plane where
// the stencil needs it (VecFused: the plane above from two loads instead of one, the +-sdx rows through LDS as before,
// rim and edge values recomputed by the lanes that need them) and stored by its owner; AS = A S; partials S.S, AS.S,
// AS.AS -- all three in the SpMV kernels' thread -> cell assignment.  Saves the 8 B per row S costs to re-read and one
// launch; the products, their order and every stored value are those of K2 followed by K3 (src/solvers.f90:31-40).
// HS (z-slab, Sweep::halo_store): an instance of its own, so that the single-GPU kernel carries nothing for it
template <int FMT, bool NT, bool ZM, bool TAIL, bool PATCH, bool HS = false>
__global__ __launch_bounds__(EC3D_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void k23_s_spmv_dots(
    MatDev<FMT> A, EC3D_SWEEP_OF(ZM) sw, RedSrc src, SolverState *st, int it, const double *__restrict__ rv,
    const double *__restrict__ ap, double *__restrict__ sv, double *__restrict__ as, double *__restrict__ part)
{
    __shared__ double lds[12];
    EC3D_TBL_DECL;
    const int slot[1] = {P_D1};
    PartialsEarly<1> pe;
    partials_request<1>(src, slot, pe);
    const double rr0 = st->rr0[it & 1];
    TableEarly te;
    table_request<FMT>(A, te);
    const int stop_it = stop_iter_of(st);
    EC3D_REQUESTS_OUT;
    if (stop_it < it) return;
    double d[1];
    partials_finish<1>(src, slot, pe, d, lds);
    const double alpha = rr0 / d[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) st->alpha = alpha;
    table_store<FMT>(A, te, tbl);
    ZRegs zr;
    PatchPos pp{};
    int pstep = 0;
    double acc[3] = {0.0, 0.0, 0.0};
    walk_spmv<FMT, ZM, FMT != FMT_SAV, PATCH>(A, sw, pp, [&](int64_t tile, auto fc) {
        EC3D_ROW_S;
        double s0, s1;
        d2 q; // S[r], S[r+1]
        spmv_pair<FMT, ZM, TAIL, NT, PATCH>(A, sw, pp, tbl, stg, pstep, VecFused{rv, ap, alpha}, r, tile, (bool)fc, zr, s0, s1, q);
        store2<NT>(sv, r, nst, q.x, q.y);
        if (as) store2<NT>(as, r, nst, s0, s1); // (nullptr: K4 runs in SpMV form and computes A S again, k4s_x_r_spmv)
        if constexpr (HS && FMT == FMT_DICT7 && ZM && PATCH) {
            // z-slab: S on the neighbours' planes, formed here from the exchanged R and AP exactly as their owners form it
            // (same expression, same operands, same alpha), goes into S's ghost rows for the stencil of K4 in SpMV form
            if (sw.hs_mask) {
                const int64_t pl = tile / sw.tpp, kdz = A.off[6];
                const VecFused xs{rv, ap, alpha};
                if ((sw.hs_mask & 1) && pl == 0) *reinterpret_cast<d2 *>(sv + r - kdz) = xs.pair(r - kdz);
                if ((sw.hs_mask & 2) && pl == sw.hs_last) *reinterpret_cast<d2 *>(sv + r + kdz) = xs.pair(r + kdz);
            }
        }
        double q0 = q.x, q1 = q.y;
        EC3D_MASK2(r, sw, q0, q1);
        EC3D_IDLE2(q0, q1);
        EC3D_MASK2(r, sw, s0, s1);
        EC3D_IDLE2(s0, s1);
        acc[0] = acc[0] + q0 * q0;
        acc[0] = acc[0] + q1 * q1;
        acc[1] = acc[1] + s0 * q.x;
        acc[1] = acc[1] + s1 * q.y;
        acc[2] = acc[2] + s0 * s0;
        acc[2] = acc[2] + s1 * s1;
    });
    block_sum<3>(acc, lds);
    {
        const int pslot_[3] = {P_SS, P_D2, P_D3};
        publish_partials<3>(sw, part, pslot_, acc, lds);
    }
}

// K5 + the next iteration's K1 in one launch (2-D tiles only, single rank): the exits and the restart rule of K5
// (src/solvers.f90:43-49), then P = R + beta*(P - omega*AP) formed per plane where the stencil of AP = A P needs it
// (VecFusedP) and stored by its owner, AP.R0 for iteration it + 1 (:30-32).  P and AP of iteration `it` are read from
// (p_old, ap_old) and those of it + 1 written to (p_new, ap_new): other workgroups read the old values of cells this
// one owns (rim, edge, the planes at a segment's ends), so the update cannot be in place.  Saves the 8 B per row P
// costs to re-read and one launch; every stored value and every product is K5's followed by K1's.
template <int FMT, bool NT, bool ZM, bool TAIL, bool PATCH, bool HS = false>
__global__ __launch_bounds__(EC3D_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void k51_p_spmv_dot(
    MatDev<FMT> A, EC3D_SWEEP_OF(ZM) sw, RedSrc src, SolverState *st, int it, const double *__restrict__ rv,
    const double *__restrict__ p_old, const double *__restrict__ ap_old, double *__restrict__ p_new,
    double *__restrict__ ap_new, double *__restrict__ r0, double *__restrict__ part, double *hist, int64_t hist_cap)
{
    __shared__ double lds[8];
    EC3D_TBL_DECL;
    const int slot[2] = {P_RR, P_RR0N};
    PartialsEarly<2> pe;
    partials_request<2>(src, slot, pe);
    const double bnorm = st->bnorm, tol = st->tol, alpha = st->alpha, omega = st->omega, rr0 = st->rr0[it & 1];
    TableEarly te;
    table_request<FMT>(A, te);
    {
        int si, kind;
        stop_read(st, si, kind);
        EC3D_REQUESTS_OUT;
        if (si < it || (si == it && kind == 1)) return;
    }
    double d[2];
    partials_finish<2>(src, slot, pe, d, lds);
    const double rnorm = sqrt(d[0]);
    // (the second launch of a split K5-in-K1 -- z-slab, interior planes behind the boundary planes -- leaves the state
    // alone: the first one has written it, and the restart counter must count once)
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0 && sw.part_off == 0;
    if (lead && hist && it <= hist_cap) hist[2 * (int64_t)(it - 1) + 1] = rnorm;
    if (lead) st->rnorm = rnorm;
    if (rnorm / bnorm < tol) {
        if (lead) stop_publish(st, it, 2);
        return;
    }
    const double rr0_new = d[1];
    const double beta = (alpha / omega) * rr0_new / rr0;
    const bool restart = fabs(rr0_new) / bnorm < tol;
    if (lead) st->rr0[(it + 1) & 1] = restart ? d[0] : rr0_new;
    if (lead && restart) st->restarts = st->restarts + 1;
    table_store<FMT>(A, te, tbl);
    ZRegs zr;
    PatchPos pp{};
    int pstep = 0;
    double acc[1] = {0.0};
    // (Round 6 measured AP.R0's products a step LATER -- R0's pair still requested behind the step, but waited for behind the
    // next step's plane above: 1185-1190 us either way at 512^3, profiles/r06_patch_requests_512.log; not kept.)
    walk_spmv<FMT, ZM, FMT != FMT_SAV, PATCH>(A, sw, pp, [&](int64_t tile, auto fc) {
        EC3D_ROW_S;
        double s0, s1;
        d2 pc; // the new P[r], P[r+1]
        spmv_pair<FMT, ZM, TAIL, NT, PATCH>(A, sw, pp, tbl, stg, pstep, VecFusedP{rv, p_old, ap_old, beta, omega, restart}, r, tile,
                                            (bool)fc, zr, s0, s1, pc);
        // R0's pair BEHIND the step, a memory round trip of its own.  Requested with the step's first requests it loses
        // (round 3: 1176 -> 1206 us), and so it does behind the arrival of the plane above, with the rim row (round 6:
        // 1120-1185 -> 1215-1218 us; AP.R0 summed before the step's stores instead of behind them: no difference) --
        // profiles/r06_patch_requests_512.log
        d2 q = restart ? pc : load2<NT>(r0 + r); // after a restart R0 = R = the new P
        store2<NT>(p_new, r, nst, pc.x, pc.y);
        if (restart) store2<NT>(r0, r, nst, pc.x, pc.y);
        if constexpr (HS && FMT == FMT_DICT7 && ZM && PATCH) {
            // z-slab: the new P on the neighbours' planes (formed from the exchanged R and AP and the old P kept there the
            // same way) into the new P's ghost rows: the next K5-in-K1 reads it as ITS old P, so P is never exchanged
            if (sw.hs_mask) {
                const int64_t pl = tile / sw.tpp, kdz = A.off[6];
                const VecFusedP xp{rv, p_old, ap_old, beta, omega, restart};
                if ((sw.hs_mask & 1) && pl == 0) *reinterpret_cast<d2 *>(p_new + r - kdz) = xp.pair(r - kdz);
                if ((sw.hs_mask & 2) && pl == sw.hs_last) *reinterpret_cast<d2 *>(p_new + r + kdz) = xp.pair(r + kdz);
            }
        }
        store2<NT>(ap_new, r, nst, s0, s1);
        EC3D_MASK2(r, sw, s0, s1);
        EC3D_IDLE2(s0, s1);
        acc[0] = acc[0] + s0 * q.x;
        acc[0] = acc[0] + s1 * q.y;
    });
    block_sum<1>(acc, lds);
    {
        const int pslot_[1] = {P_D1};
        publish_partials<1>(sw, part, pslot_, acc, lds);
    }
}

// K4: if ‖S‖/Bnorm < tol: X += alpha*P, exit (src/solvers.f90:34-38); else
//     omega = (AS·S)/(AS·AS) ; X = X + alpha*P + omega*S ; R = S - omega*AS ;
//     partials R·R and R·R0   (:40-44)
template <bool NT>
__global__ __launch_bounds__(EC3D_THREADS) void k4_x_r_update(SweepV sw, RedSrc src_ss, RedSrc src, SolverState *st,
                                                              int it, const double *__restrict__ p,
                                                              const double *__restrict__ sv,
                                                              const double *__restrict__ as,
                                                              const double *__restrict__ r0, double *__restrict__ x,
                                                              double *__restrict__ rv, double *__restrict__ part,
                                                              double *hist, int64_t hist_cap)
{
    __shared__ double lds[8];
    // S.S (from K2 or K2-in-K3), AS.S and AS.AS (from K3), the scalars and the exit word: one round trip
    const int slot_ss[1] = {P_SS};
    const int slot[2] = {P_D2, P_D3};
    PartialsEarly<1> pss;
    PartialsEarly<2> pd;
    partials_request<1>(src_ss, slot_ss, pss);
    partials_request<2>(src, slot, pd);
    const double alpha = st->alpha, bnorm = st->bnorm, tol = st->tol;
    const int stop_it = stop_iter_of(st);
    EC3D_REQUESTS_OUT;
    if (stop_it < it) return; // (this kernel is the one that may publish the exit (it, 1))
    double ss[1];
    partials_finish<1>(src_ss, slot_ss, pss, ss, lds);
    const double snorm = sqrt(ss[0]);
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead && hist && it <= hist_cap) hist[2 * (int64_t)(it - 1)] = snorm;
    if (snorm / bnorm < tol) {
        struct OpsX { d2 xv, pv; };
        walk_vec(sw, [&](int64_t tile) {
            EC3D_ROW;
            return OpsX{*reinterpret_cast<const d2 *>(x + r), *reinterpret_cast<const d2 *>(p + r)};
        }, [&](int64_t tile, const OpsX &o) {
            EC3D_ROW;
            store2<false>(x, r, sw.n, o.xv.x + alpha * o.pv.x, o.xv.y + alpha * o.pv.y);
        });
        // every workgroup has read the exit word above; K5 of this iteration tests it again.  Published by the
        // lead thread only; other workgroups of THIS launch may already have passed their entry test, which
        // only compares against earlier iterations.
        if (lead) stop_publish(st, it, 1);
        return;
    }
    double d[2];
    partials_finish<2>(src, slot, pd, d, lds);
    const double omega = d[0] / d[1];
    if (lead) st->omega = omega;
    double acc[2] = {0.0, 0.0};
    struct Ops { d2 xv, pv, s, a, q; };
    walk_vec(sw, [&](int64_t tile) {
        EC3D_ROW;
        return Ops{load2<NT>(x + r), load2<NT>(p + r), load2<NT>(sv + r), load2<NT>(as + r), load2<NT>(r0 + r)};
    }, [&](int64_t tile, const Ops &o) {
        EC3D_ROW;
        const d2 xv = o.xv, pv = o.pv, s = o.s, a = o.a, q = o.q;
        store2<NT>(x, r, sw.n, (xv.x + alpha * pv.x) + omega * s.x, (xv.y + alpha * pv.y) + omega * s.y);
        double e0 = s.x - omega * a.x, e1 = s.y - omega * a.y;
        store2k<NT>(rv, r, sw.n, e0, e1, keep_of(sw) & EC3D_KEEP_R);
        EC3D_MASK2(r, sw, e0, e1);
        acc[0] = acc[0] + e0 * e0;
        acc[0] = acc[0] + e1 * e1;
        acc[1] = acc[1] + e0 * q.x;
        acc[1] = acc[1] + e1 * q.y;
    });
    block_sum<2>(acc, lds);
    {
        const int pslot_[2] = {P_RR, P_RR0N};
        publish_partials<2>(sw, part, pslot_, acc, lds);
    }
}

// K4 with the X update DEFERRED (single rank, three-launch iteration, vectors far beyond the caches).
// X = X + alpha*P + omega*S (src/solvers.f90:41) is the only statement of the loop that reads or writes X, and nothing in
// the loop reads X: the update of iteration k may be carried out LATER, as long as the updates are applied in order and
// each as the same two rounded additions -- X is then the same bits.  P(k) and S(k) stay untouched for D iterations in
// rings of D buffers (ec3d_ctx::pbuf / sbuf), alpha(k) and omega(k) wait in the SolverState; D - 1 of D iterations run
// K4 WITHOUT X (NE = 0: 24 B read, 8 B written per row instead of 40 / 16) and the D-th applies the D updates at once
// (NE = D: 24 + 16 D B read, 16 written).  Reads are unchanged in total (S is read a second time instead of X), D - 1 of D
// writes of X never happen: 50 B per row and iteration at D = 4 instead of 56.  An exit with updates pending (the ||S||
// exit here, the ||R|| exit in K51) leaves them to k_x_flush, which the host launches once it has seen the exit word.
struct XRing {
    const double *p[EC3D_XD_MAX]; // P and S of the pending iterations, oldest first; the last entry is this iteration's
    const double *s[EC3D_XD_MAX];
};
template <bool NT, int NE>
__global__ __launch_bounds__(EC3D_THREADS) void k4d_x_r_update(SweepV sw, RedSrc src_ss, RedSrc src, SolverState *st, int it,
                                                               int xm, XRing ring, const double *__restrict__ as,
                                                               const double *__restrict__ r0, double *__restrict__ x,
                                                               double *__restrict__ rv, double *__restrict__ part,
                                                               double *hist, int64_t hist_cap)
{
    __shared__ double lds[8];
    constexpr int NC = NE > 0 ? NE - 1 : 0; // entry of this iteration's P and S in the ring argument
    const int slot_ss[1] = {P_SS};
    const int slot[2] = {P_D2, P_D3};
    PartialsEarly<1> pss;
    PartialsEarly<2> pd;
    partials_request<1>(src_ss, slot_ss, pss);
    partials_request<2>(src, slot, pd);
    const double alpha = st->alpha, bnorm = st->bnorm, tol = st->tol;
    double pa[EC3D_XD_MAX], po[EC3D_XD_MAX];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        pa[j] = st->pend_alpha[j];
        po[j] = st->pend_omega[j];
    }
    const int stop_it = stop_iter_of(st);
    EC3D_REQUESTS_OUT;
    if (stop_it < it) return;
    double ss[1];
    partials_finish<1>(src_ss, slot_ss, pss, ss, lds);
    const double snorm = sqrt(ss[0]);
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead && hist && it <= hist_cap) hist[2 * (int64_t)(it - 1)] = snorm;
    if (snorm / bnorm < tol) { // src/solvers.f90:34-38: X = X + alpha*P joins the pending updates as a half one
        if (lead) {
            st->pend_alpha[xm] = alpha;
            st->pend_omega[xm] = 0.0;
            st->npend = xm + 1;
            st->pend_half = 1;
            stop_publish(st, it, 1);
        }
        return;
    }
    double d[2];
    partials_finish<2>(src, slot, pd, d, lds);
    const double omega = d[0] / d[1];
    if (lead) {
        st->omega = omega;
        if (NE == 0) {
            st->pend_alpha[xm] = alpha;
            st->pend_omega[xm] = omega;
            st->npend = xm + 1;
        } else {
            st->npend = 0;
        }
    }
    pa[NC] = alpha;
    po[NC] = omega;
    double acc[2] = {0.0, 0.0};
    struct Ops { d2 xv, pv[NE > 0 ? NE : 1], s[NE > 0 ? NE : 1], a, q; };
    walk_vec(sw, [&](int64_t tile) {
        EC3D_ROW;
        Ops o;
        if (NE > 0) o.xv = load2<NT>(x + r);
#pragma unroll
        for (int j = 0; j < NE; ++j) o.pv[j] = load2<NT>(ring.p[j] + r);
#pragma unroll
        for (int j = 0; j < (NE > 0 ? NE : 1); ++j) o.s[j] = load2<NT>(ring.s[j] + r);
        o.a = load2<NT>(as + r);
        o.q = load2<NT>(r0 + r);
        return o;
    }, [&](int64_t tile, const Ops &o) {
        EC3D_ROW;
        if (NE > 0) {
            d2 xv = o.xv;
#pragma unroll
            for (int j = 0; j < NE; ++j) { // one (X + alpha*P) + omega*S per pending iteration, oldest first
                xv.x = (xv.x + pa[j] * o.pv[j].x) + po[j] * o.s[j].x;
                xv.y = (xv.y + pa[j] * o.pv[j].y) + po[j] * o.s[j].y;
            }
            store2<NT>(x, r, sw.n, xv.x, xv.y);
        }
        const d2 s = o.s[NC], a = o.a, q = o.q;
        double e0 = s.x - omega * a.x, e1 = s.y - omega * a.y;
        store2k<NT>(rv, r, sw.n, e0, e1, keep_of(sw) & EC3D_KEEP_R);
        EC3D_MASK2(r, sw, e0, e1);
        acc[0] = acc[0] + e0 * e0;
        acc[0] = acc[0] + e1 * e1;
        acc[1] = acc[1] + e0 * q.x;
        acc[1] = acc[1] + e1 * q.y;
    });
    block_sum<2>(acc, lds);
    {
        const int pslot_[2] = {P_RR, P_RR0N};
        publish_partials<2>(sw, part, pslot_, acc, lds);
    }
}

// K4 as an SpMV kernel (2-D tiles, single rank, vectors far beyond the caches): AS = A S is COMPUTED AGAIN here, from the
// S that K4 reads anyway, instead of being written by K23 and read back -- 8 B written and 8 B read per row less, for 13
// flops per row on a memory-bound machine.  The same spmv_pair on the same tiles gives the same AS bit for bit, so omega
// (from K23's AS.S and AS.AS), R = S - omega*AS and everything after it are unchanged; R.R and R.R0 are summed in the SpMV
// kernels' thread -> cell assignment (the library reports it as geometry 0, the twin follows).  The X update is the
// deferred one of k4d_x_r_update: NEMAX = 0 leaves X alone, NEMAX = 4 applies ne (1 .. 4) pending updates, oldest first.
// Both exits leave pending updates to k_x_flush (the ||S|| exit's X = X + alpha*P as a half update).
template <int FMT, bool NT, int NEMAX>
__global__ __launch_bounds__(EC3D_THREADS) __attribute__((amdgpu_waves_per_eu(NEMAX > 0 ? 2 : 4))) void k4s_x_r_spmv(
    MatDev<FMT> A, SweepZ sw, RedSrc src_ss, RedSrc src, SolverState *st, int it, int ne, int xm, XRing ring,
    const double *__restrict__ r0, double *__restrict__ x, double *__restrict__ rv, double *__restrict__ part, double *hist,
    int64_t hist_cap)
{
    constexpr bool ZM = true, TAIL = false, PATCH = true;
    __shared__ double lds[8];
    EC3D_TBL_DECL;
    const int slot_ss[1] = {P_SS};
    const int slot[2] = {P_D2, P_D3};
    PartialsEarly<1> pss;
    PartialsEarly<2> pd;
    partials_request<1>(src_ss, slot_ss, pss);
    partials_request<2>(src, slot, pd);
    const double alpha = st->alpha, bnorm = st->bnorm, tol = st->tol;
    double pa[EC3D_XD_MAX], po[EC3D_XD_MAX];
#pragma unroll
    for (int j = 0; j < EC3D_XD_MAX - 1; ++j) {
        pa[j] = NEMAX > 0 ? st->pend_alpha[j] : 0.0;
        po[j] = NEMAX > 0 ? st->pend_omega[j] : 0.0;
    }
    TableEarly te;
    table_request<FMT>(A, te);
    const int stop_it = stop_iter_of(st);
    EC3D_REQUESTS_OUT;
    if (stop_it < it) return;
    double ss[1];
    partials_finish<1>(src_ss, slot_ss, pss, ss, lds);
    const double snorm = sqrt(ss[0]);
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0 && sw.part_off == 0; // (split launch: the first one writes the state)
    if (lead && hist && it <= hist_cap) hist[2 * (int64_t)(it - 1)] = snorm;
    if (snorm / bnorm < tol) {
        if (lead) {
            st->pend_alpha[xm] = alpha;
            st->pend_omega[xm] = 0.0;
            st->npend = xm + 1;
            st->pend_half = 1;
            stop_publish(st, it, 1);
        }
        return;
    }
    double d[2];
    partials_finish<2>(src, slot, pd, d, lds);
    const double omega = d[0] / d[1];
    if (lead) {
        st->omega = omega;
        if (NEMAX == 0) {
            st->pend_alpha[xm] = alpha;
            st->pend_omega[xm] = omega;
            st->npend = xm + 1;
        } else {
            st->npend = 0;
        }
    }
    // entry ne - 1 is this iteration's: its alpha and omega are known only now
#pragma unroll
    for (int j = 0; j < EC3D_XD_MAX; ++j)
        if (NEMAX > 0 && j == ne - 1) {
            pa[j] = alpha;
            po[j] = omega;
        }
    table_store<FMT>(A, te, tbl);
    ZRegs zr;
    PatchPos pp{};
    int pstep = 0;
    double acc[2] = {0.0, 0.0};
    const double *scur = ring.s[NEMAX > 0 ? ne - 1 : 0];
    walk_spmv<FMT, ZM, FMT != FMT_SAV, PATCH>(A, sw, pp, [&](int64_t tile, auto fc) {
        EC3D_ROW_S;
        // the operands that do not go through the stencil first: they are in flight while it runs
        const d2 q = load2<NT>(r0 + r);
        d2 xv{0.0, 0.0}, pv[NEMAX > 0 ? NEMAX : 1], so[NEMAX > 0 ? NEMAX : 1];
        if (NEMAX > 0) {
            xv = load2<NT>(x + r);
#pragma unroll
            for (int j = 0; j < NEMAX; ++j)
                if (j < ne) pv[j] = load2<NT>(ring.p[j] + r);
#pragma unroll
            for (int j = 0; j < NEMAX - 1; ++j)
                if (j < ne - 1) so[j] = load2<NT>(ring.s[j] + r);
        }
        double a0, a1;
        d2 sc; // S[r], S[r+1]
        spmv_pair<FMT, ZM, TAIL, NT, PATCH>(A, sw, pp, tbl, stg, pstep, VecPlain{scur}, r, tile, (bool)fc, zr, a0, a1, sc);
        if (NEMAX > 0) {
#pragma unroll
            for (int j = 0; j < NEMAX; ++j)
                if (j < ne) { // one (X + alpha*P) + omega*S per pending iteration, oldest first
                    const d2 sj = (j == ne - 1) ? sc : so[j < NEMAX - 1 ? j : 0];
                    xv.x = (xv.x + pa[j] * pv[j].x) + po[j] * sj.x;
                    xv.y = (xv.y + pa[j] * pv[j].y) + po[j] * sj.y;
                }
            store2<NT>(x, r, nst, xv.x, xv.y);
        }
        double e0 = sc.x - omega * a0, e1 = sc.y - omega * a1;
        store2k<NT>(rv, r, nst, e0, e1, 0);
        EC3D_MASK2(r, sw, e0, e1);
        EC3D_IDLE2(e0, e1);
        acc[0] = acc[0] + e0 * e0;
        acc[0] = acc[0] + e1 * e1;
        acc[1] = acc[1] + e0 * q.x;
        acc[1] = acc[1] + e1 * q.y;
    });
    block_sum<2>(acc, lds);
    {
        const int pslot_[2] = {P_RR, P_RR0N};
        publish_partials<2>(sw, part, pslot_, acc, lds);
    }
}

// the pending X updates after an exit: entries 0 .. npend-1 of the ring, the last one without its omega*S term when the
// exit was the ||S|| one (src/solvers.f90:34-38).  Launched by the host after the exit word has been seen.
__global__ __launch_bounds__(EC3D_THREADS) void k_x_flush(SweepV sw, const SolverState *st, XRing ring, double *__restrict__ x)
{
    const int np = st->npend, half = st->pend_half;
    double pa[EC3D_XD_MAX], po[EC3D_XD_MAX];
#pragma unroll
    for (int j = 0; j < EC3D_XD_MAX; ++j) {
        pa[j] = st->pend_alpha[j];
        po[j] = st->pend_omega[j];
    }
    if (np <= 0) return;
    struct Ops { d2 xv, pv[EC3D_XD_MAX], s[EC3D_XD_MAX]; };
    walk_vec(sw, [&](int64_t tile) {
        EC3D_ROW;
        Ops o;
        o.xv = *reinterpret_cast<const d2 *>(x + r);
#pragma unroll
        for (int j = 0; j < EC3D_XD_MAX; ++j)
            if (j < np) {
                o.pv[j] = *reinterpret_cast<const d2 *>(ring.p[j] + r);
                o.s[j] = (half && j == np - 1) ? d2{0.0, 0.0} : *reinterpret_cast<const d2 *>(ring.s[j] + r);
            }
        return o;
    }, [&](int64_t tile, const Ops &o) {
        EC3D_ROW;
        d2 xv = o.xv;
#pragma unroll
        for (int j = 0; j < EC3D_XD_MAX; ++j)
            if (j < np) {
                if (half && j == np - 1) { // X = X + alpha*P alone: no second addition (X + 0.0 could turn -0 into +0)
                    xv.x = xv.x + pa[j] * o.pv[j].x;
                    xv.y = xv.y + pa[j] * o.pv[j].y;
                } else {
                    xv.x = (xv.x + pa[j] * o.pv[j].x) + po[j] * o.s[j].x;
                    xv.y = (xv.y + pa[j] * o.pv[j].y) + po[j] * o.s[j].y;
                }
            }
        store2<false>(x, r, sw.n, xv.x, xv.y);
    });
}

// K5: if ‖R‖/Bnorm < tol exit (src/solvers.f90:43) ; beta = (alpha/omega)*rr0_new/rr0 (:45) ;
//     P = R + beta*(P - omega*AP) (:46) ; restart R0 = R, P = R when |rr0_new|/Bnorm < tol (:47-49)
//     p_old == p: in place; p_old != p: the new P goes to the next buffer of the ring (deferred X update: the old P is
//     still wanted by a later K4)
template <bool NT>
__global__ __launch_bounds__(EC3D_THREADS) void k5_p_update(SweepV sw, RedSrc src, SolverState *st, int it,
                                                            const double *__restrict__ rv,
                                                            const double *__restrict__ ap, const double *p_old,
                                                            double *p, double *__restrict__ r0, double *hist,
                                                            int64_t hist_cap)
{
    __shared__ double lds[8];
    // Only exits taken by EARLIER launches end this one: the ||S|| exit of this iteration's K4 (it, kind 1) or
    // anything before.  The lead thread of THIS launch publishes (it, 2) below while other workgroups may still
    // be at this test; a wave that returned on seeing it would leave its workgroup's barriers in
    // reduce_partials short of a wave.  The pair is one word (stop_publish), so (it, 1) is K4's and nothing else.
    const int slot[2] = {P_RR, P_RR0N};
    PartialsEarly<2> pe;
    partials_request<2>(src, slot, pe);
    const double bnorm = st->bnorm, tol = st->tol, alpha = st->alpha, omega = st->omega, rr0 = st->rr0[it & 1];
    {
        int si, kind;
        stop_read(st, si, kind);
        EC3D_REQUESTS_OUT;
        if (si < it || (si == it && kind == 1)) return;
    }
    double d[2];
    partials_finish<2>(src, slot, pe, d, lds);
    const double rnorm = sqrt(d[0]);
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead && hist && it <= hist_cap) hist[2 * (int64_t)(it - 1) + 1] = rnorm;
    if (lead) st->rnorm = rnorm;
    if (rnorm / bnorm < tol) {
        if (lead) stop_publish(st, it, 2);
        return;
    }
    const double rr0_new = d[1];
    const double beta = (alpha / omega) * rr0_new / rr0;
    const bool restart = fabs(rr0_new) / bnorm < tol;
    // next iteration's R·R0: after a restart R0 == R, so it is R·R in the same summation order
    if (lead) st->rr0[(it + 1) & 1] = restart ? d[0] : rr0_new;
    if (lead && restart && sw.part_off == 0) st->restarts = st->restarts + 1; // (a split K5 -- boundary + interior launch -- counts once)
    if (restart) {
        walk_vec(sw, [&](int64_t tile) {
            EC3D_ROW;
            return load2<NT>(rv + r);
        }, [&](int64_t tile, const d2 &q) {
            EC3D_ROW;
            store2<NT>(r0, r, sw.n, q.x, q.y);
            store2<NT>(p, r, sw.n, q.x, q.y);
        });
    } else {
        struct Ops { d2 q, pv, a; };
        walk_vec(sw, [&](int64_t tile) {
            EC3D_ROW;
            return Ops{load2<NT>(rv + r), load2<NT>(p_old + r), load2<NT>(ap + r)};
        }, [&](int64_t tile, const Ops &o) {
            EC3D_ROW;
            const d2 q = o.q, pv = o.pv, a = o.a;
            store2k<NT>(p, r, sw.n, q.x + beta * (pv.x - omega * a.x), q.y + beta * (pv.y - omega * a.y), keep_of(sw) & EC3D_KEEP_P);
        });
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
static inline int fmt_of(const MatView &A)
{
    if (A.sav) return FMT_SAV;
    if (A.nb == 7 && A.ncls > 0) return FMT_DICT7;
    if (A.nb == 7) return FMT_DIA7;
    return FMT_GENERIC;
}
// streaming policy: vectors of >= 32 MiB each (n_pad >= 4 Mi rows) cannot live in the caches
static inline bool nt_of(const Sweep &sw) { return (sw.nt & 1) != 0; }

static inline TailDev tail_of(const MatView &A) { return TailDev{A.tail_id, A.tile_flag, A.chunk_ptr, A.tcol, A.tval}; }
template <int FMT> static MatDev<FMT> mat_dev(const MatView &A);
template <> MatDev<FMT_GENERIC> mat_dev<FMT_GENERIC>(const MatView &A)
{
    MatDev<FMT_GENERIC> m;
    for (int b = 0; b < EC3D_MAXB; ++b) {
        m.band[b] = A.band[b];
        m.off[b] = A.off[b];
    }
    m.nb = A.nb;
    m.t = tail_of(A);
    return m;
}
template <> MatDev<FMT_DIA7> mat_dev<FMT_DIA7>(const MatView &A)
{
    MatDev<FMT_DIA7> m;
    for (int b = 0; b < 7; ++b) {
        m.band[b] = A.band[b];
        m.off[b] = A.off[b];
    }
    m.pm1 = A.pm1;
    m.t = tail_of(A);
    return m;
}
template <> MatDev<FMT_DICT7> mat_dev<FMT_DICT7>(const MatView &A)
{
    MatDev<FMT_DICT7> m;
    m.cls = A.cls;
    m.table = A.table;
    for (int b = 0; b < 7; ++b) m.off[b] = A.off[b];
    m.ncls = A.ncls;
    m.pm1 = A.pm1;
    m.t = tail_of(A);
    return m;
}
template <> MatDev<FMT_SAV> mat_dev<FMT_SAV>(const MatView &A)
{
    MatDev<FMT_SAV> m;
    m.cls = A.cls;
    m.tile_flag = A.tile_flag;
    m.table = A.table;
    m.nC = A.sav_nC;
    m.sdx = A.sav_step[1];
    m.pitch = A.sav_step[2];
    m.planes = A.sav_step[2] > 0 ? A.sav_nC / A.sav_step[2] : 0;
    m.ncls = A.ncls;
    m.pm1 = A.pm1;
    m.a0 = A.sav_a0;
    m.u0 = A.sav_u0;
    m.zero = A.sav_zero;
    return m;
}
static inline SweepZ sweep_z(const Sweep &sw)
{
    SweepZ z;
    z.ntiles = sw.ntiles;
    z.n = sw.n;
    z.tpp = sw.zm_tpp;
    z.pps = sw.zm_pps;
    z.npl = sw.zm_npl;
    z.pl0 = sw.zm_pl0;
    z.plstep = sw.zm_plstep;
    z.hs_mask = sw.halo_store;
    z.hs_last = sw.zm_tpp > 0 ? (int)(sw.ntiles / sw.zm_tpp) - 1 : 0;
    z.pstride = sw.pstride;
    z.part_off = sw.part_off;
    z.ulist_n = sw.ulist_n;
    z.ulist = sw.ulist;
    z.win_npo = sw.win_nt > 0 ? sw.win_nt / sw.zm_tpp : 0;
    z.win_npb = sw.win_nt > 0 ? sw.win_blk / sw.zm_tpp : 0;
    z.win_p0 = sw.win_nt > 0 ? sw.win_t0 / sw.zm_tpp : 0;
    z.patch_npx = sw.patch_npx;
    z.patch_sdx = sw.patch_sdx;
    z.zm_tpp = sw.zm_tpp;
    z.keep = sw.nt >> 1;
    z.rp_px = sw.rp_px;
    z.rp_py = sw.rp_py;
    z.rp_npx = sw.rp_npx;
    z.rp_sdy = sw.rp_sdy;
    z.rp_flag = sw.rp_flag;
    z.il_planes = sw.il_planes;
    z.il_nw = sw.il_nw;
    z.il_umask = sw.il_umask;
    z.il_seg = sw.il_seg;
    return z;
}
// dynamic LDS: the class table (a full 256-class table of the structured form would be 32 KiB and cap the CU
// at 4 workgroups; real problems have 64 classes = 8 KiB) and, for the z-marching structured kernels, the
// staging slots behind it (16 KiB): 24.6 KiB per workgroup, six of them fit a CU's 160 KiB
static inline size_t tbl_bytes(const MatView &A, int F, bool zm, bool patch, bool il = false)
{
    // structured form, interleaved z-march: the table alone (walk_zm_il stages nothing)
    if (il && F == FMT_SAV) return (size_t)A.ncls * EC3D_SAV_STRIDE * 8;
    // structured form on runtime-shaped 2-D tiles: table, two exchange buffers, two staging slots (sav_patch_step)
    if (patch && F == FMT_SAV) return (size_t)A.ncls * EC3D_SAV_STRIDE * 8 + (size_t)(2 + EC3D_NSTAGE_RT) * EC3D_TILE * 8;
    if (patch) return (size_t)((F == FMT_DICT7 ? A.ncls * 7 + 1 : 0) & ~1) * 8 + (size_t)2 * EC3D_TILE * 8;
    if (F == FMT_DICT7) return (size_t)A.ncls * 7 * 8;
    if (F == FMT_SAV) return (size_t)A.ncls * EC3D_SAV_STRIDE * 8 + (zm ? (size_t)EC3D_NSTAGE * EC3D_TILE * 8 : 0);
    return 0;
}
#define EC3D_LAUNCH_ZT(F, NT_, KERNEL, ...)                                                                         \
    do {                                                                                                            \
        const bool tail_ = F != FMT_SAV && A.has_tail;                                                              \
        const bool patch_ = zm_ && !tail_ && ((sw.patch_npx > 0 && F == FMT_DICT7) || (sw.rp_px > 0 && F == FMT_SAV)); /* choose_sweep */ \
        const size_t lds_ = tbl_bytes(A, F, zm_, patch_, zm_ && !patch_ && sw.il_planes > 0);                       \
        const MatDev<F> Ad = mat_dev<F>(A);                                                                         \
        if constexpr (F != FMT_GENERIC) {                                                                           \
            if (zm_) {                                                                                              \
                const SweepZ swz = sweep_z(sw);                                                                     \
                if (patch_) { if constexpr (F == FMT_DICT7 || F == FMT_SAV) KERNEL<F, NT_, true, false, true><<<sw.nblk, EC3D_THREADS, lds_, s>>>(Ad, swz, __VA_ARGS__); } \
                else if (tail_) { if constexpr (F != FMT_SAV) KERNEL<F, NT_, true, true, false><<<sw.nblk, EC3D_THREADS, lds_, s>>>(Ad, swz, __VA_ARGS__); } \
                else if (F == FMT_SAV && sw.il_planes > 0) { if constexpr (F == FMT_SAV) KERNEL<F, NT_, true, true, false><<<sw.nblk, EC3D_THREADS, lds_, s>>>(Ad, swz, __VA_ARGS__); } /* interleaved z-march: EC3D_IL */ \
                else KERNEL<F, NT_, true, false, false><<<sw.nblk, EC3D_THREADS, lds_, s>>>(Ad, swz, __VA_ARGS__);   \
                break;                                                                                              \
            }                                                                                                       \
        }                                                                                                           \
        if (tail_) { if constexpr (F != FMT_SAV) KERNEL<F, NT_, false, true, false><<<sw.nblk, EC3D_THREADS, lds_, s>>>(Ad, sw, __VA_ARGS__); } \
        else KERNEL<F, NT_, false, false, false><<<sw.nblk, EC3D_THREADS, lds_, s>>>(Ad, sw, __VA_ARGS__);           \
    } while (0)
#define EC3D_LAUNCH_FMT(F, KERNEL, ...)                                                        \
    do {                                                                                       \
        const bool zm_ = sw.zm_tpp > 0 && sw.bnd_last < 0 && F != FMT_GENERIC;                 \
        if (nt_of(sw))                                                                         \
            EC3D_LAUNCH_ZT(F, true, KERNEL, __VA_ARGS__);                                      \
        else                                                                                   \
            EC3D_LAUNCH_ZT(F, false, KERNEL, __VA_ARGS__);                                     \
    } while (0)
#define EC3D_DISPATCH(A, KERNEL, ...)                                                          \
    do {                                                                                       \
        switch (fmt_of(A)) {                                                                   \
        case FMT_SAV: EC3D_LAUNCH_FMT(FMT_SAV, KERNEL, __VA_ARGS__); break;                    \
        case FMT_DICT7: EC3D_LAUNCH_FMT(FMT_DICT7, KERNEL, __VA_ARGS__); break;                \
        case FMT_DIA7: EC3D_LAUNCH_FMT(FMT_DIA7, KERNEL, __VA_ARGS__); break;                  \
        default: EC3D_LAUNCH_FMT(FMT_GENERIC, KERNEL, __VA_ARGS__);                            \
        }                                                                                      \
    } while (0)
static inline SweepV sweep_v(const Sweep &sw)
{
    SweepV v;
    v.ntiles = sw.ntiles;
    v.n = sw.n;
    v.S = sw.S;
    v.nblk = sw.nblk;
    v.pstride = sw.pstride;
    v.part_off = sw.part_off;
    v.ulist_n = sw.ulist_n;
    v.two = sw.vec_depth;
    v.ulist = sw.ulist;
    v.win_nt = sw.win_nt;
    v.win_blk = sw.win_blk;
    v.win_t0 = sw.win_t0;
    v.nown = sw.nown;
    v.keep = sw.nt >> 1;
    for (int q = 0; q < 4; ++q) {
        v.own_lo[q] = sw.own_lo[q];
        v.own_hi[q] = sw.own_hi[q];
    }
    return v;
}
#define EC3D_LAUNCH_VEC(KERNEL, ...)                                                           \
    do {                                                                                       \
        const SweepV swv = sweep_v(sw);                                                        \
        if (nt_of(sw))                                                                         \
            KERNEL<true><<<sw.nblk, EC3D_THREADS, 0, s>>>(swv, __VA_ARGS__);                   \
        else                                                                                   \
            KERNEL<false><<<sw.nblk, EC3D_THREADS, 0, s>>>(swv, __VA_ARGS__);                  \
    } while (0)

void ec3d_launch_spmv(const MatView &A, const Sweep &sw, const double *x, double *y, hipStream_t s)
{
    EC3D_DISPATCH(A, k_spmv, x, y);
}

void ec3d_launch_residual(const MatView &A, const Sweep &sw, const double *x, const double *b, double *r,
                          double *r0, double *p, double *part, hipStream_t s)
{
    EC3D_DISPATCH(A, k_residual, x, b, r, r0, p, part);
}

void ec3d_launch_finalize(const RedSrc &src, double *lsum, unsigned mask, hipStream_t s)
{
    k_finalize<<<1, EC3D_THREADS, 0, s>>>(src, lsum, mask);
}
// two producers' partials in ONE launch (K2's S.S waits for K3's AS.S and AS.AS: nobody reads it before the gather behind K3)
void ec3d_launch_finalize2(const RedSrc &a, unsigned mask_a, const RedSrc &b, unsigned mask_b, double *lsum, hipStream_t s)
{
    k_finalize2<<<1, EC3D_THREADS, 0, s>>>(a, mask_a, b, mask_b, lsum);
}

void ec3d_launch_setup(SolverState *st, const RedSrc &src, double tol, hipStream_t s)
{
    k_setup<<<1, EC3D_THREADS, 0, s>>>(st, src, tol);
}

void ec3d_launch_k1(const MatView &A, const Sweep &sw, const SolverState *st, int it, const double *p,
                    const double *r0, double *ap, double *part, hipStream_t s)
{
    EC3D_DISPATCH(A, k1_spmv_dot, st, it, p, r0, ap, part);
}

void ec3d_launch_k2(const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r, const double *ap,
                    double *sv, double *part, hipStream_t s)
{
    EC3D_LAUNCH_VEC(k2_s_update, src, st, it, r, ap, sv, part);
}

void ec3d_launch_k3(const MatView &A, const Sweep &sw, SolverState *st, int it, const double *sv, double *as,
                    double *part, hipStream_t s)
{
    EC3D_DISPATCH(A, k3_spmv_dots, st, it, sv, as, part);
}

void ec3d_launch_k23(const MatView &A, const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r,
                     const double *ap, double *sv, double *as, double *part, hipStream_t s)
{
    // the 2-D-tile kernels only (ec3d_fused23): dictionary cube and structured A-V form, one instance per cache policy
    const SweepZ swz = sweep_z(sw);
    if (A.sav) {
        const MatDev<FMT_SAV> Ad = mat_dev<FMT_SAV>(A);
        const size_t lds = tbl_bytes(A, FMT_SAV, true, true);
        if (nt_of(sw))
            k23_s_spmv_dots<FMT_SAV, true, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, ap, sv, as, part);
        else
            k23_s_spmv_dots<FMT_SAV, false, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, ap, sv, as, part);
        return;
    }
    const MatDev<FMT_DICT7> Ad = mat_dev<FMT_DICT7>(A);
    const size_t lds = tbl_bytes(A, FMT_DICT7, true, true);
    if (sw.halo_store) { // a z-slab running the three-launch iteration: S is also stored on the halo planes
        if (nt_of(sw))
            k23_s_spmv_dots<FMT_DICT7, true, true, false, true, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, ap, sv, as, part);
        else
            k23_s_spmv_dots<FMT_DICT7, false, true, false, true, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, ap, sv, as, part);
        return;
    }
    if (nt_of(sw))
        k23_s_spmv_dots<FMT_DICT7, true, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, ap, sv, as, part);
    else
        k23_s_spmv_dots<FMT_DICT7, false, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, ap, sv, as, part);
}

void ec3d_launch_k51(const MatView &A, const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r,
                     const double *p_old, const double *ap_old, double *p_new, double *ap_new, double *r0, double *part,
                     double *hist, int64_t hist_cap, hipStream_t s)
{
    const SweepZ swz = sweep_z(sw);
    if (A.sav) {
        const MatDev<FMT_SAV> Ad = mat_dev<FMT_SAV>(A);
        const size_t lds = tbl_bytes(A, FMT_SAV, true, true);
        if (nt_of(sw))
            k51_p_spmv_dot<FMT_SAV, true, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, p_old, ap_old, p_new, ap_new, r0, part, hist, hist_cap);
        else
            k51_p_spmv_dot<FMT_SAV, false, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, p_old, ap_old, p_new, ap_new, r0, part, hist, hist_cap);
        return;
    }
    const MatDev<FMT_DICT7> Ad = mat_dev<FMT_DICT7>(A);
    const size_t lds = tbl_bytes(A, FMT_DICT7, true, true);
    if (sw.halo_store) { // a z-slab running the three-launch iteration: the new P is also stored on the halo planes
        if (nt_of(sw))
            k51_p_spmv_dot<FMT_DICT7, true, true, false, true, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, p_old, ap_old, p_new, ap_new, r0, part, hist, hist_cap);
        else
            k51_p_spmv_dot<FMT_DICT7, false, true, false, true, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, p_old, ap_old, p_new, ap_new, r0, part, hist, hist_cap);
        return;
    }
    if (nt_of(sw))
        k51_p_spmv_dot<FMT_DICT7, true, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, p_old, ap_old, p_new, ap_new, r0, part, hist, hist_cap);
    else
        k51_p_spmv_dot<FMT_DICT7, false, true, false, true><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src, st, it, r, p_old, ap_old, p_new, ap_new, r0, part, hist, hist_cap);
}

void ec3d_launch_k4(const Sweep &sw, const RedSrc &src_ss, const RedSrc &src, SolverState *st, int it,
                    const double *p, const double *sv, const double *as, const double *r0, double *x, double *r,
                    double *part, double *hist, int64_t hist_cap, hipStream_t s)
{
    EC3D_LAUNCH_VEC(k4_x_r_update, src_ss, src, st, it, p, sv, as, r0, x, r, part, hist, hist_cap);
}

// K4 of an iteration whose X update is deferred (ne = 0) or which applies ne >= 2 pending updates; p[j], sv[j]: the
// pending iterations' P and S, oldest first, this iteration's last (ne = 0: only sv[0], this iteration's S)
void ec3d_launch_k4d(const Sweep &sw, const RedSrc &src_ss, const RedSrc &src, SolverState *st, int it, int ne, int xm,
                     const double *const *p, const double *const *sv, const double *as, const double *r0, double *x,
                     double *r, double *part, double *hist, int64_t hist_cap, hipStream_t s)
{
    if (ne == 1) { // one update, this iteration's own: that is the classic K4
        ec3d_launch_k4(sw, src_ss, src, st, it, p[0], sv[0], as, r0, x, r, part, hist, hist_cap, s);
        return;
    }
    XRing ring{};
    for (int j = 0; j < (ne > 0 ? std::min(ne, EC3D_XD_MAX) : 1); ++j) {
        ring.p[j] = p[j];
        ring.s[j] = sv[j];
    }
    SweepV swv = sweep_v(sw);
    {   // tiles in flight per wave: the launch without X has 4 streams, the applying one 7 + 4 (ne - 1) -- not K4's 7.
        // Big grids (vector plan of two tiles in flight, nothing cached; 512^3, profiles/r04_deferred_x_512.log): without X
        // 1 / 2 / 4 / 8 tiles 1.00 / 0.81 / 0.75 / 0.76 ms, the applying launch better with one tile at ne >= 3, with two at
        // ne = 2.  Mid sizes (plan of one tile, XCD-aware map; r04_deferred_x_mid_sizes.log): one without X, two applying.
        static const int off_env = getenv("EC3D_XD_OFF_DEPTH") ? atoi(getenv("EC3D_XD_OFF_DEPTH")) : 0;
        static const int on_env = getenv("EC3D_XD_ON_DEPTH") ? atoi(getenv("EC3D_XD_ON_DEPTH")) : 0;
        const bool big = sw.vec_depth >= 2;
        const int want = ne == 0 ? (off_env > 0 ? off_env : big ? 4 : 1) : (on_env > 0 ? on_env : big ? (ne == 2 ? 2 : 1) : 2);
        if (swv.win_nt == 0) swv.two = (want == 2 || want == 4) ? want : 1;
    }
#define EC3D_K4D(NE)                                                                                                         \
    do {                                                                                                                     \
        if (nt_of(sw))                                                                                                       \
            k4d_x_r_update<true, NE><<<sw.nblk, EC3D_THREADS, 0, s>>>(swv, src_ss, src, st, it, xm, ring, as, r0, x, r, part, hist, hist_cap); \
        else                                                                                                                 \
            k4d_x_r_update<false, NE><<<sw.nblk, EC3D_THREADS, 0, s>>>(swv, src_ss, src, st, it, xm, ring, as, r0, x, r, part, hist, hist_cap); \
    } while (0)
    switch (ne) {
    case 0: EC3D_K4D(0); break;
    case 2: EC3D_K4D(2); break;
    case 3: EC3D_K4D(3); break;
    default: EC3D_K4D(4); break;
    }
#undef EC3D_K4D
}

// K4 in SpMV form (k4s_x_r_spmv): dictionary cube on 2-D tiles.  ne = 0: X left alone; 1 .. 4: that many updates applied
void ec3d_launch_k4s(const MatView &A, const Sweep &sw, const RedSrc &src_ss, const RedSrc &src, SolverState *st, int it,
                     int ne, int xm, const double *const *p, const double *const *sv, const double *r0, double *x, double *r,
                     double *part, double *hist, int64_t hist_cap, hipStream_t s)
{
    XRing ring{};
    for (int j = 0; j < (ne > 0 ? ne : 1); ++j) {
        ring.p[j] = p[j];
        ring.s[j] = sv[j];
    }
    const SweepZ swz = sweep_z(sw);
    const MatDev<FMT_DICT7> Ad = mat_dev<FMT_DICT7>(A);
    const size_t lds = tbl_bytes(A, FMT_DICT7, true, true);
#define EC3D_K4S(NT_, NE_) \
    k4s_x_r_spmv<FMT_DICT7, NT_, NE_><<<sw.nblk, EC3D_THREADS, lds, s>>>(Ad, swz, src_ss, src, st, it, ne, xm, ring, r0, x, r, part, hist, hist_cap)
    if (nt_of(sw)) {
        if (ne == 0) EC3D_K4S(true, 0);
        else EC3D_K4S(true, 4);
    } else {
        if (ne == 0) EC3D_K4S(false, 0);
        else EC3D_K4S(false, 4);
    }
#undef EC3D_K4S
}

// A GROUP of X updates -- iterations first .. first + count - 1 -- applied by a launch of its own on a second stream, beside
// the iterations that follow (ec3d_xasync: every K4 then leaves X alone).  alpha / omega of iteration it wait in entry
// it % nent of the SolverState (nent = two groups), P and S in rings of nent buffers; the iterations' own kernels only come
// back to those entries and buffers after this launch has finished (the host makes the main stream wait).  An exit
// INSIDE the group ends it there -- the ||S|| exit's X = X + alpha*P without the omega*S term (src/solvers.f90:34-38) --
// and an exit before it leaves nothing to do: the stop word any kernel of the group's iterations published is final when
// this launch starts (it is ordered behind the group's last K4; a later ||R|| exit of that same iteration means "all").
// One (X + alpha*P) + omega*S per iteration, oldest first: the bits of k4d_x_r_update's applying form.
__global__ __launch_bounds__(EC3D_THREADS) void k_x_group(SweepV sw, const SolverState *st, XRing ring, int first, int count, int nent,
                                                          double *__restrict__ x)
{
    int si, kind;
    stop_read(st, si, kind);
    double pa[EC3D_XD_MAX], po[EC3D_XD_MAX];
#pragma unroll
    for (int j = 0; j < EC3D_XD_MAX; ++j) {
        const int e = (first + (j < count ? j : 0)) % nent;
        pa[j] = st->pend_alpha[e];
        po[j] = st->pend_omega[e];
    }
    if (si < first) return;
    int np = count, half = 0;
    if (si - first < count) {
        np = si - first + 1;
        half = kind == 1;
    }
    struct Ops { d2 xv, pv[EC3D_XD_MAX], s[EC3D_XD_MAX]; };
    walk_vec(sw, [&](int64_t tile) {
        EC3D_ROW;
        Ops o;
        o.xv = load2<true>(x + r);
#pragma unroll
        for (int j = 0; j < EC3D_XD_MAX; ++j)
            if (j < np) {
                o.pv[j] = load2<true>(ring.p[j] + r);
                o.s[j] = (half && j == np - 1) ? d2{0.0, 0.0} : load2<true>(ring.s[j] + r);
            }
        return o;
    }, [&](int64_t tile, const Ops &o) {
        EC3D_ROW;
        d2 xv = o.xv;
#pragma unroll
        for (int j = 0; j < EC3D_XD_MAX; ++j)
            if (j < np) {
                if (half && j == np - 1) { // X = X + alpha*P alone: no second addition (X + 0.0 could turn -0 into +0)
                    xv.x = xv.x + pa[j] * o.pv[j].x;
                    xv.y = xv.y + pa[j] * o.pv[j].y;
                } else {
                    xv.x = (xv.x + pa[j] * o.pv[j].x) + po[j] * o.s[j].x;
                    xv.y = (xv.y + pa[j] * o.pv[j].y) + po[j] * o.s[j].y;
                }
            }
        store2<true>(x, r, sw.n, xv.x, xv.y);
    });
}

void ec3d_launch_x_group(const Sweep &sw, const SolverState *st, const double *const *p, const double *const *sv, int first,
                         int count, int d2, double *x, int nblk, hipStream_t s)
{
    XRing ring{};
    for (int j = 0; j < EC3D_XD_MAX; ++j) {
        ring.p[j] = p[j];
        ring.s[j] = sv[j];
    }
    SweepV swv = sweep_v(sw);
    // fewer workgroups than the iteration's kernels when asked for (a smaller share of the bandwidth while they run);
    // the tile walk is a stride over whatever grid it is given
    const int wg = nblk > 0 ? std::min(nblk, sw.nblk) : sw.nblk;
    if (wg != sw.nblk) { // (a plain stride over the tiles: the XCD-aware deal is tied to the full grid)
        swv.nblk = wg;
        swv.S = 0;
    }
    k_x_group<<<wg, EC3D_THREADS, 0, s>>>(swv, st, ring, first, count, d2, x);
}

void ec3d_launch_x_flush(const Sweep &sw, const SolverState *st, const double *const *p, const double *const *sv,
                         double *x, hipStream_t s)
{
    XRing ring{};
    for (int j = 0; j < EC3D_XD_MAX; ++j) {
        ring.p[j] = p[j];
        ring.s[j] = sv[j];
    }
    k_x_flush<<<sw.nblk, EC3D_THREADS, 0, s>>>(sweep_v(sw), st, ring, x);
}

void ec3d_launch_k5(const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r, const double *ap,
                    const double *p_old, double *p, double *r0, double *hist, int64_t hist_cap, hipStream_t s)
{
    EC3D_LAUNCH_VEC(k5_p_update, src, st, it, r, ap, p_old, p, r0, hist, hist_cap);
}
