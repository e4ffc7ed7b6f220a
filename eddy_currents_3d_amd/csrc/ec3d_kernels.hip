// ec3d_kernels.hip — gfx950 kernels of the BiCGSTAB-with-restart hot path.
//
// Reference arithmetic restated (src/solvers.f90:3-61, see the per-kernel notes) with these
// MI355X-first choices:
//   * matrix = DIA bands (SoA, one fp64 stream per band) + sliced-ELL tail; no column indices
//     on the band part, every stream is read with 16-byte accesses, one tile = 512 rows;
//   * one iteration = 5 launches (K1..K5); every vector op is fused into the kernel that
//     produces its operand, every dot product into the kernel that produces its vector;
//   * reductions are deterministic: per-thread sequential over its tiles, 64-lane shuffle tree,
//     4 wave sums left to right, one partial per workgroup; the NEXT kernel's workgroups each
//     re-reduce the partials in the same order (a few KB from L2), so there is no atomics, no
//     inter-workgroup handshake and no host round trip; scalars (alpha, omega, beta, rr0) live
//     in a device-resident SolverState;
//   * convergence is decided on the device: an exit writes stop_iter, later launches become
//     no-ops, the host polls asynchronously (ec3d_solve.hip);
//   * blockIdx -> tile map is XCD aware: the 8 XCD labels (blockIdx % 8) sweep disjoint
//     contiguous groups of S tiles of one moving window, so x[r ± sdx] re-reads hit the L2 of
//     the XCD that fetched them and the window's planes stay in the Infinity Cache.
// Built with -ffp-contract=off: products and sums are rounded separately, exactly as the
// reference's x86-64 object code does; the oracle's "GPU order" twin reproduces every bit.
#include "ec3d_internal.hpp"

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
// rows that take part in the dot products: all rows < n, or -- for one z-slab of the A-V system held
// on an extended grid (own planes + halo planes of every component) -- the owned index ranges only
__device__ __forceinline__ bool row_owned(const Sweep &sw, int64_t r)
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

// The vector a row kernel multiplies by, behind a small accessor (pair = two consecutive entries from
// an 8-byte-aligned address, at = one gathered entry).
struct VecPlain {
    const double *__restrict__ x;
    __device__ __forceinline__ d2 pair(int64_t i) const { return *reinterpret_cast<const d2u *>(x + i); }
    __device__ __forceinline__ double at(int64_t i) const { return x[i]; }
};
// Tail of one row: s += tval[e] * x[tcol[e]] over the row's slots of its 64-row slice, in stored order.
// The loads are issued in batches (all values/columns of a batch, then all gathers, then the adds in
// order): a plain loop is a chain of two dependent global loads per entry, ~1-2 us each, and a 13-entry
// U row then costs more than the whole banded part of the tile.
template <int B, class V>
__device__ __forceinline__ double tail_batch(const MatView &A, const V &x, int64_t e0, int cnt, double s)
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
__device__ __forceinline__ double tail_add(const MatView &A, const V &x, int t, double s)
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

// Matrix formats the row kernel is specialised for (template parameter FMT):
//   FMT_GENERIC  any number of bands, one fp64 stream per band
//   FMT_DIA7     7 bands, unrolled (72 B/row: 56 coefficients + x + y)
//   FMT_DICT7    7 bands whose coefficient 7-tuples take <= 256 distinct values ("stencil classes"):
//                one class byte per row + a table staged in LDS (17 B/row: 1 + x + y).  The values
//                multiplied are the same doubles, so results are bit-identical to FMT_DIA7.
enum { FMT_GENERIC = 0, FMT_DIA7 = 7, FMT_DICT7 = 107, FMT_SAV = 207 };
#define EC3D_SAV_STRIDE 16

template <int FMT>
__device__ __forceinline__ void stage_table(const MatView &A, double *tbl)
{
    if (FMT == FMT_DICT7 || FMT == FMT_SAV) {
        const int cnt = A.ncls * (FMT == FMT_SAV ? EC3D_SAV_STRIDE : 7);
        for (int i = threadIdx.x; i < cnt; i += EC3D_THREADS) tbl[i] = A.table[i];
        __syncthreads();
    }
}

// Structured A-V form: the coupling slots of rows r, r+1 (see MatView), added in slot order = ascending
// column order = the reference's row-sum order.  A slot whose coefficient is zero is not an entry of the
// reference's row: it is neither loaded nor added.  Both rows take their operands from one 16-byte load
// per slot; most rows use 2 of the 5 (A rows) or 6 of the 9 (U rows) slots.
// the two running sums are final here and no load moves across (register pressure, see the callers)
#define EC3D_PIN(a, b) asm volatile("" : "+v"(a), "+v"(b)::"memory")
template <class V>
__device__ __forceinline__ void sav_u_pre(const MatView &A, const double *t0, const double *t1, const V &x,
                                          int64_t r, double &s0, double &s1)
{
    // per component: the (up to) three loads first, then the adds in slot order -- a load inside the
    // add chain would cost one memory round trip per slot
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int64_t base = r - (3 - d) * A.sav_nC;
        double v0[3], v1[3];
        d2 xx[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            v0[j] = t0[7 + 3 * d + j];
            v1[j] = t1[7 + 3 * d + j];
            xx[j] = d2{0.0, 0.0};
            if (v0[j] != 0.0 || v1[j] != 0.0) xx[j] = x.pair(base + (j - 1) * A.sav_step[d]);
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
__device__ __forceinline__ void sav_a_post(const MatView &A, const double *t0, const double *t1, const V &x,
                                           int64_t r, int d, double &s0, double &s1)
{
    const int64_t base = r + (3 - d) * A.sav_nC, step = A.sav_step[d];
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
};

// rows r, r+1 of A*x (src/solvers.f90:58-59): bands in ascending column order, then the tail.
// ZM: band 0 / 3 / 6 (offsets -kdz, 0, +kdz) come from / go to the registers `z`.
// `ctr` returns x[r], x[r+1] (the centre band's operand).
template <int FMT, bool ZM, class V>
__device__ __forceinline__ void spmv_pair(const MatView &A, const double *tbl, const V &x, int64_t r, int64_t tile,
                                          bool first, ZRegs &z, double &s0, double &s1, d2 &ctr)
{
    uint8_t tflag = 0; // per tile: any tail row (bands + tail) / any coupled row (structured form)
    if (FMT == FMT_DIA7 || FMT == FMT_DICT7 || FMT == FMT_SAV) {
        d2 xv[7];
        // the +-1 neighbours (bands 2 and 4 of the 7-point operator) are the centre pairs of the
        // adjacent lanes: take them by lane shuffle instead of two unaligned 16-byte loads; only the
        // first/last lane of a wave reads its outer neighbour from memory.  The SpMV kernels are
        // bound by L1/TA load issue, not by HBM, so 3 full-wave loads per step instead of 5 matter.
        const bool pm1 = A.pm1 != 0;
        // every load of the step is issued before the first use: class bytes, the two edge-lane
        // neighbours, then the band operands (one round trip per step instead of three)
        unsigned short cc = 0;
        if (FMT == FMT_DICT7 || FMT == FMT_SAV) cc = *reinterpret_cast<const unsigned short *>(A.cls + r);
        if (FMT == FMT_SAV || A.has_tail) tflag = A.tile_flag[tile];
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
            xv[b] = x.pair(r + A.off[b]);
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
        if (FMT == FMT_DIA7) {
            d2 c[7];
#pragma unroll
            for (int b = 0; b < 7; ++b) c[b] = *reinterpret_cast<const d2 *>(A.band[b] + r);
            s0 = c[0].x * xv[0].x;
            s1 = c[0].y * xv[0].y;
#pragma unroll
            for (int b = 1; b < 7; ++b) {
                s0 = s0 + c[b].x * xv[b].x;
                s1 = s1 + c[b].y * xv[b].y;
            }
        } else if (FMT == FMT_DICT7) {
            const double *t0 = tbl + (cc & 0xFF) * 7, *t1 = tbl + (cc >> 8) * 7;
            s0 = t0[0] * xv[0].x;
            s1 = t1[0] * xv[0].y;
#pragma unroll
            for (int b = 1; b < 7; ++b) {
                s0 = s0 + t0[b] * xv[b].x;
                s1 = s1 + t1[b] * xv[b].y;
            }
        } else { // FMT_SAV: U rows take their A couplings first, A rows their U couplings last
            const int c0 = cc & 0xFF, c1 = cc >> 8;
            const double *t0 = tbl + c0 * EC3D_SAV_STRIDE, *t1 = tbl + c1 * EC3D_SAV_STRIDE;
            const bool cpl = tflag != 0; // any coupled row in this tile (uniform)
            // the coupling slots of an A class mean U columns, those of a U class A columns: a row of the
            // other kind goes through the all-zero class
            const double *zt = tbl + A.sav_zero * EC3D_SAV_STRIDE;
            const int64_t nA = 3 * A.sav_nC;
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
                const int d0 = (r >= A.sav_nC) + (r >= 2 * A.sav_nC);
                const int d1 = (r + 1 >= A.sav_nC) + (r + 1 >= 2 * A.sav_nC);
                const bool in0 = c0 >= A.sav_a0 && c0 < A.sav_u0, in1 = c1 >= A.sav_a0 && c1 < A.sav_u0 && r + 1 < nA;
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
        if (A.has_tail) tflag = A.tile_flag[tile];
    }
    if (A.has_tail && tflag) {
        i2 t = *reinterpret_cast<const i2 *>(A.tail_id + r);
        if (t.x >= 0) s0 = tail_add(A, x, t.x, s0);
        if (t.y >= 0) s1 = tail_add(A, x, t.y, s1);
    }
}

// the SpMV grids are sized for 6 workgroups per CU (choose_sweep): keep the register count within that
// (the structured form without z-marching only runs on grids too small for plane-aligned tiles: it takes the
// registers it needs -- 5 per CU -- instead of spilling)
#define EC3D_SPMV_OCC __attribute__((amdgpu_waves_per_eu((FMT == FMT_SAV && !ZM) ? 5 : 6)))
#define EC3D_SWEEP_BEGIN_(MODE)                                                                \
    bool need_first_ = true;                                                                   \
    int64_t lst_ = -1; /* >= 0: walking the list of occupied U tiles */                        \
    for (int64_t it_ = 0;; ++it_) {                                                            \
        int64_t tile = 0;                                                                      \
        if (lst_ < 0) {                                                                        \
            tile = ec3d_tile_of<MODE>(sw, blockIdx.x, it_);                                    \
            if (tile >= sw.ntiles) {                                                           \
                if (sw.ulist_n == 0) break;                                                    \
                lst_ = blockIdx.x;                                                             \
            }                                                                                  \
        }                                                                                      \
        if (lst_ >= 0) {                                                                       \
            if (lst_ >= sw.ulist_n) break;                                                     \
            tile = sw.ulist[lst_];                                                             \
            lst_ += sw.nblk;                                                                   \
            need_first_ = true;                                                                \
        }                                                                                      \
        const bool first_ = need_first_;                                                       \
        need_first_ = false;                                                                   \
        (void)first_;                                                                          \
        const int64_t r = tile * EC3D_TILE + 2 * (int64_t)threadIdx.x;
#define EC3D_SWEEP_BEGIN EC3D_SWEEP_BEGIN_(-1)             /* vector kernels */
#define EC3D_SWEEP_BEGIN_A EC3D_SWEEP_BEGIN_((ZM ? 1 : 0)) /* SpMV-type kernels: ZM is a template parameter */
#define EC3D_SWEEP_END }
#define EC3D_TBL_DECL                                                                          \
    extern __shared__ double tbl[] /* the class table, sized at launch (EC3D_TBL_BYTES) */

// ---------------------------------------------------------------------------------------------
// plain y = A x  (src/solvers.f90:54-61)
template <int FMT, bool NT, bool ZM>
__global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k_spmv(MatView A, Sweep sw, const double *__restrict__ x,
                                                       double *__restrict__ y)
{
    EC3D_TBL_DECL;
    stage_table<FMT>(A, tbl);
    ZRegs zr;
    EC3D_SWEEP_BEGIN_A
    double s0, s1;
    d2 ctr;
    spmv_pair<FMT, ZM>(A, tbl, VecPlain{x}, r, tile, first_, zr, s0, s1, ctr);
    store2<NT>(y, r, sw.n, s0, s1);
    EC3D_SWEEP_END
}

// setup: R = B - A X ; R0 = R ; P = R ; partials of B·B and R·R   (src/solvers.f90:14-21)
template <int FMT, bool NT, bool ZM>
__global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k_residual(MatView A, Sweep sw, const double *__restrict__ x,
                                                           const double *__restrict__ b, double *__restrict__ rv,
                                                           double *__restrict__ r0, double *__restrict__ p,
                                                           double *__restrict__ part)
{
    __shared__ double lds[8];
    EC3D_TBL_DECL;
    stage_table<FMT>(A, tbl);
    ZRegs zr;
    double acc[2] = {0.0, 0.0};
    EC3D_SWEEP_BEGIN_A
    double s0, s1;
    d2 ctr;
    spmv_pair<FMT, ZM>(A, tbl, VecPlain{x}, r, tile, first_, zr, s0, s1, ctr);
    d2 bv = load2<NT>(b + r);
    double e0 = bv.x - s0, e1 = bv.y - s1, b0 = bv.x, b1 = bv.y;
    store2<NT>(rv, r, sw.n, e0, e1);
    store2<NT>(r0, r, sw.n, e0, e1);
    store2<NT>(p, r, sw.n, e0, e1);
    EC3D_MASK2(r, sw, e0, e1);
    EC3D_MASK2(r, sw, b0, b1);
    acc[0] = acc[0] + b0 * b0;
    acc[0] = acc[0] + b1 * b1;
    acc[1] = acc[1] + e0 * e0;
    acc[1] = acc[1] + e1 * e1;
    EC3D_SWEEP_END
    block_sum<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[P_BB * sw.pstride + sw.part_off + blockIdx.x] = acc[0];
        part[P_RR_INIT * sw.pstride + sw.part_off + blockIdx.x] = acc[1];
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
        st->stop_kind = 0;
        st->stop_iter = (bnorm == 0.0) ? 0 : INT_MAX;
    }
}

// K1: AP = A P ; partial AP·R0    (src/solvers.f90:30, :32 denominator)
template <int FMT, bool NT, bool ZM>
__global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k1_spmv_dot(MatView A, Sweep sw, const SolverState *st, int it,
                                                            const double *__restrict__ p,
                                                            const double *__restrict__ r0,
                                                            double *__restrict__ ap, double *__restrict__ part)
{
    __shared__ double lds[4];
    EC3D_TBL_DECL;
    if (st->stop_iter < it) return;
    stage_table<FMT>(A, tbl);
    ZRegs zr;
    double acc[1] = {0.0};
    EC3D_SWEEP_BEGIN_A
    double s0, s1;
    d2 ctr;
    spmv_pair<FMT, ZM>(A, tbl, VecPlain{p}, r, tile, first_, zr, s0, s1, ctr);
    d2 q = load2<NT>(r0 + r);
    store2<NT>(ap, r, sw.n, s0, s1);
    EC3D_MASK2(r, sw, s0, s1);
    acc[0] = acc[0] + s0 * q.x;
    acc[0] = acc[0] + s1 * q.y;
    EC3D_SWEEP_END
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) part[P_D1 * sw.pstride + sw.part_off + blockIdx.x] = acc[0];
}

// K2: alpha = rr0 / (AP·R0) ; S = R - alpha*AP ; partial S·S   (src/solvers.f90:31-34)
template <bool NT>
__global__ __launch_bounds__(EC3D_THREADS) void k2_s_update(Sweep sw, RedSrc src, SolverState *st, int it,
                                                            const double *__restrict__ rv,
                                                            const double *__restrict__ ap, double *__restrict__ sv,
                                                            double *__restrict__ part)
{
    __shared__ double lds[4];
    if (st->stop_iter < it) return;
    const int slot[1] = {P_D1};
    double d[1];
    reduce_partials<1>(src, slot, d, lds);
    const double alpha = st->rr0[it & 1] / d[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) st->alpha = alpha;
    double acc[1] = {0.0};
    EC3D_SWEEP_BEGIN
    d2 a = load2<NT>(ap + r);
    d2 q = load2<NT>(rv + r);
    double s0 = q.x - alpha * a.x, s1 = q.y - alpha * a.y;
    store2<NT>(sv, r, sw.n, s0, s1);
    EC3D_MASK2(r, sw, s0, s1);
    acc[0] = acc[0] + s0 * s0;
    acc[0] = acc[0] + s1 * s1;
    EC3D_SWEEP_END
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) part[P_SS * sw.pstride + sw.part_off + blockIdx.x] = acc[0];
}

// K3: AS = A S ; partials AS·S and AS·AS (src/solvers.f90:39-40).  Launched before ‖S‖ is known
// (one global reduction point less per iteration): when the ‖S‖ exit of :34-38 is then
// taken by K4, AS is simply never used -- results are unchanged.
template <int FMT, bool NT, bool ZM>
__global__ __launch_bounds__(EC3D_THREADS) EC3D_SPMV_OCC void k3_spmv_dots(MatView A, Sweep sw, SolverState *st, int it,
                                                             const double *__restrict__ sv,
                                                             double *__restrict__ as, double *__restrict__ part)
{
    __shared__ double lds[8];
    EC3D_TBL_DECL;
    if (st->stop_iter < it) return;
    stage_table<FMT>(A, tbl);
    ZRegs zr;
    double acc[2] = {0.0, 0.0};
    EC3D_SWEEP_BEGIN_A
    double s0, s1;
    d2 q;
    spmv_pair<FMT, ZM>(A, tbl, VecPlain{sv}, r, tile, first_, zr, s0, s1, q);
    store2<NT>(as, r, sw.n, s0, s1);
    EC3D_MASK2(r, sw, s0, s1);
    acc[0] = acc[0] + s0 * q.x;
    acc[0] = acc[0] + s1 * q.y;
    acc[1] = acc[1] + s0 * s0;
    acc[1] = acc[1] + s1 * s1;
    EC3D_SWEEP_END
    block_sum<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[P_D2 * sw.pstride + sw.part_off + blockIdx.x] = acc[0];
        part[P_D3 * sw.pstride + sw.part_off + blockIdx.x] = acc[1];
    }
}

// K4: if ‖S‖/Bnorm < tol: X += alpha*P, exit (src/solvers.f90:34-38); else
//     omega = (AS·S)/(AS·AS) ; X = X + alpha*P + omega*S ; R = S - omega*AS ;
//     partials R·R and R·R0   (:40-44)
template <bool NT>
__global__ __launch_bounds__(EC3D_THREADS) void k4_x_r_update(Sweep sw, RedSrc src_ss, RedSrc src, SolverState *st,
                                                              int it, const double *__restrict__ p,
                                                              const double *__restrict__ sv,
                                                              const double *__restrict__ as,
                                                              const double *__restrict__ r0, double *__restrict__ x,
                                                              double *__restrict__ rv, double *__restrict__ part,
                                                              double *hist, int64_t hist_cap)
{
    __shared__ double lds[8];
    if (st->stop_iter < it) return; // (this kernel is the one that may set stop_iter = it)
    const int slot_ss[1] = {P_SS};
    double ss[1];
    reduce_partials<1>(src_ss, slot_ss, ss, lds);
    const double snorm = sqrt(ss[0]);
    const double alpha = st->alpha;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead && hist && it <= hist_cap) hist[2 * (int64_t)(it - 1)] = snorm;
    if (snorm / st->bnorm < st->tol) {
        EC3D_SWEEP_BEGIN
        d2 xv = *reinterpret_cast<const d2 *>(x + r);
        d2 pv = *reinterpret_cast<const d2 *>(p + r);
        store2<false>(x, r, sw.n, xv.x + alpha * pv.x, xv.y + alpha * pv.y);
        EC3D_SWEEP_END
        if (lead) st->stop_kind = 1;
        // every workgroup has read stop_iter above; K5 of this iteration tests stop_iter <= it.
        // Written last and only by the lead thread; other workgroups of THIS launch may already
        // have passed their entry test, which only compares against earlier iterations.
        if (lead) st->stop_iter = it;
        return;
    }
    const int slot[2] = {P_D2, P_D3};
    double d[2];
    reduce_partials<2>(src, slot, d, lds);
    const double omega = d[0] / d[1];
    if (lead) st->omega = omega;
    double acc[2] = {0.0, 0.0};
    EC3D_SWEEP_BEGIN
    d2 xv = load2<NT>(x + r);
    d2 pv = load2<NT>(p + r);
    d2 s = load2<NT>(sv + r);
    d2 a = load2<NT>(as + r);
    d2 q = load2<NT>(r0 + r);
    store2<NT>(x, r, sw.n, (xv.x + alpha * pv.x) + omega * s.x, (xv.y + alpha * pv.y) + omega * s.y);
    double e0 = s.x - omega * a.x, e1 = s.y - omega * a.y;
    store2<NT>(rv, r, sw.n, e0, e1);
    EC3D_MASK2(r, sw, e0, e1);
    acc[0] = acc[0] + e0 * e0;
    acc[0] = acc[0] + e1 * e1;
    acc[1] = acc[1] + e0 * q.x;
    acc[1] = acc[1] + e1 * q.y;
    EC3D_SWEEP_END
    block_sum<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[P_RR * sw.pstride + sw.part_off + blockIdx.x] = acc[0];
        part[P_RR0N * sw.pstride + sw.part_off + blockIdx.x] = acc[1];
    }
}

// K5: if ‖R‖/Bnorm < tol exit (src/solvers.f90:43) ; beta = (alpha/omega)*rr0_new/rr0 (:45) ;
//     P = R + beta*(P - omega*AP) (:46) ; restart R0 = R, P = R when |rr0_new|/Bnorm < tol (:47-49)
template <bool NT>
__global__ __launch_bounds__(EC3D_THREADS) void k5_p_update(Sweep sw, RedSrc src, SolverState *st, int it,
                                                            const double *__restrict__ rv,
                                                            const double *__restrict__ ap, double *__restrict__ p,
                                                            double *__restrict__ r0, double *hist, int64_t hist_cap)
{
    __shared__ double lds[8];
    // Only exits taken by EARLIER launches end this one: the ||S|| exit of this iteration's K4
    // (stop_iter == it, kind 1) or anything before.  The lead thread of THIS launch writes stop_iter = it
    // (kind 2) below while other workgroups may still be at this test; a wave that returned on seeing it
    // would leave its workgroup's barriers in reduce_partials short of a wave.  Every workgroup reaches the
    // same ||R|| decision from the same sums anyway.  (kind is written before stop_iter, by one thread.)
    {
        const int si = st->stop_iter;
        if (si < it || (si == it && st->stop_kind == 1)) return;
    }
    const int slot[2] = {P_RR, P_RR0N};
    double d[2];
    reduce_partials<2>(src, slot, d, lds);
    const double rnorm = sqrt(d[0]);
    const double bnorm = st->bnorm, tol = st->tol;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead && hist && it <= hist_cap) hist[2 * (int64_t)(it - 1) + 1] = rnorm;
    if (rnorm / bnorm < tol) {
        if (lead) {
            st->stop_kind = 2;
            st->stop_iter = it;
        }
        return;
    }
    const double rr0_new = d[1];
    const double alpha = st->alpha, omega = st->omega;
    const double beta = (alpha / omega) * rr0_new / st->rr0[it & 1];
    const bool restart = fabs(rr0_new) / bnorm < tol;
    // next iteration's R·R0: after a restart R0 == R, so it is R·R in the same summation order
    if (lead) st->rr0[(it + 1) & 1] = restart ? d[0] : rr0_new;
    EC3D_SWEEP_BEGIN
    d2 q = load2<NT>(rv + r);
    if (restart) {
        store2<NT>(r0, r, sw.n, q.x, q.y);
        store2<NT>(p, r, sw.n, q.x, q.y);
    } else {
        d2 pv = load2<NT>(p + r);
        d2 a = load2<NT>(ap + r);
        store2<NT>(p, r, sw.n, q.x + beta * (pv.x - omega * a.x), q.y + beta * (pv.y - omega * a.y));
    }
    EC3D_SWEEP_END
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
static inline bool nt_of(const Sweep &sw) { return sw.nt != 0; }
// dynamic LDS: the class table only (a full 256-class table of the structured form would be 32 KiB and
// cap the CU at 4 workgroups; real problems have ~100 classes)
#define EC3D_TBL_BYTES(F)                                                                      \
    ((F) == FMT_DICT7 ? (size_t)A.ncls * 7 * 8 : ((F) == FMT_SAV ? (size_t)A.ncls * EC3D_SAV_STRIDE * 8 : 0))
#define EC3D_LAUNCH_FMT(F, KERNEL, ...)                                                        \
    do {                                                                                       \
        const bool zm_ = sw.zm_tpp > 0 && sw.bnd_last < 0 && F != FMT_GENERIC;                 \
        if (nt_of(sw) && zm_)                                                                  \
            KERNEL<F, true, (F != FMT_GENERIC)><<<sw.nblk, EC3D_THREADS, EC3D_TBL_BYTES(F), s>>>(__VA_ARGS__); \
        else if (nt_of(sw))                                                                    \
            KERNEL<F, true, false><<<sw.nblk, EC3D_THREADS, EC3D_TBL_BYTES(F), s>>>(__VA_ARGS__);              \
        else if (zm_)                                                                          \
            KERNEL<F, false, (F != FMT_GENERIC)><<<sw.nblk, EC3D_THREADS, EC3D_TBL_BYTES(F), s>>>(__VA_ARGS__);\
        else                                                                                   \
            KERNEL<F, false, false><<<sw.nblk, EC3D_THREADS, EC3D_TBL_BYTES(F), s>>>(__VA_ARGS__);             \
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
#define EC3D_LAUNCH_VEC(KERNEL, ...)                                                           \
    do {                                                                                       \
        if (nt_of(sw))                                                                         \
            KERNEL<true><<<sw.nblk, EC3D_THREADS, 0, s>>>(__VA_ARGS__);                        \
        else                                                                                   \
            KERNEL<false><<<sw.nblk, EC3D_THREADS, 0, s>>>(__VA_ARGS__);                       \
    } while (0)

void ec3d_launch_spmv(const MatView &A, const Sweep &sw, const double *x, double *y, hipStream_t s)
{
    EC3D_DISPATCH(A, k_spmv, A, sw, x, y);
}

void ec3d_launch_residual(const MatView &A, const Sweep &sw, const double *x, const double *b, double *r,
                          double *r0, double *p, double *part, hipStream_t s)
{
    EC3D_DISPATCH(A, k_residual, A, sw, x, b, r, r0, p, part);
}

void ec3d_launch_finalize(const RedSrc &src, double *lsum, unsigned mask, hipStream_t s)
{
    k_finalize<<<1, EC3D_THREADS, 0, s>>>(src, lsum, mask);
}

void ec3d_launch_setup(SolverState *st, const RedSrc &src, double tol, hipStream_t s)
{
    k_setup<<<1, EC3D_THREADS, 0, s>>>(st, src, tol);
}

void ec3d_launch_k1(const MatView &A, const Sweep &sw, const SolverState *st, int it, const double *p,
                    const double *r0, double *ap, double *part, hipStream_t s)
{
    EC3D_DISPATCH(A, k1_spmv_dot, A, sw, st, it, p, r0, ap, part);
}

void ec3d_launch_k2(const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r, const double *ap,
                    double *sv, double *part, hipStream_t s)
{
    EC3D_LAUNCH_VEC(k2_s_update, sw, src, st, it, r, ap, sv, part);
}

void ec3d_launch_k3(const MatView &A, const Sweep &sw, SolverState *st, int it, const double *sv, double *as,
                    double *part, hipStream_t s)
{
    EC3D_DISPATCH(A, k3_spmv_dots, A, sw, st, it, sv, as, part);
}

void ec3d_launch_k4(const Sweep &sw, const RedSrc &src_ss, const RedSrc &src, SolverState *st, int it,
                    const double *p, const double *sv, const double *as, const double *r0, double *x, double *r,
                    double *part, double *hist, int64_t hist_cap, hipStream_t s)
{
    EC3D_LAUNCH_VEC(k4_x_r_update, sw, src_ss, src, st, it, p, sv, as, r0, x, r, part, hist, hist_cap);
}

void ec3d_launch_k5(const Sweep &sw, const RedSrc &src, SolverState *st, int it, const double *r, const double *ap,
                    double *p, double *r0, double *hist, int64_t hist_cap, hipStream_t s)
{
    EC3D_LAUNCH_VEC(k5_p_update, sw, src, st, it, r, ap, p, r0, hist, hist_cap);
}
