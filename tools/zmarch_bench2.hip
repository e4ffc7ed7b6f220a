// zmarch_bench2.hip -- on-box probe (round 6): what bounds K2-in-K3 (k23_s_spmv_dots) at 512^3?  Its traffic is 16 B
// read (R, AP) + 8 B written (S) per row; a linear sweep of that shape (K2) takes reads/7.1 + writes/4.5 TB/s = 541 us,
// K2-in-K3 takes 697.  This file times that traffic under the z-march's structure, one ingredient at a time:
//   lin      linear sweep (K2's shape)
//   zreg     z-march over 128x4 patches, plane above into registers, no LDS, no barrier
//   zbar     + the centre plane through LDS with one barrier per step and the rim rows from memory (patch_pair's shape)
//   zbar2    the same, two planes per trip (both planes' loads requested together)
//   lin_st   the linear sweep with nontemporal loads and the store's cache policy spelled out; wr_only / rd_only
//   zcopy    two launches in a row walking up / down (Infinity Cache)
//   zfull<PF> the whole step of K2-in-K3 (classes, table, shuffles, edge lanes, rim rows, dots), requests PF steps ahead
// (an LDS-DMA ring of P+2 plane slots was timed too, 577-617 us with sums that were never verified: removed)
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/zmarch_bench2.hip -o /tmp/zb2 && /tmp/zb2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ d2 ld(const double *p) { return *reinterpret_cast<const d2 *>(p); }
__device__ __forceinline__ void stnt(double *p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p)); }
__device__ __forceinline__ d2 form(d2 q, d2 a, double alpha) { return d2{q.x - alpha * a.x, q.y - alpha * a.y}; }

constexpr int T = 256, PX = 128, PY = 4, HX = PX / 2;

template <int U>
__global__ __launch_bounds__(T) void lin(int64_t ntiles, double alpha, const double *__restrict__ r,
                                         const double *__restrict__ ap, double *__restrict__ s, double *__restrict__ acc)
{
    double sum = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * U; t < ntiles; t += (int64_t)gridDim.x * U) {
        d2 a[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (t + u) * 2 * T + 2 * threadIdx.x;
            a[u] = ld(ap + row);
            q[u] = ld(r + row);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (t + u) * 2 * T + 2 * threadIdx.x;
            d2 v = form(q[u], a[u], alpha);
            stnt(s + row, v);
            sum += v.x * v.y;
        }
    }
    acc[(int64_t)blockIdx.x * T + threadIdx.x] = sum;
}

// the linear sweep with the store's cache policy spelled out (gfx950: nt / sc0 / sc1 bits of global_store_dwordx4)
template <int POL>
__global__ __launch_bounds__(T) void lin_st(int64_t ntiles, double alpha, const double *__restrict__ r,
                                            const double *__restrict__ ap, double *__restrict__ s, double *__restrict__ acc)
{
    double sum = 0.0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row = t * 2 * T + 2 * threadIdx.x;
        const d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(ap + row));
        const d2 q = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(r + row));
        const d2 v = form(q, a, alpha);
        double *p = s + row;
        if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
        sum += v.x * v.y;
    }
    acc[(int64_t)blockIdx.x * T + threadIdx.x] = sum;
}
// write only / read only: what the part gives either way
template <int POL>
__global__ __launch_bounds__(T) void wr_only(int64_t ntiles, double *__restrict__ s)
{
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        double *p = s + t * 2 * T + 2 * threadIdx.x;
        const d2 v = d2{(double)t, 1.0};
        if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
        if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    }
}
__global__ __launch_bounds__(T) void rd_only(int64_t ntiles, const double *__restrict__ r, double *__restrict__ acc)
{
    double sum = 0.0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const d2 q = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(r + t * 2 * T + 2 * threadIdx.x));
        sum += q.x + q.y;
    }
    acc[(int64_t)blockIdx.x * T + threadIdx.x] = sum;
}

struct Geo {
    int64_t kdz, sdx;
    int npx, ncol, pps, nplanes;
};
__device__ __forceinline__ bool place(const Geo &g, int64_t &r0, int &p0, int &p1)
{
    const int cpx = (g.ncol + 7) >> 3, c = blockIdx.x & 7, sg = blockIdx.x >> 3;
    const int col = c * cpx + sg % cpx, seg = sg / cpx;
    if (col >= g.ncol) return false;
    p0 = seg * g.pps;
    p1 = min(g.nplanes, p0 + g.pps);
    const int py = col / g.npx, px = col % g.npx;
    const int t = threadIdx.x, y = t / HX, q = t % HX;
    r0 = (int64_t)(py * PY + y) * g.sdx + px * PX + 2 * q;
    return p0 < p1;
}

// plane above into registers, nothing else
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4))) void zreg(Geo g, double alpha,
                                                                                   const double *__restrict__ r,
                                                                                   const double *__restrict__ ap,
                                                                                   double *__restrict__ s, double *__restrict__ acc)
{
    int64_t r0;
    int p0, p1;
    if (!place(g, r0, p0, p1)) return;
    int64_t row = r0 + (int64_t)p0 * g.kdz;
    d2 xm = form(ld(r + row - g.kdz), ld(ap + row - g.kdz), alpha), xc = form(ld(r + row), ld(ap + row), alpha);
    double sum = 0.0;
    for (int p = p0; p < p1; ++p, row += g.kdz) {
        const d2 zp = form(ld(r + row + g.kdz), ld(ap + row + g.kdz), alpha);
        stnt(s + row, xc);
        sum += (xm.x + xc.x) * zp.x + (xm.y + xc.y) * zp.y;
        xm = xc;
        xc = zp;
    }
    acc[(int64_t)blockIdx.x * T + threadIdx.x] = sum;
}

// patch_pair's shape: centre plane through LDS (two buffers, one raw barrier per step), rim rows one plane ahead
template <int UNR>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4))) void zbar(Geo g, double alpha,
                                                                                   const double *__restrict__ r,
                                                                                   const double *__restrict__ ap,
                                                                                   double *__restrict__ s, double *__restrict__ acc)
{
    __shared__ double pbuf[2 * 2 * T];
    int64_t r0;
    int p0, p1;
    if (!place(g, r0, p0, p1)) return;
    const int t = threadIdx.x, y = t / HX;
    const bool rimrow = y == 0 || y == PY - 1;
    const int64_t roff = y == 0 ? -g.sdx : g.sdx;
    int64_t row = r0 + (int64_t)p0 * g.kdz;
    d2 xm = form(ld(r + row - g.kdz), ld(ap + row - g.kdz), alpha), xc = form(ld(r + row), ld(ap + row), alpha);
    d2 rim = d2{0, 0};
    if (rimrow) rim = form(ld(r + row + roff), ld(ap + row + roff), alpha);
    *reinterpret_cast<d2 *>(pbuf + 2 * t) = xc;
    double sum = 0.0;
    int step = 0;
    auto one = [&](d2 zp, d2 rimn) {
        double *cur = pbuf + (step & 1) * 2 * T, *nxt = pbuf + ((step + 1) & 1) * 2 * T;
        ++step;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        d2 ym = rim, yp = rim;
        if (y > 0) ym = *reinterpret_cast<const d2 *>(cur + 2 * (t - HX));
        if (y < PY - 1) yp = *reinterpret_cast<const d2 *>(cur + 2 * (t + HX));
        stnt(s + row, xc);
        sum += (xm.x + xc.x + ym.x + yp.x) * zp.x + (xm.y + xc.y + ym.y + yp.y) * zp.y;
        *reinterpret_cast<d2 *>(nxt + 2 * t) = zp;
        xm = xc;
        xc = zp;
        rim = rimn;
        row += g.kdz;
    };
    int p = p0;
    if constexpr (UNR == 2) {
        for (; p + 1 < p1; p += 2) {
            const d2 q1 = ld(r + row + g.kdz), a1 = ld(ap + row + g.kdz);
            const d2 q2 = ld(r + row + 2 * g.kdz), a2 = ld(ap + row + 2 * g.kdz);
            d2 rq1 = d2{0, 0}, ra1 = rq1, rq2 = rq1, ra2 = rq1;
            if (rimrow) {
                rq1 = ld(r + row + g.kdz + roff);
                ra1 = ld(ap + row + g.kdz + roff);
                rq2 = ld(r + row + 2 * g.kdz + roff);
                ra2 = ld(ap + row + 2 * g.kdz + roff);
            }
            one(form(q1, a1, alpha), form(rq1, ra1, alpha));
            one(form(q2, a2, alpha), form(rq2, ra2, alpha));
        }
    }
    for (; p < p1; ++p) {
        const d2 q1 = ld(r + row + g.kdz), a1 = ld(ap + row + g.kdz);
        d2 rq1 = d2{0, 0}, ra1 = rq1;
        if (rimrow) {
            rq1 = ld(r + row + g.kdz + roff);
            ra1 = ld(ap + row + g.kdz + roff);
        }
        one(form(q1, a1, alpha), form(rq1, ra1, alpha));
    }
    acc[(int64_t)blockIdx.x * T + threadIdx.x] = sum;
}

// The whole step of K2-in-K3 on 2-D tiles (patch_pair + k23's body): class bytes, the 28-class table in LDS, +-1 by lane
// shuffle with the edge lanes' loads, rim rows, the 7 products summed in order, S stored, three dot products -- and the
// step's requests made PF steps AHEAD of their use, the loop unrolled PF+1 times so that a request's registers are the ones
// the step that uses them reads (no loop-carried copy of data still in flight).
struct Raw {
    d2 q, a;      // R, AP of the thread's two cells, plane above the centre
    d2 rq, ra;    // ... of the rim row beside them (first / last patch row)
    double eq, ea; // ... of the cell beside the patch row's end (edge lanes)
    unsigned short cc;
};
template <int PF, int WPE>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(WPE))) void zfull(Geo g, double alpha,
                                                                                       const double *__restrict__ r,
                                                                                       const double *__restrict__ ap,
                                                                                       const unsigned char *__restrict__ cls,
                                                                                       const double *__restrict__ tab,
                                                                                       double *__restrict__ s, double *__restrict__ acc)
{
    __shared__ double pbuf[2 * 2 * T];
    __shared__ double tbl[28 * 7];
    for (int k = threadIdx.x; k < 28 * 7; k += T) tbl[k] = tab[k];
    int64_t r0;
    int p0, p1;
    if (!place(g, r0, p0, p1)) return;
    const int t = threadIdx.x, y = t / HX, q = t % HX;
    const bool rimrow = y == 0 || y == PY - 1;
    const int64_t roff = y == 0 ? -g.sdx : g.sdx;
    const bool edge = q == 0 || q == HX - 1;
    const int eoff = q == 0 ? -1 : 2;
    int64_t row = r0 + (int64_t)p0 * g.kdz;
    // what step `pl` (centre plane pl) takes from memory: plane pl+1's pair, its rim row, and the centre plane's edge cells and classes
    auto issue = [&](Raw &w, int64_t rw) { // rw = the thread's row in the centre plane of the step
        // the requests not every wave makes FIRST: the wait for an older request is counted as if they had not been
        // made, so in the waves that made them it also waits for as many of the oldest younger ones -- these
        if (rimrow) {
            w.rq = ld(r + rw + g.kdz + roff);
            w.ra = ld(ap + rw + g.kdz + roff);
        }
        if (edge) {
            w.eq = r[rw + eoff];
            w.ea = ap[rw + eoff];
        }
        w.cc = *reinterpret_cast<const unsigned short *>(cls + rw);
        w.q = ld(r + rw + g.kdz);
        w.a = ld(ap + rw + g.kdz);
    };
    d2 xm = form(ld(r + row - g.kdz), ld(ap + row - g.kdz), alpha), xc = form(ld(r + row), ld(ap + row), alpha);
    d2 rim = d2{0, 0};
    if (rimrow) rim = form(ld(r + row + roff), ld(ap + row + roff), alpha);
    *reinterpret_cast<d2 *>(pbuf + 2 * t) = xc;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    int step = 0;
    auto consume = [&](Raw &w) {
        double *cur = pbuf + (step & 1) * 2 * T, *nxt = pbuf + ((step + 1) & 1) * 2 * T;
        ++step;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        d2 ym = rim, yp = rim;
        if (y > 0) ym = *reinterpret_cast<const d2 *>(cur + 2 * (t - HX));
        if (y < PY - 1) yp = *reinterpret_cast<const d2 *>(cur + 2 * (t + HX));
        double left = 0.0, right = 0.0;
        const double ev = w.eq - alpha * w.ea;
        if (q == 0) left = ev;
        if (q == HX - 1) right = ev;
        const double l = __shfl_up(xc.y, 1, 64), rr = __shfl_down(xc.x, 1, 64);
        if (q != 0) left = l;
        if (q != HX - 1) right = rr;
        const d2 zp = form(w.q, w.a, alpha);
        const double *t0 = tbl + (w.cc & 0xFF) * 7, *t1 = tbl + (w.cc >> 8) * 7;
        double s0 = t0[0] * xm.x, s1 = t1[0] * xm.y;
        s0 = s0 + t0[1] * ym.x;
        s1 = s1 + t1[1] * ym.y;
        s0 = s0 + t0[2] * left;
        s1 = s1 + t1[2] * xc.x;
        s0 = s0 + t0[3] * xc.x;
        s1 = s1 + t1[3] * xc.y;
        s0 = s0 + t0[4] * xc.y;
        s1 = s1 + t1[4] * right;
        s0 = s0 + t0[5] * yp.x;
        s1 = s1 + t1[5] * yp.y;
        s0 = s0 + t0[6] * zp.x;
        s1 = s1 + t1[6] * zp.y;
        stnt(s + row, xc);
        acc0 = acc0 + xc.x * xc.x;
        acc0 = acc0 + xc.y * xc.y;
        acc1 = acc1 + s0 * xc.x;
        acc1 = acc1 + s1 * xc.y;
        acc2 = acc2 + s0 * s0;
        acc2 = acc2 + s1 * s1;
        *reinterpret_cast<d2 *>(nxt + 2 * t) = zp;
        xm = xc;
        xc = zp;
        // (taken out of the registers by every wave: where only the rim waves read them the others keep them "pending" in
        // the compiler's books, and the next request into them waits for nearly everything in flight)
        asm volatile("" : "+v"(w.rq), "+v"(w.ra));
        if (rimrow) rim = form(w.rq, w.ra, alpha);
        row += g.kdz;
    };
    Raw raw[PF + 1];
#pragma unroll
    for (int k = 0; k < PF; ++k) issue(raw[k], row + k * g.kdz);
    // (whole trips only -- up to PF planes at the segment's end are left out of this probe: an early way out of the
    // unrolled trip leaves its requests unread on one path into the loop header, and the compiler then waits for them
    // before the next trip's first request)
#pragma unroll 1
    for (int p = p0; p + PF < p1; p += PF + 1) {
#pragma unroll
        for (int u = 0; u <= PF; ++u) {
            issue(raw[(u + PF) % (PF + 1)], row + PF * g.kdz);
            consume(raw[u]);
        }
    }
    acc[(int64_t)blockIdx.x * T + t] = acc0 + acc1 + acc2;
}

// One stream in, one out, z-march, planes walked upwards (DIR = +1) or downwards (-1): does a kernel that walks DOWN find the
// planes the kernel before it wrote LAST (walking up) in the 256 MiB Infinity Cache?  (1 GiB per vector at 512^3.)
template <int DIR, bool NTST>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4))) void zcopy(Geo g, const double *__restrict__ in,
                                                                                    double *__restrict__ out, double *__restrict__ acc)
{
    int64_t r0;
    int p0, p1;
    if (!place(g, r0, p0, p1)) return;
    double sum = 0.0;
    for (int k = 0; k < p1 - p0; ++k) {
        const int p = DIR > 0 ? p0 + k : p1 - 1 - k;
        const int64_t row = r0 + (int64_t)p * g.kdz;
        const d2 v = ld(in + row);
        const d2 w = d2{v.x * 1.0000001, v.y * 0.9999999};
        if (NTST) stnt(out + row, w);
        else *reinterpret_cast<d2 *>(out + row) = w;
        sum += w.x + w.y;
    }
    acc[(int64_t)blockIdx.x * T + threadIdx.x] = sum;
}

__global__ void fill(double *r, double *ap, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        r[i] = (double)(i % 1009) * 1e-3;
        ap[i] = (double)((i * 7) % 1013) * 1e-3;
    }
}

int main()
{
    const int N = 512;
    const int64_t n = (int64_t)N * N * N, kdz = (int64_t)N * N, ghost = 8 * kdz;
    double *v[3], *acc;
    for (auto &p : v) {
        CK(hipMalloc(&p, (n + 2 * ghost) * 8));
        CK(hipMemset(p, 0, (n + 2 * ghost) * 8));
        p += ghost;
    }
    fill<<<1024, 256>>>(v[0] - ghost, v[1] - ghost, n + 2 * ghost);
    unsigned char *cls;
    CK(hipMalloc(&cls, n + 2 * ghost));
    {
        std::vector<unsigned char> hc(n + 2 * ghost);
        for (size_t i = 0; i < hc.size(); ++i) hc[i] = (unsigned char)((i * 2654435761u >> 7) % 28);
        CK(hipMemcpy(cls, hc.data(), hc.size(), hipMemcpyHostToDevice));
        cls += ghost;
    }
    double *tab;
    CK(hipMalloc(&tab, 28 * 7 * 8));
    {
        double ht[28 * 7];
        for (int i = 0; i < 28 * 7; ++i) ht[i] = 0.01 * (i % 13) - 0.05;
        CK(hipMemcpy(tab, ht, sizeof ht, hipMemcpyHostToDevice));
    }
    const size_t nacc = 4096 * (size_t)T;
    CK(hipMalloc(&acc, nacc * 8));
    std::vector<double> h(nacc);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        CK(hipMemset(acc, 0, nacc * 8));
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        CK(hipMemcpy(h.data(), acc, nacc * 8, hipMemcpyDeviceToHost));
        double tot = 0.0;
        for (double x : h) tot += x;
        printf("%-52s %8.1f us  %6.2f TB/s   sum %.12e\n", name, ms * 1e3, 24.0 * n / ms / 1e9, tot);
        fflush(stdout);
    };
    char nm[128];
    Geo g{kdz, N, N / PX, (int)(kdz / (PX * PY)), 0, N};
    for (int G : {512, 768, 1024}) {
        snprintf(nm, sizeof nm, "lin U=1 G=%d", G);
        run(nm, [&] { lin<1><<<G, T>>>(n / (2 * T), 0.3, v[0], v[1], v[2], acc); });
        snprintf(nm, sizeof nm, "lin U=2 G=%d", G);
        run(nm, [&] { lin<2><<<G, T>>>(n / (2 * T), 0.3, v[0], v[1], v[2], acc); });
    }
    {
        const int64_t nt_ = n / (2 * T);
        for (int G : {512, 1024}) {
#define LST(P_, NAME_)                                                                                                 \
    snprintf(nm, sizeof nm, "lin (nt loads) stores " NAME_ " G=%d", G);                                                \
    run(nm, [&] { lin_st<P_><<<G, T>>>(nt_, 0.3, v[0], v[1], v[2], acc); });
            LST(0, "plain") LST(1, "nt") LST(2, "sc1") LST(3, "sc0 sc1") LST(4, "sc0") LST(5, "sc1 nt") LST(6, "sc0 sc1 nt")
            snprintf(nm, sizeof nm, "write only (8 B/row; rate column x3) plain G=%d", G);
            run(nm, [&] { wr_only<0><<<G, T>>>(nt_, v[2]); });
            snprintf(nm, sizeof nm, "write only (8 B/row; rate column x3) nt G=%d", G);
            run(nm, [&] { wr_only<1><<<G, T>>>(nt_, v[2]); });
            snprintf(nm, sizeof nm, "write only (8 B/row; rate column x3) sc0 sc1 G=%d", G);
            run(nm, [&] { wr_only<3><<<G, T>>>(nt_, v[2]); });
            snprintf(nm, sizeof nm, "read only (8 B/row; rate column x3) G=%d", G);
            run(nm, [&] { rd_only<<<G, T>>>(nt_, v[0], acc); });
        }
    }
    {
        g.pps = N / 2;
        const int G = (g.ncol + 7) / 8 * 8 * 2;
        // a -> b by one launch, b -> c by the next: both walking up, or the second walking down; 16 B per row and launch
        auto pair = [&](const char *name, auto second) {
            run(name, [&] {
                zcopy<1, true><<<G, T>>>(g, v[0], v[1], acc);
                second();
            });
        };
        pair("pair: up (nt stores) then up     [2 launches]", [&] { zcopy<1, true><<<G, T>>>(g, v[1], v[2], acc); });
        pair("pair: up (nt stores) then DOWN   [2 launches]", [&] { zcopy<-1, true><<<G, T>>>(g, v[1], v[2], acc); });
        run("pair: up (plain stores) then up   [2 launches]", [&] {
            zcopy<1, false><<<G, T>>>(g, v[0], v[1], acc);
            zcopy<1, false><<<G, T>>>(g, v[1], v[2], acc);
        });
        run("pair: up (plain stores) then DOWN [2 launches]", [&] {
            zcopy<1, false><<<G, T>>>(g, v[0], v[1], acc);
            zcopy<-1, false><<<G, T>>>(g, v[1], v[2], acc);
        });
        run("alternating up / DOWN over a <-> b, plain stores [2 launches]", [&] {
            zcopy<1, false><<<G, T>>>(g, v[0], v[1], acc);
            zcopy<-1, false><<<G, T>>>(g, v[1], v[0], acc);
        });
        run("alternating up / DOWN over a <-> b, nt stores [2 launches]", [&] {
            zcopy<1, true><<<G, T>>>(g, v[0], v[1], acc);
            zcopy<-1, true><<<G, T>>>(g, v[1], v[0], acc);
        });
        run("up / up over a <-> b, nt stores [2 launches]", [&] {
            zcopy<1, true><<<G, T>>>(g, v[0], v[1], acc);
            zcopy<1, true><<<G, T>>>(g, v[1], v[0], acc);
        });
    }
    for (int nseg : {2}) {
        g.pps = (N + nseg - 1) / nseg;
        const int G = (g.ncol + 7) / 8 * 8 * nseg;
        snprintf(nm, sizeof nm, "zreg nseg=%d G=%d", nseg, G);
        run(nm, [&] { zreg<<<G, T>>>(g, 0.3, v[0], v[1], v[2], acc); });
        snprintf(nm, sizeof nm, "zbar nseg=%d G=%d", nseg, G);
        run(nm, [&] { zbar<1><<<G, T>>>(g, 0.3, v[0], v[1], v[2], acc); });
        snprintf(nm, sizeof nm, "zbar2 nseg=%d G=%d", nseg, G);
        run(nm, [&] { zbar<2><<<G, T>>>(g, 0.3, v[0], v[1], v[2], acc); });
#define FULL(PF_, W_)                                                                                                  \
    snprintf(nm, sizeof nm, "zfull PF=%d waves/SIMD<=%d nseg=%d G=%d", PF_, W_, nseg, G);                               \
    run(nm, [&] { zfull<PF_, W_><<<G, T>>>(g, 0.3, v[0], v[1], cls, tab, v[2], acc); });
        FULL(0, 4) FULL(0, 2) FULL(1, 3) FULL(1, 2) FULL(2, 2)
    }
    return 0;
}
