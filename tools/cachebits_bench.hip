// cachebits_bench.hip — which gfx950 cache-policy bits stream best for the K4 shape (5 reads, 2 writes)?
// Loads/stores as inline asm with every combination of sc0 / sc1 / nt; 768 workgroups, 512^3 rows.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/cachebits_bench.hip -o /tmp/cb && /tmp/cb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define DEF_LD(NAME, BITS)                                                                     \
    __device__ __forceinline__ d2 NAME(const double *p)                                        \
    {                                                                                          \
        d2 v;                                                                                  \
        asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(v) : "v"(p) : "memory");   \
        return v;                                                                              \
    }
#define DEF_ST(NAME, BITS)                                                                     \
    __device__ __forceinline__ void NAME(double *p, d2 v)                                      \
    {                                                                                          \
        asm volatile("global_store_dwordx4 %0, %1, off " BITS : : "v"(p), "v"(v) : "memory");  \
    }
DEF_LD(ld_none, "")
DEF_LD(ld_nt, "nt")
DEF_LD(ld_sc1, "sc1")
DEF_LD(ld_sc0sc1, "sc0 sc1")
DEF_LD(ld_sc1nt, "sc1 nt")
DEF_LD(ld_all, "sc0 sc1 nt")
DEF_ST(st_none, "")
DEF_ST(st_nt, "nt")
DEF_ST(st_sc1, "sc1")
DEF_ST(st_sc0sc1, "sc0 sc1")
DEF_ST(st_sc1nt, "sc1 nt")
DEF_ST(st_all, "sc0 sc1 nt")

template <int LD, int ST>
__global__ __launch_bounds__(256) void k4like(int64_t ntiles, double alpha, double omega, const double *p,
                                              const double *sv, const double *as, const double *r0, double *x,
                                              double *rv, double *part)
{
    auto ld = [](const double *q) {
        return LD == 0 ? ld_none(q) : LD == 1 ? ld_nt(q) : LD == 2 ? ld_sc1(q) : LD == 3 ? ld_sc0sc1(q)
               : LD == 4 ? ld_sc1nt(q) : ld_all(q);
    };
    auto st = [](double *q, d2 v) {
        if (ST == 0) st_none(q, v); else if (ST == 1) st_nt(q, v); else if (ST == 2) st_sc1(q, v);
        else if (ST == 3) st_sc0sc1(q, v); else if (ST == 4) st_sc1nt(q, v); else st_all(q, v);
    };
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row = t * 512 + 2 * threadIdx.x;
        d2 xv = ld(x + row), pv = ld(p + row), s = ld(sv + row), a = ld(as + row), q = ld(r0 + row);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        d2 xn = d2{(xv.x + alpha * pv.x) + omega * s.x, (xv.y + alpha * pv.y) + omega * s.y};
        d2 rn = d2{s.x - omega * a.x, s.y - omega * a.y};
        st(x + row, xn);
        st(rv + row, rn);
        acc0 = acc0 + rn.x * rn.x; acc0 = acc0 + rn.y * rn.y;
        acc1 = acc1 + rn.x * q.x;  acc1 = acc1 + rn.y * q.y;
    }
    acc0 += acc1;
    for (int off = 32; off > 0; off >>= 1) acc0 += __shfl_down(acc0, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x], acc0);
}

template <int LD, int ST> double run(int64_t n, double **v, double *part)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto f = [&] { k4like<LD, ST><<<768, 256>>>(n / 512, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part); };
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10;
}

int main()
{
    const int64_t n = 512ll * 512 * 512;
    double *v[6], *part;
    for (auto &p : v) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
    CK(hipMalloc(&part, 65536 * 8)); CK(hipMemset(part, 0, 65536 * 8));
    const char *names[6] = {"none", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
#define ROW(L)                                                                                         \
    { double t[6] = {run<L, 0>(n, v, part), run<L, 1>(n, v, part), run<L, 2>(n, v, part),               \
                     run<L, 3>(n, v, part), run<L, 4>(n, v, part), run<L, 5>(n, v, part)};              \
      printf("load %-10s:", names[L]); for (int s = 0; s < 6; ++s) printf("  %s %.3f", names[s], t[s]); printf("  ms\n"); }
    printf("columns: store policy\n");
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
    return 0;
}
