// stream_bench.hip — on-box sweep of streaming-kernel shapes for the vector stages (K2/K4/K5).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/stream_bench.hip -o /tmp/sb && /tmp/sb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ d2 ld(const double *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
    return *reinterpret_cast<const d2 *>(p);
}
template <bool NT> __device__ __forceinline__ void st(double *p, d2 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p));
    else *reinterpret_cast<d2 *>(p) = v;
}

// K2 shape: 2 reads, 1 write, 1 dot.  U = tiles per loop trip, T = threads
template <int T, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(T) void k2like(int64_t ntiles, double alpha, const double *__restrict__ r,
                                            const double *__restrict__ ap, double *__restrict__ s, double *part)
{
    double acc = 0.0;
    const int64_t tile_rows = 2 * T;
    for (int64_t t = (int64_t)blockIdx.x * U; t < ntiles; t += (int64_t)gridDim.x * U) {
        d2 a[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (t + u) * tile_rows + 2 * threadIdx.x;
            if (t + u < ntiles) { a[u] = ld<NTL>(ap + row); q[u] = ld<NTL>(r + row); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t + u >= ntiles) break;
            const int64_t row = (t + u) * tile_rows + 2 * threadIdx.x;
            d2 v = d2{q[u].x - alpha * a[u].x, q[u].y - alpha * a[u].y};
            st<NTS>(s + row, v);
            acc = acc + v.x * v.x;
            acc = acc + v.y * v.y;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x], acc);
}

// K4 shape: 5 reads, 2 writes, 2 dots
template <int T, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(T) void k4like(int64_t ntiles, double alpha, double omega, const double *__restrict__ p,
                                            const double *__restrict__ sv, const double *__restrict__ as,
                                            const double *__restrict__ r0, double *__restrict__ x,
                                            double *__restrict__ rv, double *part)
{
    double acc0 = 0.0, acc1 = 0.0;
    const int64_t tile_rows = 2 * T;
    for (int64_t t = (int64_t)blockIdx.x * U; t < ntiles; t += (int64_t)gridDim.x * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t + u >= ntiles) break;
            const int64_t row = (t + u) * tile_rows + 2 * threadIdx.x;
            d2 xv = ld<NTL>(x + row), pv = ld<NTL>(p + row), s = ld<NTL>(sv + row), a = ld<NTL>(as + row),
               q = ld<NTL>(r0 + row);
            d2 xn = d2{(xv.x + alpha * pv.x) + omega * s.x, (xv.y + alpha * pv.y) + omega * s.y};
            d2 rn = d2{s.x - omega * a.x, s.y - omega * a.y};
            st<NTS>(x + row, xn);
            st<NTS>(rv + row, rn);
            acc0 = acc0 + rn.x * rn.x; acc0 = acc0 + rn.y * rn.y;
            acc1 = acc1 + rn.x * q.x;  acc1 = acc1 + rn.y * q.y;
        }
    }
    acc0 += acc1;
    for (int off = 32; off > 0; off >>= 1) acc0 += __shfl_down(acc0, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x], acc0);
}

// K4 shape with 4 CONSECUTIVE rows per thread (two adjacent 16-byte accesses per stream): a wave touches 2 KB of
// every stream per pair of instructions instead of 1 KB
template <int T, bool NTL, bool NTS>
__global__ __launch_bounds__(T) void k4wide(int64_t ntiles, double alpha, double omega, const double *__restrict__ p,
                                            const double *__restrict__ sv, const double *__restrict__ as,
                                            const double *__restrict__ r0, double *__restrict__ x,
                                            double *__restrict__ rv, double *part)
{
    double acc0 = 0.0, acc1 = 0.0;
    const int64_t tile_rows = 4 * T;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row = t * tile_rows + 4 * threadIdx.x;
        d2 xv[2], pv[2], s[2], a[2], q[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xv[h] = ld<NTL>(x + row + 2 * h); pv[h] = ld<NTL>(p + row + 2 * h); s[h] = ld<NTL>(sv + row + 2 * h);
            a[h] = ld<NTL>(as + row + 2 * h); q[h] = ld<NTL>(r0 + row + 2 * h);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d2 xn = d2{(xv[h].x + alpha * pv[h].x) + omega * s[h].x, (xv[h].y + alpha * pv[h].y) + omega * s[h].y};
            d2 rn = d2{s[h].x - omega * a[h].x, s[h].y - omega * a[h].y};
            st<NTS>(x + row + 2 * h, xn);
            st<NTS>(rv + row + 2 * h, rn);
            acc0 = acc0 + rn.x * rn.x; acc0 = acc0 + rn.y * rn.y;
            acc1 = acc1 + rn.x * q[h].x; acc1 = acc1 + rn.y * q[h].y;
        }
    }
    acc0 += acc1;
    for (int off = 32; off > 0; off >>= 1) acc0 += __shfl_down(acc0, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x], acc0);
}

// pure copy (float4-equivalent): the box's practical ceiling
template <int T> __global__ __launch_bounds__(T) void copyk(int64_t ntiles, const double *__restrict__ a, double *__restrict__ b)
{
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row = t * 2 * T + 2 * threadIdx.x;
        *reinterpret_cast<d2 *>(b + row) = *reinterpret_cast<const d2 *>(a + row);
    }
}
template <int T> __global__ __launch_bounds__(T) void readk(int64_t ntiles, const double *__restrict__ a, double *part)
{
    double acc = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row = t * 2 * T + 2 * threadIdx.x;
        d2 v = *reinterpret_cast<const d2 *>(a + row);
        acc += v.x + v.y;
    }
    if (acc == 1.2345) part[0] = acc;
}

template <class F> double timeit(F f, int reps = 10)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 512;
    const int64_t n = N * N * N;
    double *v[8], *part;
    for (auto &p : v) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
    CK(hipMalloc(&part, 65536 * 8)); CK(hipMemset(part, 0, 65536 * 8));
    const int grids[] = {256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 8192};
    printf("n=%lld\n", (long long)n);
#define RUN(NAME, BYTES, T, ...)                                                               \
    for (int g : grids) {                                                                      \
        const int64_t ntiles = n / (2 * T);                                                    \
        double ms = timeit([&] { __VA_ARGS__; });                                              \
        printf("%-28s T=%d grid=%5d  %.3f ms  %.0f GB/s\n", NAME, T, g, ms, BYTES * (double)n / ms / 1e6); \
    }
    RUN("copy", 16, 256, (copyk<256><<<g, 256>>>(ntiles, v[0], v[1])))
    RUN("read", 8, 256, (readk<256><<<g, 256>>>(ntiles, v[0], part)))
    RUN("k2 base", 24, 256, (k2like<256, 1, false, false><<<g, 256>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k2 U2", 24, 256, (k2like<256, 2, false, false><<<g, 256>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k2 U4", 24, 256, (k2like<256, 4, false, false><<<g, 256>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k2 ntstore", 24, 256, (k2like<256, 1, false, true><<<g, 256>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k2 ntload+store", 24, 256, (k2like<256, 1, true, true><<<g, 256>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k2 U2 nt", 24, 256, (k2like<256, 2, true, true><<<g, 256>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k2 T512", 24, 512, (k2like<512, 1, false, false><<<g, 512>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k2 T1024", 24, 1024, (k2like<1024, 1, false, false><<<g, 1024>>>(ntiles, 0.5, v[0], v[1], v[2], part)))
    RUN("k4 base", 56, 256, (k4like<256, 1, false, false><<<g, 256>>>(ntiles, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part)))
    RUN("k4 U2", 56, 256, (k4like<256, 2, false, false><<<g, 256>>>(ntiles, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part)))
    RUN("k4 nt", 56, 256, (k4like<256, 1, true, true><<<g, 256>>>(ntiles, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part)))
    RUN("k4 ntstore", 56, 256, (k4like<256, 1, false, true><<<g, 256>>>(ntiles, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part)))
    for (int g : grids) {
        const int64_t ntiles = n / (4 * 256);
        double ms = timeit([&] { k4wide<256, true, true><<<g, 256>>>(ntiles, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part); });
        printf("%-28s T=%d grid=%5d  %.3f ms  %.0f GB/s\n", "k4 wide4 nt", 256, g, ms, 56 * (double)n / ms / 1e6);
    }
    RUN("k4 T512", 56, 512, (k4like<512, 1, false, false><<<g, 512>>>(ntiles, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part)))
    return 0;
}
