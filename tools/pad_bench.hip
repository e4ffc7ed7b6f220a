// pad_bench.hip — does the spacing between the 7 streams of K4 (5 reads, 2 writes) matter?
// One buffer, vector k at base + k*(n*8 + pad) bytes; K4-shaped nontemporal loop, 768 workgroups.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/pad_bench.hip -o /tmp/pb && /tmp/pb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k4like(int64_t ntiles, double alpha, double omega, const double *__restrict__ p,
                                              const double *__restrict__ sv, const double *__restrict__ as,
                                              const double *__restrict__ r0, double *__restrict__ x,
                                              double *__restrict__ rv, double *part)
{
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row = t * 512 + 2 * threadIdx.x;
        d2 xv = __builtin_nontemporal_load((const d2 *)(x + row)), pv = __builtin_nontemporal_load((const d2 *)(p + row)),
           s = __builtin_nontemporal_load((const d2 *)(sv + row)), a = __builtin_nontemporal_load((const d2 *)(as + row)),
           q = __builtin_nontemporal_load((const d2 *)(r0 + row));
        d2 xn = d2{(xv.x + alpha * pv.x) + omega * s.x, (xv.y + alpha * pv.y) + omega * s.y};
        d2 rn = d2{s.x - omega * a.x, s.y - omega * a.y};
        __builtin_nontemporal_store(xn, (d2 *)(x + row));
        __builtin_nontemporal_store(rn, (d2 *)(rv + row));
        acc0 = acc0 + rn.x * rn.x; acc0 = acc0 + rn.y * rn.y;
        acc1 = acc1 + rn.x * q.x;  acc1 = acc1 + rn.y * q.y;
    }
    acc0 += acc1;
    for (int off = 32; off > 0; off >>= 1) acc0 += __shfl_down(acc0, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x], acc0);
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 512;
    const int64_t n = N * N * N;
    const int64_t pads[] = {0, 256, 1024, 2048, 4096, 4096 + 256, 8192, 16384 + 1024, 65536, 65536 + 4096 + 256,
                            1 << 20, (1 << 20) + 65536 + 4096 + 256, (4 << 20) + 1024, (4 << 20) + 2048,
                            (4 << 20) + 1024 + 65536, 3 * 4096 + 768, 7 * 8192 + 1280, 33 * 4096, 129 * 2048};
    const int64_t maxpad = 8 << 20;
    char *base;
    double *part;
    CK(hipMalloc(&base, 6 * (n * 8 + maxpad) + maxpad));
    CK(hipMemset(base, 0, 6 * (n * 8 + maxpad) + maxpad));
    CK(hipMalloc(&part, 65536 * 8));
    CK(hipMemset(part, 0, 65536 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        for (int64_t pad : pads) {
            double *v[6];
            for (int k = 0; k < 6; ++k) v[k] = (double *)(base + k * (n * 8 + pad));
            auto run = [&] { k4like<<<768, 256>>>(n / 512, 0.5, 0.25, v[0], v[1], v[2], v[3], v[4], v[5], part); };
            run(); run();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) run();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("pad %9lld B  %.3f ms  %.0f GB/s\n", (long long)pad, ms / 10, 56.0 * n / (ms / 10) / 1e6);
        }
    return 0;
}
