// zmarch_bench.hip — on-box probe: what does the memory system deliver for the z-marching access pattern of the fused
// SpMV kernels (K2-in-K3: read R and AP, write S and AS, 32 B per row) against a linear sweep of the same four vectors,
// and does the size of the piece a workgroup touches per plane matter?
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/zmarch_bench.hip -o /tmp/zb && /tmp/zb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ d2 ld(const double *p) { return *reinterpret_cast<const d2 *>(p); }
__device__ __forceinline__ void stnt(double *p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p)); }

// linear: workgroup b takes tiles b, b + G, ... of 2*T rows, U tiles per trip
template <int T, int U>
__global__ __launch_bounds__(T) void lin(int64_t ntiles, double alpha, const double *__restrict__ r,
                                         const double *__restrict__ ap, double *__restrict__ s, double *__restrict__ as)
{
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
            d2 v = d2{q[u].x - alpha * a[u].x, q[u].y - alpha * a[u].y};
            stnt(s + row, v);
            stnt(as + row, d2{v.x * 0.5, v.y * 0.5});
        }
    }
}
// z-march: a workgroup owns U adjacent pieces of 2*T rows of the plane ("column") and walks pps planes; XCD label
// b & 7 owns a contiguous range of columns (as ec3d_tile_of does)
template <int T, int U>
__global__ __launch_bounds__(T) void zm(int64_t kdz, int ncol, int pps, int nplanes, double alpha,
                                        const double *__restrict__ r, const double *__restrict__ ap,
                                        double *__restrict__ s, double *__restrict__ as)
{
    const int cpx = (ncol + 7) >> 3, c = blockIdx.x & 7, sg = blockIdx.x >> 3;
    const int col = c * cpx + sg % cpx, seg = sg / cpx;
    if (col >= ncol) return;
    const int p0 = seg * pps, p1 = min(nplanes, p0 + pps);
    for (int p = p0; p < p1; ++p) {
        d2 a[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (int64_t)p * kdz + ((int64_t)col * U + u) * 2 * T + 2 * threadIdx.x;
            a[u] = ld(ap + row);
            q[u] = ld(r + row);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (int64_t)p * kdz + ((int64_t)col * U + u) * 2 * T + 2 * threadIdx.x;
            d2 v = d2{q[u].x - alpha * a[u].x, q[u].y - alpha * a[u].y};
            stnt(s + row, v);
            stnt(as + row, d2{v.x * 0.5, v.y * 0.5});
        }
    }
}

int main()
{
    const int N = 512;
    const int64_t n = (int64_t)N * N * N, kdz = (int64_t)N * N;
    double *v[4];
    for (auto &p : v) {
        CK(hipMalloc(&p, n * 8));
        CK(hipMemset(p, 0, n * 8));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, 32.0 * n / ms / 1e9);
        fflush(stdout);
    };
    char nm[128];
#define LIN(T, U, G)                                                                                                   \
    snprintf(nm, sizeof nm, "linear T=%d U=%d G=%d", T, U, G);                                                         \
    run(nm, [&] { lin<T, U><<<G, T>>>(n / (2 * T), 0.3, v[0], v[1], v[2], v[3]); });
#define ZM(T, U, NSEG)                                                                                                 \
    {                                                                                                                  \
        const int ncol = (int)(kdz / (2 * T * U)), cols8 = (ncol + 7) / 8 * 8, pps = (N + NSEG - 1) / NSEG;              \
        snprintf(nm, sizeof nm, "z-march T=%d U=%d nseg=%d (G=%d, %d KiB/plane/wg)", T, U, NSEG, cols8 * NSEG,          \
                 2 * T * U * 8 / 1024);                                                                                \
        run(nm, [&] { zm<T, U><<<cols8 * NSEG, T>>>(kdz, ncol, pps, N, 0.3, v[0], v[1], v[2], v[3]); });               \
    }
    LIN(256, 1, 768) LIN(256, 2, 768) LIN(256, 2, 256) LIN(256, 2, 1024) LIN(1024, 1, 256)
    ZM(256, 1, 1) ZM(256, 1, 2) ZM(256, 1, 3) ZM(256, 1, 4)
    ZM(256, 2, 2) ZM(256, 2, 4) ZM(256, 4, 4) ZM(256, 4, 8)
    ZM(512, 1, 2) ZM(512, 1, 4) ZM(1024, 1, 2) ZM(1024, 1, 4) ZM(1024, 1, 8)
    ZM(1024, 2, 4) ZM(1024, 2, 8)
    return 0;
}
