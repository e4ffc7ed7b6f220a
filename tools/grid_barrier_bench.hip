// grid_barrier_bench.hip — what does a SCALAR-ONLY rendezvous inside a kernel cost against a kernel boundary?
//
// The reference's shipped problems (0.4-0.8 M unknowns) run five launches of ~9 us per BiCGSTAB iteration
// (src/solvers.f90:24-50): launch-bound.  Three of the five boundaries only carry SCALARS across (alpha after K1, omega after
// K3, beta after K4: the vectors a following K2 / K4 / K5 touches are the rows the same thread just produced), so K1+K2 and
// K3+K4+K5 could each be ONE launch with a rendezvous in between: every workgroup publishes its partial sum, waits until
// all have, reduces them.  No bulk data crosses, so no cache needs flushing: write-through partials, a monotonic arrival
// counter, agent-scope polling.  This measures that rendezvous in isolation, on the shape of the real thing:
//   (A) a chain of dependent launches, each: reduce the previous stage's W partials, sweep its rows (y = a*x + y), leave a partial;
//   (B) ONE launch running the same stages in a loop with the rendezvous between them.
// W workgroups of 256 threads, all co-resident (W <= 256 CUs x 8); the polling loop gives up after a bounded number of
// polls (a flag is set and every stage after it is skipped), so the grid always drains.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o /tmp/gbb && /tmp/gbb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double block_sum(double v, double *lds)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    v = ((lds[0] + lds[1]) + lds[2]) + lds[3];
    __syncthreads();
    return v;
}

// one stage: alpha from the producer's partials, y = alpha*x + y on this workgroup's rows, partial of y.y
__device__ __forceinline__ double stage_body(int64_t n, double alpha, const double *__restrict__ x, double *__restrict__ y)
{
    double acc = 0.0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i + 1 < n; i += (int64_t)gridDim.x * 512) {
        const double2 xv = *reinterpret_cast<const double2 *>(x + i);
        double2 yv = *reinterpret_cast<double2 *>(y + i);
        yv.x = alpha * xv.x + yv.x;
        yv.y = alpha * xv.y + yv.y;
        *reinterpret_cast<double2 *>(y + i) = yv;
        acc += yv.x * yv.x + yv.y * yv.y;
    }
    return acc;
}

__global__ __launch_bounds__(256) void stage_kernel(int64_t n, const double *part_in, double *part_out, const double *x, double *y)
{
    __shared__ double lds[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) a += part_in[i];
    const double alpha = 1e-30 * block_sum(a, lds);
    const double acc = block_sum(stage_body(n, alpha, x, y), lds);
    if (threadIdx.x == 0) part_out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void persistent_kernel(int64_t n, int stages, double *part /* 2 x W */, unsigned *counter,
                                                         unsigned base /* arrivals before this launch: the counter is monotonic */,
                                                         unsigned *gave_up, const double *x, double *y)
{
    __shared__ double lds[4];
    __shared__ int ok;
    const unsigned W = gridDim.x;
    for (int s = 0; s < stages; ++s) {
        double *pin = part + (size_t)(s & 1) * W, *pout = part + (size_t)((s + 1) & 1) * W;
        // ---- rendezvous: every workgroup's partial of stage s-1 is in memory
        if (threadIdx.x == 0) {
            int good = 1;
            if (s > 0) {
                const unsigned want = base + (unsigned)s * W;
                unsigned polls = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base < want - base) {
                    if (++polls > (1u << 22)) { good = 0; *gave_up = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) good = 0;
            ok = good;
        }
        __syncthreads();
        if (!ok) return; // (uniform per workgroup; a workgroup that leaves no longer arrives: the others give up too)
        double a = 0.0;
        for (int i = threadIdx.x; i < (int)W; i += 256) a += __hip_atomic_load(&pin[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double alpha = 1e-30 * block_sum(a, lds);
        const double acc = block_sum(stage_body(n, alpha, x, y), lds);
        if (threadIdx.x == 0) {
            __hip_atomic_store(&pout[blockIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the write-through partial has been acknowledged
            asm volatile("" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int64_t n : {400000ll, 800000ll, 3200000ll}) {
        for (int W : {512, 768, 1024}) {
            double *x, *y, *part;
            unsigned *cnt;
            CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&part, 2 * W * 8)); CK(hipMalloc(&cnt, 8));
            CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8)); CK(hipMemset(part, 0, 2 * W * 8)); CK(hipMemset(cnt, 0, 8));
            const int stages = 50, reps = 40;
            auto chain = [&] { for (int s = 0; s < stages; ++s) stage_kernel<<<W, 256, 0, st>>>(n, part + (size_t)(s & 1) * W, part + (size_t)((s + 1) & 1) * W, x, y); };
            chain(); CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) chain();
            CK(hipStreamSynchronize(st));
            const double us_launch = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * stages);
            unsigned base = 0;
            persistent_kernel<<<W, 256, 0, st>>>(n, stages, part, cnt, base, cnt + 1, x, y);
            base += (unsigned)stages * (unsigned)W;
            CK(hipStreamSynchronize(st));
            t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) {
                persistent_kernel<<<W, 256, 0, st>>>(n, stages, part, cnt, base, cnt + 1, x, y);
                base += (unsigned)stages * (unsigned)W;
            }
            CK(hipStreamSynchronize(st));
            const double us_pers = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * stages);
            unsigned h[2];
            CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
            printf("n=%8lld W=%4d: %.2f us per stage as dependent launches, %.2f us per stage inside one launch (rendezvous)%s\n",
                   (long long)n, W, us_launch, us_pers, h[1] ? "  [a poll gave up: NOT all workgroups were resident]" : "");
            CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(part)); CK(hipFree(cnt));
        }
    }
    return 0;
}
