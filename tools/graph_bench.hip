// graph_bench.hip — is a launch-bound BiCGSTAB iteration (5 dependent small kernels) faster as a hipGraph?
// Dependent chain of K4-shaped kernels over n rows, 40 kernels per batch: stream launches vs graph replay.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/graph_bench.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = a * x[i] + y[i];
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int64_t n : {100000ll, 800000ll, 3000000ll, 12000000ll}) {
        double *x, *y;
        CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
        CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
        const int per = 40, reps = 50;
        auto batch = [&] { for (int k = 0; k < per; ++k) axpy<<<768, 256, 0, s>>>(n, 0.5, (k & 1) ? x : y, (k & 1) ? y : x); };
        batch(); CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) batch();
        CK(hipStreamSynchronize(s));
        double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * per);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        batch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * per);
        printf("n=%9lld: per kernel %.2f us (stream launches)  %.2f us (graph replay)\n", (long long)n, us_stream, us_graph);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
