// Which HIP streams share a hardware pipe on MI355X?  N streams are created in order; every pair (i, j) then runs two chains of DEPENDENT launches at the same
// time (one chain per stream, each launch ~8 us on a quarter of the chip) and the wall time is compared with one chain alone.  Streams whose queues sit on the
// same compute pipe take ~2x (the pipe advances one queue at a time); streams on different pipes overlap.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/queue_pipe_probe tools/probe/queue_pipe_probe.hip && /tmp/queue_pipe_probe [n_streams] [launches]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(float* p, int iters) {
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0000001f, 1e-7f);
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

static double run(const std::vector<hipStream_t>& ss, float** buf, int launches, int iters) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int l = 0; l < launches; ++l)
        for (size_t k = 0; k < ss.size(); ++k) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, ss[k], buf[k], iters);
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8, launches = argc > 2 ? atoi(argv[2]) : 300, iters = 4000;
    std::vector<hipStream_t> s(n);
    for (int i = 0; i < n; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    float* buf[2];
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&buf[k], 64 * 256 * 4)); CK(hipMemset(buf[k], 0, 64 * 256 * 4)); }
    for (int i = 0; i < n; ++i) run({s[i]}, buf, 20, iters);  // every queue exists and is warm
    double alone = 1e30;
    for (int r = 0; r < 3; ++r) alone = std::min(alone, run({s[0]}, buf, launches, iters));
    printf("one chain of %d dependent launches alone: %.2f ms (%.1f us per launch)\npair time / alone (rows i, columns j):\n      ", launches, alone, alone * 1e3 / launches);
    for (int j = 0; j < n; ++j) printf("  s%-3d", j);
    printf("\n");
    for (int i = 0; i < n; ++i) {
        printf("s%-3d  ", i);
        for (int j = 0; j < n; ++j) {
            if (j <= i) { printf("   .  "); continue; }
            double t = 1e30;
            for (int r = 0; r < 2; ++r) t = std::min(t, run({s[i], s[j]}, buf, launches, iters));
            printf(" %5.2f", t / alone);
        }
        printf("\n");
    }
    // the legacy default stream against each
    printf("null  ");
    for (int j = 0; j < n; ++j) {
        double t = 1e30;
        for (int r = 0; r < 2; ++r) t = std::min(t, run({(hipStream_t)0, s[j]}, buf, launches, iters));
        printf(" %5.2f", t / alone);
    }
    printf("\n");
    return 0;
}
