// Probe (round 5): what one tagged exchange costs between G cooperating workgroups, per step, as a function of the payload -- the number a cooperative
// decoder-step kernel (VERDICT r4 #3c) stands or falls with.  Every workgroup publishes V values per step as 8-byte words (16 + 16 value bits and the
// step's 16-bit tag in both halves, relaxed agent-scope stores: the protocol of bilstm_group_ks_kernel) and gathers the G * V values of its group by
// polling the words themselves (all of a thread's words in flight at once, re-read until every tag is the step's).  No arithmetic in between: the
// time per step IS the exchange.  Build: hipcc --offload-arch=gfx950 -O3 ll_exchange_probe.hip -o ll_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void ll_store(unsigned long long* p, float v, unsigned int tag) {
    const unsigned int u = __float_as_uint(v), t16 = tag & 0xffffu;
    __hip_atomic_store(p, (unsigned long long)((u & 0xffff0000u) | t16) | ((unsigned long long)((u << 16) | t16) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NMAX>
__global__ __launch_bounds__(512) void xchg_kernel(unsigned long long* buf, int G, int V, int steps, float* sink, unsigned int* err) {
    const int group = blockIdx.x / G, p = blockIdx.x % G, j = threadIdx.x, GV = G * V;
    unsigned long long* base = buf + (size_t)group * 2 * GV;
    float acc = (float)j;
    for (int s = 0; s < steps; ++s) {
        unsigned long long* b = base + (size_t)(s & 1) * GV;
        const unsigned int t16 = (unsigned int)(s + 1) & 0xffffu;
        for (int i = j; i < V; i += 512) ll_store(b + p * V + i, acc + (float)i, (unsigned int)(s + 1));
        unsigned long long w[NMAX];
        bool all = false;
        for (int spin = 0; spin < (1 << 20) && !all; ++spin) {
            all = true;
#pragma unroll
            for (int k = 0; k < NMAX; ++k) {
                const int i = j + 512 * k;
                w[k] = i < GV ? __hip_atomic_load(b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (int k = 0; k < NMAX; ++k) {
                const int i = j + 512 * k;
                if (i < GV && (((unsigned int)w[k] & 0xffffu) != t16 || ((unsigned int)(w[k] >> 32) & 0xffffu) != t16)) all = false;
            }
        }
        if (!all) atomicOr(err, 1u);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NMAX; ++k) sum += __uint_as_float(((unsigned int)w[k] & 0xffff0000u) | ((unsigned int)(w[k] >> 32) >> 16));
        acc = acc * 0.5f + sum * 1e-6f;
        __syncthreads();
    }
    sink[blockIdx.x * 512 + j] = acc;
}

int main() {
    const int steps = 200;
    unsigned long long* buf;
    float* sink;
    unsigned int* err;
    const size_t cap = (size_t)64 << 20;
    (void)hipMalloc(&buf, cap);
    (void)hipMalloc(&sink, 256 * 512 * sizeof(float));
    (void)hipMalloc(&err, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    struct Cfg { int G, V, groups; const char* what; };
    const Cfg cfgs[] = {
        {4, 64, 32, "BiLSTM H=256: 64 units per workgroup"},
        {4, 2048, 38, "decoder, 32-row tile x 64 units"},
        {4, 4096, 38, "decoder, 64-row tile x 64 units"},
        {4, 4096, 60, "same, 240 workgroups"},
        {8, 2048, 30, "decoder, 64-row tile x 32 units, 8 workgroups"},
        {2, 8192, 64, "decoder, 64-row tile x 128 units, 2 workgroups"},
        {4, 8192, 38, "decoder, 128-row tile x 64 units"},
    };
    for (const Cfg& c : cfgs) {
        const int GV = c.G * c.V, n = (GV + 511) / 512;
        if ((size_t)c.groups * 2 * GV * 8 > cap || n > 64) { printf("skip %s\n", c.what); continue; }
        float best = 1e30f;
        unsigned int herr = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemset(buf, 0, (size_t)c.groups * 2 * GV * 8);
            (void)hipMemset(err, 0, 4);
            (void)hipEventRecord(e0, 0);
            if (n <= 1) hipLaunchKernelGGL((xchg_kernel<1>), dim3(c.G * c.groups), dim3(512), 0, 0, buf, c.G, c.V, steps, sink, err);
            else if (n <= 16) hipLaunchKernelGGL((xchg_kernel<16>), dim3(c.G * c.groups), dim3(512), 0, 0, buf, c.G, c.V, steps, sink, err);
            else if (n <= 32) hipLaunchKernelGGL((xchg_kernel<32>), dim3(c.G * c.groups), dim3(512), 0, 0, buf, c.G, c.V, steps, sink, err);
            else hipLaunchKernelGGL((xchg_kernel<64>), dim3(c.G * c.groups), dim3(512), 0, 0, buf, c.G, c.V, steps, sink, err);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        }
        printf("G=%d workgroups x V=%5d values published each (%6.1f KB gathered per workgroup and step), %3d groups = %3d workgroups: %7.2f us per exchange%s   [%s]\n", c.G,
               c.V, GV * 8 / 1024.0, c.groups, c.G * c.groups, best * 1e3f / steps, herr ? "  TIMEOUT" : "", c.what);
    }
    return 0;
}
