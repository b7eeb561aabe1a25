// barrier_probe.hip -- what a grid barrier costs on this part (256 workgroups of 512 threads, one per CU), for the
// register-resident sweep (rri_onchip_kernels.hpp).  Variants:
//   0  one counter, atomic add (release) + every workgroup polling it, acquire fence in every wave    (first version)
//   1  the same, acquire fence in the polling wave only
//   2  flag array: workgroup b stores epoch to flag[b] (release), wave 0 polls all G flags, acquire in wave 0
//   3  flag array with relaxed agent-scope accesses only: no L2 write-back / invalidate at all (the data that crosses
//      workgroups must then be written and read with agent-scope accesses itself)
//   4  two levels: per-XCD counters (blockIdx % 8), the last arrival of an XCD adds to the global one, the last of
//      those publishes the epoch in 8 per-XCD flags (one cache line each) that the workgroups of that XCD poll
//   hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define AG __HIP_MEMORY_SCOPE_AGENT

template <int V>
__global__ __launch_bounds__(512) void k_bar(unsigned* mem, int iters, int G, double* sink) {
    unsigned* counter = mem;            // [0]
    unsigned* flags = mem + 64;         // [G]            (variant 2, 3)
    unsigned* xcnt = mem + 1024;        // [8][32]        per-XCD counters, one 128-byte line each
    unsigned* xflag = mem + 2048;       // [8][32]        per-XCD release flags
    const int tid = threadIdx.x, b = blockIdx.x;
    double acc = 0.0;
    for (int it = 1; it <= iters; ++it) {
        // a little cross-workgroup data: every workgroup writes one double, reads its neighbour's after the barrier
        if (tid == 0) {
            if (V == 3) __hip_atomic_store(reinterpret_cast<unsigned long long*>(sink + 512 + 256 * (it & 1)) + b, (unsigned long long)(it * 1000 + b), __ATOMIC_RELAXED, AG);
            else sink[512 + 256 * (it & 1) + b] = (double)(it * 1000 + b);
        }
        __syncthreads();
        if (V == 0 || V == 1) {
            if (tid == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, AG);
                const unsigned target = (unsigned)it * (unsigned)G;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, AG) < target) __builtin_amdgcn_s_sleep(2);
                if (V == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (V == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else if (V == 2 || V == 3) {
            if (tid == 0) __hip_atomic_store(flags + b, (unsigned)it, V == 2 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, AG);
            if (tid < 64) {
                bool done = false;
                while (!done) {
                    bool ok = true;
                    for (int q = tid; q < G; q += 64) ok = ok && (__hip_atomic_load(flags + q, __ATOMIC_RELAXED, AG) >= (unsigned)it);
                    done = __all(ok);
                    if (!done) __builtin_amdgcn_s_sleep(1);
                }
                if (V == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        } else if (V == 4) {
            if (tid == 0) {
                const int x = b & 7;
                const unsigned in_x = (unsigned)((G - x + 7) / 8);      // workgroups with blockIdx % 8 == x
                const unsigned prev = __hip_atomic_fetch_add(xcnt + 32 * x, 1u, __ATOMIC_RELEASE, AG);
                if (prev + 1 == (unsigned)it * in_x) {
                    const unsigned p2 = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, AG);
                    if (p2 + 1 == (unsigned)it * 8u)
                        for (int y = 0; y < 8; ++y) __hip_atomic_store(xflag + 32 * y, (unsigned)it, __ATOMIC_RELEASE, AG);
                }
                while (__hip_atomic_load(xflag + 32 * x, __ATOMIC_RELAXED, AG) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int nb = (b + 1) % G;
            if (V == 3) acc += (double)__hip_atomic_load(reinterpret_cast<unsigned long long*>(sink + 512 + 256 * (it & 1)) + nb, __ATOMIC_RELAXED, AG);
            else acc += sink[512 + 256 * (it & 1) + nb];
        }
        __syncthreads();      // (slots alternate with the iteration's parity: a workgroup one iteration ahead writes the other one)
    }
    if (tid == 0) sink[b] = acc;
}

template <int V>
void run(unsigned* mem, double* sink, int G, int iters) {
    hipMemset(mem, 0, 4096 * sizeof(unsigned));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_bar<V>, dim3(G), dim3(512), 0, 0, mem, 10, G, sink);
    hipDeviceSynchronize();
    hipMemset(mem, 0, 4096 * sizeof(unsigned));
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_bar<V>, dim3(G), dim3(512), 0, 0, mem, iters, G, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<double> h(G);
    hipMemcpy(h.data(), sink, G * sizeof(double), hipMemcpyDeviceToHost);
    // expected checksum of workgroup b: sum_it (it * 1000 + (b+1) % G) when every read saw the value of ITS iteration
    int stale = 0;
    for (int w = 0; w < G; ++w) {
        double want = 0;
        for (int it = 1; it <= iters; ++it) want += it * 1000 + (w + 1) % G;
        if (h[w] != want) ++stale;
    }
    printf("variant %d: %.2f us per barrier (%d iterations, %d workgroups), workgroups that read a stale value: %d\n", V,
           1e3 * ms / iters, iters, G, stale);
}

int main() {
    unsigned* mem; double* sink;
    hipMalloc(&mem, 4096 * sizeof(unsigned));
    hipMalloc(&sink, 2048 * sizeof(double));
    hipMemset(sink, 0, 2048 * sizeof(double));
    const int G = 256, iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(mem, sink, G, iters);
        run<1>(mem, sink, G, iters);
        run<2>(mem, sink, G, iters);
        run<3>(mem, sink, G, iters);
        run<4>(mem, sink, G, iters);
    }
    return 0;
}
