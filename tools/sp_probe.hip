// What bounds the pattern-only passes (k_sp_blk)?  The same streams (uint16 offsets + fp32 values, quads per lane,
// 1024-thread workgroups with 120 KiB of LDS tables, one per CU) with the work taken apart:
//   0  stream only: load the quads, add the values
//   1  + three LDS gathers per entry (random offsets), added as floats
//   2  + the float64 arithmetic of pass C (convert, two corrections, round to fp32, two sums)
//   3  = 2 + write the value back (pass C)
//   hipcc --offload-arch=gfx950 -O3 tools/sp_probe.hip -o /tmp/sp_probe && /tmp/sp_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNR>
__global__ __launch_bounds__(1024) void k(const us4* __restrict__ idx, f4* __restrict__ val, long long nquads,
                                          const double* __restrict__ tab, int bw, double* __restrict__ out) {
    extern __shared__ float lds[];
    float* t1 = lds;
    float* t2 = t1 + bw;
    float* t3 = t2 + bw;
    for (int g = threadIdx.x; g < bw; g += 1024) {
        t1[g] = (float)tab[g];
        t2[g] = (float)tab[bw + g];
        t3[g] = (float)tab[2 * bw + g];
    }
    __syncthreads();
    const long long per = (nquads + gridDim.x - 1) / gridDim.x;
    const long long q0 = per * blockIdx.x, q1 = min(nquads, q0 + per);
    double s1 = 0.0, s2 = 0.0;
    const double c1 = 1e-3, c2 = -2e-3;
    for (long long q = q0 + threadIdx.x; q < q1; q += 1024LL * UNR) {
        us4 g[UNR];
        f4 e[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long qq = min(q + 1024LL * u, q1 - 1);
            g[u] = __builtin_nontemporal_load(idx + qq);
            e[u] = __builtin_nontemporal_load(val + qq);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            f4 o = e[u];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (MODE == 0) { s1 += (double)e[u][m]; continue; }
                const int gi = g[u][m];
                if (MODE == 1) { s1 += (double)(t1[gi] + t2[gi] + t3[gi] + e[u][m]); continue; }
                double corr = c1 * (double)t1[gi];
                corr = fma(c2, (double)t2[gi], corr);
                double x = (double)e[u][m] - corr;
                const float r = (float)x;
                o[m] = r;
                x = (double)r;
                const double v = (double)t3[gi];
                s1 = fma(x, v, s1);
                s2 = fma(v, v, s2);
            }
            if (MODE == 3 && q + 1024LL * u < q1) __builtin_nontemporal_store(o, val + q + 1024LL * u);
        }
    }
    out[(long long)blockIdx.x * 1024 + threadIdx.x] = s1 + s2;
}

template <int MODE, int UNR>
static void run(const char* name, const us4* idx, f4* val, long long nquads, const double* tab, int bw, double* out,
                int blocks, double bytes_per_entry) {
    const size_t sh = 3 * (size_t)bw * sizeof(float);
    hipFuncSetAttribute((const void*)k<MODE, UNR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, UNR>), dim3(blocks), dim3(1024), sh, 0, idx, val, nquads, tab, bw, out);
    hipDeviceSynchronize();
    const int reps = 20;
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, UNR>), dim3(blocks), dim3(1024), sh, 0, idx, val, nquads, tab, bw, out);
    hipDeviceSynchronize();
    auto t1 = std::chrono::high_resolution_clock::now();
    const double us = std::chrono::duration<double>(t1 - t0).count() * 1e6 / reps;
    printf("%-44s blocks %4d  %7.1f us  %6.2f TB/s\n", name, blocks, us, bytes_per_entry * nquads * 4 / us * 1e-6);
}

int main() {
    const long long n = 50000000LL, nquads = n / 4;
    const int bw = 10240;
    std::vector<unsigned short> hi((size_t)n);
    std::vector<float> hv((size_t)n);
    srand(1);
    for (long long i = 0; i < n; ++i) { hi[i] = (unsigned short)(rand() % bw); hv[i] = (float)(rand() % 1000) * 1e-3f; }
    std::vector<double> ht(3 * bw);
    for (int i = 0; i < 3 * bw; ++i) ht[i] = (rand() % 1000) * 1e-3;
    us4* idx; f4* val; double *tab, *out;
    hipMalloc(&idx, n * 2); hipMalloc(&val, n * 4); hipMalloc(&tab, 3 * bw * 8); hipMalloc(&out, 4096 * 1024 * 8);
    hipMemcpy(idx, hi.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(val, hv.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(tab, ht.data(), 3 * bw * 8, hipMemcpyHostToDevice);
    for (int blocks : {256, 768, 1536}) {
        run<0, 6>("0 stream only", idx, val, nquads, tab, bw, out, blocks, 6);
        run<1, 6>("1 + LDS gathers (float adds)", idx, val, nquads, tab, bw, out, blocks, 6);
        run<2, 6>("2 + float64 arithmetic of pass C, no write", idx, val, nquads, tab, bw, out, blocks, 6);
        run<3, 6>("3 + write back (pass C)", idx, val, nquads, tab, bw, out, blocks, 10);
    }
    run<0, 8>("0 stream only, 8 quads in flight", idx, val, nquads, tab, bw, out, 768, 6);
    run<2, 8>("2 arithmetic, 8 quads in flight", idx, val, nquads, tab, bw, out, 768, 6);
    run<2, 4>("2 arithmetic, 4 quads in flight", idx, val, nquads, tab, bw, out, 768, 6);
    return 0;
}
