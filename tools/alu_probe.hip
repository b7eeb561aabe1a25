// Issue rates of the f64 instructions the sparse passes are made of (per SIMD, 4 waves resident):
//   hipcc --offload-arch=gfx950 -O3 tools/alu_probe.hip -o /tmp/alu_probe && /tmp/alu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float f[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { f[i] = seed + threadIdx.x * 0.001f + i; d[i] = f[i] * 1.0001; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])); }
            if (OP == 1) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i])); }
            if (OP == 2) { asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7])); }
            if (OP == 3) { asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7])); }
            if (OP == 4) { asm volatile("v_mul_f64 %0, %1, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7])); }
            if (OP == 5) { asm volatile("v_add_f64 %0, %1, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7])); }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name) {
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    const int iters = 20000, blocks = 256 * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    auto t1 = std::chrono::high_resolution_clock::now();
    const double sec = std::chrono::duration<double>(t1 - t0).count();
    const double instr_per_simd = 4.0 * iters * 8;           // wave-instructions issued on one SIMD
    printf("%-16s %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", name, sec * 1e3,
           sec * 1e9 / instr_per_simd, sec * 2.4e9 / instr_per_simd);
    hipFree(out);
}

int main() {
    run<3>("v_fma_f32");
    run<2>("v_fma_f64");
    run<4>("v_mul_f64");
    run<5>("v_add_f64");
    run<0>("v_cvt_f64_f32");
    run<1>("v_cvt_f32_f64");
    return 0;
}
