// bw_probe.hip -- achievable HBM read / copy bandwidth on MI355X for candidate access patterns of the
// RRI pass kernel.  Build: hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o tools/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef long long i64;
typedef float f4 __attribute__((ext_vector_type(4)));
#define float4 f4
#define make_float4(a,b,c,d) f4{a,b,c,d}

// A: grid-stride read, U independent 16-B loads per thread per iteration, sum to keep it alive
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd_stride(const float4* __restrict__ p, i64 nvec, float* out) {
    float acc = 0.f;
    const i64 stride = (i64)gridDim.x * 256;
    i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < nvec; i += stride) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
// B: each block owns a contiguous chunk
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd_chunk(const float4* __restrict__ p, i64 nvec, float* out) {
    const i64 per = (nvec + gridDim.x - 1) / gridDim.x;
    const i64 lo = (i64)blockIdx.x * per, hi = min(nvec, lo + per);
    float acc = 0.f;
    i64 i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < hi; i += 256) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
// C: the pass pattern: wave streams rows of a column panel (panel = 64 lanes * NCH vec), rows round-robin over 4 waves
template <int U, int NCH, bool NT>
__global__ __launch_bounds__(256) void rd_panel(const float* __restrict__ X, i64 ldx, int n, int rpb, int npanels, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int panel = blockIdx.x % npanels, rb = blockIdx.x / npanels;
    const int row0 = rb * rpb, row1 = min(n, row0 + rpb);
    const int colb = panel * 64 * 4 * NCH + lane * 4;
    float acc = 0.f;
    for (int r = row0 + wave; r < row1; r += 4 * U) {
        float4 v[U][NCH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + 4 * u;
            if (rr < row1) {
                const float4* xp = (const float4*)(X + (i64)rr * ldx + colb);
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[u][c] = NT ? __builtin_nontemporal_load(xp + c * 64) : xp[c * 64];
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[u][c] = make_float4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc += v[u][c].x + v[u][c].y + v[u][c].z + v[u][c].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// D: same but a wave takes a CONTIGUOUS run of rows (rows_per_wave) instead of round-robin
template <int U, int NCH, bool NT>
__global__ __launch_bounds__(256) void rd_panel_contig(const float* __restrict__ X, i64 ldx, int n, int rpb, int npanels, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int panel = blockIdx.x % npanels, rb = blockIdx.x / npanels;
    const int row0 = rb * rpb, row1 = min(n, row0 + rpb);
    const int rpw = (row1 - row0 + 3) / 4;
    const int w0 = row0 + wave * rpw, w1 = min(row1, w0 + rpw);
    const int colb = panel * 64 * 4 * NCH + lane * 4;
    float acc = 0.f;
    for (int r = w0; r < w1; r += U) {
        float4 v[U][NCH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u;
            if (rr < w1) {
                const float4* xp = (const float4*)(X + (i64)rr * ldx + colb);
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[u][c] = NT ? __builtin_nontemporal_load(xp + c * 64) : xp[c * 64];
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[u][c] = make_float4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc += v[u][c].x + v[u][c].y + v[u][c].z + v[u][c].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// copy variants
template <int U, bool NT>
__global__ __launch_bounds__(256) void cp_stride(const float4* __restrict__ s, float4* __restrict__ d, i64 nvec) {
    const i64 stride = (i64)gridDim.x * 256;
    i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
    for (; i < nvec; i += stride) d[i] = s[i];
}

// in-place read-modify-write (the access pattern of the rank-one residual update), grid-stride
template <int U, bool NT>
__global__ __launch_bounds__(256) void rmw_stride(float4* __restrict__ d, i64 nvec, float c) {
    const i64 stride = (i64)gridDim.x * 256;
    i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(d + i + u * stride) : d[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = v[u] - c; if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
    for (; i < nvec; i += stride) d[i] = d[i] - c;
}
// in-place, the pass pattern: block = 4 waves = 4 adjacent 1 KiB panels x a row block, U rows in flight per wave
template <int U, bool NT>
__global__ __launch_bounds__(256) void rmw_panel(float* __restrict__ X, i64 ldx, int n, int rpb, int npg, float c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = blockIdx.x % npg, rb = blockIdx.x / npg;
    const int row0 = rb * rpb, row1 = min(n, row0 + rpb);
    const i64 col = (i64)(pg * 4 + wave) * 256 + lane * 4;
    if (col >= ldx) return;
    for (int r = row0; r < row1; r += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4* xp = (float4*)(X + (i64)(r + u) * ldx + col);
            v[u] = (r + u < row1) ? (NT ? __builtin_nontemporal_load(xp) : *xp) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4* xp = (float4*)(X + (i64)(r + u) * ldx + col);
            v[u] = v[u] - c;
            if (r + u < row1) { if (NT) __builtin_nontemporal_store(v[u], xp); else *xp = v[u]; }
        }
    }
}

// the pass pattern with INTERLEAVED row chunks: block rb takes the chunks of U rows number rb, rb + nrb, rb + 2 nrb, ...
// so that the blocks running at one time cover one contiguous window of the matrix (as a linear stream does)
template <int U, bool NT, bool WRITE>
__global__ __launch_bounds__(256) void rmw_panel_il(float* __restrict__ X, i64 ldx, int n, int nrb, int npg, float c, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = blockIdx.x % npg, rb = blockIdx.x / npg;
    const i64 col = (i64)(pg * 4 + wave) * 256 + lane * 4;
    if (col >= ldx) return;
    float acc = 0.f;
    for (int r = rb * U; r < n; r += nrb * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4* xp = (float4*)(X + (i64)(r + u) * ldx + col);
            v[u] = (r + u < n) ? (NT ? __builtin_nontemporal_load(xp) : *xp) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4* xp = (float4*)(X + (i64)(r + u) * ldx + col);
            if (WRITE) {
                v[u] = v[u] - c;
                if (r + u < n) { if (NT) __builtin_nontemporal_store(v[u], xp); else *xp = v[u]; }
            } else {
                acc += v[u].x + v[u].y + v[u].z + v[u].w;
            }
        }
    }
    if (!WRITE && acc == 123.456f) out[0] = acc;
}

template <typename F>
double timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000, d = argc > 2 ? atoi(argv[2]) : 10000;
    const i64 elems = (i64)n * d, nvec = elems / 4;
    const double GB = elems * 4.0 / 1e9;
    float *X, *Y, *out;
    CK(hipMalloc(&X, elems * 4)); CK(hipMalloc(&Y, elems * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(X, 0x3c, elems * 4)); CK(hipMemset(Y, 0, elems * 4));
    const int reps = 10;
    printf("buffer %.2f GB (%d x %d fp32)\n", GB, n, d);
#define RUN(name, bytes, ...) { double ms = timeit([&] { __VA_ARGS__; }, reps); printf("%-46s %8.4f ms  %8.1f GB/s\n", name, ms, (bytes) / ms / 1e6); }
    for (int g : {1024, 2048, 4096, 8192}) {
        char nm[96];
        snprintf(nm, 96, "rd_stride U4 grid %d", g); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_stride<4, false>), dim3(g), dim3(256), 0, 0, (const float4*)X, nvec, out));
        snprintf(nm, 96, "rd_stride U8 grid %d", g); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_stride<8, false>), dim3(g), dim3(256), 0, 0, (const float4*)X, nvec, out));
        snprintf(nm, 96, "rd_stride U8 NT grid %d", g); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_stride<8, true>), dim3(g), dim3(256), 0, 0, (const float4*)X, nvec, out));
        snprintf(nm, 96, "rd_chunk U8 grid %d", g); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_chunk<8, false>), dim3(g), dim3(256), 0, 0, (const float4*)X, nvec, out));
    }
    {
        const int npanels = (d + 1023) / 1024;
        for (int wgs : {512, 1024, 2048, 4096}) {
            int nrb = wgs / npanels; if (nrb < 1) nrb = 1;
            int rpb = (n + nrb - 1) / nrb; nrb = (n + rpb - 1) / rpb;
            char nm[96];
            snprintf(nm, 96, "rd_panel U4 NCH4 wgs~%d (rpb %d)", wgs, rpb); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_panel<4, 4, false>), dim3(nrb * npanels), dim3(256), 0, 0, X, (i64)d, n, rpb, npanels, out));
            snprintf(nm, 96, "rd_panel U4 NCH4 NT wgs~%d", wgs); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_panel<4, 4, true>), dim3(nrb * npanels), dim3(256), 0, 0, X, (i64)d, n, rpb, npanels, out));
            snprintf(nm, 96, "rd_panel U8 NCH4 wgs~%d", wgs); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_panel<8, 4, false>), dim3(nrb * npanels), dim3(256), 0, 0, X, (i64)d, n, rpb, npanels, out));
            snprintf(nm, 96, "rd_panel_contig U4 NCH4 wgs~%d", wgs); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_panel_contig<4, 4, false>), dim3(nrb * npanels), dim3(256), 0, 0, X, (i64)d, n, rpb, npanels, out));
            snprintf(nm, 96, "rd_panel_contig U8 NCH4 NT wgs~%d", wgs); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_panel_contig<8, 4, true>), dim3(nrb * npanels), dim3(256), 0, 0, X, (i64)d, n, rpb, npanels, out));
        }
        // narrower panels: NCH 2 (512 cols) and 1 (256 cols)
        for (int nch : {2, 1}) {
            const int np2 = (d + 256 * nch - 1) / (256 * nch);
            int nrb = 2048 / np2; if (nrb < 1) nrb = 1;
            int rpb = (n + nrb - 1) / nrb; nrb = (n + rpb - 1) / rpb;
            char nm[96];
            snprintf(nm, 96, "rd_panel U8 NCH%d wgs~2048", nch);
            if (nch == 2) RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_panel<8, 2, false>), dim3(nrb * np2), dim3(256), 0, 0, X, (i64)d, n, rpb, np2, out))
            else RUN(nm, elems * 4.0, hipLaunchKernelGGL((rd_panel<8, 1, false>), dim3(nrb * np2), dim3(256), 0, 0, X, (i64)d, n, rpb, np2, out))
        }
    }
    for (int g : {2048, 8192}) {
        char nm[96];
        snprintf(nm, 96, "cp_stride U4 grid %d (r+w bytes)", g); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((cp_stride<4, false>), dim3(g), dim3(256), 0, 0, (const float4*)X, (float4*)Y, nvec));
        snprintf(nm, 96, "cp_stride U4 NT grid %d (r+w bytes)", g); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((cp_stride<4, true>), dim3(g), dim3(256), 0, 0, (const float4*)X, (float4*)Y, nvec));
    }
    for (int g : {2048, 8192, 32768}) {
        char nm[96];
        snprintf(nm, 96, "rmw_stride U4 grid %d (r+w bytes)", g); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_stride<4, false>), dim3(g), dim3(256), 0, 0, (float4*)Y, nvec, 1.0f));
        snprintf(nm, 96, "rmw_stride U4 NT grid %d (r+w bytes)", g); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_stride<4, true>), dim3(g), dim3(256), 0, 0, (float4*)Y, nvec, 1.0f));
        snprintf(nm, 96, "rmw_stride U8 NT grid %d (r+w bytes)", g); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_stride<8, true>), dim3(g), dim3(256), 0, 0, (float4*)Y, nvec, 1.0f));
        snprintf(nm, 96, "cp_stride U8 NT grid %d (r+w bytes)", g); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((cp_stride<8, true>), dim3(g), dim3(256), 0, 0, (const float4*)X, (float4*)Y, nvec));
    }
    {
        const int npg = (d + 1023) / 1024;
        for (int wgs : {2048, 8192}) {
            int nrb = wgs / npg; if (nrb < 1) nrb = 1;
            int rpb = (n + nrb - 1) / nrb; nrb = (n + rpb - 1) / rpb;
            char nm[96];
            snprintf(nm, 96, "rmw_panel U8 NT wgs~%d (rpb %d)", wgs, rpb); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_panel<8, true>), dim3(nrb * npg), dim3(256), 0, 0, Y, (i64)d, n, rpb, npg, 1.0f));
            snprintf(nm, 96, "rmw_panel U8 wgs~%d", wgs); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_panel<8, false>), dim3(nrb * npg), dim3(256), 0, 0, Y, (i64)d, n, rpb, npg, 1.0f));
            snprintf(nm, 96, "rmw_panel U4 NT wgs~%d", wgs); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_panel<4, true>), dim3(nrb * npg), dim3(256), 0, 0, Y, (i64)d, n, rpb, npg, 1.0f));
            snprintf(nm, 96, "rmw_panel U16 NT wgs~%d", wgs); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_panel<16, true>), dim3(nrb * npg), dim3(256), 0, 0, Y, (i64)d, n, rpb, npg, 1.0f));
        }
    }
    {
        const int npg = (d + 1023) / 1024;
        for (int wgs : {1024, 2048, 4096}) {
            const int nrb = wgs / npg;
            char nm[96];
            snprintf(nm, 96, "rmw_panel_il U8 NT wgs~%d (r+w bytes)", wgs); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_panel_il<8, true, true>), dim3(nrb * npg), dim3(256), 0, 0, Y, (i64)d, n, nrb, npg, 1.0f, out));
            snprintf(nm, 96, "rmw_panel_il U4 NT wgs~%d (r+w bytes)", wgs); RUN(nm, 2 * elems * 4.0, hipLaunchKernelGGL((rmw_panel_il<4, true, true>), dim3(nrb * npg), dim3(256), 0, 0, Y, (i64)d, n, nrb, npg, 1.0f, out));
            snprintf(nm, 96, "rd_panel_il U8 NT wgs~%d (read only)", wgs); RUN(nm, elems * 4.0, hipLaunchKernelGGL((rmw_panel_il<8, true, false>), dim3(nrb * npg), dim3(256), 0, 0, Y, (i64)d, n, nrb, npg, 1.0f, out));
        }
    }
    return 0;
}
