// lds_dma_probe.hip -- does an LDS-DMA ring (global_load_lds_dwordx4: HBM -> LDS with no VGPR destination) move the streaming
// pass of the library beyond what register staging reaches?  The pass's float64 work (row dots, column sums, and for the
// read-modify-write form the two rank-one terms and the store) is attached; the library's own k_pass runs beside it on the
// same buffer and launch geometry.  ("next" 3a / 4.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Irri_nmf_amd/csrc tools/lds_dma_probe.hip -o tools/lds_dma_probe
//   tools/lds_dma_probe [n d]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "rri_kernels.hpp"
using namespace rri;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// one 1 KiB wave-instruction: lane l's 16 bytes at gsrc land at LDS byte address lds_base + 16 l (M0 carries the base; the
// compiler reserves M0, so it is saved and restored inside the statement).  NT: non-temporal (the matrix is read once per pass).
template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The pass with its rows staged through a per-wave LDS ring of S slots (1 KiB each: one row of the wave's panel), C rows per
// chunk.  Geometry as k_pass: 4 waves = 4 adjacent panels x one row block of rpb rows (interleaved chunks when nrb_il > 0).
// UPD = 0: row dots + column sums of X.  UPD = 2: X <- X - a b^T - a2 b2^T first, stored, products of the new X.
// RS: the 8 row dots of a chunk are added through an LDS tile (wave_rowsum8_*: ~4 vector instructions per row) instead of six
// DPP steps per row (30); the tile is the chunk's own ring slots, free between the reads of the chunk and their refill.
template <int UPD, int S, int C, bool NT, bool RS = false>
__global__ __launch_bounds__(256) void k_pass_dma(float* __restrict__ X, i64 ldx, int n, int ncols, const double* __restrict__ trow,
                                                  const double* __restrict__ wcol, double* __restrict__ Ypart,
                                                  double* __restrict__ Zpart, i64 ldz, int rpb, int npg,
                                                  const double* __restrict__ avec, const double* __restrict__ bvec,
                                                  const double* __restrict__ avec2, const double* __restrict__ bvec2, int nrb_il) {
    constexpr int K = S / C;          // chunks the ring holds
    static_assert(S % C == 0 && K >= 2, "ring = whole chunks, at least two");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* ring = reinterpret_cast<f32x4*>(smem);                         // [4 waves][S][64 lanes]
    double* ysh = reinterpret_cast<double*>(smem + 4 * S * 1024);         // [4][rpb]
    double* wsh = ysh + 4 * rpb;
    double* ash = wsh + rpb;
    double* ash2 = ash + rpb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pg = blockIdx.x % npg, rb = blockIdx.x / npg;
    auto grow = [&](int li) -> int { return nrb_il > 0 ? ((li / C) * nrb_il + rb) * C + (li % C) : rb * rpb + li; };
    for (int i = threadIdx.x; i < rpb; i += 256) {
        const int g = grow(i);
        wsh[i] = g < n ? wcol[g] : 0.0;
        if (UPD > 0) { ash[i] = g < n ? avec[g] : 0.0; ash2[i] = g < n ? avec2[g] : 0.0; }
    }
    __syncthreads();
    const int col = (pg * 4 + wave) * 256 + lane * 4;
    if ((pg * 4 + wave) * 256 >= ncols) {               // wave-uniform: a panel beyond the matrix
        for (int i = lane; i < rpb; i += 64) ysh[wave * rpb + i] = 0.0;
    } else {
        const bool ok = col < ncols;
        const int colc = ok ? col : 0;                  // lanes beyond the last column load (and ignore) column 0
        double tv[4], zacc[4], bv[4], bv2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            zacc[e] = 0.0;
            tv[e] = ok ? trow[col + e] : 0.0;
            bv[e] = (UPD > 0 && ok) ? bvec[col + e] : 0.0;
            bv2[e] = (UPD > 1 && ok) ? bvec2[col + e] : 0.0;
        }
        // the compiler defers the wait for these ordinary loads to their first use -- inside the loop, where its vmcnt(0) would
        // drain the ring on every iteration: use them here, before the first LDS-DMA is issued
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(tv[e]), "v"(bv[e]), "v"(bv2[e]));
        f32x4* myring = ring + (size_t)wave * S * 64;
        // LDS byte offset of the wave's ring (address space 3 pointers are 32-bit offsets into the workgroup's allocation)
        const unsigned ring_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)wave * S * 1024u;
        // number of chunks of this block that start inside the matrix
        int nchunks = 0;
        for (int l0 = 0; l0 < rpb; l0 += C) { if (grow(l0) < n) nchunks = l0 / C + 1; else break; }
        auto issue = [&](int q) {                       // chunk q of the block into ring slots (q mod K) C ...
            const int r = grow(q * C);
#pragma unroll
            for (int u = 0; u < C; ++u) {
                const int rr = min(r + u, n - 1);       // rows beyond the matrix: the last row again (ignored below)
                glds16<NT>(X + (i64)rr * ldx + colc, ring_base + (unsigned)(((q % K) * C + u) * 1024));
            }
        };
        for (int q = 0; q < K && q < nchunks; ++q) issue(q);
        for (int q = 0; q < nchunks; ++q) {
            // chunk q must have landed.  Issued after it: in steady state K-1 chunks of loads and (UPD) K-1 chunks of stores;
            // near the ends fewer -- waiting for more than necessary is always safe
            if (q + K <= nchunks && q >= K) { if constexpr (UPD > 0) wait_vm<2 * C * (K - 1)>(); else wait_vm<C * (K - 1)>(); }
            else if (q + K <= nchunks) wait_vm<C * (K - 1)>();
            else wait_vm<0>();
            const int l0 = q * C, r = grow(l0);
            f32x4 x[C];
#pragma unroll
            for (int u = 0; u < C; ++u) x[u] = myring[((q % K) * C + u) * 64 + lane];
            double* tile = reinterpret_cast<double*>(myring + ((q % K) * C) * 64);     // RS: 8 x 72 doubles of the 8 KiB just read
            if constexpr (RS) { static_assert(!RS || C == 8, "8 rows per tile"); asm volatile("" ::: "memory"); }
#pragma unroll
            for (int u = 0; u < C; ++u) {
                const int rr = r + u;
                double wv = 0.0, na = 0.0, na2 = 0.0;
                if (rr < n) { wv = wsh[l0 + u]; if (UPD > 0) { na = -ash[l0 + u]; na2 = -ash2[l0 + u]; } }
                double xe[4] = {(double)x[u][0], (double)x[u][1], (double)x[u][2], (double)x[u][3]};
                if constexpr (UPD > 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { xe[e] = fma(na, bv[e], xe[e]); xe[e] = fma(na2, bv2[e], xe[e]); }
                    const f32x4 rounded = f32x4{(float)xe[0], (float)xe[1], (float)xe[2], (float)xe[3]};
                    if (rr < n && ok) stream_store<NT>(reinterpret_cast<f32x4*>(X + (i64)rr * ldx + col), rounded);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xe[e] = (double)rounded[e];
                }
                double yp = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e) { yp = fma(xe[e], tv[e], yp); zacc[e] = fma(wv, xe[e], zacc[e]); }
                if (!(rr < n && ok)) yp = 0.0;
                if constexpr (RS) wave_rowsum8_park(tile, u, lane, yp);
                else {
                    const double tot = wave_sum_lane63<double>(yp);
                    if (lane == 63) ysh[wave * rpb + l0 + u] = tot;
                }
            }
            if constexpr (RS) {
                const double tot = wave_rowsum8_finish(tile, lane);
                if ((lane & 7) == 0) ysh[wave * rpb + l0 + (lane >> 3)] = tot;
            }
            // the slots are free once the reads above have returned (their values were consumed): refill them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (q + K < nchunks) issue(q + K);
        }
        if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Zpart[(i64)rb * ldz + col + e] = zacc[e];
        }
        for (int i = nchunks * C + lane; i < rpb; i += 64) ysh[wave * rpb + i] = 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rpb; i += 256) {
        const int g = grow(i);
        if (g < n) Ypart[(i64)pg * n + g] = (ysh[i] + ysh[rpb + i]) + (ysh[2 * rpb + i] + ysh[3 * rpb + i]);
    }
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
    const i64 ld = (d + 3) / 4 * 4, elems = (i64)n * ld;
    const int npg = (int)((ld + 1023) / 1024);
    float* X; double *trow, *wcol, *avec, *bvec, *avec2, *bvec2, *Yp, *Zp, *Yr, *Zr;
    DevState* st;
    CK(hipMalloc(&X, elems * 4)); CK(hipMalloc(&st, sizeof(DevState))); CK(hipMemset(st, 0, sizeof(DevState)));
    std::vector<float> hx((size_t)std::min<i64>(elems, 1 << 24));
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u % 1000) / 1000.0);
    for (i64 off = 0; off < elems; off += (i64)hx.size())
        CK(hipMemcpy(X + off, hx.data(), (size_t)std::min<i64>((i64)hx.size(), elems - off) * 4, hipMemcpyHostToDevice));
    auto dvec = [&](double** p, i64 m, double scale) {
        std::vector<double> h((size_t)m);
        for (i64 i = 0; i < m; ++i) h[(size_t)i] = scale * ((i * 40503u % 997) / 997.0);
        CK(hipMalloc(p, m * 8)); CK(hipMemcpy(*p, h.data(), m * 8, hipMemcpyHostToDevice));
    };
    dvec(&trow, ld, 1.0); dvec(&wcol, n, 1.0); dvec(&avec, n, 1e-4); dvec(&bvec, ld, 1e-4); dvec(&avec2, n, -1e-4); dvec(&bvec2, ld, 1e-4);
    const int reps = 10;
    printf("X %d x %d fp32 (%.2f GB); row dots + column sums in float64 (k_pass's work)\n", n, d, elems * 4.0 / 1e9);
    const bool geo = argc > 3;            // tools/lds_dma_probe n d geo: the read-only pass over row-block sizes and ring shapes
    for (int upd : {0, 2}) {
        if (geo && upd) break;
        for (int wgs : {2048, 4096, 8192, 64, 96, 128, 160, 192, 248}) {
            if (!geo && wgs < 1000) break;
            if (geo && wgs >= 1000 && wgs != 2048) continue;
            int nrb = std::max(1, wgs / npg);
            int rpb = (int)(((n + nrb - 1) / nrb + 15) / 16 * 16);
            if (wgs < 1000) rpb = wgs;     // geo: the value is the row-block size itself
            nrb = (n + rpb - 1) / rpb;
            CK(hipMalloc(&Yp, (size_t)npg * n * 8)); CK(hipMalloc(&Zp, (size_t)nrb * ld * 8));
            CK(hipMalloc(&Yr, (size_t)npg * n * 8)); CK(hipMalloc(&Zr, (size_t)nrb * ld * 8));
            const int il = upd ? nrb : 0;
            const double bytes = elems * 4.0 * (upd ? 2 : 1);
            char nm[160];
            // the library's kernel on the same geometry
            {
                const size_t sh = ((5 + upd) * (size_t)rpb + 4 * 8 * 72) * sizeof(double);
                double ms;
                if (upd) ms = timeit([&] { hipLaunchKernelGGL((k_pass<float, true, true, 2, 16, true, false>), dim3(npg * nrb), dim3(256), sh, 0, X, ld, n, (int)ld, trow, wcol, Yr, Zr, ld, rpb, npg, avec, bvec, avec2, bvec2, (const double*)bvec /* b2 - b2sub */, (const DevState*)st, TgramJob{}, il); }, reps);
                else ms = timeit([&] { hipLaunchKernelGGL((k_pass<float, true, true, 0, 8, true, true>), dim3(npg * nrb), dim3(256), sh, 0, (const float*)X, ld, n, (int)ld, trow, wcol, Yr, Zr, ld, rpb, npg, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const DevState*)st, TgramJob{}, il); }, reps);
                snprintf(nm, 160, "UPD %d  library k_pass (registers)      wgs %5d rpb %4d", upd, npg * nrb, rpb);
                printf("%-64s %8.4f ms  %7.1f GB/s\n", nm, ms, bytes / ms / 1e6);
            }
#define RUN_DMA(UPD_, S_, C_, NT_) RUN_DMA2(UPD_, S_, C_, NT_, false)
#define RUN_DMA2(UPD_, S_, C_, NT_, RS_)                                                                                            \
    {                                                                                                                           \
        const size_t sh = 4 * (size_t)(S_) * 1024 + ((5 + UPD_) * (size_t)rpb) * sizeof(double);                                         \
        CK(hipFuncSetAttribute((const void*)k_pass_dma<UPD_, S_, C_, NT_, RS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        const double ms = timeit([&] { hipLaunchKernelGGL((k_pass_dma<UPD_, S_, C_, NT_, RS_>), dim3(npg * nrb), dim3(256), sh, 0, X, ld, n, (int)ld, trow, wcol, Yp, Zp, ld, rpb, npg, avec, bvec, avec2, bvec2, il); }, reps); \
        snprintf(nm, 160, "UPD %d  LDS-DMA ring S %2d C %d %s %s LDS %3zu KB wgs %5d", UPD_, S_, C_, NT_ ? "nt" : "  ", RS_ ? "tile" : "dpp ", sh / 1024, npg * nrb);            \
        printf("%-64s %8.4f ms  %7.1f GB/s\n", nm, ms, bytes / ms / 1e6);                                                       \
    }
            if (geo) {
                RUN_DMA(0, 16, 8, true) RUN_DMA2(0, 16, 8, true, true) RUN_DMA(0, 12, 4, true) RUN_DMA(0, 8, 4, true) RUN_DMA(0, 16, 4, true)
            } else if (upd == 0) {
                RUN_DMA2(0, 16, 8, true, true) RUN_DMA2(0, 24, 8, true, true) RUN_DMA(0, 8, 4, true) RUN_DMA(0, 16, 8, true) RUN_DMA(0, 16, 4, true) RUN_DMA(0, 24, 8, true) RUN_DMA(0, 32, 8, true) RUN_DMA(0, 16, 8, false) RUN_DMA2(0, 16, 8, true, true)
            } else {
                RUN_DMA2(2, 16, 8, true, true) RUN_DMA2(2, 24, 8, true, true) RUN_DMA(2, 8, 4, true) RUN_DMA(2, 16, 8, true) RUN_DMA(2, 16, 4, true) RUN_DMA(2, 24, 8, true) RUN_DMA(2, 32, 8, true) RUN_DMA(2, 16, 8, false)
            }
            // results: the read-only form must give the library's sums (same order per lane; the row dots meet in another order
            // only across the 4 waves -- identical here); the read-modify-write forms have moved X, so only finiteness is checked
            if (upd == 0) {
                std::vector<double> a((size_t)ld), b((size_t)ld);
                CK(hipMemcpy(a.data(), Zp, ld * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), Zr, ld * 8, hipMemcpyDeviceToHost));
                double dz = 0, nz = 0;
                for (i64 j = 0; j < d; ++j) { dz += (a[j] - b[j]) * (a[j] - b[j]); nz += b[j] * b[j]; }
                std::vector<double> ya((size_t)n), yb((size_t)n);
                CK(hipMemcpy(ya.data(), Yp, (size_t)n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(yb.data(), Yr, (size_t)n * 8, hipMemcpyDeviceToHost));
                double dy = 0, ny = 0;
                for (int i = 0; i < n; ++i) { dy += (ya[i] - yb[i]) * (ya[i] - yb[i]); ny += yb[i] * yb[i]; }
                printf("        check (last ring variant vs library, first row block / first panel): column sums %.2e, row dots %.2e\n", std::sqrt(dz / nz), std::sqrt(dy / ny));
            }
            CK(hipFree(Yp)); CK(hipFree(Zp)); CK(hipFree(Yr)); CK(hipFree(Zr));
        }
    }
    return 0;
}
