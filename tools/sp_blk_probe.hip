// sp_blk_probe.hip -- the pattern-only read-modify-write pass (k_sp_blk, rri_sparse_kernels.hpp) on a realistic blocked
// copy of a random pattern (100000 x 10000, 5 % observed: BASELINE config 5), outside the library: the shipped kernel
// against candidate kernels on the SAME copy, work list and factors; results compared entry by entry, times from HIP
// events over alternating launches.  Both orientations (rows as segments / columns as segments).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Irri_nmf_amd/csrc -Iinclude -Itools tools/sp_blk_probe.hip -o tools/sp_blk_probe
//   tools/sp_blk_probe [n d density reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "sp_blk_candidates.hpp"

using namespace rri;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

struct Copy {
    int nblk, bw, lps;
    i64 nseg, gdim, count;
    std::vector<i64> segptr;
    std::vector<unsigned short> idx_old, idx_new;   // pads: SP_PAD (shipped kernel) / the zero slot bw (candidates)
    std::vector<SpWork> work_old, work_new;
};

static int g_items = 3 * 256;
static Copy build(const std::vector<i64>& indptr, const std::vector<int>& indices, i64 n, i64 d, int w, int max_segs_new) {
    Copy cp;
    const i64 nnz = indptr[n];
    cp.gdim = w == 0 ? d : n;
    cp.nseg = w == 0 ? n : d;
    const int cap = SP_BLOCK_BYTES / (3 * 4);
    cp.nblk = (int)((cp.gdim + cap - 1) / cap);
    cp.bw = (int)(((cp.gdim + cp.nblk - 1) / cp.nblk + 3) / 4 * 4);
    if (cp.bw > cap) cp.bw = cap;
    const i64 nseg = cp.nseg, stride = nseg + 1;
    std::vector<i64>& sp = cp.segptr;
    sp.assign((size_t)cp.nblk * stride, 0);
    for (i64 r = 0; r < n; ++r)
        for (i64 p = indptr[r]; p < indptr[r + 1]; ++p) {
            const i64 j = indices[p];
            const i64 g = w == 0 ? j : r, sgm = w == 0 ? r : j;
            sp[(size_t)((g / cp.bw) * stride + sgm + 1)] += 1;
        }
    i64 run = 0;
    for (int b = 0; b < cp.nblk; ++b) {
        i64* row = sp.data() + (size_t)b * stride;
        row[0] = run;
        for (i64 q = 1; q <= nseg; ++q) { run += (row[q] + 3) / 4 * 4; row[q] = run; }
    }
    cp.count = run;
    cp.idx_old.assign((size_t)run, SP_PAD);
    cp.idx_new.assign((size_t)run, (unsigned short)cp.bw);
    std::vector<i64> fill((size_t)cp.nblk * nseg);
    for (int b = 0; b < cp.nblk; ++b)
        for (i64 q = 0; q < nseg; ++q) fill[(size_t)b * nseg + q] = sp[(size_t)b * stride + q];
    for (i64 r = 0; r < n; ++r)
        for (i64 p = indptr[r]; p < indptr[r + 1]; ++p) {
            const i64 j = indices[p];
            const i64 g = w == 0 ? j : r, sgm = w == 0 ? r : j;
            const i64 b = g / cp.bw;
            const i64 q = fill[(size_t)(b * nseg + sgm)]++;
            cp.idx_old[(size_t)q] = cp.idx_new[(size_t)q] = (unsigned short)(g - b * cp.bw);
        }
    const i64 per_item = std::max<i64>(4096, nnz / g_items);
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<SpWork>& work = variant == 0 ? cp.work_old : cp.work_new;
        const i64 max_segs = variant == 0 ? 8192 : max_segs_new;
        for (int b = 0; b < cp.nblk; ++b) {
            const i64* row = sp.data() + (size_t)b * stride;
            i64 s0 = 0;
            while (s0 < nseg) {
                i64 s1 = s0 + 1;
                while (s1 < nseg && s1 - s0 < max_segs && row[s1 + 1] - row[s0] <= per_item) ++s1;
                work.push_back(SpWork{b, (int)s0, (int)s1, 0});
                s0 = s1;
            }
        }
    }
    cp.idx_new.resize((size_t)run + (size_t)4 * SP2_DUMP_QUADS * cp.work_new.size(), (unsigned short)cp.bw);   // the dump quads of k_sp_blk2: pads
    const i64 avg = nnz / std::max<i64>(1, (i64)cp.nblk * nseg);
    cp.lps = avg >= 768 ? 64 : avg >= 384 ? 32 : avg >= 192 ? 16 : 8;
    return cp;
}

template <typename T>
static T* to_dev(const std::vector<T>& v) {
    T* p = nullptr;
    CK(hipMalloc((void**)&p, std::max<size_t>(1, v.size()) * sizeof(T)));
    if (!v.empty()) CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

struct Dev {
    i64* segptr;
    unsigned short *idx_old, *idx_new;
    SpWork *work_old, *work_new;
    float *val0, *val_a, *val_b;
    double *B1, *B2, *V, *A1, *A2, *S1a, *S2a, *S1b, *S2b;
};

template <int LPS, bool DO_S = true, bool WRITE = true>
static void launch_old(const Copy& cp, const Dev& dv, float* val, double* S1, double* S2, hipStream_t st, const DevState* ds) {
    const size_t sh = 3 * (size_t)cp.bw * sizeof(float);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)k_sp_blk_r2<float, DO_S, true, WRITE, LPS>, hipFuncAttributeMaxDynamicSharedMemorySize, SP_BLOCK_BYTES)); set = true; }
    hipLaunchKernelGGL((k_sp_blk_r2<float, DO_S, true, WRITE, LPS>), dim3((unsigned)cp.work_old.size()), dim3(1024), sh, st,
                       (const SpWork*)dv.work_old, (const i64*)dv.segptr, cp.nseg, (const unsigned short*)dv.idx_old, val, cp.bw,
                       cp.gdim, (const double*)dv.B1, (const double*)dv.B2, (const double*)dv.V, (const double*)dv.A1,
                       (const double*)dv.A2, S1, S2, cp.nseg, ds);
}
template <int LPS, int Q, bool DO_S = true, bool WRITE = true, int MEM = 0>
static void launch_new(const Copy& cp, const Dev& dv, float* val, double* S1, double* S2, hipStream_t st, const DevState* ds) {
    const size_t sh = sp2_lds_bytes<float>(cp.bw);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)k_sp_blk2<float, DO_S, true, WRITE, LPS, Q, MEM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL((k_sp_blk2<float, DO_S, true, WRITE, LPS, Q, MEM>), dim3((unsigned)cp.work_new.size()), dim3(1024), sh, st,
                       (const SpWork*)dv.work_new, (const i64*)dv.segptr, cp.nseg, (const unsigned short*)dv.idx_new, val, cp.count, cp.bw,
                       cp.gdim, (const double*)dv.B1, (const double*)dv.B2, (const double*)dv.V, (const double*)dv.A1,
                       (const double*)dv.A2, S1, S2, cp.nseg, ds);
}

template <int LPS>
static void launch_lib(const Copy& cp, const Dev& dv, float* val, double* S1, double* S2, hipStream_t st, const DevState* ds) {
    const size_t sh = sp_lds_bytes<float>(cp.bw);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)k_sp_blk<float, true, true, true, LPS>, hipFuncAttributeMaxDynamicSharedMemorySize, SP_BLOCK_BYTES + 64)); set = true; }
    hipLaunchKernelGGL((k_sp_blk<float, true, true, true, LPS>), dim3((unsigned)cp.work_old.size()), dim3(1024), sh, st,
                       (const SpWork*)dv.work_old, (const i64*)dv.segptr, cp.nseg, (const unsigned short*)dv.idx_new, val, cp.bw,
                       cp.gdim, (const double*)dv.B1, (const double*)dv.B2, (const double*)dv.V, (const double*)dv.A1,
                       (const double*)dv.A2, S1, S2, cp.nseg, ds);
}
template <int LPS, bool PK, bool F32C, int MEM = 0>
static void launch_3(const Copy& cp, const Dev& dv, float* val, double* S1, double* S2, hipStream_t st, const DevState* ds) {
    const size_t sh = 3 * (size_t)(cp.bw + 1) * sizeof(float);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)k_sp_blk3<float, true, true, true, LPS, PK, F32C, MEM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL((k_sp_blk3<float, true, true, true, LPS, PK, F32C, MEM>), dim3((unsigned)cp.work_old.size()), dim3(1024), sh, st,
                       (const SpWork*)dv.work_old, (const i64*)dv.segptr, cp.nseg, (const unsigned short*)dv.idx_new, val, cp.bw,
                       cp.gdim, (const double*)dv.B1, (const double*)dv.B2, (const double*)dv.V, (const double*)dv.A1,
                       (const double*)dv.A2, S1, S2, cp.nseg, ds);
}

int main(int argc, char** argv) {
    const i64 n = argc > 1 ? atoll(argv[1]) : 100000, d = argc > 2 ? atoll(argv[2]) : 10000;
    const double dens = argc > 3 ? atof(argv[3]) : 0.05;
    const int reps = argc > 4 ? atoi(argv[4]) : 10;
    if (argc > 5) g_items = atoi(argv[5]);
    std::mt19937_64 rng(7);
    std::vector<i64> indptr(n + 1, 0);
    std::vector<int> indices;
    indices.reserve((size_t)(n * d * dens * 1.05));
    std::geometric_distribution<i64> gap(dens);
    for (i64 r = 0; r < n; ++r) {
        for (i64 j = gap(rng); j < d; j += 1 + gap(rng)) indices.push_back((int)j);
        indptr[r + 1] = (i64)indices.size();
    }
    const i64 nnz = indptr[n];
    printf("pattern %lld x %lld, %lld observed (%.2f %%)\n", (long long)n, (long long)d, (long long)nnz, 100.0 * nnz / ((double)n * d));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    DevState* ds;
    CK(hipMalloc((void**)&ds, sizeof(DevState)));
    CK(hipMemset(ds, 0, sizeof(DevState)));
    for (int w = 0; w < 2; ++w) {
        Copy cp = build(indptr, indices, n, d, w, SP2_MAX_SEGS);
        printf("copy %d (%s as segments): %d blocks of %d, %lld segments per block, %lld entries with padding, lanes per segment %d, work items %zu (shipped) / %zu (candidate)\n",
               w, w == 0 ? "rows" : "columns", cp.nblk, cp.bw, (long long)cp.nseg, (long long)cp.count, cp.lps, cp.work_old.size(), cp.work_new.size());
        Dev dv;
        dv.segptr = to_dev(cp.segptr);
        dv.idx_old = to_dev(cp.idx_old); dv.idx_new = to_dev(cp.idx_new);
        dv.work_old = to_dev(cp.work_old); dv.work_new = to_dev(cp.work_new);
        std::vector<float> val(cp.idx_new.size(), 0.f);
        std::uniform_real_distribution<float> uf(-1.f, 1.f);
        for (size_t q = 0; q < (size_t)cp.count; ++q) val[q] = cp.idx_old[q] == SP_PAD ? 0.f : uf(rng);
        dv.val0 = to_dev(val); dv.val_a = to_dev(val); dv.val_b = to_dev(val);
        std::uniform_real_distribution<double> ud(0.0, 1.0);
        std::vector<double> B1((size_t)cp.gdim), B2((size_t)cp.gdim), V((size_t)cp.gdim), A1((size_t)cp.nseg), A2((size_t)cp.nseg);
        for (auto& x : B1) x = ud(rng);
        for (auto& x : B2) x = ud(rng) - 0.5;
        for (auto& x : V) x = ud(rng);
        for (auto& x : A1) x = 1e-3 * ud(rng);
        for (auto& x : A2) x = 1e-3 * (ud(rng) - 0.5);
        dv.B1 = to_dev(B1); dv.B2 = to_dev(B2); dv.V = to_dev(V); dv.A1 = to_dev(A1); dv.A2 = to_dev(A2);
        const size_t ns = (size_t)cp.nblk * cp.nseg;
        std::vector<double> zeros(ns, 0.0);
        dv.S1a = to_dev(zeros); dv.S2a = to_dev(zeros); dv.S1b = to_dev(zeros); dv.S2b = to_dev(zeros);
        auto run_old = [&](float* v, double* s1, double* s2) {
            switch (cp.lps) {
                case 8: launch_old<8>(cp, dv, v, s1, s2, st, ds); break;
                case 16: launch_old<16>(cp, dv, v, s1, s2, st, ds); break;
                case 32: launch_old<32>(cp, dv, v, s1, s2, st, ds); break;
                default: launch_old<64>(cp, dv, v, s1, s2, st, ds); break;
            }
        };
        struct Cand { const char* name; int lps, q; };
        const Cand cands[] = {{"k_sp_blk2 lanes 32, 4 quads per lane", 32, 4}, {"k_sp_blk2 lanes 32, 2 quads per lane", 32, 2},
                              {"k_sp_blk2 lanes 16, 4 quads per lane", 16, 4}, {"k_sp_blk2 lanes 64, 2 quads per lane", 64, 2},
                              {"k_sp_blk2 lanes 16, 8 quads per lane", 16, 8}};
        auto run_new = [&](const Cand& c, float* v, double* s1, double* s2) {
            if (c.lps == 32 && c.q == 4) launch_new<32, 4>(cp, dv, v, s1, s2, st, ds);
            else if (c.lps == 32 && c.q == 2) launch_new<32, 2>(cp, dv, v, s1, s2, st, ds);
            else if (c.lps == 16 && c.q == 4) launch_new<16, 4>(cp, dv, v, s1, s2, st, ds);
            else if (c.lps == 64 && c.q == 2) launch_new<64, 2>(cp, dv, v, s1, s2, st, ds);
            else launch_new<16, 8>(cp, dv, v, s1, s2, st, ds);
        };
        // correctness: one launch each from the same values
        run_old(dv.val_a, dv.S1a, dv.S2a);
        CK(hipStreamSynchronize(st));
        std::vector<float> va(val.size()), vb(val.size());
        std::vector<double> s1a(ns), s2a(ns), s1b(ns), s2b(ns);
        CK(hipMemcpy(va.data(), dv.val_a, va.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s1a.data(), dv.S1a, ns * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s2a.data(), dv.S2a, ns * 8, hipMemcpyDeviceToHost));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const double bytes = 10.0 * (double)cp.count;
        auto time_it = [&](auto&& fn) {
            fn(); fn();
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) fn();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return 1e3 * ms / reps;
        };
        const double t_old = time_it([&] { run_old(dv.val_a, dv.S1a, dv.S2a); });
        printf("  %-44s %7.1f us  %5.2f TB/s (%.3f of 8)\n", "k_sp_blk_r2 (round 2)", t_old, bytes / t_old * 1e-6, bytes / t_old * 1e-6 / 8.0);
        for (const Cand& c : cands) {
            CK(hipMemcpy(dv.val_b, dv.val0, val.size() * 4, hipMemcpyDeviceToDevice));
            CK(hipMemset(dv.S1b, 0, ns * 8)); CK(hipMemset(dv.S2b, 0, ns * 8));
            run_new(c, dv.val_b, dv.S1b, dv.S2b);
            CK(hipStreamSynchronize(st));
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) { printf("  %-44s launch failed: %s\n", c.name, hipGetErrorString(le)); continue; }
            CK(hipMemcpy(vb.data(), dv.val_b, vb.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(s1b.data(), dv.S1b, ns * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(s2b.data(), dv.S2b, ns * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t q = 0; q < (size_t)cp.count; ++q) bad += va[q] != vb[q];
            for (size_t q = (size_t)cp.count; q < vb.size(); ++q) bad += vb[q] != 0.f;     // the dump quads stay zero
            double e1m = 0.0, e2m = 0.0;
            for (size_t q = 0; q < ns; ++q) {
                e1m = std::max(e1m, std::fabs(s1a[q] - s1b[q]) / (1e-300 + std::fabs(s1a[q]) + 1.0));
                e2m = std::max(e2m, std::fabs(s2a[q] - s2b[q]) / (1e-300 + std::fabs(s2a[q]) + 1.0));
            }
            const double t = time_it([&] { run_new(c, dv.val_b, dv.S1b, dv.S2b); });
            printf("  %-44s %7.1f us  %5.2f TB/s (%.3f of 8)   values differing %zu, sums rel. diff %.1e / %.1e\n", c.name, t,
                   bytes / t * 1e-6, bytes / t * 1e-6 / 8.0, bad, e1m, e2m);
            fflush(stdout);
        }
        if (cp.lps == 32) {       // the library's kernel of round 3
            CK(hipMemcpy(dv.val_b, dv.val0, val.size() * 4, hipMemcpyDeviceToDevice));
            launch_lib<32>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(vb.data(), dv.val_b, vb.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(s1b.data(), dv.S1b, ns * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            double e1m = 0.0;
            for (size_t q = 0; q < (size_t)cp.count; ++q) bad += va[q] != vb[q];
            for (size_t q = 0; q < ns; ++q) e1m = std::max(e1m, std::fabs(s1a[q] - s1b[q]) / (std::fabs(s1a[q]) + 1.0));
            const double t = time_it([&] { launch_lib<32>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            printf("  %-44s %7.1f us  %5.2f TB/s (%.3f of 8)   values differing %zu, sums rel. diff %.1e\n", "k_sp_blk (library, round 3)", t,
                   bytes / t * 1e-6, bytes / t * 1e-6 / 8.0, bad, e1m);
        }
        if (cp.lps == 32) {       // k_sp_blk3: the shipped structure with less work per entry
            for (int v = 0; v < 4; ++v) {
                auto run3 = [&](float* vv, double* s1, double* s2) {
                    if (v == 0) launch_3<32, false, false>(cp, dv, vv, s1, s2, st, ds);
                    else if (v == 1) launch_3<32, true, false>(cp, dv, vv, s1, s2, st, ds);
                    else if (v == 2) launch_3<32, false, true>(cp, dv, vv, s1, s2, st, ds);
                    else launch_3<32, true, true>(cp, dv, vv, s1, s2, st, ds);
                };
                CK(hipMemcpy(dv.val_b, dv.val0, val.size() * 4, hipMemcpyDeviceToDevice));
                run3(dv.val_b, dv.S1b, dv.S2b);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(vb.data(), dv.val_b, vb.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(s1b.data(), dv.S1b, ns * 8, hipMemcpyDeviceToHost));
                size_t bad = 0;
                double worst = 0.0, e1m = 0.0;
                for (size_t q = 0; q < (size_t)cp.count; ++q)
                    if (va[q] != vb[q]) { ++bad; worst = std::max(worst, (double)std::fabs(va[q] - vb[q]) / (std::fabs((double)va[q]) + 1e-30)); }
                for (size_t q = 0; q < ns; ++q) e1m = std::max(e1m, std::fabs(s1a[q] - s1b[q]) / (std::fabs(s1a[q]) + 1.0));
                const double t = time_it([&] { run3(dv.val_b, dv.S1b, dv.S2b); });
                printf("  k_sp_blk3 lanes 32%s%s %*s %7.1f us  %5.2f TB/s (%.3f of 8)   values differing %zu (worst rel. %.1e), sums rel. diff %.1e\n",
                       (v & 1) ? ", {b1,b2} in one table" : "", (v & 2) ? ", fp32 corrections" : "", (int)(22 - ((v & 1) ? 22 : 0)), "", t,
                       bytes / t * 1e-6, bytes / t * 1e-6 / 8.0, bad, worst, e1m);
                fflush(stdout);
            }
        }
        if (cp.lps == 32) {       // memory-instruction flavours: 1 = plain stores, 2 = plain loads, 4 = (k_sp_blk2) the stores of an item after its arithmetic
            const double m1 = time_it([&] { launch_3<32, true, false, 1>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double m3 = time_it([&] { launch_3<32, true, false, 3>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double n1 = time_it([&] { launch_new<32, 4, true, true, 1>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double n3 = time_it([&] { launch_new<32, 4, true, true, 3>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double n4 = time_it([&] { launch_new<32, 4, true, true, 4>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double n7 = time_it([&] { launch_new<32, 4, true, true, 7>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double g1 = time_it([&] { launch_3<32, true, false, 3 + 8>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double g2 = time_it([&] { launch_3<32, true, false, 3 + 16>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double g3 = time_it([&] { launch_3<32, true, false, 3 + 24>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double h1 = time_it([&] { launch_3<32, true, false, 3 + 32>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double h2 = time_it([&] { launch_3<32, true, false, 3 + 8 + 32>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double h3 = time_it([&] { launch_3<32, true, false, 0 + 8 + 32>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            printf("  k_sp_blk3 next segment's bounds requested ahead: plain loads + stores %.1f us, + waves 8-15 late %.1f us, non-temporal + late %.1f us\n", h1, h2, h3);
            printf("  k_sp_blk3 plain loads + stores, waves 8-15 starting late by 1 / 2 / 3 sleeps: %.1f / %.1f / %.1f us\n", g1, g2, g3);
            printf("  memory flavours: k_sp_blk3 plain stores %.1f us, plain loads + stores %.1f us;  k_sp_blk2 plain stores %.1f, plain loads + stores %.1f, stores after the arithmetic %.1f, all three %.1f us\n", m1, m3, n1, n3, n4, n7);
        }
        if (cp.lps == 32) {       // the pass taken apart, both kernels: no sums / no write-back
            const double a1 = time_it([&] { launch_old<32, false, true>(cp, dv, dv.val_a, dv.S1a, dv.S2a, st, ds); });
            const double b1 = time_it([&] { launch_new<32, 4, false, true>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            const double a2 = time_it([&] { launch_old<32, true, false>(cp, dv, dv.val_a, dv.S1a, dv.S2a, st, ds); });
            const double b2 = time_it([&] { launch_new<32, 4, true, false>(cp, dv, dv.val_b, dv.S1b, dv.S2b, st, ds); });
            printf("  parts (lanes 32): corrections + write-back, no sums: shipped %.1f us, candidate %.1f us;  corrections + sums, no write-back: shipped %.1f us, candidate %.1f us\n", a1, b1, a2, b2);
        }
        void* fr[] = {dv.segptr, dv.idx_old, dv.idx_new, dv.work_old, dv.work_new, dv.val0, dv.val_a, dv.val_b, dv.B1, dv.B2, dv.V,
                      dv.A1, dv.A2, dv.S1a, dv.S2a, dv.S1b, dv.S2b};
        for (void* p : fr) CK(hipFree(p));
    }
    return 0;
}
