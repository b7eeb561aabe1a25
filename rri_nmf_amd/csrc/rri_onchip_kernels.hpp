// rri_onchip_kernels.hpp -- the plain (unweighted, Gram-form) sweep for problems whose X fits the chip's REGISTERS.
//
// A 10000 x 1000 fp32 X is 40 MB: 156 KB per CU of an MI355X, against 512 KB of vector registers and 160 KB of LDS per
// CU.  At that size the launch-per-phase schedule (k_trow_small, k_pass, k_wcol: three dependent launches per topic
// step, 22-24 us) spends its time on launch latencies and on re-reading a cache-resident X, not on work.  Here ONE
// persistent kernel -- one 512-thread workgroup per CU (two waves per SIMD, 256 registers each), all co-resident --
// loads its rows of X into registers once and runs whole sweeps; what crosses workgroups goes through small arrays on
// the memory side and two flag hand-overs per topic step:
//
//   every workgroup b   rows [b rows_wg, (b+1) rows_wg) of X and W   (X: registers, W: LDS; the global k-major W kept current)
//   worker w < NA       also columns [w CWA, (w+1) CWA) of T         (all k rows of that slice in LDS); NA = ceil(LD / CWA)
//
//   phase B (t)  all    T T[t]^T = sum_w mkP[w], row checks (nmf.py:751-769), T[t,:] into registers, y = X t for the own
//                       rows, W-column update (nmf.py:464-469, 728-734); then for tn = t+1:
//                       mkZ[b] = w_tn^T X, mkG[b] = (w_tn^T W, ||w_tn||^2, column sum of the update) over the own rows
//                       -> flagB[b]; the workers wait for all G of them, the others go on to wait for the workers
//   phase A (tn) workers  g = sum_b mkG[b], column check (nmf.py:471-476), z_j = sum_b mkZ[b][j], numer_T, qf_min for the
//                       own columns (nmf.py:437-447, 670-676) -> T[tn, slice], mkP[w] = partial T T[tn]^T and row sum
//                       -> flagA[w] (top bit: the step halts); everybody waits for the NA of them
//
// Only NA workgroups read the G x (k+2) Gram partials and everybody reads NA x (k+1) Gram partials of T: with every
// workgroup reducing everything itself the partial sums alone moved 23 MB per topic step through the memory side, more
// than half of the X that no longer moves.
//
// Wave layout of a workgroup: CG column groups of 256 columns x RG = 8 / CG row groups; lane l of a wave holds the 4
// adjacent columns cg*256 + 4l .. +3 (one 16-byte load per row) of the rows rg, rg + RG, ... of the workgroup's block:
// RPW float4 registers.  Row dots are lane-local over 4 columns, 8 rows at a time through a wave-private LDS tile, and CG
// partials per row meet in LDS; column sums are lane-local over the rows, RG partials per column meet in LDS.  All sums
// float64, fixed order.
//
// Same arithmetic as the launch-per-phase kernels (same branches of qf_min, same checks, same halt protocol:
// DevState.halt with the position of the detecting step), another order of the row / column partial sums.
// Both halves free, 2 <= k <= 22.  With the topic-model flags (T rows projected onto the simplex at every step) the workers
// exchange their slices of the closed-form row and each finishes the projection of the whole row: one more hand-over among
// the 32 workers per step.
//
// Hand-overs (tools/barrier_probe.hip, profiles/r02_grid_barrier_variants.log): a workgroup stores the number of the
// step in its flag; one wave of each waiting workgroup polls the flags it depends on.  Everything that crosses workgroups
// -- the flags, mkZ / mkG / mkP and the T row of the step -- is written and read with agent-scope accesses (they go to the
// memory side, past the per-XCD L2s), so no L2 write-back / invalidate is needed: 4.1 us per grid-wide barrier against
// 7.9 us for a counter with release / acquire fences and 22 us when every wave issues the acquire.  The number of polls
// is BOUNDED: a grid that cannot make progress (workgroups not co-resident) raises an abort word that every poll also
// reads, so every wave reaches the end of the kernel and the host sees HALT_ERR_GRID_SYNC instead of a hang.
#pragma once
#include "rri_kernels.hpp"

namespace rri {

enum { HALT_ERR_GRID_SYNC = -8 };
constexpr int ONCHIP_THREADS = 512, ONCHIP_WAVES = ONCHIP_THREADS / 64;
constexpr int ONCHIP_MAX_K = 22;      // k + 2 Gram entries = 8 waves x 3 in flight: one round trip in phase A.  Beyond that the
                                      // per-topic cost of the kernel grows faster than that of the launch-per-phase schedule
                                      // (5000 x 1000: -3 % at k = 24 and 32, -15 % at k = 64; profiles/r02_onchip_sizes.log)
constexpr int ONCHIP_CWA = 32;          // columns of T per worker
constexpr int ONCHIP_PG = ONCHIP_THREADS / ONCHIP_CWA;   // groups of workgroup partials in the column-sum reduction

struct OnchipArgs {
    const void* X; i64 ldx; int n, d, LD, k;      // X in the handle's storage type (the kernel is instantiated per type)
    double* Wt; i64 ldw; double* T; i64 ldt;
    double* mkZ;                   // [2][G][LD]   column-sum partials of the carried topic (two buffers: by the parity of the step that reads)
    double* mkG;                   // [2][k+2][G]  Gram-row partials | ||w||^2 | column sum of the last update, entry-major
    double* mkP;                   // [k+1][64]  T T[t]^T partials | row sum of the new T row, entry-major (NA <= 64 workers)
    double* mkX;                   // [2][LD]  the T row before its projection (topic-model flags): slices from the workers, by the
                                   // parity of the step; an element that has not arrived holds ONCHIP_ABSENT
    double* xyp; int xy_stride;    // <w_t, X t_t> partials for the objective (XYpart[t * xy_stride + b])
    unsigned* bar;                 // [0] abort word, [64 + w] flagA of worker w, [64 + 64 + b] flagB of workgroup b (zero at launch)
    int G, NA, rows_wg, CG, RG, kS;
    int s0, t0, ph0, s_end;        // cursor (sweep, topic, phase) and end sweep (exclusive)
    int skip_row_finish;           // a resumed W half whose T-row checks already ran (after a T-row reset)
    unsigned spin_limit;
    unsigned entry_spin_limit;     // polls of the all-grid hand-over at kernel entry (short: a grid that is not resident as a whole shows here)
    long long* dbg;                // diagnostics build only: [2][16] accumulated 100 MHz ticks per section (workgroup 0, workgroup G-1)
    KParams p; DevState* st;
};

#define RRI_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, RRI_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, RRI_AGENT));
}

// An exchange slot that has not been written in this step: a NaN payload no arithmetic produces.  The readers poll the DATA
// (one trip to the memory side) instead of a flag and then the data (two).
constexpr unsigned long long ONCHIP_ABSENT = 0xfff7a5a5fff7a5a5ULL;
__device__ __forceinline__ bool onchip_absent(double v) { return (unsigned long long)__double_as_longlong(v) == ONCHIP_ABSENT; }
__device__ __forceinline__ double onchip_absent_value() { return __longlong_as_double((long long)ONCHIP_ABSENT); }

// this workgroup's data of the step is on the memory side: publish `value` in `mine`
__device__ __forceinline__ void onchip_signal(unsigned* mine, unsigned value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's agent-scope stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(mine, value, __ATOMIC_RELAXED, RRI_AGENT);
}
// wait until the `count` flags at `flags` have reached `epoch` (low 31 bits).  Returns 0 = go on, 1 = a flag carries the
// halt bit, 2 = the grid gave up.  Called by every thread; wave 0 polls.
__device__ __forceinline__ int onchip_wait(unsigned* bar, const unsigned* flags, int count, unsigned epoch, unsigned spin_limit) {
    __shared__ int verdict_sh;
    if (threadIdx.x < 64) {
        int verdict = 0;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            unsigned top = 0u;
            for (int q = threadIdx.x; q < count; q += 64) {
                const unsigned f = __hip_atomic_load(flags + q, __ATOMIC_RELAXED, RRI_AGENT);
                ok = ok && ((f & 0x7fffffffu) >= epoch);
                top |= f & 0x80000000u;
            }
            if (__all(ok)) { verdict = __any(top != 0u) ? 1 : 0; break; }
            if (__hip_atomic_load(bar, __ATOMIC_RELAXED, RRI_AGENT) != 0u) { verdict = 2; break; }
            if (++spins > spin_limit) {
                if (threadIdx.x == 0) __hip_atomic_store(bar, 1u, __ATOMIC_RELAXED, RRI_AGENT);
                verdict = 2;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (threadIdx.x == 0) verdict_sh = verdict;
    }
    __syncthreads();
    const int v = verdict_sh;
    __syncthreads();                     // verdict_sh is rewritten by the next wait
    return v;
}

// sums over `np` workgroup partials part[q * stride + e] for the entries e = wave, wave + 8, ... < ne, into out[e]: the
// loads of up to 3 entries (4 partials per lane each) are in flight together -- one round trip for k <= 22 -- and every
// workgroup adds in the same order
// (entry-major arrays: the np partials of an entry are contiguous, `stride` doubles per entry -- with workgroup-major rows every
// 8-byte load of a partial touched a sector of its own, eight times the bytes)
__device__ __forceinline__ void onchip_entry_sums(const double* __restrict__ part, int stride, int ne, int np, double* out,
                                                  int wave, int lane) {
    constexpr int EB = 3;
    for (int e0 = wave; e0 < ne; e0 += ONCHIP_WAVES * EB) {
        double acc[EB];
#pragma unroll
        for (int m = 0; m < EB; ++m) acc[m] = 0.0;
#pragma unroll 1
        for (int q0 = lane; q0 < np; q0 += 256) {
            double v[EB][4];
#pragma unroll
            for (int m = 0; m < EB; ++m) {
                const int e = e0 + ONCHIP_WAVES * m;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + 64 * u;
                    v[m][u] = (e < ne && q < np) ? ld_agent(part + (unsigned)(e * stride + q)) : 0.0;
                }
            }
#pragma unroll
            for (int m = 0; m < EB; ++m)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m] += v[m][u];
        }
#pragma unroll
        for (int m = 0; m < EB; ++m) {
            const int e = e0 + ONCHIP_WAVES * m;
            const double tot = wave_sum<double>(acc[m]);
            if (lane == 0 && e < ne) out[e] = tot;
        }
    }
}

// Michelot's fixed point for the simplex projection of a row held two elements per thread (v0 = row[tid], v1 = row[tid + 512];
// -inf where there is none): the same iteration as simplex_theta (rri_kernels.hpp), with the sum and the count of the active
// set reduced together -- two barriers per iteration instead of eight; every thread returns the same theta
__device__ __forceinline__ double onchip_simplex_theta(double v0, double v1, double s, double* sh, int* iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double theta = -1.0e300;
    i64 cnt_prev = -1;
    int it = 0;
    for (; it < 2 * ONCHIP_THREADS + 2; ++it) {
        double sum = 0.0, cnt = 0.0;
        if (v0 > theta) { sum += v0; cnt += 1.0; }
        if (v1 > theta) { sum += v1; cnt += 1.0; }
        sum = wave_sum<double>(sum);
        cnt = wave_sum<double>(cnt);
        __syncthreads();                       // sh may still be read from the iteration before
        if (lane == 0) { sh[wave] = sum; sh[8 + wave] = cnt; }
        __syncthreads();
        double S = 0.0, C = 0.0;
#pragma unroll
        for (int q = 0; q < ONCHIP_WAVES; ++q) { S += sh[q]; C += sh[8 + q]; }
        const i64 ci = (i64)C;
        if (ci == cnt_prev || ci == 0) break;
        theta = (S - s) / C;
        cnt_prev = ci;
    }
    *iters = it;
    return theta;
}

// The same fixed point by ONE wave, the row left in LDS (LD <= 1024: 16 elements per lane, re-read in every pass:
// 8 KB at 128 B per clock, no registers held) -- no workgroup barrier inside the iteration.  The 8-wave form above costs
// two barriers per iteration plus two block sums around it, ~30 barriers per projected row, on the critical path of every
// topic step of the topic-model flags; this one needs two (before and after).  The elements are w_j = row[j] when !shifted,
// max(row[j] - shift, 0) when shifted (the second projection of nmf.py:759-761 runs on the first one's result without
// storing it in between).
//
// Every pass over the row costs ~0.35 us on this one wave (16 LDS reads, two dependent wave sums, a float64 division), so the
// passes are what is counted here:
//   * the iteration climbs to theta* from any lower bound and ends at the same active set, hence the same theta (the sum
//     over that set in this lane order).  Two lower bounds are known before it starts: Michelot's own first iterate
//     (sum of all - s) / d, and max(w) - s (the largest element alone already reaches s there).  From the larger of the two
//     an ill-scaled row -- closed-form rows of the first sweeps sum to thousands where the simplex wants 1, three elements
//     stay active -- needs 1-2 iterations instead of 9;
//   * sum_j max(w_j - theta, 0) is accumulated lane-locally in every pass: the pass that finds the active set unchanged has
//     computed it for the final theta (the same terms in the same order as a pass of its own);
//   * the second projection starts from the first one's outputs: the sum of its elements IS the first one's sum_proj and
//     their maximum is max(vmax - shift, 0) -- no opening pass (start->have).
struct OnchipThetaStart { bool have; double all, vmax; };
__device__ __forceinline__ double onchip_wave_theta(const double* row, int d, double s, bool shifted, double shift,
                                                    OnchipThetaStart start, double* sum_all, double* vmax_out, double* sum_proj, int* iters) {
    const int lane = threadIdx.x & 63;
    auto elem = [&](int q) -> double {
        const int j = lane + 64 * q;
        if (j >= d) return -1.0e300;
        const double v = row[j];
        return shifted ? fmax(v - shift, 0.0) : v;
    };
    double all = start.all, vmax = start.vmax;
    if (!start.have) {
        all = 0.0; vmax = -1.0e300;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double v = elem(q);
            if (v > -1.0e299) { all += v; vmax = fmax(vmax, v); }
        }
        all = wave_sum<double>(all);
        vmax = wave_max(vmax);
    }
    double theta = fmax((all - s) / (double)d, vmax - s);
    i64 cnt_prev = -1;
    int it = 1;
    double sp = 0.0;
    for (; it < 2 * 1024 + 2; ++it) {
        double sum = 0.0;
        int cnt_i = 0;
        sp = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double v = elem(q);
            if (v > theta) { sum += v; cnt_i += 1; sp += v - theta; }
        }
        sum = wave_sum<double>(sum);
        cnt_i = wave_sum_i32(cnt_i);
        const i64 ci = (i64)cnt_i;
        if (ci == cnt_prev || ci == 0) break;
        theta = (sum - s) / (double)cnt_i;
        cnt_prev = ci;
    }
    if (it >= 2 * 1024 + 2) {                  // not reached (the active set shrinks in every pass): the sum for the last theta
        sp = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double v = elem(q);
            if (v > theta) sp += v - theta;
        }
    }
    *sum_all = all;
    *vmax_out = vmax;
    *sum_proj = wave_sum<double>(sp);
    *iters = it;
    return theta;
}

// DBG: sections of a topic step timed by thread 0 of workgroup 0 (a worker) and of the last workgroup (RRI_ONCHIP_TIMING,
// tools/onchip_probe.py): 0 phase A loads, 1 phase A rest + signal, 2 wait for the workers, 3 phase B loads, 4 row dots,
// 5 W update, 6 carry, 7 hand-over to the workers
// PROJ: the instantiation for the topic-model flags (the projection stage costs the plain one registers it does not have)
template <typename SX, int RPW, bool DBG = false, bool PROJ = false>
__global__ __launch_bounds__(ONCHIP_THREADS) void k_onchip_sweeps(OnchipArgs a) {
    constexpr int NTH = ONCHIP_THREADS, NWV = ONCHIP_WAVES, CWA = ONCHIP_CWA, PG = ONCHIP_PG;
    DevState* st = a.st;
    if (st->halt) return;
    long long dacc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long dlast = DBG ? wall_clock64() : 0;
#define RRI_STAMP(i)                                                           \
    do {                                                                       \
        if (DBG && threadIdx.x == 0) {                                         \
            const long long now_ = wall_clock64();                             \
            dacc[i] += now_ - dlast;                                           \
            dlast = now_;                                                      \
        }                                                                      \
    } while (0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, G = a.G, NA = a.NA, k = a.k, kS = a.kS, CG = a.CG, RG = a.RG;
    const int cg = wave % CG, rg = wave / CG;
    const int col0 = cg * 256 + lane * 4;
    const int row0 = b * a.rows_wg;
    const int rows_here = max(0, min(a.rows_wg, a.n - row0));
    const bool worker = b < NA;
    const int j0 = b * CWA;                                 // first column of a worker's T slice
    const KParams p = a.p;
    unsigned* flagA = a.bar + 64;
    unsigned* flagB = a.bar + 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* Wl = reinterpret_cast<double*>(smem);           // [rows_wg][kS]   own rows of W
    double* Tl = Wl + (size_t)a.rows_wg * kS;               // [k][CWA]        a worker's column slice of T
    double* gsh = Tl + (size_t)k * CWA;                     // [k + 2]
    double* tts = gsh + k + 2;                              // [k + 1]
    double* zred = tts + k + 1;                             // [PG][CWA]       partial column sums of a worker
    double* ysh = zred + PG * CWA;                          // [CG][rows_wg]
    double* xyl = ysh + (size_t)CG * a.rows_wg;             // [rows_wg]
    double* zsh = xyl + a.rows_wg;                          // [RG][CG * 256] = 8 x 256
    double* tiles = zsh + NWV * 256;                        // [8 waves][8 x 72]   row-sum tiles (wave_rowsum8)
    const int LDp = CG * 256;
    double* tile = tiles + wave * (8 * 72);
    double* rowsh = tiles + NWV * 8 * 72;                   // [LD]  the whole T row of the step (workers, topic-model flags)
    double* scratch = rowsh + 1024;                         // [40]  block sums
    constexpr bool project = PROJ;                          // T rows on the simplex at every step (nmf.py:447, 751-761):
                                                            // the host picks the instantiation from project_T && has_t_row_sum

    // ---- residents: X rows -> registers, W rows and (workers) the T slice -> LDS ------------------------------------
    SX xr[RPW][4];       // float: 4 registers per row, RPW <= 20; double: 8 per row, RPW <= 10
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int lr = rg + RG * r;
#pragma unroll
        for (int c = 0; c < 4; ++c) xr[r][c] = SX(0);
        if (lr < rows_here && col0 < a.LD) {
            const SX* src = static_cast<const SX*>(a.X) + (i64)(row0 + lr) * a.ldx + col0;
            if constexpr (sizeof(SX) == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                for (int c = 0; c < 4; ++c) xr[r][c] = v[c];
            } else {                                      // LD is a multiple of 2 only: the second pair may lie past the row
                const f64x2 v0 = *reinterpret_cast<const f64x2*>(src);
                xr[r][0] = v0[0]; xr[r][1] = v0[1];
                if (col0 + 2 < a.LD) {
                    const f64x2 v1 = *reinterpret_cast<const f64x2*>(src + 2);
                    xr[r][2] = v1[0]; xr[r][3] = v1[1];
                }
            }
        }
    }
    // fp32 X stays fp32 in the registers and is widened where it is used: the compiler must not hoist the conversions out
    // of the topic loop (it would keep a float64 copy of every element: three times the registers)
    auto keep_fp32 = [&]() {
#pragma unroll
        for (int r = 0; r < RPW; ++r) asm volatile("" : "+v"(xr[r][0]), "+v"(xr[r][1]), "+v"(xr[r][2]), "+v"(xr[r][3]));
    };
    for (int e = tid; e < rows_here * k; e += NTH) {
        const int l = e / rows_here, i = e - l * rows_here;  // k-major global copy: consecutive threads, consecutive rows
        Wl[(size_t)i * kS + l] = a.Wt[(i64)l * a.ldw + row0 + i];
    }
    if (worker)
        for (int e = tid; e < k * CWA; e += NTH) {
            const int l = e / CWA, jl = e - l * CWA;
            Tl[e] = (j0 + jl < a.d) ? a.T[(i64)l * a.ldt + j0 + jl] : 0.0;
        }
    __syncthreads();

    unsigned epoch = 0;       // number of the last hand-over; advances identically in every workgroup

    // The carry of topic tn -- column sums w_tn^T X and Gram-row partials of w_tn over the own rows -- in two parts.
    // carry_pre: everything that does not depend on the W column `texcl` of the running step (column tn itself is not
    // touched by that step): the column sums, ||w_tn||^2 and the Gram entries against every other column.  It runs while
    // the workgroup waits for the workers' flags, off the chain of dependent hand-overs (19.1 -> 18.7 us per topic step at
    // 10000 x 1000; running it two steps ahead, in the workers' own wait, measured the same).  carry_post: the two
    // entries that need the updated column t.  The arrays are double-buffered by the parity of the step that READS them,
    // so the workers may still be reading the running step's buffer while the next one is written.
    auto carry_pre = [&](int tn, int texcl, int buf) {
        double* mkZb = a.mkZ + (size_t)buf * G * a.LD;
        double* mkGb = a.mkG + (size_t)buf * (k + 2) * G;
        keep_fp32();
        double za[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int lr = rg + RG * r;
            if (lr < rows_here) {
                const double wn = Wl[(size_t)lr * kS + tn];
#pragma unroll
                for (int c = 0; c < 4; ++c) za[c] = fma(wn, (double)xr[r][c], za[c]);
            }
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four rows' conversions live at a time, not all RPW
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) zsh[(size_t)rg * LDp + col0 + c] = za[c];
        __syncthreads();
#pragma unroll 1
        for (int j = tid; j < a.LD; j += NTH) {
            double zq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) zq[q] = q < RG ? zsh[(size_t)q * LDp + j] : 0.0;
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < RG) s += zq[q];
            st_agent(mkZb + (unsigned)(b * a.LD + j), s);
        }
#pragma unroll 1
        for (int e = wave; e < k + 1; e += NWV) {
            if (e == texcl) continue;                              // wave-uniform
            double acc = 0.0;
#pragma unroll 1
            for (int i = lane; i < rows_here; i += 64) {
                const double wn = Wl[(size_t)i * kS + tn];
                acc = fma(wn, e < k ? Wl[(size_t)i * kS + e] : wn, acc);
            }
            acc = wave_sum<double>(acc);
            if (lane == 0) st_agent(mkGb + (unsigned)(e * G + b), acc);
        }
        __syncthreads();                                           // zsh is free again
    };
    auto carry_post = [&](int tn, int t, int buf) {
        double* mkGb = a.mkG + (size_t)buf * (k + 2) * G;
        if (wave < 2) {                                            // wave 0: <w_tn, w_t new>, wave 1: sum of the new column t
            double acc = 0.0;
#pragma unroll 1
            for (int i = lane; i < rows_here; i += 64) {
                const double wt = Wl[(size_t)i * kS + t];
                acc = wave == 0 ? fma(Wl[(size_t)i * kS + tn], wt, acc) : acc + wt;
            }
            acc = wave_sum<double>(acc);
            if (lane == 0) st_agent(mkGb + (unsigned)((wave == 0 ? t : k + 1) * G + b), acc);
        }
    };
    // phase B of every workgroup is done -> flagB; the workers wait for all of them
    auto hand_to_workers = [&]() -> int {
        epoch += 1u;
        onchip_signal(flagB + b, epoch);
        return worker ? onchip_wait(a.bar, flagB, G, epoch, a.spin_limit) : 0;
    };

    bool have_carry = false;
    int chk = 0, tprev = -1;
    unsigned stepq = 0;       // topic steps of this launch so far

    // Entry hand-over: every workgroup reports in and waits for all others BEFORE anything is written to W, T or the partial
    // arrays.  A grid that is not resident as a whole -- two processes' persistent grids dispatched at the same moment each
    // hold a part of the CUs (tools/onchip_two_processes.py: 5-13 times in 400 calls with four processes on one GPU) -- fails
    // HERE, after entry_spin_limit polls (~tens of milliseconds) instead of the seconds of the in-run bound, with nothing to
    // undo; the host reruns the range launch by launch and backs the process off the persistent path for a while.
    if (project && worker && tid < 2 * CWA) {              // both buffers of the row exchange start "absent" (own slice)
        const int which = tid / CWA, jl = tid % CWA;
        if (j0 + jl < a.LD) st_agent(a.mkX + (size_t)which * a.LD + (unsigned)(j0 + jl), onchip_absent_value());
    }
    epoch += 1u;
    onchip_signal(flagB + b, epoch);
    if (onchip_wait(a.bar, flagB, G, epoch, a.entry_spin_limit) == 2) goto sync_failed;
    for (int s = a.s0; s < a.s_end; ++s) {
        for (int t = (s == a.s0) ? a.t0 : 0; t < k; ++t) {
            const int ph = (s == a.s0 && t == a.t0) ? a.ph0 : 0;
            const bool update_T = ph == 0;
            int mode = 0;
            const int buf = (int)(stepq & 1u);             // buffers phase A of this step reads; the next step's are buf ^ 1
            const bool last_step = (s == a.s_end - 1) && (t == k - 1);
            if (update_T && !have_carry) {
                carry_pre(t, -1, buf);
                if (hand_to_workers() == 2) goto sync_failed;
            }
            // ---------------- phase A (workers): T row t on the own column slice -----------------------------------
            RRI_STAMP(7);
            unsigned halt_bit = 0u;
            if (worker) {
                if (update_T) {
                    // the partials of the own columns: thread = (column jl, group pg of workgroups pg, pg + PG, ...)
                    const int jl = tid % CWA, pg = tid / CWA;
                    const double* mkZr = a.mkZ + (size_t)buf * G * a.LD;
                    double zp[16];
                    const bool col_ok = j0 + jl < a.LD;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int q = pg + PG * u;
                        zp[u] = (col_ok && q < G) ? ld_agent(mkZr + (unsigned)(q * a.LD + j0 + jl)) : 0.0;
                    }
                    onchip_entry_sums(a.mkG + (size_t)buf * (k + 2) * G, G, k + 2, G, gsh, wave, lane);
                    double zacc = 0.0;
#pragma unroll
                    for (int u = 0; u < 16; ++u) zacc += zp[u];
#pragma unroll 1
                    for (int q = pg + PG * 16; q < G; q += PG) zacc += col_ok ? ld_agent(mkZr + (unsigned)(q * a.LD + j0 + jl)) : 0.0;
                    zred[pg * CWA + jl] = zacc;
                    __syncthreads();
                    RRI_STAMP(0);
                    const double nw = gsh[k];
                    int code = 0;
                    if (chk) {
                        const double sw = gsh[k + 1];
                        const bool ev = (sw <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
                        const bool err = !ev && !(sw > 0.0);
                        if (ev || err) {
                            code = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
                            if (b == 0 && tid == 0) { st->halt = code; st->halt_topic = tprev; st->halt_sweep = s; st->halt_pos = t; }
                        }
                    }
                    const double c = nw + p.reg_t_l2;          // denom = nw + reg_t_l2 (nmf.py:438)
                    if (code == 0 && !(c > 0.0)) {             // scalar c <= 0 (optimization.py:60-73)
                        if (project) mode = (p.t_row_sum == 1.0) ? 2 : HALT_ERR_NOT_IMPLEMENTED;     // one-hot at the arg-max
                        else if (p.has_trs && p.t_row_sum != 0.0) mode = 1;                          // entries at a bound
                        else mode = HALT_ERR_UNBOUNDED;
                        if (mode < 0) {
                            code = mode;
                            if (b == 0 && tid == 0) { st->halt = code; st->halt_topic = t; st->halt_sweep = s; st->halt_pos = t; }
                        }
                    }
                    if (code != 0) halt_bit = 0x80000000u;     // every worker takes the same verdict from the same sums
                    else if (tid < 8 * CWA) {
                        // the closed form for the own columns: 8 lanes per column share the PG partial column sums and the
                        // k-term product with the Gram row (all their LDS reads in flight at once), three DPP steps add the parts
                        const int jc = tid >> 3, sub = tid & 7;
                        constexpr int TERMS = (ONCHIP_MAX_K + 7) / 8;
                        double zpart = 0.0, acc = 0.0;
                        double gv[TERMS], tvv[TERMS], zq[PG / 8];
#pragma unroll
                        for (int q = 0; q < PG / 8; ++q) zq[q] = zred[(sub + 8 * q) * CWA + jc];
#pragma unroll
                        for (int q = 0; q < TERMS; ++q) {
                            const int l = sub + 8 * q;
                            gv[q] = (l < k && l != t) ? gsh[l] : 0.0;
                            tvv[q] = (l < k) ? Tl[l * CWA + jc] : 0.0;
                        }
#pragma unroll
                        for (int q = 0; q < PG / 8; ++q) zpart += zq[q];
#pragma unroll
                        for (int q = 0; q < TERMS; ++q) acc = fma(gv[q], tvv[q], acc);
                        zpart = group8_sum(zpart);
                        acc = group8_sum(acc);
                        if (sub == 0 && j0 + jc < a.d) {
                            const double numer = (zpart - acc) - p.reg_t_l1;
                            double x;
                            if (mode == 0) x = fmax(numer, 0.0) / (c + p.eps);
                            else if (mode == 1) x = (-numer + c < 0.0) ? p.t_row_sum : 0.0;
                            else x = numer;                        // mode 2: the arg-max of the numerator takes it all
                            if (project) st_agent(a.mkX + (size_t)buf * a.LD + (unsigned)(j0 + jc), onchip_absent(x) ? __longlong_as_double(0x7ff8000000000000LL) : x);
                            else {
                                Tl[t * CWA + jc] = x;
                                st_agent(a.T + (i64)t * a.ldt + j0 + jc, x);
                            }
                        }
                    }
                    __syncthreads();
                    if (b == 0 && tid == 0 && code == 0) st->tmode = mode;
                    if (project && code == 0) {
                        // the closed-form row is complete only across the workers: every worker takes all slices (a hand-over
                        // among the NA workers) and finishes qf_min for the whole row itself -- Michelot's fixed point for the
                        // simplex projection, or the one-hot row -- and the checks of _project_and_check_reset_t
                        // (nmf.py:751-769), exactly as k_trow_final does; all workers come to the same row and verdict
                        RRI_STAMP(8);                          // closed form of the own columns
                        // No flag for this exchange: the slices were stored into the step's buffer of mkX, whose slots held
                        // ONCHIP_ABSENT, and every thread polls the two elements it stages until both are there -- store, then
                        // ONE trip to the memory side, where a flag costs acknowledge + barrier + flag store + flag poll + the
                        // load of the data.  The polls are bounded like those of the flags.
                        {
                            const double* xin = a.mkX + (size_t)buf * a.LD;
                            const bool need0 = tid < a.d, need1 = tid + NTH < a.d;
                            double r0 = need0 ? onchip_absent_value() : 0.0, r1 = need1 ? onchip_absent_value() : 0.0;
                            unsigned spins = 0;
                            int failed = 0;
                            for (;;) {
                                if (need0 && onchip_absent(r0)) r0 = ld_agent(xin + (unsigned)tid);
                                if (need1 && onchip_absent(r1)) r1 = ld_agent(xin + (unsigned)(tid + NTH));
                                if (__all(!onchip_absent(r0) && !onchip_absent(r1))) break;
                                if ((++spins & 63u) == 0u && __hip_atomic_load(a.bar, __ATOMIC_RELAXED, RRI_AGENT) != 0u) { failed = 1; break; }
                                if (spins > a.spin_limit) {
                                    if (lane == 0) __hip_atomic_store(a.bar, 1u, __ATOMIC_RELAXED, RRI_AGENT);
                                    failed = 1;
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(1);
                            }
                            if (tid < a.LD) rowsh[tid] = r0;
                            if (tid + NTH < a.LD) rowsh[tid + NTH] = r1;
                            if (__syncthreads_or(failed)) goto sync_failed;
                            // the other buffer is the next step's: its slots of the own slice go back to "absent".  Every worker
                            // has stored this step's slice, so all of them are past their reads of the step before; the stores
                            // are acknowledged before this workgroup's flagA below, which every reader of the next step waits for
                            if (tid < CWA && j0 + tid < a.LD) st_agent(a.mkX + (size_t)(buf ^ 1) * a.LD + (unsigned)(j0 + tid), onchip_absent_value());
                        }
                        RRI_STAMP(10);                         // the whole row in LDS
                        double nx = 1.0, sumT = 0.0;
                        int iters = 0;
                        if (mode == 2) {
                            double mx = -1.0e300;
                            i64 idx = (i64)0x7fffffffffffffffLL;
                            for (int j = tid; j < a.d; j += NTH)
                                if (rowsh[j] > mx) { mx = rowsh[j]; idx = j; }     // ascending j per thread: first index on ties
                            wave_argmax(mx, idx);
                            __syncthreads();
                            if (lane == 0) { scratch[wave] = mx; scratch[8 + wave] = (double)idx; }
                            __syncthreads();
                            double bm = scratch[0];
                            i64 bi = (i64)scratch[8];
                            for (int q = 1; q < NWV; ++q) {
                                const double om = scratch[q];
                                const i64 oi = (i64)scratch[8 + q];
                                if (om > bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
                            }
                            __syncthreads();
                            for (int j = tid; j < a.d; j += NTH) rowsh[j] = (j == bi) ? 1.0 : 0.0;
                            sumT = 1.0;                       // the unit vector: nothing to re-project, no reset
                        } else {
                            // wave 0 alone runs both fixed points on the row in LDS (onchip_wave_theta) and leaves
                            // {theta1, theta2, second projection taken, sum of the row, sum after the first projection,
                            // iterations} in scratch; everybody applies them.  Two barriers instead of ~30 per row.
                            if (wave == 0) {
                                double all = 0.0, vmax = 0.0, sp = 0.0, sp2 = 0.0, all2 = 0.0, vmax2 = 0.0;
                                int it1 = 0, it2 = 0;
                                const double th1 = onchip_wave_theta(rowsh, a.d, p.t_row_sum, false, 0.0, OnchipThetaStart{false, 0.0, 0.0},
                                                                     &all, &vmax, &sp, &it1);
                                double th2 = 0.0, again = 0.0;
                                if ((sp > 1e-10 || p.reset_method == RESET_NONE) && p.t_row_sum != 0.0 && fabs(sp - p.t_row_sum) > 1e-15) {
                                    again = 1.0;                              // nmf.py:759-761: project again
                                    th2 = onchip_wave_theta(rowsh, a.d, p.t_row_sum, true, th1, OnchipThetaStart{true, sp, fmax(vmax - th1, 0.0)},
                                                            &all2, &vmax2, &sp2, &it2);
                                }
                                if (lane == 0) {
                                    scratch[0] = th1; scratch[1] = th2; scratch[2] = again; scratch[3] = all; scratch[4] = sp;
                                    scratch[5] = (double)(it1 + it2);
                                }
                            }
                            __syncthreads();
                            const double th1 = scratch[0], th2 = scratch[1];
                            const bool again = scratch[2] != 0.0;
                            nx = scratch[3];
                            sumT = scratch[4];
                            iters = (int)scratch[5];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int j = tid + NTH * q;
                                if (j < a.d) {
                                    double w = fmax(rowsh[j] - th1, 0.0);
                                    if (again) w = fmax(w - th2, 0.0);
                                    rowsh[j] = w;
                                }
                            }
                            if (b == 0 && tid == 0) st->theta = th1;
                            if (!(sumT > 1e-10 || p.reset_method == RESET_NONE) && p.resets_left > 0) {
                                code = HALT_EVENT_RESET_T;
                                halt_bit = 0x80000000u;
                            }
                        }
                        __syncthreads();
                        RRI_STAMP(12);                         // projected
                        if (b == 0 && tid == 0) {
                            st->nt1 = nx; st->sumT = sumT; st->proj_iters = iters;
                            if (code != 0) { st->halt = code; st->halt_topic = t; st->halt_sweep = s; st->halt_pos = t; }
                        }
                        if (code == 0 && tid < CWA && j0 + tid < a.d) {
                            const double x = rowsh[j0 + tid];
                            Tl[t * CWA + tid] = x;
                            st_agent(a.T + (i64)t * a.ldt + j0 + tid, x);
                        }
                        __syncthreads();
                    }
                }
                if (halt_bit == 0u) {
                    // T T[t]^T over the own slice; [k] = sum of the row.  Thread = (entry e, quarter of the 32 columns): 8 LDS
                    // reads in flight each, the four quarters of an entry sit in adjacent lanes and are added in order
                    for (int e0 = 0; e0 < k + 1; e0 += NTH / 4) {
                        const int e = e0 + (tid >> 2), part = tid & 3;
                        double xv[8], lv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int jl = part * 8 + q;
                            xv[q] = Tl[t * CWA + jl];
                            lv[q] = (e < k) ? Tl[e * CWA + jl] : 1.0;
                        }
                        double acc = 0.0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc = fma(lv[q], xv[q], acc);
                        const double a1 = dpp<0x55, 0xf>(acc), a2 = dpp<0xAA, 0xf>(acc), a3 = dpp<0xFF, 0xf>(acc);   // lanes 1, 2, 3 of the quad
                        if (part == 0 && e < k + 1) st_agent(a.mkP + (unsigned)(e * 64 + b), ((acc + a1) + a2) + a3);
                    }
                }
                epoch += 1u;
                onchip_signal(flagA + b, epoch | halt_bit);
                RRI_STAMP(1);
            } else {
                epoch += 1u;
            }
            if (halt_bit == 0u && !last_step) carry_pre((t + 1) % k, t, buf ^ 1);   // while the flags travel
            {
                const int v = onchip_wait(a.bar, flagA, NA, epoch, a.spin_limit);
                if (v == 2) goto sync_failed;
                if (v == 1) return;                        // the workers found an event or an error: DevState says which
            }
            RRI_STAMP(2);
            // mode of the T row for the diagnostics below: only workgroup 0 (a worker) reports it

            // ---------------- phase B: row checks, W column t on the own rows, carry of topic t + 1 ---------------
            {
                double tv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (col0 + c < a.LD) tv[c] = ld_agent(a.T + (unsigned)(t * (int)a.ldt + col0 + c));
                onchip_entry_sums(a.mkP, 64, k + 1, NA, tts, wave, lane);
                __syncthreads();
                RRI_STAMP(3);
                const bool row_checks = !project && (update_T || !a.skip_row_finish);     // with a projection: done in phase A
                if (row_checks) {                          // _project_and_check_reset_t without a projection (nmf.py:751-769)
                    const double ps = tts[k];
                    if (b == 0 && tid == 0) {
                        st->nt1 = (mode == 0) ? ps : 1.0;
                        st->sumT = ps;
                    }
                    if (!(ps > 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0) {
                        if (b == 0 && tid == 0) {
                            st->halt = HALT_EVENT_RESET_T; st->halt_topic = t; st->halt_sweep = s; st->halt_pos = t;
                        }
                        return;                            // every workgroup: the same sums, the same verdict
                    }
                }
                const double cden = tts[t] + p.reg_w_l2;   // denom = nt + reg_w_l2 (nmf.py:465)
                int wmode = 0;
                if (!(cden > 0.0)) {
                    if (p.has_wrs && p.w_row_sum != 0.0) wmode = 1;
                    else {
                        if (b == 0 && tid == 0) {
                            st->halt = HALT_ERR_UNBOUNDED; st->halt_topic = t; st->halt_sweep = s; st->halt_pos = t;
                        }
                        return;
                    }
                }
                if (b == 0 && tid == 0) st->nt = tts[t];
                keep_fp32();
                // row dots of the own rows, 8 rows per round through the wave's LDS tile (wave_rowsum8)
#pragma unroll
                for (int r0 = 0; r0 < RPW; r0 += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        double part = 0.0;
                        if (r0 + u < RPW) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) part = fma((double)xr[r0 + u < RPW ? r0 + u : 0][c], tv[c], part);
                        }
                        wave_rowsum8_park(tile, u, lane, part);
                    }
                    const double tot = wave_rowsum8_finish(tile, lane);
                    const int lr = rg + RG * (r0 + (lane >> 3));
                    if ((lane & 7) == 0 && r0 + (lane >> 3) < RPW && lr < rows_here) ysh[(size_t)cg * a.rows_wg + lr] = tot;
                }
                __syncthreads();
                RRI_STAMP(4);
                // W column t for the own rows: 8 lanes per row share the k-term dot (lane sub takes the topics sub, sub + 8,
                // sub + 16: all their LDS reads in flight at once) and the CG row-dot partials; three DPP steps add the parts
                for (int r0 = 0; r0 < rows_here; r0 += NTH / 8) {
                    const int i = r0 + (tid >> 3), sub = tid & 7;
                    constexpr int TERMS = (ONCHIP_MAX_K + 7) / 8;
                    double part = 0.0, y = 0.0;
                    if (i < rows_here) {
                        double wv[TERMS], sv[TERMS];
#pragma unroll
                        for (int q = 0; q < TERMS; ++q) {
                            const int l = sub + 8 * q;
                            wv[q] = (l < k) ? Wl[(size_t)i * kS + l] : 0.0;
                            sv[q] = (l < k && l != t) ? tts[l] : 0.0;
                        }
                        if (sub < CG) y = ysh[(size_t)sub * a.rows_wg + i];
#pragma unroll
                        for (int q = 0; q < TERMS; ++q) part = fma(wv[q], sv[q], part);
                    }
                    part = group8_sum(part);
                    y = group4_sum(y);                                           // CG <= 4 partials, in lanes 0 .. 3
                    if (i < rows_here && sub == 0) {
                        const double numer = (y - part) - p.reg_w_l1;
                        double wnew;
                        if (wmode == 0) wnew = fmax(numer, 0.0) / (cden + p.eps);
                        else wnew = (-numer + cden < 0.0) ? p.w_row_sum : 0.0;
                        Wl[(size_t)i * kS + t] = wnew;
                        a.Wt[(i64)t * a.ldw + row0 + i] = wnew;
                        xyl[i] = wnew * y;
                    }
                }
                __syncthreads();
                if (wave == 0) {                           // <w_t, X t_t> over the own rows: the objective's cross term
                    double acc = 0.0;
#pragma unroll 1
                    for (int i = lane; i < rows_here; i += 64) acc += xyl[i];
                    acc = wave_sum<double>(acc);
                    if (lane == 0) a.xyp[(i64)t * a.xy_stride + b] = acc;
                }
                RRI_STAMP(5);
                carry_post((t + 1) % k, t, buf ^ 1);
                RRI_STAMP(6);
                chk = 1;
                tprev = t;
                have_carry = true;
            }
            if (hand_to_workers() == 2) goto sync_failed;
            stepq += 1u;
        }
    }
    RRI_STAMP(7);
    if (DBG && a.dbg && tid == 0 && (b == 0 || b == G - 1))
        for (int i = 0; i < 14; ++i) a.dbg[(b == 0 ? 0 : 16) + i] = dacc[i];
    // the column check of the last W update of the call (position of the next step: sweep s_end, topic 0): workgroup 0 is
    // a worker and has waited for every workgroup's partials
    if (chk && b == 0) {
        const double v = [&]() {
            double acc = 0.0;
#pragma unroll 1
            for (int q = lane; q < G; q += 64) acc += ld_agent(a.mkG + (size_t)(stepq & 1u) * (k + 2) * G + (unsigned)((k + 1) * G + q));
            return wave_sum<double>(acc);
        }();
        if (tid == 0) {
            const bool ev = (v <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
            const bool err = !ev && !(v > 0.0);
            if (ev || err) {
                st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
                st->halt_topic = tprev; st->halt_sweep = a.s_end; st->halt_pos = 0;
            }
        }
    }
    return;
sync_failed:
    if (b == 0 && tid == 0 && st->halt == 0) {
        st->halt = HALT_ERR_GRID_SYNC; st->halt_topic = -1; st->halt_sweep = 0; st->halt_pos = 0;
    }
}

#undef RRI_STAMP

}  // namespace rri
