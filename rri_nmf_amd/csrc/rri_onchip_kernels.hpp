// rri_onchip_kernels.hpp -- the plain (unweighted, Gram-form) sweep for problems whose X fits the chip's REGISTERS.
//
// A 10000 x 1000 fp32 X is 40 MB: 156 KB per CU of an MI355X, against 512 KB of vector registers and 160 KB of LDS per
// CU.  At that size the launch-per-phase schedule (k_trow_small, k_pass, k_wcol: three dependent launches per topic
// step, 22-24 us) spends its time on launch latencies and on re-reading a cache-resident X, not on work.  Here ONE
// persistent kernel -- one 512-thread workgroup per CU (two waves per SIMD, 256 registers each), all co-resident --
// loads its rows of X into registers once and runs whole sweeps; what crosses workgroups goes through small exchange arrays
// on the memory side, twice per topic step:
//
//   every workgroup b   rows [b rows_wg, (b+1) rows_wg) of X and W   (X: registers, W: LDS; the global k-major W kept current)
//   worker w < NA       also columns [w CWA, (w+1) CWA) of T         (all k rows of that slice in LDS); NA = ceil(LD / CWA)
//
//   phase B (t)  all    T T[t]^T = sum_w mkP[w], row checks (nmf.py:751-769), T[t,:] (mkT) into registers, y = X t for the
//                       own rows, W-column update (nmf.py:464-469, 728-734); then for tn = t+1:
//                       mkZ[b] = w_tn^T X, mkG[b] = (w_tn^T W, ||w_tn||^2, column sum of the update) over the own rows
//   phase A (tn) workers  g = sum_b mkG[b], column check (nmf.py:471-476), z_j = sum_b mkZ[b][j], numer_T, qf_min for the
//                       own columns (nmf.py:437-447, 670-676) -> T[tn, slice] (mkT), mkP[w] = partial T T[tn]^T and row sum
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
// Both halves free, 2 <= k <= 64, d <= 2048.  With the topic-model flags (T rows projected onto the simplex at every step;
// d <= 1024) the workers exchange their slices of the closed-form row (mkX) and each finishes the projection of the whole row:
// one more exchange among the workers per step.
//
// The exchanges (tools/barrier_probe.hip, profiles/r02_grid_barrier_variants.log, r03_onchip_sections.log): everything that
// crosses workgroups is written and read with agent-scope accesses (they go to the memory side, past the per-XCD L2s), so no
// L2 write-back / invalidate is needed.  Round 2 published a step with a flag per workgroup -- wait for the stores'
// acknowledgement, barrier, flag store; the consumer polls the flags, then loads the data: two dependent trips per hand-over,
// 2.7 us.  Now the DATA is its own hand-over: every exchange slot holds an "absent" marker until its value of the step is
// stored, consumers poll the values themselves (one trip), and the owner of a slot marks it absent again at a point where
// every reader is known to be past it (comments at the stores).  The only flags left are those of the all-grid hand-over at
// kernel entry.  The number of polls is BOUNDED: a grid that cannot make progress (workgroups not co-resident) raises an abort
// word that every poll also reads, so every wave reaches the end of the kernel and the host sees HALT_ERR_GRID_SYNC instead
// of a hang.
#pragma once
#include "rri_kernels.hpp"

namespace rri {

enum { HALT_ERR_GRID_SYNC = -8 };
constexpr int ONCHIP_THREADS = 512, ONCHIP_WAVES = ONCHIP_THREADS / 64;
constexpr int ONCHIP_SMALL_K = 22;    // k + 2 Gram entries = 8 waves x 3 in flight: one round of loads in phase A (KT = 3)
constexpr int ONCHIP_MAX_K = 64;      // KT = 8: two rounds beyond k = 46
constexpr int ONCHIP_CWA = 32;          // columns of T per worker
constexpr int ONCHIP_PG = ONCHIP_THREADS / ONCHIP_CWA;   // groups of workgroup partials in the column-sum reduction

struct OnchipArgs {
    const void* X; i64 ldx; int n, d, LD, k;      // X in the handle's storage type (the kernel is instantiated per type)
    double* Wt; i64 ldw; double* T; i64 ldt;
    // exchange arrays: two buffers each, by the parity of the step that reads; a slot holds ONCHIP_ABSENT until its value of the
    // step is stored
    double* mkZ;                   // [2][G][LD]   column-sum partials of the carried topic
    double* mkG;                   // [2][k+2][G]  Gram-row partials | ||w||^2 | column sum of the last update, entry-major
    double* mkP;                   // [2][k+1][64]  T T[t]^T partials | row sum of the new T row, entry-major (NA <= 64 workers)
    double* mkT;                   // [2][LD]  the T row of the step, slices from the workers
    double* mkX;                   // [2][LD]  the T row before its projection (topic-model flags): slices from the workers, by the
                                   // parity of the step; an element that has not arrived holds ONCHIP_ABSENT
    double* xyp; int xy_stride;    // <w_t, X t_t> partials for the objective (XYpart[t * xy_stride + b])
    double* objE;                  // [slots][G]  exchange slots: every workgroup's share of the objective of a sweep
    int track;                     // 1: the objective of the launch's LAST sweep (slot 0), without the constant 1/2 ||X||^2, is
                                   //    left in DevState.obj_track
                                   // 2: also every sweep's objective in objhist[sweep - s0] (slot sweep - s0), and the stop rule
                                   //    of nmf.py:510 / optimization.py:284-291 after every sweep but the last: a launch that
                                   //    finds |o_s - o_{s-1}| <= stop_scale ends there (HALT_EVENT_STOP, halt_sweep = s + 1)
    double* objhist;               // [sweeps of the launch]  (track == 2)
    double* dec;                   // [sweeps of the launch]  exchange slots: workgroup 0's verdict after sweep s0 + i (0 go on, 1 stop)
    double obj_prev, stop_scale, half_xsq;     // the objective before the launch, eps_stop |o_0 - o_1|, 1/2 ||X||^2
    unsigned* bar;                 // [0] abort word, [128 + b] the entry flag of workgroup b (zero at launch)
    int G, NA, rows_wg, CG, RG, kS;
    int s0, t0, ph0, s_end;        // cursor (sweep, topic, phase) and end sweep (exclusive)
    int skip_row_finish;           // a resumed W half whose T-row checks already ran (after a T-row reset)
    unsigned spin_limit;
    int nap_eighths;               // share of an observed wait slept through before the first load of the next one, in eighths
    unsigned entry_spin_limit;     // polls of the all-grid hand-over at kernel entry (short: a grid that is not resident as a whole shows here)
    unsigned jitter;               // tests (RRI_ONCHIP_JITTER = seed, 0 = off): every wave sleeps a seeded pseudo-random 0-5 us before each of
                                   // its exchange stores and first polls -- another ORDER of arrivals at every hand-over, the same results
    int fail_step;                 // tests (RRI_ONCHIP_FAIL_STEP): every workgroup gives up in phase B of this topic step of the launch, as if its
                                   // polls had run out -- a launch that fails IN the run, after W, T and the objective slots have been written; -1: never
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

// What crosses workgroups inside the sweep carries its own "it is there": every exchange slot holds ONCHIP_ABSENT until the value
// of the step is stored into it, and the readers poll the DATA until none of what they loaded is absent -- store, then ONE trip
// to the memory side, where a flag costs acknowledge + barrier + flag store + flag poll + the load of the data behind it
// (profiles/r03_onchip_tm_sections.log: 3.75 -> 1.9 us for the exchange of the row slices).  The markers are NaNs with a
// payload no arithmetic produces; they are told by their high word, and a store of real data that happened to carry one of
// them (an input NaN of that very pattern) is rewritten to the canonical NaN.  ONCHIP_HALTED in a slot: the workers ended the
// step with an event or an error (DevState says which) and every reader returns.
constexpr unsigned ONCHIP_ABSENT_HI = 0xfff7a5a5u, ONCHIP_HALTED_HI = 0xfff7a5a6u;
__device__ __forceinline__ bool onchip_absent(double v) { return (unsigned)__double2hiint(v) == ONCHIP_ABSENT_HI; }
__device__ __forceinline__ bool onchip_halted(double v) { return (unsigned)__double2hiint(v) == ONCHIP_HALTED_HI; }
__device__ __forceinline__ double onchip_absent_value() { return __hiloint2double((int)ONCHIP_ABSENT_HI, (int)ONCHIP_ABSENT_HI); }
__device__ __forceinline__ double onchip_halted_value() { return __hiloint2double((int)ONCHIP_HALTED_HI, (int)ONCHIP_HALTED_HI); }
__device__ __forceinline__ void st_data(double* p, double v) {
    const unsigned hi = (unsigned)__double2hiint(v);
    st_agent(p, (hi == ONCHIP_ABSENT_HI || hi == ONCHIP_HALTED_HI) ? __longlong_as_double(0x7ff8000000000000LL) : v);
}
// this wave's agent-scope stores so far have been acknowledged (what a later store -- after a workgroup barrier: of any wave
// -- may rely on having landed first)
__device__ __forceinline__ void onchip_stores_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Diagnostics (OnchipArgs.jitter): what a protocol of polled slots can break on is the ORDER in which stores and polls of
// different workgroups reach the memory side, and a quiet machine shows few orders.  With a seed, a wave sleeps 0 .. 9 naps of
// ~0.55 us -- a hash of (seed, workgroup, wave, topic step, site) -- before every exchange store, every "absent" re-mark and every
// first poll: each hand-over then sees its arrivals in another order, step after step (tests: the same bits as undisturbed).
__device__ __forceinline__ void onchip_jitter(unsigned seed, int b, int wave, unsigned step, unsigned site) {
    if (seed == 0u) return;
    unsigned h = seed ^ ((unsigned)b * 0x9E3779B1u) ^ ((unsigned)wave * 0x85EBCA77u) ^ (step * 0xC2B2AE3Du) ^ (site * 0x27D4EB2Fu);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    const int naps = (int)(h % 10u);
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(20);      // 1280 cycles each
}

// Loads v[u] = base[off(u)] (0.0 where off(u) is ONCHIP_NONE) until none is absent: `base` is wave-uniform, the offsets are
// recomputed for every round of loads rather than kept (registers), and a round that found something absent is repeated as a
// whole after a short sleep.  Every lane of the wave calls; returns, wave-uniform, 0 = all there, 2 = the grid gave up (the
// abort word, or this wave ran out of polls and raised it).
// (issue / finish apart: the first rounds of two polls can be in flight together)
constexpr unsigned ONCHIP_NONE = 0xffffffffu;
// The steps of a sweep are alike: what a wave waited for in the step before it will wait for again, about as long.  2048 waves
// that poll a few kilobytes while the workers work keep the memory side busy with exactly the channels the workers' own exchange
// goes through, so a wave sleeps through the first part of a wait it has reason to expect before its first load (measured: 3 %
// of a plain topic step at 10000 x 1000, nothing with the projection stage).  The estimate only moves on evidence, and
// never close to the expected arrival -- a wave that oversleeps delays the workgroups that wait for IT, whose waits then look
// longer, and with naps that track the whole wait that feeds on itself (measured: 500 us per topic step):
//   a wait that needed more than one round of loads saw the data arrive: nap 5/8 of that wait next time;
//   a wait whose first round found everything there may have overslept: nap 3/4 of the last nap.
// (ticks of 100 MHz)
struct OnchipNap {
    int ticks = 0;
    long long t0 = 0;
    __device__ __forceinline__ void before() {
        t0 = wall_clock64();
        const long long until = t0 + ticks;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(4);
    }
    __device__ __forceinline__ void after(int rounds, int eighths) {      // eighths: 5 (RRI_ONCHIP_NAP_EIGHTHS; 0 = no naps)
        // capped at 20 us: the waits of a step are 3-15 us; one that took milliseconds (the grid was preempted: a device shared
        // with another process) says nothing about the next, and a nap of that length would take dozens of steps to decay
        const int waited = (int)min(wall_clock64() - t0, 0x3fffffffLL);
        ticks = rounds > 0 ? (min(waited, 3276) * eighths) >> 3 : (ticks * 3) >> 2;
    }
};
// (off(u, z): z is a zero the compiler cannot see through, to be added to the lane-dependent term of the offset -- otherwise the
// offsets of the repeated round are loop-invariant, get hoisted out of the polling loop and are kept in registers after all)
template <int N, typename OffFn>
__device__ __forceinline__ void onchip_poll_issue(const double* base, OffFn off, double (&v)[N]) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const unsigned o = off(u, z);
        v[u] = o != ONCHIP_NONE ? ld_agent(base + o) : 0.0;
    }
}
template <int N, typename OffFn>
__device__ __forceinline__ int onchip_poll_finish(const double* base, OffFn off, double (&v)[N], unsigned* bar, unsigned spin_limit,
                                                  int* rounds = nullptr) {
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < N; ++u) ok = ok && !onchip_absent(v[u]);
        if (__all(ok)) {
            if (rounds) *rounds = (int)spins;
            return 0;
        }
        if ((++spins & 15u) == 0u && __hip_atomic_load(bar, __ATOMIC_RELAXED, RRI_AGENT) != 0u) return 2;
        if (spins > spin_limit) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(bar, 1u, __ATOMIC_RELAXED, RRI_AGENT);
            return 2;
        }
        __builtin_amdgcn_s_sleep(2);
        onchip_poll_issue<N>(base, off, v);
    }
}

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

// sums over `np` <= 64 PPL workgroup partials part[e * stride + q] for the entries e = wave, wave + 8, ... < ne, into out[e]: a
// lane takes PPL partials of an entry (4 for the G <= 256 partials of the carries, 1 for the NA <= 64 of the workers) and NL
// loads are in flight together, i.e. 8 NL / PPL entries per round of loads -- one round for k <= 22 at NL = 12 -- and every
// workgroup adds in the same order.  The partials are exchange slots: polled until they are there (onchip_poll_*).  Returns 0,
// or 2 = the grid gave up; *halted is raised when a partial carries the workers' halt marker.
// (entry-major arrays: the np partials of an entry are contiguous, `stride` doubles per entry -- with workgroup-major rows every
// 8-byte load of a partial touched a sector of its own, eight times the bytes)
template <int NL, int PPL>
__device__ __forceinline__ int onchip_entry_sums(const double* part, int stride, int ne, int np, double* out, int wave, int lane,
                                                 unsigned* bar, unsigned spin_limit, int* halted) {
    constexpr int EB = NL / PPL;
    int failed = 0;
    for (int e0 = wave; e0 < ne; e0 += ONCHIP_WAVES * EB) {
        double v[NL];
        auto off = [&](int i, int z) -> unsigned {
            const int e = e0 + ONCHIP_WAVES * (i / PPL), q = lane + z + 64 * (i % PPL);
            return (e < ne && q < np) ? (unsigned)(e * stride + q) : ONCHIP_NONE;
        };
        onchip_poll_issue<NL>(part, off, v);
        failed |= onchip_poll_finish<NL>(part, off, v, bar, spin_limit);
#pragma unroll
        for (int m = 0; m < EB; ++m) {
            const int e = e0 + ONCHIP_WAVES * m;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < PPL; ++u) {
                acc += v[m * PPL + u];
                if (onchip_halted(v[m * PPL + u])) *halted = 1;
            }
            const double tot = wave_sum<double>(acc);
            if (lane == 0 && e < ne) out[e] = tot;
        }
    }
    return failed;
}

// Michelot's fixed point for the simplex projection (the iteration of simplex_theta, rri_kernels.hpp) by ONE wave, the row left
// in LDS (LD <= 1024: 16 elements per lane, re-read in every pass: 8 KB at 128 B per clock, no registers held) -- no workgroup
// barrier inside the iteration.  Round 2's 8-wave form (two elements per thread) cost two barriers per iteration plus two block
// sums around it, ~30 barriers per projected row, on the critical path of every topic step of the topic-model flags; this one
// needs two (before and after).  The elements are w_j = row[j] when !shifted,
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
    // row[d .. 1024) holds -1e300 ("no element"): no bound checks here -- sixteen lane masks kept across the passes were
    // sixteen scratch reloads inside every pass
    auto elem = [&](int q) -> double {
        const double v = row[lane + 64 * q];
        return shifted ? (v < -1.0e299 ? v : fmax(v - shift, 0.0)) : v;
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
// tools/onchip_probe.py): 0 wait for the carries + their load, 1 phase A rest, 2 next carry + wait for the workers, 3 marking the
// consumed slots absent, 4 row dots, 5 W update, 6 carry_post, 7 to the top of the next step; with the projection: 8 closed
// form, 10 slices arrive, 12 projected
// PROJ: the instantiation for the topic-model flags (the projection stage costs the plain one registers it does not have)
// KT: k-term dots take KT terms per lane of an 8-lane group: 3 for k <= 24, 8 for k <= 64 (same sums: the terms past k are zeros)
template <typename SX, int RPW, bool DBG = false, bool PROJ = false, int KT = 3>
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
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;          // taken anew in every topic step: see zstep below
    const int b = blockIdx.x, G = a.G, NA = a.NA, k = a.k, kS = a.kS, CG = a.CG, RG = a.RG;
    int cg = wave % CG, rg = wave / CG;
    int col0 = cg * 256 + lane * 4;
    const int row0 = b * a.rows_wg;
    const int rows_here = max(0, min(a.rows_wg, a.n - row0));
    const bool worker = b < NA;
    const int j0 = b * CWA;                                 // first column of a worker's T slice
    const KParams p = a.p;
    unsigned* flagB = a.bar + 128;                          // the entry hand-over

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* Wl = reinterpret_cast<double*>(smem);           // [rows_wg][kS]   own rows of W
    double* Tl = Wl + (size_t)a.rows_wg * kS;               // [k][CWA]        a worker's column slice of T
    double* gsh = Tl + (size_t)k * CWA;                     // [k + 2]
    double* tts = gsh + k + 2;                              // [k + 1]
    double* zred = tts + k + 1;                             // [PG][CWA]       partial column sums of a worker
    double* ysh = zred + PG * CWA;                          // [CG][rows_wg]
    double* xyl = ysh + (size_t)CG * a.rows_wg;             // [rows_wg]
    double* qyl = xyl + a.rows_wg;                          // [rows_wg]  a row's share of the objective's W-side terms of the step
    double* zsh = qyl + a.rows_wg;                          // [RG][CG * 256] = 8 x 256
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

    // The carry of topic tn -- column sums w_tn^T X and Gram-row partials of w_tn over the own rows -- in two parts.
    // carry_pre: everything that does not depend on the W column `texcl` of the running step (column tn itself is not
    // touched by that step): the column sums, ||w_tn||^2 and the Gram entries against every other column.  It runs while
    // the workgroup waits for the workers' data, off the chain of dependent exchanges (19.1 -> 18.7 us per topic step at
    // 10000 x 1000; running it two steps ahead, in the workers' own wait, measured the same).  carry_post: the two
    // entries that need the updated column t.  The arrays are double-buffered by the parity of the step that READS them,
    // so the workers may still be reading the running step's buffer while the next one is written.
    // zstep: a zero the compiler cannot see through, taken anew in every topic step and added to the thread index, from which
    // lane, wave, row group and column are then derived again.  Without it every per-thread LDS offset, global offset and lane
    // mask of the step (the unrolled loops over 20 rows alone: 20 registers + 20 mask pairs each) is invariant across steps,
    // gets hoisted out of the sweep loop, does not fit next to the resident X and comes back from scratch one by one inside the
    // loops it was hoisted from: a scratch load and a wait per row on the critical path (100-130 spilled registers in the
    // topic-model instantiation; none in the plain one, 60 with the projection).
    int zstep = 0;
    unsigned stepq = 0;       // topic steps of this launch so far
#define RRI_JIT(site) onchip_jitter(a.jitter, b, wave, stepq, site)
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
        RRI_JIT(1u);
#pragma unroll 1
        for (int j = tid; j < a.LD; j += NTH) {
            double zq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) zq[q] = q < RG ? zsh[(size_t)q * LDp + j] : 0.0;
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < RG) s += zq[q];
            st_data(mkZb + (unsigned)(b * a.LD + j), s);
        }
        RRI_JIT(2u);
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
            if (lane == 0) st_data(mkGb + (unsigned)(e * G + b), acc);
        }
        __syncthreads();                                           // zsh is free again
    };
    auto carry_post = [&](int tn, int t, int buf) {
        double* mkGb = a.mkG + (size_t)buf * (k + 2) * G;
        if (wave < 2) {                                            // wave 0: <w_tn, w_t new>, wave 1: sum of the new column t
            RRI_JIT(3u);
            double acc = 0.0;
#pragma unroll 1
            for (int i = lane; i < rows_here; i += 64) {
                const double wt = Wl[(size_t)i * kS + t];
                acc = wave == 0 ? fma(Wl[(size_t)i * kS + tn], wt, acc) : acc + wt;
            }
            acc = wave_sum<double>(acc);
            if (lane == 0) st_data(mkGb + (unsigned)((wave == 0 ? t : k + 1) * G + b), acc);
        }
    };
    bool have_carry = false;
    OnchipNap napA, napB;          // this wave's waits for the carries (workers) / for the workers
    // The objective of a sweep without a pass over X or a Gram kernel afterwards (nmf() asks for it after every sweep,
    // nmf.py:488-490, 510): 1/2 ||X - W T||^2 = 1/2 ||X||^2 - sum_t <w_t, X t_t> + 1/2 <W^T W, T T^T>, and with A = W^T W,
    // B = T T^T of the END of the sweep, 1/2 <A, B> = sum_t (sum_{l<t} A_tl B_tl + 1/2 A_tt B_tt): at step t the columns l < t
    // of W and the rows l <= t of T are final, B_tl is what every workgroup holds in tts[l], and A_tl is a sum over rows -- so a
    // ROW's share of step t is w_it (-y_i + sum_{l<t} B_tl w_il + 1/2 B_tt w_it) (+ its share of the penalties), one more
    // masked k-term dot next to the one the update needs.  eacc: this workgroup's sum over the steps of the running sweep
    // (wave 1; workgroup 0 adds the T-side penalties), stored at the end of the launch's last sweep.
    double eacc = 0.0;
    double obj_before = a.obj_prev;      // workgroup 0: the objective of the sweep before the one that has just ended
    int chk = 0, tprev = -1;

    // Entry hand-over: every workgroup reports in and waits for all others BEFORE anything is written to W, T or the partial
    // arrays.  A grid that is not resident as a whole -- two processes' persistent grids dispatched at the same moment each
    // hold a part of the CUs (tools/onchip_two_processes.py: 5-13 times in 400 calls with four processes on one GPU) -- fails
    // HERE, after entry_spin_limit polls (~tens of milliseconds) instead of the seconds of the in-run bound, with nothing to
    // undo; the host reruns the range launch by launch and backs the process off the persistent path for a while.
    // Every exchange slot this workgroup owns starts "absent", in both buffers.
    for (int e = tid; e < 2 * a.LD; e += NTH) st_agent(a.mkZ + (size_t)(e / a.LD) * G * a.LD + (unsigned)(b * a.LD + e % a.LD), onchip_absent_value());
    if (tid < 2 * (k + 2)) st_agent(a.mkG + (size_t)(tid / (k + 2)) * (k + 2) * G + (unsigned)((tid % (k + 2)) * G + b), onchip_absent_value());
    if (a.track == 1 && tid == NTH - 1) st_agent(a.objE + (unsigned)b, onchip_absent_value());
    if (a.track == 2)
        for (int i = tid; i < a.s_end - a.s0; i += NTH) {
            st_agent(a.objE + (unsigned)(i * G + b), onchip_absent_value());
            if (b == 0) st_agent(a.dec + (unsigned)i, onchip_absent_value());
        }
    if (worker) {
        if (tid < 2 * CWA) {
            const int which = tid / CWA, jl = tid % CWA;
            if (j0 + jl < a.LD) {
                st_agent(a.mkT + (size_t)which * a.LD + (unsigned)(j0 + jl), onchip_absent_value());
                st_agent(a.mkX + (size_t)which * a.LD + (unsigned)(j0 + jl), onchip_absent_value());
            }
        }
        if (tid >= 64 && tid < 64 + 2 * (k + 1)) {
            const int e = tid - 64;
            st_agent(a.mkP + (size_t)(e / (k + 1)) * (k + 1) * 64 + (unsigned)((e % (k + 1)) * 64 + b), onchip_absent_value());
        }
    }
    onchip_signal(flagB + b, 1u);
    if (onchip_wait(a.bar, flagB, G, 1u, a.entry_spin_limit) == 2) goto sync_failed;
    for (int s = a.s0; s < a.s_end; ++s) {
        for (int t = (s == a.s0) ? a.t0 : 0; t < k; ++t) {
            asm volatile("v_mov_b32 %0, 0" : "=v"(zstep));
            tid = (int)threadIdx.x + zstep; lane = tid & 63; wave = tid >> 6;
            cg = wave % CG; rg = wave / CG; col0 = cg * 256 + lane * 4;
            tile = tiles + wave * (8 * 72);
            const int ph = (s == a.s0 && t == a.t0) ? a.ph0 : 0;
            const bool update_T = ph == 0;
            int mode = 0;
            const int buf = (int)(stepq & 1u);             // buffers phase A of this step reads; the next step's are buf ^ 1
            const bool last_step = (s == a.s_end - 1) && (t == k - 1);
            const double* mkZr = a.mkZ + (size_t)buf * G * a.LD;
            const double* mkGr = a.mkG + (size_t)buf * (k + 2) * G;
            double* mkPw = a.mkP + (size_t)buf * (k + 1) * 64;
            double* mkTw = a.mkT + (size_t)buf * a.LD;
            if (update_T && !have_carry) carry_pre(t, -1, buf);      // the first step of a launch: no check pending (chk == 0), the
                                                                     // column-sum entry k + 1 is neither written nor read
            // ---------------- phase A (workers): T row t on the own column slice -----------------------------------
            RRI_STAMP(7);
            unsigned halt_bit = 0u;
            if (worker) {
                if (update_T) {
                    // the partials of the own columns: thread = (column jl, group pg of workgroups pg, pg + PG, ...); G <= 16 PG.
                    // Polled until every workgroup's carry of the step is there -- this is the hand-over from phase B
                    const int jl = tid % CWA, pg = tid / CWA;
                    double zp[16];
                    const bool col_ok = j0 + jl < a.LD;
                    auto zoff = [&](int u, int z) -> unsigned {
                        const int q = pg + z + PG * u;
                        return (col_ok && q < G) ? (unsigned)(q * a.LD + j0 + jl) : ONCHIP_NONE;
                    };
                    RRI_JIT(7u);
                    napA.before();
                    onchip_poll_issue<16>(mkZr, zoff, zp);
                    int never = 0;
                    int failed = onchip_entry_sums<KT == 3 ? 12 : 24, 4>(mkGr, G, k + 1 + chk, G, gsh, wave, lane, a.bar, a.spin_limit, &never);
                    int roundsA = 0;
                    failed |= onchip_poll_finish<16>(mkZr, zoff, zp, a.bar, a.spin_limit, &roundsA);
                    napA.after(roundsA, a.nap_eighths);
                    double zacc = 0.0;
#pragma unroll
                    for (int u = 0; u < 16; ++u) zacc += zp[u];
                    zred[pg * CWA + jl] = zacc;
                    if (__syncthreads_or(failed)) goto sync_failed;
                    // Every workgroup's carry of this step is here, so all of them are past their reads of the step before: this
                    // worker's slots of the OTHER buffers of mkT / mkP (the next step's) go back to "absent".  The stores are
                    // acknowledged (onchip_stores_landed, then a barrier) before anything of this step is stored for the
                    // others, so whoever has seen this step's values cannot find last-but-one step's in those slots.
                    RRI_JIT(11u);
                    if (tid < CWA && j0 + tid < a.LD) st_agent(a.mkT + (size_t)(buf ^ 1) * a.LD + (unsigned)(j0 + tid), onchip_absent_value());
                    if (tid >= 64 && tid < 64 + k + 1) st_agent(a.mkP + (size_t)(buf ^ 1) * (k + 1) * 64 + (unsigned)((tid - 64) * 64 + b), onchip_absent_value());
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
                    if (a.track == 2 && t == 0 && s > a.s0 && code == 0) {
                        // A sweep has just ended (its last column check passed above): workgroup 0 adds up its objective, applies
                        // the stop rule and tells the other workers, which hold their T row back until they know.  "Stop" ends
                        // the launch like any event -- nothing of sweep s has been stored for good.
                        const int slot = s - 1 - a.s0;
                        double es[4];
                        const double* src = b == 0 ? a.objE + (size_t)slot * G : a.dec + slot;
                        const int cnt = b == 0 ? G : 1;
                        auto eoff = [&](int u, int z) -> unsigned { return lane + z + 64 * u < cnt ? (unsigned)(lane + z + 64 * u) : ONCHIP_NONE; };
                        RRI_JIT(10u);
                        onchip_poll_issue<4>(src, eoff, es);
                        const int efailed = onchip_poll_finish<4>(src, eoff, es, a.bar, a.spin_limit);
                        if (__syncthreads_or(efailed)) goto sync_failed;
                        const double ev = wave_sum<double>(((es[0] + es[1]) + es[2]) + es[3]);
                        bool stop;
                        if (b == 0) {
                            const double obj = a.half_xsq + ev;
                            stop = fabs(obj - obj_before) <= a.stop_scale;
                            obj_before = obj;
                            if (tid == 0) {
                                a.objhist[slot] = obj;
                                st_data(a.dec + slot, stop ? 1.0 : 0.0);
                            }
                        } else {
                            stop = ev != 0.0;
                        }
                        if (stop) {
                            code = HALT_EVENT_STOP;
                            if (b == 0 && tid == 0) { st->halt = code; st->halt_topic = -1; st->halt_sweep = s; st->halt_pos = 0; }
                        }
                    }
                    RRI_JIT(4u);
                    if (code != 0) halt_bit = 0x80000000u;     // every worker takes the same verdict from the same sums
                    else if (tid < 8 * CWA) {
                        // the closed form for the own columns: 8 lanes per column share the PG partial column sums and the
                        // k-term product with the Gram row (all their LDS reads in flight at once), three DPP steps add the parts
                        const int jc = tid >> 3, sub = tid & 7;
                        constexpr int TERMS = KT;
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
                            if (project) st_data(a.mkX + (size_t)buf * a.LD + (unsigned)(j0 + jc), x);
                            else {
                                Tl[t * CWA + jc] = x;
                                st_agent(a.T + (i64)t * a.ldt + j0 + jc, x);
                            }
                        }
                    }
                    onchip_stores_landed();
                    __syncthreads();
                    if (b == 0 && tid == 0 && code == 0) st->tmode = mode;
                    if (project && code == 0) {
                        // the closed-form row is complete only across the workers: every worker takes all slices (an exchange
                        // among the NA workers) and finishes qf_min for the whole row itself -- Michelot's fixed point for the
                        // simplex projection, or the one-hot row -- and the checks of _project_and_check_reset_t
                        // (nmf.py:751-769), exactly as k_trow_final does; all workers come to the same row and verdict
                        RRI_STAMP(8);                          // closed form of the own columns
                        // the slices are exchange slots of the step's buffer of mkX: every thread polls the two elements it stages
                        {
                            const double* xin = a.mkX + (size_t)buf * a.LD;
                            double r[2];
                            auto xoff = [&](int u, int z) -> unsigned { return tid + z + NTH * u < a.d ? (unsigned)(tid + z + NTH * u) : ONCHIP_NONE; };
                            RRI_JIT(8u);
                            onchip_poll_issue<2>(xin, xoff, r);
                            const int failed = onchip_poll_finish<2>(xin, xoff, r, a.bar, a.spin_limit);
                            rowsh[tid] = tid < a.d ? r[0] : -1.0e300;                    // past d: "no element" for the fixed point
                            rowsh[tid + NTH] = tid + NTH < a.d ? r[1] : -1.0e300;
                            if (__syncthreads_or(failed)) goto sync_failed;
                            // the other buffer is the next step's: its slots of the own slice go back to "absent".  Every worker
                            // has stored this step's slice, so all of them are past their reads of the step before; the stores
                            // are acknowledged before this workgroup's carry of the next step (phase B), which every reader of the
                            // next step's slices has polled first
                            RRI_JIT(12u);
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
                RRI_JIT(5u);
                if (halt_bit == 0u) {
                    // the row of the step for everybody (columns past d: zeros), then T T[t]^T over the own slice; [k] = sum of
                    // the row.  Thread = (entry e, quarter of the 32 columns): 8 LDS reads in flight each, the four quarters of
                    // an entry sit in adjacent lanes and are added in order
                    if (tid < CWA && j0 + tid < a.LD) st_data(mkTw + (unsigned)(j0 + tid), Tl[t * CWA + tid]);
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
                        if (part == 0 && e < k + 1) st_data(mkPw + (unsigned)(e * 64 + b), ((acc + a1) + a2) + a3);
                    }
                } else {
                    // an event or an error (DevState says which; every worker has come to the same verdict from the same sums):
                    // the halt marker in every slot the others poll
                    if (tid < CWA && j0 + tid < a.LD) st_agent(mkTw + (unsigned)(j0 + tid), onchip_halted_value());
                    if (tid >= 64 && tid < 64 + k + 1) st_agent(mkPw + (unsigned)((tid - 64) * 64 + b), onchip_halted_value());
                }
                RRI_STAMP(1);
            }
            // (measured: the same call at the TOP of the step, where a worker could do it while this step's carries are on their
            // way, costs 16.0 instead of 13.8 us per topic step at 10000 x 1000 -- there its stores and barriers sit in front of
            // the worker's poll instead of inside a wait)
            if (halt_bit == 0u && !last_step) carry_pre((t + 1) % k, t, buf ^ 1);   // while the workers' stores travel

            // ---------------- phase B: row checks, W column t on the own rows, carry of topic t + 1 ---------------
            {
                // the hand-over from the workers: the T row (this lane's 4 columns) and the partials of T T[t]^T, polled until
                // every worker's are there
                double tv[4];
                auto toff = [&](int c, int z) -> unsigned { return col0 + z + c < a.LD ? (unsigned)(col0 + z + c) : ONCHIP_NONE; };
                RRI_JIT(9u);
                napB.before();
                onchip_poll_issue<4>(mkTw, toff, tv);
                int halted = 0;
                int failed = onchip_entry_sums<KT == 3 ? 3 : 9, 1>(mkPw, 64, k + 1, NA, tts, wave, lane, a.bar, a.spin_limit, &halted);
                if (a.fail_step >= 0 && (int)stepq == a.fail_step) failed = 2;      // tests: the in-run give-up, at a chosen step
                int roundsB = 0;
                failed |= onchip_poll_finish<4>(mkTw, toff, tv, a.bar, a.spin_limit, &roundsB);
                napB.after(roundsB, a.nap_eighths);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (onchip_halted(tv[c])) halted = 1;
                if (__syncthreads_or(failed | halted)) {
                    if (__syncthreads_or(failed)) goto sync_failed;
                    return;                                // the workers found an event or an error: DevState says which
                }
                RRI_STAMP(2);
                // every worker has stored its row of the step, so all of them are past their reads of this step's carry: the own
                // slots of THIS step's buffers of mkZ / mkG (read again two steps on) go back to "absent", acknowledged before
                // the last entries of the next step's carry are stored (carry_post, after the barrier below)
                RRI_JIT(13u);
                for (int j = tid; j < a.LD; j += NTH) st_agent(a.mkZ + (size_t)buf * G * a.LD + (unsigned)(b * a.LD + j), onchip_absent_value());
                if (tid < k + 2) st_agent(a.mkG + (size_t)buf * (k + 2) * G + (unsigned)(tid * G + b), onchip_absent_value());
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
                    constexpr int TERMS = KT;
                    double part = 0.0, part_lt = 0.0, y = 0.0;
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
#pragma unroll
                        for (int q = 0; q < TERMS; ++q) part_lt = fma(wv[q], (sub + 8 * q < t) ? sv[q] : 0.0, part_lt);
                    }
                    part = group8_sum(part);
                    part_lt = group8_sum(part_lt);
                    y = group8_sum(y);                                           // CG <= 8 partials, in lanes 0 .. CG - 1
                    if (i < rows_here && sub == 0) {
                        const double numer = (y - part) - p.reg_w_l1;
                        double wnew;
                        if (wmode == 0) wnew = fmax(numer, 0.0) / (cden + p.eps);
                        else wnew = (-numer + cden < 0.0) ? p.w_row_sum : 0.0;
                        Wl[(size_t)i * kS + t] = wnew;
                        a.Wt[(i64)t * a.ldw + row0 + i] = wnew;
                        xyl[i] = wnew * y;
                        qyl[i] = wnew * (part_lt + 0.5 * tts[t] * wnew + 0.5 * p.reg_w_l2 * wnew + p.reg_w_l1);
                    }
                }
                onchip_stores_landed();                    // the "absent" marks above (and a worker's of phase A) before carry_post
                __syncthreads();
                if (wave == 0) {                           // <w_t, X t_t> over the own rows: the objective's cross term
                    double acc = 0.0;
#pragma unroll 1
                    for (int i = lane; i < rows_here; i += 64) acc += xyl[i];
                    acc = wave_sum<double>(acc);
                    if (lane == 0) a.xyp[(i64)t * a.xy_stride + b] = acc;
                } else if (wave == 1 && a.track) {         // this step's share of the objective (see eacc)
                    double acc = 0.0;
#pragma unroll 1
                    for (int i = lane; i < rows_here; i += 64) acc += qyl[i] - xyl[i];
                    acc = wave_sum<double>(acc);
                    if (t == 0) eacc = 0.0;
                    eacc += acc;
                    if (b == 0) eacc += 0.5 * p.reg_t_l2 * tts[t] + p.reg_t_l1 * tts[k];
                    if (t == k - 1) RRI_JIT(6u);
                    if (t == k - 1 && lane == 0 && (a.track == 2 || last_step))
                        st_data(a.objE + (unsigned)((a.track == 2 ? (s - a.s0) * G : 0) + b), eacc);
                }
                RRI_STAMP(5);
                carry_post((t + 1) % k, t, buf ^ 1);
                RRI_STAMP(6);
                chk = 1;
                tprev = t;
                have_carry = true;
            }
            stepq += 1u;
        }
    }
    RRI_STAMP(7);
    if (DBG && a.dbg && tid == 0 && (b == 0 || b == G - 1))
        for (int i = 0; i < 14; ++i) a.dbg[(b == 0 ? 0 : 16) + i] = dacc[i];
    // the column check of the last W update of the call (position of the next step: sweep s_end, topic 0): workgroup 0
    // polls every workgroup's partial of the column sum (G <= 256: four per lane)
    if (chk && b == 0) {
        double cs[4];
        const double* last = a.mkG + (size_t)(stepq & 1u) * (k + 2) * G + (unsigned)((k + 1) * G);
        auto coff = [&](int u, int z) -> unsigned { return lane + z + 64 * u < G ? (unsigned)(lane + z + 64 * u) : ONCHIP_NONE; };
        onchip_poll_issue<4>(last, coff, cs);
        const int failed = onchip_poll_finish<4>(last, coff, cs, a.bar, a.spin_limit);
        if (__syncthreads_or(failed)) goto sync_failed;
        const double v = wave_sum<double>(((cs[0] + cs[1]) + cs[2]) + cs[3]);
        if (tid == 0) {
            const bool ev = (v <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
            const bool err = !ev && !(v > 0.0);
            if (ev || err) {
                st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
                st->halt_topic = tprev; st->halt_sweep = a.s_end; st->halt_pos = 0;
            }
        }
    }
    // the objective of the launch's last sweep (minus the constant 1/2 ||X||^2): the workgroups' shares, polled like everything
    // else that crosses workgroups, added in workgroup order
    if (a.track && chk && b == 0) {
        double es[4];
        const int slot = a.track == 2 ? a.s_end - 1 - a.s0 : 0;
        const double* src = a.objE + (size_t)slot * G;
        auto eoff = [&](int u, int z) -> unsigned { return lane + z + 64 * u < G ? (unsigned)(lane + z + 64 * u) : ONCHIP_NONE; };
        onchip_poll_issue<4>(src, eoff, es);
        const int failed = onchip_poll_finish<4>(src, eoff, es, a.bar, a.spin_limit);
        if (__syncthreads_or(failed)) goto sync_failed;
        const double v = wave_sum<double>(((es[0] + es[1]) + es[2]) + es[3]);
        if (tid == 0) {
            st->obj_track = v;
            if (a.track == 2) a.objhist[slot] = a.half_xsq + v;
        }
    }
    return;
sync_failed:
    if (b == 0 && tid == 0 && st->halt == 0) {
        st->halt = HALT_ERR_GRID_SYNC; st->halt_topic = -1; st->halt_sweep = 0; st->halt_pos = 0;
    }
}

#undef RRI_STAMP
#undef RRI_JIT

}  // namespace rri
