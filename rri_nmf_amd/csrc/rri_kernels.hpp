// rri_kernels.hpp -- hand-written gfx950 kernels of the RRI topic step (Gram form).
//
// One topic step t of nmf.py:415-476 is, in the reference, two BLAS2 products over X:
//     w_t^T X  (nmf.py:672)   and   X t_t  (nmf.py:729)
// Here ONE streaming pass over X (k_pass) computes  X t_t  (row dots, for the W-column
// update of topic t) AND  w_{t+1}^T X  (column sums, for the T-row update of topic t+1):
// column t+1 of W is not touched between the two (only column t changes in step t), so
// the fused pass is exact and X is read k (+1) times per sweep instead of 2k.
//
// Chain per topic (all on one stream, no host sync):
//   k_reduce      partial column sums / Gram row  -> reduce buffer (the multi-GPU all-reduce point)
//   k_trow_numer  numer_T = w^T X - (w^T W)_{-t} T - reg ; closed-form qf_min       (nmf.py:437-447)
//   k_trow_final  simplex projection / one-hot / reset check; writes T[t,:]          (nmf.py:447,751-761)
//   k_tgram       T T[t,:]^T (k-vector, entry t zeroed) and ||T[t,:]||^2               (nmf.py:730-734)
//   k_pass        the X pass (HBM-bound, the roofline kernel)
//   k_wcol        numer_W = X t - W (T t)_{-t} - reg ; qf_min ; writes W[:,t];
//                 Gram row / norm partials of the NEXT topic                             (nmf.py:464-469,673-676)
// Precision: X (and the mask / masked residual) live in HBM in the handle's storage type SX
// (fp32 or fp64).  EVERYTHING else -- W, T, partial sums, the closed-form updates -- is float64:
// the streaming pass converts each loaded X element and accumulates with v_fma_f64.  The pass is
// HBM-bound with the vector ALU ~85 % idle, so float64 arithmetic costs no time, X traffic is
// unchanged, and the result follows the reference's float64 numpy to summation-order rounding
// instead of the ~1e-4 an fp32 accumulation gives on these iterations.
// Rare branches (reset conditions, unbounded problems, dead columns) set DevState::halt;
// every later kernel of the queue then returns at once and the host resolves the event.
#pragma once
#include <type_traits>

#include "rri_device.hpp"
#include "rri_hip.h"

namespace rri {

enum { HALT_EVENT_RESET_T = 1, HALT_EVENT_RESET_W = 2,
       HALT_EVENT_STOP = 3,    // the persistent sweep: the stop rule of nmf.py:510 held at the end of sweep halt_sweep - 1
       HALT_ERR_UNBOUNDED = -4, HALT_ERR_W_COL_ZERO = -5, HALT_ERR_NOT_IMPLEMENTED = -6 };
enum { RESET_NONE = 0, RESET_MAX_RESID = 1, RESET_RANDOM = 2 };

struct DevState {
    int halt;        // 0 = running, >0 event, <0 error
    int halt_topic;  // topic the event refers to
    int halt_sweep;  // position of the DETECTING step
    int halt_pos;
    int tmode;       // qf_min branch of the current T row: 0 c>0, 1 c<=0 bounds, 2 c<=0 one-hot
    int proj_iters;  // Michelot iterations of the last projection (diagnostic)
    int pad0;        // k_wsweep_verdict: first column k_wsweep_repair restores after a reset event (0: none); cleared with halt
    int pad1;
    double nt1;      // 1-norm of the unprojected T-row solution (qf_min's nx, nmf.py:447)
    double nt;       // ||T[t,:]||^2
    double sumT;
    double theta;
    double obj_track;   // the persistent sweep: objective of the launch's last sweep minus 1/2 ||X||^2 (OnchipArgs.track)
};

struct KParams {
    int fix_W, fix_T, project_T, has_trs, has_wrs, reset_method, resets_left, pad;
    double t_row_sum, w_row_sum, reg_w_l1, reg_w_l2, reg_t_l1, reg_t_l2, eps;
};

// =========================================================================================
// k_pass: the fused streaming pass over X.
//   grid = npg * nrb workgroups of 256 threads (4 waves); geometry in the comment at the kernel.
//   Per row a wave issues one 16-byte load per lane (1 KiB coalesced; non-temporal: X is never
//   reused inside a pass), keeps U rows in flight, holds its slice of the active T row (DO_Y)
//   and its column-sum accumulators (DO_Z) in registers (float64), and reads the active W-column
//   entries from LDS.  Narrow per-wave panels keep the register count low (more waves resident,
//   more bytes in flight): measured 6.0 TB/s against 4.8 TB/s for 4 KiB-wide panels.
//   UPD: the explicit-residual form north_star names: X is the residual R and the pass first
//   applies the rank-one update R <- R - a b^T (a from LDS, b slice in registers), writes R back
//   and takes the row dots / column sums of the UPDATED residual in the same sweep over memory.
// =========================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename SX> struct XVec;
template <> struct XVec<float> {
    typedef f32x4 type;
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const f32x4& v, double (&o)[4]) {
        o[0] = (double)v[0]; o[1] = (double)v[1]; o[2] = (double)v[2]; o[3] = (double)v[3];
    }
    static __device__ __forceinline__ f32x4 pack(const double (&o)[4]) {
        return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
    }
    static __device__ __forceinline__ f32x4 zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
};
template <> struct XVec<double> {
    typedef f64x2 type;
    static constexpr int N = 2;
    static __device__ __forceinline__ void unpack(const f64x2& v, double (&o)[2]) { o[0] = v[0]; o[1] = v[1]; }
    static __device__ __forceinline__ f64x2 pack(const double (&o)[2]) { return f64x2{o[0], o[1]}; }
    static __device__ __forceinline__ f64x2 zero() { return f64x2{0.0, 0.0}; }
};

// streaming load of one 16-byte vector; NT = non-temporal (X is read once per pass, never reused)
template <bool NT, typename V>
__device__ __forceinline__ V stream_load(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT, typename V>
__device__ __forceinline__ void stream_store(V* p, const V& v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// A small job that rides in the grid of a streaming pass instead of costing a launch of its own: the Gram row
// T T[t,:]^T of the topic whose row dots the pass takes (k_tgram below describes it).  T[t,:] is final when the
// pass starts and the result is first read by the k_wcol after the pass, so the two have no order to keep.
// The first `nblocks` workgroups of the grid do the job, the others the pass.
struct TgramJob {
    const double* T; i64 ldt; int d, k, t;
    double* Ttpart; const double* tpart; int ntb, nsplit, finish, sweep;
    KParams p; DevState* st;
    int nblocks;          // k * nsplit, or 0: no job
};
__device__ __forceinline__ void tgram_block(const double* __restrict__ T, i64 ldt, int d, int k, int t, int l, int split,
                                            int ns, double* __restrict__ Ttpart, const double* __restrict__ tpart,
                                            int nblk, int finish, int sweep, const KParams& p, DevState* st,
                                            double* scratch) {
    const int chunk = (d + ns - 1) / ns;
    const int j0 = split * chunk, j1 = min(d, j0 + chunk);
    double acc = 0.0;
    for (int j = j0 + threadIdx.x; j < j1; j += 256) acc = fma(T[(i64)l * ldt + j], T[(i64)t * ldt + j], acc);
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) Ttpart[split * k + l] = acc;
    if (finish && l == 0 && split == 0) {
        double ps = 0.0;
        ps = ordered_sum<4>(tpart, 1, threadIdx.x, nblk, 256);
        ps = block_sum(ps, scratch);
        if (threadIdx.x == 0) {
            const int mode = st->tmode;
            st->nt1 = (mode == 0) ? ps : 1.0;
            st->sumT = ps;
            if (!(ps > 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0) {
                st->halt = HALT_EVENT_RESET_T; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t;
            }
        }
    }
}

// Block = 4 waves = 4 ADJACENT column panels (one per wave, 64 lanes * 16 B each) x one row block.
// Every wave walks all rows of the block, U rows in flight; the 4 row-dot partials of a row meet in
// LDS slots [wave][row] and are added in a fixed order at the end (Ypart has one slice per 4 panels);
// column sums belong to one wave each and go straight to Zpart.
// UPD = 0: X is read only.  UPD = 1: R <- R - a b^T.  UPD = 2: R <- R - a b^T - a2 (b2 - b2sub)^T, the two pending
// rank-one terms of one topic step of the explicit-residual schedule (dw_{t-1} t_{t-1}^T and w_t dt_t^T; b2sub = the
// T row before its update, so dt is formed in registers).  a, a2 come through LDS, b, b2 live in registers.
template <typename SX, bool DO_Y, bool DO_Z, int UPD, int U, bool NT, bool RS>
__global__ __launch_bounds__(256) void k_pass(typename std::conditional<(UPD > 0), SX, const SX>::type* __restrict__ X,
                                              i64 ldx, int n, int ncols,
                                              const double* __restrict__ trow, const double* __restrict__ wcol,
                                              double* __restrict__ Ypart, double* __restrict__ Zpart, i64 ldz,
                                              int rpb, int npg, const double* __restrict__ avec,
                                              const double* __restrict__ bvec, const double* __restrict__ avec2,
                                              const double* __restrict__ bvec2, const double* __restrict__ bsub2,
                                              const DevState* __restrict__ st, const TgramJob job, int nrb_il_rot) {
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int VN = XV::N;
    constexpr int PW = 64 * VN;          // columns per wave
    if (st->halt) return;
    if ((int)blockIdx.x < job.nblocks) {
        __shared__ double jscratch[40];
        tgram_block(job.T, job.ldt, job.d, job.k, job.t, (int)blockIdx.x % job.k, (int)blockIdx.x / job.k, job.nsplit,
                    job.Ttpart, job.tpart, job.ntb, job.finish, job.sweep, job.p, job.st, jscratch);
        return;
    }
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx.x (which XCD block 0 gets is not fixed).  `rot` (top bits of
    // the argument; RRI_PASS_ROT, diagnostics) rotates the tile a workgroup takes inside its group of 8: another XCD for
    // every tile, nothing else changed.
    const int nrb_il = nrb_il_rot & 0x07ffffff;
    const int rot = (int)((unsigned)nrb_il_rot >> 27);
    int bid = (int)blockIdx.x - job.nblocks;
    if (rot != 0 && (bid | 7) < (int)gridDim.x - job.nblocks) bid = (bid & ~7) | ((bid + rot) & 7);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ysh = reinterpret_cast<double*>(smem);            // [4 waves][rpb]
    double* wsh = ysh + 4 * rpb;                              // [rpb]
    double* ash = wsh + rpb;                                  // [rpb], only with UPD
    double* ash2 = ash + rpb;                                 // [rpb], only with UPD == 2
    static_assert(!RS || U == 8, "the LDS row-sum path reduces 8 rows at a time");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* tile = wsh + (1 + UPD) * rpb + wave * (8 * 72);   // [4][8*72] private row-sum tiles (RS)
    const int pg = bid % npg;
    const int rb = bid / npg;
    // Rows of this block, local index li = q U + u  ->  global row.  Interleaved (nrb_il > 0): chunk q of U rows is
    // chunk number q nrb + rb of the matrix, so the blocks that run at one time walk ONE contiguous window of X, as a
    // linear stream does (worth 2 % at 20000 x 5000, nothing at C3: the host picks).  Contiguous (nrb_il == 0): rows
    // rb rpb .. rb rpb + rpb - 1.
    auto grow = [&](int li) -> int {
        return nrb_il > 0 ? ((li / U) * nrb_il + rb) * U + (li % U) : rb * rpb + li;
    };
    const int col = (pg * 4 + wave) * PW + lane * VN;
    if (DO_Z || UPD > 0) {
        for (int i = threadIdx.x; i < rpb; i += 256) {
            const int g = grow(i);
            if (DO_Z) wsh[i] = g < n ? wcol[g] : 0.0;
            if (UPD > 0) ash[i] = g < n ? avec[g] : 0.0;
            if (UPD > 1) ash2[i] = g < n ? avec2[g] : 0.0;
        }
        __syncthreads();
    }
    const bool ok = col < ncols;
    const bool wave_has_cols = (pg * 4 + wave) * PW < ncols;   // wave-uniform
    double tv[VN], zacc[VN], bv[VN], bv2[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) {
        zacc[e] = 0.0;
        tv[e] = (DO_Y && ok) ? trow[col + e] : 0.0;
        bv[e] = (UPD > 0 && ok) ? bvec[col + e] : 0.0;
        bv2[e] = (UPD > 1 && ok) ? bvec2[col + e] - bsub2[col + e] : 0.0;
    }
    if (wave_has_cols) {
        for (int l0 = 0; l0 < rpb; l0 += U) {
            const int r = grow(l0);                  // the U rows of a chunk are consecutive
            if (r >= n) break;                       // later chunks lie further down still
            V x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u;
                x[u] = XV::zero();
                if (rr < n && ok) x[u] = stream_load<NT>(reinterpret_cast<const V*>(X + (i64)rr * ldx + col));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u;
                double wv = 0.0, na = 0.0, na2 = 0.0;
                if (DO_Z && rr < n) wv = wsh[l0 + u];
                if (UPD > 0 && rr < n) na = -ash[l0 + u];
                if (UPD > 1 && rr < n) na2 = -ash2[l0 + u];
                double xe[VN];
                XV::unpack(x[u], xe);
                if constexpr (UPD > 0) {
#pragma unroll
                    for (int e = 0; e < VN; ++e) {
                        xe[e] = fma(na, bv[e], xe[e]);
                        if (UPD > 1) xe[e] = fma(na2, bv2[e], xe[e]);
                    }
                    const V rounded = XV::pack(xe);
                    if (rr < n && ok) stream_store<NT>(reinterpret_cast<V*>(X + (i64)rr * ldx + col), rounded);
                    // the stored residual is what later passes read: continue with the ROUNDED values
                    if constexpr (sizeof(SX) == 4) XV::unpack(rounded, xe);
                }
                double yp = 0.0;
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    if (DO_Y) yp = fma(xe[e], tv[e], yp);
                    if (DO_Z) zacc[e] = fma(wv, xe[e], zacc[e]);
                }
                if constexpr (DO_Y && RS) wave_rowsum8_park(tile, u, lane, yp);
                else if (DO_Y) {
                    // the same six DPP steps as wave_sum, the total taken where they leave it (lane 63) and stored from there
                    const double tot = wave_sum_lane63<double>(yp);
                    if (lane == 63) ysh[wave * rpb + l0 + u] = tot;
                }
            }
            if constexpr (DO_Y && RS) {
                const double tot = wave_rowsum8_finish(tile, lane);
                if ((lane & 7) == 0) ysh[wave * rpb + l0 + (lane >> 3)] = tot;
            }
        }
        if (DO_Z && ok) {
#pragma unroll
            for (int e = 0; e < VN; ++e) Zpart[(i64)rb * ldz + col + e] = zacc[e];
        }
    } else if (DO_Y) {
        for (int i = lane; i < rpb; i += 64) ysh[wave * rpb + i] = 0.0;
    }
    if (DO_Y) {
        __syncthreads();
        for (int i = threadIdx.x; i < rpb; i += 256) {
            const int g = grow(i);
            if (g < n) Ypart[(i64)pg * n + g] = (ysh[i] + ysh[rpb + i]) + (ysh[2 * rpb + i] + ysh[3 * rpb + i]);
        }
    }
}

// =========================================================================================
// k_pass_dma: the read-only pass (UPD = 0) with its rows staged through a per-wave LDS ring that LDS-DMA fills
// (global_load_lds_dwordx4: HBM -> LDS with no VGPR destination), round 4.  Same geometry, same arithmetic in the same order
// as k_pass<.., U = 8, RS = true> -- the sums are bit-identical -- but the loads of the next two chunks are in flight while a
// chunk is worked on, at no cost in registers: S = 16 slots of 1 KiB (one row of the wave's panel each), C = 8 rows per chunk,
// counted s_waitcnt vmcnt(8) instead of a drain.  OPT-IN (RRI_PASS_DMA=1), not the default: in the probe, beside k_pass on the same
// buffer in one process (tools/lds_dma_probe.hip, profiles/r04_lds_dma_probe*.log), it was +1.5 ... +6 % at 100000 x 10000; inside the
// library's sweep, engines made alternately in one process (profiles/r04_pass_dma_ab.log), it ranged from +3 % to -9 % by box,
// process and workgroup geometry -- nowhere near the 0.61 ms that would have justified inline asm in the roofline kernel.  Kept as
// the record of the attempt and as a tested, bit-identical alternative.
// What it takes:
//   * the LDS-DMA and its waits are inline asm (M0 carries the LDS base of a wave-instruction: saved and restored inside the
//     statement; hipcc does not count asm loads, so the waits are ours);
//   * ordinary loads issued before the loop (the T slice) must be CONSUMED before the first DMA: hipcc defers their wait to
//     the first use, which sits inside the loop, and that vmcnt(0) would drain the ring on every iteration;
//   * the 8 row dots of a chunk go through an LDS tile (wave_rowsum8_*) laid over the chunk's own ring slots, which are free
//     between the reads of the chunk and their refill -- with six DPP steps per row the kernel was bound by its vector ALU.
// LDS per workgroup: 64 KiB of ring + (4 + 1) rpb doubles; one or two workgroups per CU, which is enough: the ring, not the
// occupancy, keeps the bytes in flight.
// =========================================================================================
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
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int PASS_DMA_SLOTS = 16, PASS_DMA_CHUNK = 8;
template <typename SX, bool DO_Y, bool DO_Z, bool NT>
__global__ __launch_bounds__(256) void k_pass_dma(const SX* __restrict__ X, i64 ldx, int n, int ncols,
                                                  const double* __restrict__ trow, const double* __restrict__ wcol,
                                                  double* __restrict__ Ypart, double* __restrict__ Zpart, i64 ldz, int rpb,
                                                  int npg, const DevState* __restrict__ st, const TgramJob job, int nrb_il,
                                                  int sub, int nrb) {
    // sub: a workgroup walks `sub` consecutive row blocks (rb0 .. rb0 + sub - 1, contiguous rows; sub = 1 with interleaved chunks) as
    // one stream and leaves one row of Zpart per row block, as k_pass does: the ring stays full across the blocks
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int VN = XV::N;
    constexpr int PW = 64 * VN;
    constexpr int S = PASS_DMA_SLOTS, C = PASS_DMA_CHUNK, K = S / C;
    if (st->halt) return;
    if ((int)blockIdx.x < job.nblocks) {
        __shared__ double jscratch[40];
        tgram_block(job.T, job.ldt, job.d, job.k, job.t, (int)blockIdx.x % job.k, (int)blockIdx.x / job.k, job.nsplit,
                    job.Ttpart, job.tpart, job.ntb, job.finish, job.sweep, job.p, job.st, jscratch);
        return;
    }
    const int bid = (int)blockIdx.x - job.nblocks;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rows = sub * rpb;                                           // local rows of this workgroup
    V* ring = reinterpret_cast<V*>(smem);                                 // [4 waves][S][64 lanes]
    double* ysh = reinterpret_cast<double*>(smem + 4 * S * 1024);         // [4][rows]
    double* wsh = ysh + 4 * rows;                                         // [rows]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pg = bid % npg, rb = (bid / npg) * sub;                     // rb: the first row block of the group
    auto grow = [&](int li) -> int { return nrb_il > 0 ? ((li / C) * nrb_il + rb) * C + (li % C) : rb * rpb + li; };
    if (DO_Z) {
        for (int i = threadIdx.x; i < rows; i += 256) {
            const int g = grow(i);
            wsh[i] = g < n ? wcol[g] : 0.0;
        }
        __syncthreads();
    }
    const int col = (pg * 4 + wave) * PW + lane * VN;
    if ((pg * 4 + wave) * PW >= ncols) {                // wave-uniform: a panel beyond the matrix
        if (DO_Y)
            for (int i = lane; i < rows; i += 64) ysh[wave * rows + i] = 0.0;
    } else {
        const bool ok = col < ncols;
        const int colc = ok ? col : 0;                  // lanes beyond the last column load (and ignore) column 0
        double tv[VN], zacc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            zacc[e] = 0.0;
            tv[e] = (DO_Y && ok) ? trow[col + e] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) asm volatile("" ::"v"(tv[e]));      // see the header: consumed before the first DMA
        V* myring = ring + (size_t)wave * S * 64;
        // LDS byte offset of the wave's ring (address space 3 pointers are 32-bit offsets into the workgroup's allocation)
        const unsigned ring_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)wave * S * 1024u;
        int nchunks = 0;                                // chunks of this block that start inside the matrix
        for (int l0 = 0; l0 < rows; l0 += C) { if (grow(l0) < n) nchunks = l0 / C + 1; else break; }
        const int cpb = rpb / C;                        // chunks per row block (rpb is a multiple of C)
        auto issue = [&](int q) {                       // chunk q of the block into the ring slots (q mod K) C ...
            const int r = grow(q * C);
#pragma unroll
            for (int u = 0; u < C; ++u) {
                const int rr = min(r + u, n - 1);       // rows beyond the matrix: the last row again (ignored below)
                glds16<NT>(X + (i64)rr * ldx + colc, ring_base + (unsigned)(((q % K) * C + u) * 1024));
            }
        };
        for (int q = 0; q < K && q < nchunks; ++q) issue(q);
        for (int q = 0; q < nchunks; ++q) {
            // chunk q must have landed: K - 1 chunks of loads were issued after it (fewer near the end: wait for all)
            if (q + K <= nchunks) wait_vmcnt<C * (K - 1)>();
            else wait_vmcnt<0>();
            const int l0 = q * C, r = grow(l0);
            V x[C];
#pragma unroll
            for (int u = 0; u < C; ++u) x[u] = myring[((q % K) * C + u) * 64 + lane];
            double* tile = reinterpret_cast<double*>(myring + ((q % K) * C) * 64);     // 8 x 72 doubles of the 8 KiB just read
            asm volatile("" ::: "memory");             // the tile aliases the ring: no store to it may move above the reads
#pragma unroll
            for (int u = 0; u < C; ++u) {
                const int rr = r + u;
                double wv = 0.0;
                if (DO_Z && rr < n) wv = wsh[l0 + u];
                double xe[VN];
                XV::unpack(x[u], xe);
                double yp = 0.0;
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    if (DO_Y) yp = fma(xe[e], tv[e], yp);
                    if (DO_Z) zacc[e] = fma(wv, xe[e], zacc[e]);
                }
                if (!(rr < n && ok)) yp = 0.0;          // (k_pass zeroes the loaded vector there: the same zero partial)
                if constexpr (DO_Y) wave_rowsum8_park(tile, u, lane, yp);
            }
            if constexpr (DO_Y) {
                const double tot = wave_rowsum8_finish(tile, lane);
                if ((lane & 7) == 0) ysh[wave * rows + l0 + (lane >> 3)] = tot;
            }
            // the slots are free once the reads above have returned (their values were consumed): refill them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (q + K < nchunks) issue(q + K);
            // the last chunk of a row block (or of the workgroup): its column sums are complete
            if (DO_Z && ((q + 1) % cpb == 0 || q + 1 == nchunks)) {
                const int rbq = rb + q / cpb;
                if (ok && rbq < nrb) {
#pragma unroll
                    for (int e = 0; e < VN; ++e) Zpart[(i64)rbq * ldz + col + e] = zacc[e];
                }
#pragma unroll
                for (int e = 0; e < VN; ++e) zacc[e] = 0.0;
            }
        }
        // row blocks of the group that lie wholly beyond the matrix cannot occur (nrb = ceil(n / rpb)), but a group may end early
        if (DO_Y)
            for (int i = nchunks * C + lane; i < rows; i += 64) ysh[wave * rows + i] = 0.0;
    }
    if (DO_Y) {
        __syncthreads();
        for (int i = threadIdx.x; i < rows; i += 256) {
            const int g = grow(i);
            if (g < n) Ypart[(i64)pg * n + g] = (ysh[i] + ysh[rows + i]) + (ysh[2 * rows + i] + ysh[3 * rows + i]);
        }
    }
}

// =========================================================================================
// k_colsums: column sums of X against NV row-vectors at once (X^T Q for the randomized SVD behind NNDSVD,
// initialization.py:105): the geometry of k_pass -- 4 waves = 4 adjacent 1 KiB panels x a row block, 4 rows in
// flight -- with NV accumulator sets per lane, so X is read once per NV vectors instead of once per vector.
// Qt: the vectors as rows (NV x n, stride ldq).  Zmulti[v][row block][column], stride ldz.
// =========================================================================================
template <typename SX, int NV>
__global__ __launch_bounds__(256) void k_colsums(const SX* __restrict__ X, i64 ldx, int n, int ncols,
                                                 const double* __restrict__ Qt, i64 ldq, int nv,
                                                 double* __restrict__ Zmulti, i64 ldz, int rpb, int npg, int nrb) {
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int VN = XV::N;
    constexpr int PW = 64 * VN;
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* wsh = reinterpret_cast<double*>(smem);   // [NV][rpb]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pg = blockIdx.x % npg, rb = blockIdx.x / npg;
    const int row0 = rb * rpb, row1 = min(n, row0 + rpb);
    const int col = (pg * 4 + wave) * PW + lane * VN;
    for (int i = threadIdx.x; i < (row1 - row0) * NV; i += 256) {
        const int v = i / (row1 - row0), r = i - v * (row1 - row0);
        wsh[v * rpb + r] = v < nv ? Qt[(i64)v * ldq + row0 + r] : 0.0;
    }
    __syncthreads();
    if ((pg * 4 + wave) * PW >= ncols) return;       // wave-uniform
    const bool ok = col < ncols;
    double acc[NV][VN];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[v][e] = 0.0;
    for (int r = row0; r < row1; r += U) {
        V x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = (r + u < row1 && ok) ? stream_load<true>(reinterpret_cast<const V*>(X + (i64)(r + u) * ldx + col)) : XV::zero();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u >= row1) break;
            double xe[VN];
            XV::unpack(x[u], xe);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double wv = wsh[v * rpb + r + u - row0];
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[v][e] = fma(wv, xe[e], acc[v][e]);
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < VN; ++e) Zmulti[((i64)v * nrb + rb) * ldz + col + e] = acc[v][e];
    }
}

// =========================================================================================
// Preprocessing of the resident X (matrixops.py:124-179: tf-idf and row normalisation), SURVEY 8f rank 3.
//   k_col_count  df_j = #{i : X_ij > 0}, geometry of k_colsums (partials per row block, k_reduce adds them)
//   k_row_inverse  xs_i = sum of the row-dot panels + spacing(1); inv_i = 1 / xs_i; rows with xs_i < 1e-10 are flagged
//                  (inv_i = -1): they become uniform 1/d (normalize's zero_sum_fix)
//   k_scale2d    X_ij <- (X_ij * s_j) * inv_i in place, one linear stream (the fastest read-modify-write pattern)
// =========================================================================================
template <typename SX>
__global__ __launch_bounds__(256) void k_col_count(const SX* __restrict__ X, i64 ldx, int n, int ncols,
                                                   double* __restrict__ Zpart, i64 ldz, int rpb, int npg) {
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int VN = XV::N;
    constexpr int PW = 64 * VN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = blockIdx.x % npg, rb = blockIdx.x / npg;
    const int row0 = rb * rpb, row1 = min(n, row0 + rpb);
    const int col = (pg * 4 + wave) * PW + lane * VN;
    if (col >= ncols) return;
    double acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = 0.0;
    for (int r = row0; r < row1; r += 4) {
        V x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            x[u] = (r + u < row1) ? stream_load<true>(reinterpret_cast<const V*>(X + (i64)(r + u) * ldx + col)) : XV::zero();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double xe[VN];
            XV::unpack(x[u], xe);
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[e] += xe[e] > 0.0 ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) Zpart[(i64)rb * ldz + col + e] = acc[e];
}

__global__ __launch_bounds__(256) void k_row_inverse(const double* __restrict__ Ypart, int npanels, int n,
                                                     double* __restrict__ inv) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double xs = 0.0;
    xs = ordered_sum<8>(Ypart + i, n, 0, npanels, 1);
    xs += 2.220446049250313e-16;                      // np.spacing(1), matrixops.py:140
    inv[i] = xs < 1e-10 ? -1.0 : 1.0 / xs;
}

template <typename SX>
__global__ __launch_bounds__(256) void k_scale2d(SX* __restrict__ X, i64 ldx, i64 n, int d, const double* __restrict__ s,
                                                 const double* __restrict__ inv) {
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int VN = XV::N;
    const i64 vpr = ldx / VN, total = n * vpr;         // ldx is a multiple of the vector width
    const double uniform = 1.0 / (double)d;
    for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total; v += (i64)gridDim.x * 256) {
        const i64 i = v / vpr;
        const int j = (int)(v - i * vpr) * VN;
        V* p = reinterpret_cast<V*>(X + i * ldx + j);
        double xe[VN];
        XV::unpack(__builtin_nontemporal_load(p), xe);
        const double r = inv ? inv[i] : 1.0;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            if (j + e >= d) { xe[e] = 0.0; continue; }                       // pad columns stay zero
            const double t = s ? xe[e] * s[j + e] : xe[e];                    // X * idf               (:172)
            xe[e] = r < 0.0 ? uniform : r * t;                                // d * X, or the zero-sum fix (:143-147)
        }
        __builtin_nontemporal_store(XV::pack(xe), p);
    }
}

// =========================================================================================
// W is stored k-major on the device (Wt: k x ldw, ldw >= n): column t of W is the contiguous
// row Wt[t,:], so the pass reads the active column coalesced and k_wcol needs no LDS staging.
//
// k_wcol: W-column update of topic t (UPDATE) and the Gram row w_tn^T W / ||w_tn||^2 of the NEXT
// topic tn (CARRY), one thread per row of W.  Gpart[b][0..k) = sum_i wn_i W[i,:], [k] = sum wn_i^2,
// [k+1] = sum_i W[i,t] (new).  T T[t]^T arrives as nsplit partial vectors from k_tgram.
// =========================================================================================
constexpr int WCOL_TILES = 1;   // 64-row tiles per k_wcol block (1 = most blocks in flight)
constexpr int GRAM_SLICES = RRI_GRAM_SLICES;  // k_reduce sums the Gpart rows in this many slices; consumers add the slices

template <bool UPDATE, bool CARRY>
__global__ __launch_bounds__(256) void k_wcol(double* __restrict__ Wt, i64 ldw, int n, int k, int t, int tn,
                                              const double* __restrict__ Ypart, int npanels,
                                              const double* __restrict__ Ttpart, int nsplit,
                                              double* __restrict__ Gpart, double* __restrict__ xyp, int sweep,
                                              KParams p, DevState* st) {
    // Block = TILES tiles of 64 rows of W (lane = row); its 4 waves split the k columns (wave w takes
    // l = w, w+4, ...): four times the waves of a row-per-thread layout, dependent chains a quarter as
    // long.  Gram partials of the tiles add up in LDS (one owner wave per column: fixed order).
    if (st->halt) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tts = reinterpret_cast<double*>(smem);   // [k]
    double* gsh = tts + k;                           // [k+2] Gram partial of the block
    double* dsh = gsh + k + 2;                       // [4][64] partial W.Tt per wave
    double* wsh = dsh + 256;                         // [64] new column entries
    double cden = 0.0;
    int mode = 0;
    // (requesting the first tile's operands before the reduction of T T[t]^T below, so that the two round trips to L2
    // overlap, was tried: +48 VGPRs, 3 instead of 4 waves per SIMD, 18.5 -> 24.8 us at C3 and no gain at 10000 x 1000)
    for (int l = tid; l < k + 2; l += 256) gsh[l] = 0.0;
    if (UPDATE) {
        for (int l = tid; l < k; l += 256) {
            tts[l] = ordered_sum<8>(Ttpart + l, k, 0, nsplit, 1);                               // entry t holds ||T[t,:]||^2 until zeroed below
        }
        __syncthreads();
        cden = tts[t] + p.reg_w_l2;                   // denom = nt + reg_w_l2 (nmf.py:465)
        __syncthreads();
        if (tid == 0) tts[t] = 0.0;                   // Tt[t] = 0 (nmf.py:732)
        if (!(cden > 0.0)) {
            // scalar c<=0 with s=None (optimization.py:60-67): entries jump to ub, or unbounded
            if (p.has_wrs && p.w_row_sum != 0.0) mode = 1;
            else {
                if (blockIdx.x == 0 && tid == 0) {
                    st->halt = HALT_ERR_UNBOUNDED; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t;
                }
                return;
            }
        }
    }
    __syncthreads();
    constexpr int CH = 16;   // columns per batch of loads per wave (all of a wave's columns when k <= 64)
    constexpr int TILES = WCOL_TILES;
    for (int tile = 0; tile < TILES; ++tile) {
        const i64 i = ((i64)blockIdx.x * TILES + tile) * 64 + lane;
        if (((i64)blockIdx.x * TILES + tile) * 64 >= n) break;   // block-uniform
        const bool valid = i < n;
        const double wn = (CARRY && valid) ? Wt[(i64)tn * ldw + i] : 0.0;
        double y = 0.0;
        if (UPDATE && wave == 0 && valid) y = ordered_sum<8>(Ypart + i, n, 0, npanels, 1);
        double dotv = 0.0;
        for (int l0 = wave; l0 < k; l0 += 4 * CH) {
            double wl[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int l = l0 + 4 * q;
                wl[q] = (l < k && valid && !(UPDATE && l == t)) ? Wt[(i64)l * ldw + i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int l = l0 + 4 * q;
                if (l < k) {   // wave-uniform
                    if (UPDATE) dotv = fma(wl[q], tts[l], dotv);
                    if (CARRY && !(UPDATE && l == t)) {
                        const double g = wave_sum<double>(wn * wl[q]);
                        if (lane == 0) gsh[l] += g;   // column l belongs to this wave alone
                    }
                }
            }
        }
        double wnew = 0.0;
        if (UPDATE) {
            dsh[wave * 64 + lane] = dotv;
            __syncthreads();
            if (wave == 0) {
                const double dot = (dsh[lane] + dsh[64 + lane]) + (dsh[128 + lane] + dsh[192 + lane]);
                if (valid) {
                    const double numer = (y - dot) - p.reg_w_l1;
                    if (mode == 0) wnew = fmax(numer, 0.0) / (cden + p.eps);
                    else wnew = (-numer + cden < 0.0) ? p.w_row_sum : 0.0;
                    Wt[(i64)t * ldw + i] = wnew;
                }
                wsh[lane] = wnew;
            }
            __syncthreads();
            wnew = wsh[lane];
        }
        if (wave == 0) {
            const double nwp = wave_sum<double>(wn * wn);
            const double swp = wave_sum<double>(wnew);
            if (lane == 0) { gsh[k] += nwp; gsh[k + 1] += swp; }
            if (UPDATE) {   // <w_t, X t_t> of this block's rows: the cross term of the objective, for free
                const double xy = wave_sum<double>(wnew * y);
                if (lane == 0) xyp[(i64)blockIdx.x * TILES + tile] = xy;
            }
        } else if (wave == 1 && UPDATE && CARRY) {
            const double gt = wave_sum<double>(wn * wnew);   // Gram entry against the NEW column t
            if (lane == 0) gsh[t] += gt;
        }
        if (UPDATE) __syncthreads();   // dsh / wsh are reused by the next tile
    }
    __syncthreads();
    double* gp = Gpart + (i64)blockIdx.x * (k + 2);
    for (int l = tid; l < k + 2; l += 256) gp[l] = gsh[l];
}

// =========================================================================================
// k_wsweep_rows: the W half of EVERY topic of a sweep in one launch, for runs with T fixed (the fold-in of new rows,
// sklearn_interface.py:327-333; the w_row refit, nmf.py:531-539).  With T fixed the update of W[i, t] (nmf.py:464-469, 728-734)
//     numer = (X T^T)[i, t] - sum_{l != t} W[i, l] (T T^T)[l, t] - reg_w_l1 ,   denom = ||T[t,:]||^2 + reg_w_l2
// involves row i of W alone: Q = X T^T (k_xtt) and G = T T^T (k_gram) are constants of the run, taken once per T, and a lane
// walks the k topics of its row with the row and G in LDS -- k^2 fused multiply-adds per row, no pass over X, one launch where
// the launch-per-topic schedule takes three per topic (k_tgram, k_wcol, k_check_wcol).  Same branches of qf_min as k_wcol.
// What couples the rows is the column check after every update (nmf.py:471-476, 793-816): the kernel leaves the column sums as
// one partial per workgroup and topic, k_wsweep_verdict takes the verdicts in topic order after the launch, and when a
// column asks for a reset at topic t* (rare), k_wsweep_repair puts the columns after t* back to what they were before the
// sweep (Wprev, written here on the way in): the state in which the launch-per-topic schedule halts.  The cross terms
// <w_t, X t_t> of the objective go to xyp as k_wcol leaves them (one partial per 64 rows).
// =========================================================================================
// Four topics at a time: the terms of the four dots that come from columns OUTSIDE the block share one read of W[i, l] and two
// 16-byte reads of G[l, t .. t+3] (three LDS reads for four multiply-adds where a topic on its own takes eight); the six terms
// inside the block follow in registers, topic by topic.  (X T^T)[i, t .. t+3] of the NEXT block is requested before this one is
// worked on.  A dot is therefore summed as (columns before the block, ascending) + (columns after it) + (the block's others).
__host__ __device__ constexpr int wsweep_kp(int k) { return (k + 3) & ~3; }
__host__ __device__ constexpr size_t wsweep_lds_bytes(int k) { return ((size_t)k * wsweep_kp(k) + 64 * (size_t)k) * sizeof(double); }

__global__ __launch_bounds__(64) void k_wsweep_rows(double* __restrict__ Wt, double* __restrict__ Wprev, i64 ldw, int n, int k,
                                                    int t0, const double* __restrict__ Qt, const double* __restrict__ G,
                                                    double* __restrict__ colsum, int nwb, double* __restrict__ xyp,
                                                    i64 xy_stride, KParams p, const DevState* __restrict__ st) {
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int kp = wsweep_kp(k);
    double* gsh = reinterpret_cast<double*>(smem);   // [k][kp]  (symmetric to the bit: k_gram writes both halves from one sum)
    double* wsh = gsh + (size_t)k * kp;              // [k][64] the rows' entries of W, current
    const int lane = threadIdx.x;
    const i64 i = (i64)blockIdx.x * 64 + lane;
    const bool valid = i < n;
    for (int q = lane; q < k * kp; q += 64) {
        const int l = q / kp, c = q - l * kp;
        gsh[q] = c < k ? G[(i64)l * k + c] : 0.0;
    }
    constexpr int CH = 8;
    for (int l0 = 0; l0 < k; l0 += CH) {
        double wv[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) wv[q] = (l0 + q < k && valid) ? Wt[(i64)(l0 + q) * ldw + i] : 0.0;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int l = l0 + q;
            if (l < k) {
                wsh[l * 64 + lane] = wv[q];
                if (valid) Wprev[(i64)l * ldw + i] = wv[q];
            }
        }
    }
    const int tb0 = t0 & ~3;
    double yv[4], yn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) yv[j] = (tb0 + j < k && valid) ? Qt[(i64)(tb0 + j) * ldw + i] : 0.0;
    __syncthreads();
    for (int tb = tb0; tb < k; tb += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) yn[j] = (tb + 4 + j < k && valid) ? Qt[(i64)(tb + 4 + j) * ldw + i] : 0.0;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        auto outside = [&](int la, int lb) {
#pragma unroll 4
            for (int l = la; l < lb; ++l) {
                const double wl = wsh[l * 64 + lane];
                const f64x2 g01 = *reinterpret_cast<const f64x2*>(gsh + (size_t)l * kp + tb);
                const f64x2 g23 = *reinterpret_cast<const f64x2*>(gsh + (size_t)l * kp + tb + 2);
                acc[0] = fma(wl, g01[0], acc[0]);
                acc[1] = fma(wl, g01[1], acc[1]);
                acc[2] = fma(wl, g23[0], acc[2]);
                acc[3] = fma(wl, g23[1], acc[3]);
            }
        };
        outside(0, tb);
        outside(min(tb + 4, k), k);
        double wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = tb + j < k ? wsh[(tb + j) * 64 + lane] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = tb + j;
            if (t >= k || t < t0) continue;           // (wave-uniform)
            const double* gr = gsh + (size_t)t * kp;
            const double cden = gr[t] + p.reg_w_l2;   // denom = nt + reg_w_l2 (nmf.py:465)
            int mode = 0;
            if (!(cden > 0.0)) {                      // scalar c <= 0 with s = None (optimization.py:60-67): entries jump to ub, or
                if (p.has_wrs && p.w_row_sum != 0.0) mode = 1;
                else continue;                        // unbounded -- k_wsweep_verdict reports it at this topic; the column stays
            }
            double dot = acc[j];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2)
                if (j2 != j && tb + j2 < k) dot = fma(wb[j2], gr[tb + j2], dot);     // Tt[t] = 0 (nmf.py:732): j2 == j left out
            const double y = yv[j];
            const double numer = (y - dot) - p.reg_w_l1;
            double wnew = 0.0;
            if (valid) {
                if (mode == 0) wnew = fmax(numer, 0.0) / (cden + p.eps);
                else wnew = (-numer + cden < 0.0) ? p.w_row_sum : 0.0;
                Wt[(i64)t * ldw + i] = wnew;
            }
            wb[j] = wnew;
            wsh[t * 64 + lane] = wnew;
            const double sw = wave_sum<double>(wnew);
            const double xy = wave_sum<double>(wnew * y);
            if (lane == 0) {
                colsum[(i64)t * nwb + blockIdx.x] = sw;
                xyp[(i64)t * xy_stride + blockIdx.x] = xy;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = yn[j];
    }
}

// the column checks of a k_wsweep_rows launch, in topic order: stage bit 0 -- sums[t] = sum over the workgroups' partials (a
// wave per topic, fixed order); bit 1 -- the verdicts on sums (row-sharded runs all-reduce sums in between), those of k_wcol
// (unbounded: a non-positive denominator with no bound to jump to) and k_check_wcol, halting at the FIRST topic that fails
// with the position the launch-per-topic schedule reports.  A reset event also leaves pad0 = t* + 1 for k_wsweep_repair.
__global__ __launch_bounds__(1024) void k_wsweep_verdict(const double* __restrict__ colsum, int nwb, double* __restrict__ sums,
                                                         const double* __restrict__ G, int k, int t0, int sweep, int stage,
                                                         KParams p, DevState* st) {
    if (st->halt) return;
    __shared__ double ssum[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (stage & 1) {
        for (int t = t0 + wave; t < k; t += 16) {
            double a = ordered_sum<8>(colsum + (i64)t * nwb, 1, lane, nwb, 64);
            a = wave_sum<double>(a);
            if (lane == 0) { sums[t] = a; ssum[t] = a; }
        }
    }
    if (!(stage & 2)) return;
    if (!(stage & 1))
        for (int t = t0 + (int)threadIdx.x; t < k; t += 1024) ssum[t] = sums[t];
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int t = t0; t < k; ++t) {
        const double cden = G[(i64)t * k + t] + p.reg_w_l2;
        if (!(cden > 0.0) && !(p.has_wrs && p.w_row_sum != 0.0)) {
            st->halt = HALT_ERR_UNBOUNDED; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t;
            return;
        }
        const double a = ssum[t];
        const bool ev = (a <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
        const bool err = !ev && !(a > 0.0);
        if (ev || err) {
            st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
            st->halt_topic = t;
            st->halt_sweep = t + 1 == k ? sweep + 1 : sweep;     // position of the NEXT step, where a resumed run continues
            st->halt_pos = t + 1 == k ? 0 : t + 1;
            if (ev) st->pad0 = t + 1;
            return;
        }
    }
}

// after a reset event of k_wsweep_verdict at topic t*: columns t* + 1 .. k - 1 of W as they were before the sweep
__global__ __launch_bounds__(256) void k_wsweep_repair(double* __restrict__ Wt, const double* __restrict__ Wprev, i64 ldw, int n,
                                                       int k, const DevState* __restrict__ st) {
    const int from = st->pad0;
    if (from <= 0 || st->halt != HALT_EVENT_RESET_W) return;
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int l = from; l < k; ++l) Wt[(i64)l * ldw + i] = Wprev[(i64)l * ldw + i];
}

// =========================================================================================
// k_wcol_resid: W-column update of topic t in the explicit-residual schedule ("explicit-residual variant"):
//     numer_W = R t' + w_t ||t'||^2 - reg_w_l1      (R: the stored residual, which already contains w_t t'^T)
// y = R t' arrives as the row-dot panels of the pass; the new column goes to Wt[t,:], dw = new - old stays in `dw`
// as the row factor of the rank-one term  - dw t'^T  that the NEXT pass folds into R.  One thread per row.
// Gpart[b] (k+2 entries per block of 256 rows): [t] = <dw, w_tn> -- the coefficient with which that pending term enters
// the column sums of the next topic tn --, [k] = ||w_tn||^2, [k+1] = sum of the new column, every other entry 0.
// UPDATE = false (prologue): only [k] = ||w_tn||^2.
// =========================================================================================
template <bool UPDATE>
__global__ __launch_bounds__(256) void k_wcol_resid(double* __restrict__ Wt, i64 ldw, int n, int k, int t, int tn,
                                                    const double* __restrict__ Ypart, int npanels,
                                                    const double* __restrict__ Ttpart, int nsplit,
                                                    double* __restrict__ dw, double* __restrict__ Gpart, int sweep,
                                                    KParams p, DevState* st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    double cden = 1.0, nt = 0.0;
    int mode = 0;
    if (UPDATE) {
        nt = ordered_sum<8>(Ttpart + t, k, 0, nsplit, 1);      // ||T[t,:]||^2 from the slices of k_tgram
        cden = nt + p.reg_w_l2;                                // denom = nt + reg_w_l2 (nmf.py:465)
        if (!(cden > 0.0)) {
            if (p.has_wrs && p.w_row_sum != 0.0) mode = 1;     // optimization.py:60-67
            else {
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    st->halt = HALT_ERR_UNBOUNDED; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t;
                }
                return;
            }
        }
    }
    double wnew = 0.0, dwi = 0.0, wn = 0.0;
    if (i < n) {
        if (UPDATE) {
            const double y = ordered_sum<8>(Ypart + i, n, 0, npanels, 1);
            const double w0 = Wt[(i64)t * ldw + i];
            const double numer = fma(w0, nt, y) - p.reg_w_l1;
            if (mode == 0) wnew = fmax(numer, 0.0) / (cden + p.eps);
            else wnew = (-numer + cden < 0.0) ? p.w_row_sum : 0.0;
            Wt[(i64)t * ldw + i] = wnew;
            dwi = wnew - w0;
            dw[i] = dwi;
        }
        wn = (UPDATE && tn == t) ? wnew : Wt[(i64)tn * ldw + i];
    }
    const double g = block_sum(wn * dwi, scratch);
    const double nwp = block_sum(wn * wn, scratch);
    const double sw = block_sum(wnew, scratch);
    double* gp = Gpart + (i64)blockIdx.x * (k + 2);
    for (int l = threadIdx.x; l < k + 2; l += 256) gp[l] = (l == t && UPDATE) ? g : (l == k ? nwp : (l == k + 1 ? sw : 0.0));
}

// =========================================================================================
// k_reduce: fixed-order reduction of the row-block partials into the reduce buffer
//   red[0..ldz) = w^T X ; red[ldz + g*(k+2) + ...), g < GRAM_SLICES: slice sums of [w^T W | ||w||^2 | sum W[:,tprev]]
// (in the row-sharded multi-GPU run this buffer is what the ranks all-reduce).
// 1024 threads: column blocks take 32 columns x 32 row-block groups, the last block the Gram row.
// =========================================================================================
__global__ __launch_bounds__(1024) void k_reduce(const double* __restrict__ Zpart, i64 ldz, int nrb,
                                                 const double* __restrict__ Gpart, int nwb, int k,
                                                 double* __restrict__ red, const DevState* __restrict__ st) {
    if (st->halt) return;
    __shared__ double sh[32 * 33];
    const int tid = threadIdx.x;
    // Gpart == NULL: column blocks only (the grid then has no Gram blocks)
    const int ngram = Gpart ? GRAM_SLICES : 0;
    const int nzb = gridDim.x - ngram;
    if ((int)blockIdx.x < nzb) {
        const int c = tid & 31, g = tid >> 5;
        const i64 j = (i64)blockIdx.x * 32 + c;
        double a = 0.0;
        if (j < ldz) a = ordered_sum<8>(Zpart + j, ldz, g, nrb, 32);
        sh[g * 33 + c] = a;
        __syncthreads();
        if (tid < 32 && j < ldz) {   // here c == tid
            double s = 0.0;
            for (int q = 0; q < 32; ++q) s += sh[q * 33 + tid];
            red[j] = s;
        }
    } else {
        // Gram slice gs: rows [b0, b1) of Gpart, 16 waves striding over them, lanes over the k+2 entries
        const int gs = blockIdx.x - nzb;
        const int per = (nwb + GRAM_SLICES - 1) / GRAM_SLICES;
        const int b0 = gs * per, b1 = min(nwb, b0 + per);
        const int lane = tid & 63, wave = tid >> 6;  // 16 waves
        for (int l0 = 0; l0 < k + 2; l0 += 64) {
            const int l = l0 + lane;
            double a = 0.0;
            if (l < k + 2) a = ordered_sum<8>(Gpart + l, k + 2, b0 + wave, b1, 16);
            __syncthreads();
            sh[wave * 64 + lane] = a;
            __syncthreads();
            if (wave == 0 && l < k + 2) {
                double s = 0.0;
                for (int q = 0; q < 16; ++q) s += sh[q * 64 + lane];
                red[ldz + (i64)gs * (k + 2) + l] = s;
            }
        }
    }
}

// red[ldz + g*(k+2) + l], g < GRAM_SLICES -> entry l of [w^T W | ||w||^2 | sum W[:,tprev]]
__device__ __forceinline__ double red_gram(const double* __restrict__ red, i64 ldz, int k, int l) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < GRAM_SLICES; ++g) s += red[ldz + (i64)g * (k + 2) + l];
    return s;
}

// =========================================================================================
// k_trow_numer: numerator of the T-row update and the closed-form minimiser (before any
// simplex projection).  Also evaluates the pending W-column check of the previous topic
// (_check_reset_W + assert, nmf.py:471-476) from the reduced column sum.  When no projection
// follows, the new row is written straight into T.  blockDim = 128.
// =========================================================================================
__global__ __launch_bounds__(128) void k_trow_numer(double* __restrict__ T, i64 ldt, int d, int k, int t,
                                                    const double* __restrict__ red, i64 ldz,
                                                    double* __restrict__ xraw, double* __restrict__ tpart,
                                                    i64* __restrict__ tpart_idx, int check_prev, int tprev,
                                                    int sweep, KParams p, DevState* st, int resid_form,
                                                    double* __restrict__ told) {
    // resid_form (explicit-residual schedule): red[0..d) = R^T w_t over the stored residual, the Gram part holds the
    // coefficient of the one pending rank-one term (slot tprev: <dw_tprev, w_t>) and zeros; the topic's own term
    // comes back as + ||w_t||^2 T[t,:]  (numer_T = R^T w + t ||w||^2).  told: the row before its update.
    if (st->halt) return;
    const int tid = threadIdx.x;
    __shared__ double scratch[40];
    __shared__ double gsh[RRI_MAX_K];
    const double nw = red_gram(red, ldz, k, k);
    if (check_prev) {
        const double sw = red_gram(red, ldz, k, k + 1);
        const bool ev = (sw <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
        const bool err = !ev && !(sw > 0.0);
        if (ev || err) {
            if (blockIdx.x == 0 && tid == 0) {
                st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
                st->halt_topic = tprev; st->halt_sweep = sweep; st->halt_pos = t;
            }
            return;
        }
    }
    const double c = nw + p.reg_t_l2;  // denom = nw + reg_t_l2 (nmf.py:438)
    const bool project = p.project_T && p.has_trs;
    int mode = 0;
    if (!(c > 0.0)) {
        if (!project) {
            if (p.has_trs && p.t_row_sum != 0.0) mode = 1;
            else mode = HALT_ERR_UNBOUNDED;
        } else {
            mode = (p.t_row_sum == 1.0) ? 2 : HALT_ERR_NOT_IMPLEMENTED;
        }
        if (mode < 0) {
            if (blockIdx.x == 0 && tid == 0) {
                st->halt = mode; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t;
            }
            return;
        }
    }
    for (int l = tid; l < k; l += 128) gsh[l] = (l == t) ? (resid_form ? -nw : 0.0) : red_gram(red, ldz, k, l);
    __syncthreads();
    const i64 j = (i64)blockIdx.x * 128 + tid;
    double x = 0.0, mx = -1.0e300;
    if (j < d) {
        if (told) told[j] = T[(i64)t * ldt + j];
        // gsh[t] == 0, so row t may be read like the others; 16 loads are issued before the first use
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int l0 = 0; l0 < k; l0 += 16) {
            double tv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) tv[q] = (l0 + q < k) ? T[(i64)(l0 + q) * ldt + j] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                if (l0 + q < k) a0 = fma(gsh[l0 + q], tv[q], a0);
                if (l0 + q + 1 < k) a1 = fma(gsh[l0 + q + 1], tv[q + 1], a1);
                if (l0 + q + 2 < k) a2 = fma(gsh[l0 + q + 2], tv[q + 2], a2);
                if (l0 + q + 3 < k) a3 = fma(gsh[l0 + q + 3], tv[q + 3], a3);
            }
        }
        const double acc = (a0 + a1) + (a2 + a3);
        const double numer = (red[j] - acc) - p.reg_t_l1;
        if (mode == 0) x = fmax(numer, 0.0) / (c + p.eps);
        else if (mode == 1) x = (-numer + c < 0.0) ? p.t_row_sum : 0.0;
        else { x = numer; mx = numer; }
        xraw[j] = x;
        if (mode == 1 || (mode == 0 && !project)) T[(i64)t * ldt + j] = x;
    }
    if (mode != 2) {
        const double s = block_sum(x, scratch);
        if (tid == 0) tpart[blockIdx.x] = s;
    } else {
        i64 idx = (j < d) ? j : (i64)0x7fffffffffffffffLL;
        wave_argmax(mx, idx);
        __shared__ double wm[2];
        __shared__ i64 wi[2];
        if ((tid & 63) == 0) { wm[tid >> 6] = mx; wi[tid >> 6] = idx; }
        __syncthreads();
        if (tid == 0) {
            if (wm[1] > wm[0] || (wm[1] == wm[0] && wi[1] < wi[0])) { wm[0] = wm[1]; wi[0] = wi[1]; }
            tpart[blockIdx.x] = wm[0];
            tpart_idx[blockIdx.x] = wi[0];
        }
    }
    if (blockIdx.x == 0 && tid == 0) st->tmode = mode;
}

// =========================================================================================
// k_trow_small: k_reduce and k_trow_numer in ONE launch, for problems small enough that every workgroup can afford
// to reduce the Gram partials itself (nwb (k+2) doubles, L2-resident): launch-bound sizes lose a dependent launch per
// topic step.  Block = 32 columns x 32 groups (1024 threads): the groups share the row blocks of Zpart for the column
// sums, the Gpart rows for the Gram entries and the topics for (w^T W)_{-t} T; partials meet in LDS and are added in a
// fixed order.  Also writes red (columns and slice 0 of the Gram part, the other slices zero), so whoever reads the
// reduce buffer afterwards sees what k_reduce would have left up to the order of the sums.
// tpart / tpart_idx then hold one entry per 32 columns.
// =========================================================================================
__global__ __launch_bounds__(1024) void k_trow_small(double* __restrict__ T, i64 ldt, int d, int k, int t,
                                                     const double* __restrict__ Zpart, int nrb,
                                                     const double* __restrict__ Gpart, int nwb, double* __restrict__ red,
                                                     i64 ldz, double* __restrict__ xraw, double* __restrict__ tpart,
                                                     i64* __restrict__ tpart_idx, int check_prev, int tprev, int sweep,
                                                     KParams p, DevState* st, int resid_form, double* __restrict__ told) {
    if (st->halt) return;
    const int tid = threadIdx.x;
    __shared__ double sh[32 * 33];
    __shared__ double gsh[RRI_MAX_K + 2];
    __shared__ double zs[32];
    const int cc = tid & 31, g = tid >> 5;
    const i64 j = (i64)blockIdx.x * 32 + cc;
    // The three sets of operands of this kernel -- Gram partials, column-sum partials, rows of T -- do not depend on
    // each other: the column-sum partials (the first 16 per thread) and the rows of T (the first 4 per thread) are
    // requested here, before the Gram stage and its barriers, so the three round trips to L2 overlap instead of
    // following one another (launch-bound sizes: this kernel is a chain of dependent latencies, nothing else).
    constexpr int ZB = 16;
    double zpre[ZB];
#pragma unroll
    for (int q = 0; q < ZB; ++q) {
        const int r = g + 32 * q;
        zpre[q] = (j < ldz && r < nrb) ? Zpart[(i64)r * ldz + j] : 0.0;
    }
    double tkeep[4];     // T[g + 32 q][j]: the first round of the (w^T W) T sum (all there is when k <= 128)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int l = g + 32 * q;
        tkeep[q] = (j < d && l < k && (l != t || resid_form)) ? T[(i64)l * ldt + j] : 0.0;
    }
    // Gram entries [w^T W (k) | ||w||^2 | sum W[:,tprev]]: 32 groups over the Gpart rows, 32 entries per round
    for (int l0 = 0; l0 < k + 2; l0 += 32) {
        const int l = l0 + cc;
        double a = 0.0;
        if (l < k + 2) a = ordered_sum<8>(Gpart + l, k + 2, g, nwb, 32);
        __syncthreads();
        sh[g * 33 + cc] = a;
        __syncthreads();
        if (tid < 32 && l < k + 2) {
            double s = 0.0;
            for (int q = 0; q < 32; ++q) s += sh[q * 33 + tid];
            gsh[l] = s;
        }
    }
    __syncthreads();
    const double nw = gsh[k];
    if (check_prev) {
        const double sw = gsh[k + 1];
        const bool ev = (sw <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
        const bool err = !ev && !(sw > 0.0);
        if (ev || err) {
            if (blockIdx.x == 0 && tid == 0) {
                st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
                st->halt_topic = tprev; st->halt_sweep = sweep; st->halt_pos = t;
            }
            return;
        }
    }
    const double c = nw + p.reg_t_l2;  // denom = nw + reg_t_l2 (nmf.py:438)
    const bool project = p.project_T && p.has_trs;
    int mode = 0;
    if (!(c > 0.0)) {
        if (!project) {
            if (p.has_trs && p.t_row_sum != 0.0) mode = 1;
            else mode = HALT_ERR_UNBOUNDED;
        } else {
            mode = (p.t_row_sum == 1.0) ? 2 : HALT_ERR_NOT_IMPLEMENTED;
        }
        if (mode < 0) {
            if (blockIdx.x == 0 && tid == 0) {
                st->halt = mode; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t;
            }
            return;
        }
    }
    if (blockIdx.x == 0)   // the reduce buffer as k_reduce leaves it: slice 0 carries the sums, the rest zeros
        for (int i = tid; i < GRAM_SLICES * (k + 2); i += 1024) red[ldz + i] = i < k + 2 ? gsh[i] : 0.0;
    {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < ZB; ++q) a += zpre[q];                          // rows g, g + 32, ...: the same order as before
        if (j < ldz) a = ordered_sum_acc<8>(a, Zpart + j, ldz, g + 32 * ZB, nrb, 32);
        sh[g * 33 + cc] = a;
    }
    __syncthreads();
    if (tid < 32) {   // here cc == tid
        double s = 0.0;
        for (int q = 0; q < 32; ++q) s += sh[q * 33 + tid];
        zs[tid] = s;
        if (j < ldz) red[j] = s;
    }
    __syncthreads();
    {
        double a = 0.0;
        if (j < d)
            for (int l0 = g; l0 < k; l0 += 32 * 4) {   // 4 rows of T in flight, added in topic order
                double tv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int l = l0 + 32 * q;
                    if (l0 == g) tv[q] = tkeep[q];
                    else tv[q] = (l < k && (l != t || resid_form)) ? T[(i64)l * ldt + j] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int l = l0 + 32 * q;
                    if (l < k && l != t) a = fma(gsh[l], tv[q], a);
                    else if (l == t && resid_form) a = fma(-nw, tv[q], a);   // + ||w_t||^2 T[t,:] (explicit-residual form)
                }
            }
        sh[g * 33 + cc] = a;
    }
    __syncthreads();
    if (tid < 64) {   // wave 0; lanes 0..31 own a column each
        double x = 0.0, mx = -1.0e300;
        i64 idx = (i64)0x7fffffffffffffffLL;
        if (tid < 32 && j < d) {
            double acc = 0.0;
            for (int q = 0; q < 32; ++q) acc += sh[q * 33 + tid];
            if (told) told[j] = T[(i64)t * ldt + j];
            const double numer = (zs[tid] - acc) - p.reg_t_l1;
            if (mode == 0) x = fmax(numer, 0.0) / (c + p.eps);
            else if (mode == 1) x = (-numer + c < 0.0) ? p.t_row_sum : 0.0;
            else { x = numer; mx = numer; idx = j; }
            xraw[j] = x;
            if (mode == 1 || (mode == 0 && !project)) T[(i64)t * ldt + j] = x;
        }
        if (mode != 2) {
            const double s = wave_sum<double>(x);
            if (tid == 0) tpart[blockIdx.x] = s;
        } else {
            wave_argmax(mx, idx);
            if (tid == 0) { tpart[blockIdx.x] = mx; tpart_idx[blockIdx.x] = idx; }
        }
    }
    if (blockIdx.x == 0 && tid == 0) st->tmode = mode;
}

// Michelot's fixed point for the Euclidean simplex projection (same active set, hence the same
// theta = (sum_active - s)/|active|, as the sort of matrixops.py:58-63).  Single workgroup.
__device__ inline double simplex_theta(const double* v, int d, double s, double* scratch, int* iters) {
    double theta = -1.0e300;
    i64 cnt_prev = -1;
    int it = 0;
    for (; it < d + 2; ++it) {
        double sum = 0.0, cnt = 0.0;
        for (int j = threadIdx.x; j < d; j += blockDim.x) {
            const double x = v[j];
            if (x > theta) { sum += x; cnt += 1.0; }
        }
        sum = block_sum(sum, scratch);
        cnt = block_sum(cnt, scratch);
        const i64 ci = (i64)cnt;
        if (ci == cnt_prev || ci == 0) break;
        theta = (sum - s) / cnt;
        cnt_prev = ci;
    }
    *iters = it;
    return theta;
}

// =========================================================================================
// k_trow_final: finishes qf_min (simplex projection or one-hot), the row checks of
// _project_and_check_reset_t (nmf.py:751-769) and writes T[t,:].  One workgroup of 1024.
// `light` = no projection configured: the row is already in T and only the sums are needed.
// =========================================================================================
__global__ __launch_bounds__(1024) void k_trow_final(double* __restrict__ T, i64 ldt, int d, int t,
                                                     double* __restrict__ xraw, const double* __restrict__ tpart,
                                                     const i64* __restrict__ tpart_idx, int nblk, int sweep,
                                                     KParams p, DevState* st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    const int tid = threadIdx.x;
    const int mode = st->tmode;
    const bool project = p.project_T && p.has_trs;
    double nx = 1.0, sumT = 0.0;
    int iters = 0;
    bool row_dirty = false;  // xraw differs from what k_trow_numer stored in T
    if (mode == 2) {
        double bm = tpart[0];
        i64 bi = tpart_idx[0];
        for (int b = 1; b < nblk; ++b)
            if (tpart[b] > bm) { bm = tpart[b]; bi = tpart_idx[b]; }
        for (int j = tid; j < d; j += blockDim.x) xraw[j] = (j == bi) ? 1.0 : 0.0;
        row_dirty = true;
    } else {
        double ps = 0.0;
        ps = ordered_sum<4>(tpart, 1, tid, nblk, (int)blockDim.x);
        ps = block_sum(ps, scratch);
        if (mode == 0) nx = ps;
        sumT = ps;
        if (mode == 0 && project) {
            const double th = simplex_theta(xraw, d, p.t_row_sum, scratch, &iters);
            for (int j = tid; j < d; j += blockDim.x) xraw[j] = fmax(xraw[j] - th, 0.0);
            if (tid == 0) st->theta = th;
            row_dirty = true;
        }
    }
    if (row_dirty) {
        __syncthreads();
        sumT = 0.0;
        for (int j = tid; j < d; j += blockDim.x) sumT += xraw[j];
        sumT = block_sum(sumT, scratch);
    }
    bool event = false;
    if (sumT > 1e-10 || p.reset_method == RESET_NONE) {
        // nmf.py:759-761: project again when the row is not on the simplex to 1e-15
        if (p.has_trs && p.t_row_sum != 0.0 && p.project_T && fabs(sumT - p.t_row_sum) > 1e-15) {
            int it2 = 0;
            const double th = simplex_theta(xraw, d, p.t_row_sum, scratch, &it2);
            for (int j = tid; j < d; j += blockDim.x) xraw[j] = fmax(xraw[j] - th, 0.0);
            iters += it2;
            row_dirty = true;
        }
    } else if (p.resets_left > 0) {
        event = true;
    }
    if (row_dirty) {
        __syncthreads();
        for (int j = tid; j < d; j += blockDim.x) T[(i64)t * ldt + j] = xraw[j];
    }
    if (tid == 0) {
        st->nt1 = nx;
        st->sumT = sumT;
        st->proj_iters = iters;
        if (event) { st->halt = HALT_EVENT_RESET_T; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t; }
    }
}

// k_tgram: partial T T[t,:]^T over column slice blockIdx.y: Ttpart[y][l] = <T[l,slice], T[t,slice]>
// (entry l == t is the slice's share of ||T[t,:]||^2; k_wcol sums the slices, nmf.py:730-734).
// finish != 0 (no projection configured, so k_trow_numer already stored the row): block (0,0) also does
// what is left of _project_and_check_reset_t: nt1, sum(T[t,:]) and the reset decision (nmf.py:757-769).
__global__ __launch_bounds__(256) void k_tgram(const double* __restrict__ T, i64 ldt, int d, int k, int t,
                                               double* __restrict__ Ttpart, const double* __restrict__ tpart,
                                               int nblk, int finish, int sweep, KParams p, DevState* st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    tgram_block(T, ldt, d, k, t, blockIdx.x, blockIdx.y, gridDim.y, Ttpart, tpart, nblk, finish, sweep, p, st, scratch);
}

// W[:,t] *= nt1 (nmf.py:450-452); only observable when fix_W keeps the column.
__global__ __launch_bounds__(256) void k_scale_wcol(double* Wt, i64 ldw, int n, int t, const DevState* st) {
    if (st->halt) return;
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Wt[(i64)t * ldw + i] *= st->nt1;
}

// the W-column check alone, against the reduced buffer (row-sharded runs, after the last sweep)
__global__ __launch_bounds__(64) void k_check_red(const double* __restrict__ red, i64 ldz, int k, int tprev,
                                                  int sweep, int pos, KParams p, DevState* st) {
    if (st->halt) return;
    if (threadIdx.x == 0) {
        const double sw = red_gram(red, ldz, k, k + 1);
        const bool ev = (sw <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
        const bool err = !ev && !(sw > 0.0);
        if (ev || err) {
            st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
            st->halt_topic = tprev; st->halt_sweep = sweep; st->halt_pos = pos;
        }
    }
}

// stand-alone W-column check for the paths without a following T-row step
__global__ __launch_bounds__(256) void k_check_wcol(const double* __restrict__ Gpart, int nwb, int k, int tprev,
                                                    int sweep, int pos, KParams p, DevState* st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    double a = 0.0;
    a = ordered_sum<8>(Gpart + k + 1, k + 2, threadIdx.x, nwb, (int)blockDim.x);
    a = block_sum(a, scratch);
    if (threadIdx.x == 0) {
        const bool ev = (a <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
        const bool err = !ev && !(a > 0.0);
        if (ev || err) {
            st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
            st->halt_topic = tprev; st->halt_sweep = sweep; st->halt_pos = pos;
        }
    }
}

// row-sharded runs: this rank's share of sum W[:,tprev] (and no negative-denominator flag) into tail[0..1]; the
// verdict is taken on the all-reduced pair (k_wcheck_tail)
__global__ __launch_bounds__(256) void k_colsum_tail(const double* __restrict__ Gpart, int nwb, int k,
                                                     double* __restrict__ tail, const DevState* __restrict__ st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    double a = ordered_sum<8>(Gpart + k + 1, k + 2, threadIdx.x, nwb, (int)blockDim.x);
    a = block_sum(a, scratch);
    if (threadIdx.x == 0) { tail[0] = a; tail[1] = 0.0; }
}

// max-residual reset of a row-sharded run: this rank's candidate (largest row-residual norm, its GLOBAL row index)
__global__ void k_pack_candidate(const double* __restrict__ rowpos, const i64* __restrict__ idx, i64 row_offset,
                                 double* __restrict__ out) {
    const i64 i = idx[0];
    out[0] = rowpos[i];
    out[1] = (double)(row_offset + i);
}

// =========================================================================================
// k_proj_rows: Euclidean projection of every row of W on the simplex of radius s (scalar) or
// s_vec[i] (matrixops.py:72-100; nmf.py:481-484, 519-529).  One thread per row; the k entries
// of a row are k coalesced loads from the k-major Wt (re-read per Michelot iteration from L2).
// =========================================================================================
__global__ __launch_bounds__(256) void k_proj_rows(double* __restrict__ Wt, i64 ldw, int n, int k, double s_scalar,
                                                   const double* __restrict__ s_vec) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s = s_vec ? s_vec[i] : s_scalar;
    double theta = -1.0e300;
    int cnt_prev = -1;
    for (int it = 0; it < k + 2; ++it) {
        double sum = 0.0;
        int cnt = 0;
        for (int l = 0; l < k; ++l) {
            const double x = Wt[(i64)l * ldw + i];
            if (x > theta) { sum += x; ++cnt; }
        }
        if (cnt == cnt_prev || cnt == 0) break;
        theta = (sum - s) / (double)cnt;
        cnt_prev = cnt;
    }
    for (int l = 0; l < k; ++l) Wt[(i64)l * ldw + i] = fmax(Wt[(i64)l * ldw + i] - theta, 0.0);
}

// =========================================================================================
// k_resid: E = X - W T on 64x64 tiles with a 4x4 register tile per thread (a dense k-panel
// GEMM on the vector ALU; not on the per-sweep hot path).  Used for
//   - true_objective (nmf.py:77-83): rowobj[i] = sum_j (Wm_ij) E_ij^2
//   - 'max_resid_document' (nmf.py:771-773): rowpos[i] = sum_j max(E_ij,0)^2
//   - WRITE_E: the masked residual M .* E of the weighted flavour.
// grid = ceil(n/64) workgroups of 256; each walks all column tiles (deterministic sums).
// =========================================================================================
template <typename SX, bool MASKED, bool WRITE_E>
__global__ __launch_bounds__(256) void k_resid(const SX* __restrict__ X, i64 ldx, const SX* __restrict__ M, i64 ldm,
                                               const unsigned* __restrict__ Mb, i64 ldb,
                                               const double* __restrict__ Wt, i64 ldw, const double* __restrict__ T,
                                               i64 ldt, int n, int d, int k, double* __restrict__ rowobj,
                                               double* __restrict__ rowpos, SX* __restrict__ E, i64 lde,
                                               int w_resident) {
    typedef double S;
    constexpr int KC = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // w_resident: the whole 64-row tile of W (k x 64) stays in LDS for all column tiles (k <= 256); otherwise only a
    // KC-topic slice at a time, reloaded per column tile (large k: correctness path, L2 traffic n k 8 per column tile)
    S* Wsh = reinterpret_cast<S*>(smem);          // [k or KC][64]  (transposed tile of W)
    S* Tsh = Wsh + (size_t)(w_resident ? k : KC) * 64;   // [KC][64]
    double* red = reinterpret_cast<double*>(Tsh + KC * 64);  // [64][17]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const i64 row0 = (i64)blockIdx.x * 64;
    if (w_resident)
        for (int idx = tid; idx < 64 * k; idx += 256) {
            const int l = idx >> 6, r = idx & 63;
            Wsh[l * 64 + r] = (row0 + r < n) ? Wt[(i64)l * ldw + row0 + r] : S(0);
        }
    double so[4] = {0, 0, 0, 0}, sp[4] = {0, 0, 0, 0};
    for (i64 c0 = 0; c0 < d; c0 += 64) {
        S acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = S(0);
        for (int l0 = 0; l0 < k; l0 += KC) {
            const int kc = min(KC, k - l0);
            __syncthreads();
            for (int idx = tid; idx < kc * 64; idx += 256) {
                const int l = idx >> 6, c = idx & 63;
                Tsh[l * 64 + c] = (c0 + c < d) ? T[(i64)(l0 + l) * ldt + c0 + c] : S(0);
                if (!w_resident) Wsh[l * 64 + c] = (row0 + c < n) ? Wt[(i64)(l0 + l) * ldw + row0 + c] : S(0);
            }
            __syncthreads();
            const int wbase = w_resident ? l0 : 0;
            for (int l = 0; l < kc; ++l) {
                S wv[4], tv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) wv[a] = Wsh[(wbase + l) * 64 + ty * 4 + a];
#pragma unroll
                for (int b = 0; b < 4; ++b) tv[b] = Tsh[l * 64 + tx * 4 + b];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = fma(wv[a], tv[b], acc[a][b]);
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const i64 i = row0 + ty * 4 + a;
            if (i >= n) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const i64 j = c0 + tx * 4 + b;
                if (j >= d) continue;
                S e = (S)X[i * ldx + j] - acc[a][b];
                const S m = !MASKED ? S(1)
                                    : (Mb ? (S)((Mb[(i >> 3) * ldb + (j >> 2)] >> ((int)((i & 7) << 2) + (int)(j & 3))) & 1u)
                                          : (S)M[i * ldm + j]);
                if (WRITE_E) E[i * lde + j] = (SX)(m * e);
                so[a] += (double)m * (double)e * (double)e;
                const double ep = e > S(0) ? (double)e : 0.0;
                sp[a] += ep * ep;
            }
        }
    }
    __syncthreads();
    // reduce the 16 column-group partials of every row in a fixed order
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int a = 0; a < 4; ++a) red[(ty * 4 + a) * 17 + tx] = pass == 0 ? so[a] : sp[a];
        __syncthreads();
        if (tid < 64 && row0 + tid < n) {
            double s = 0.0;
            for (int q = 0; q < 16; ++q) s += red[tid * 17 + q];
            if (pass == 0) { if (rowobj) rowobj[row0 + tid] = s; }
            else { if (rowpos) rowpos[row0 + tid] = s; }
        }
        __syncthreads();
    }
}

// =========================================================================================
// k_resid_mfma: the same residual on the matrix cores, for k <= 64: E tile (64 rows x 64 columns per step) =
// X - W T with v_mfma_f64_16x16x4_f64.  Wave w of the block owns rows 16w .. 16w+15: its A fragments (W rows,
// one double per lane and k-step: lane l holds W[row l&15][4s + (l>>4)]) stay in registers for the whole row
// block; the T tile of a step (4 KS x 64 doubles) is shared by the 4 waves through LDS, double-buffered: the loads
// of step j+1 (T tile and X) are in flight while the 4 KS MFMAs of step j run.
// The 4 MFMA tiles of a step take INTERLEAVED columns -- tile c owns columns c0 + 4m + c, m < 16 -- so the 4 results
// a lane holds for one row (f64 MFMA result layout: lane l, register r = row (l>>4) + 4r, tile column l&15) are 4
// CONSECUTIVE columns of X: one 16-byte load (and store) per lane and row, 256 contiguous bytes per 16 lanes, and
// one mask nibble.  The T tile is parked in LDS permuted to match (column 4m + c at slot 16c + m), so a B fragment
// read is 16 consecutive doubles: no bank conflicts.
// =========================================================================================
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int RESID_TS = 64;   // row stride of k_resid_mfma's T tile in LDS, in doubles (80 -- the two k-rows a half-wave reads then lie
                               // 32 banks apart instead of on the same banks -- measured the same: 2.81 ms at BASELINE config 5)

// SUMS = false (a rebuild of the stored residual that wants neither row sums): the epilogue is convert, subtract, store.
template <typename SX, bool MASKED, bool WRITE_E, int KS, int WAVES, bool SUMS = true>   // KS = k-steps of 4 (4, 8, 12, 13 or 16): k <= 4 KS
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2))) void k_resid_mfma(const SX* __restrict__ X, i64 ldx, const SX* __restrict__ M,
                                                    i64 ldm, const unsigned* __restrict__ Mb, i64 ldb,
                                                    const double* __restrict__ Wt, i64 ldw,
                                                    const double* __restrict__ T, i64 ldt, int n, int d, int k,
                                                    double* __restrict__ rowobj, double* __restrict__ rowpos,
                                                    SX* __restrict__ E, i64 lde, int dchunk) {
    // blockIdx.y: the column range [y dchunk, (y+1) dchunk) of the row block (dchunk a multiple of 64; the builds that take row
    // sums run with one range).  A row block over ALL columns is a long workgroup, and 1563 of them on 256 CUs x 2 are 3.05
    // rounds: the last 26 ran a fourth round on an empty chip.
    constexpr int kp = 4 * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tsh = reinterpret_cast<double*>(smem);   // [2][kp][RESID_TS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 row0 = (i64)blockIdx.x * (16 * WAVES) + wave * 16;
    const int lr = lane & 15, lk = lane >> 4;
    double afrag[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int l = 4 * s + lk;
        afrag[s] = (l < k && row0 + lr < n) ? Wt[(i64)l * ldw + row0 + lr] : 0.0;
    }
    // cooperative load of a T tile: thread -> column tid & 63 (parked at its permuted slot), rows (tid >> 6) + WAVES i
    const int tc = tid & 63, tr = tid >> 6;
    const int slot = 16 * (tc & 3) + (tc >> 2);
    constexpr int NST = (kp + WAVES - 1) / WAVES;
    double stage[NST];
    typedef typename std::conditional<sizeof(SX) == 4, float, double>::type XR;
    XR xq[4][4];                                     // the lane's 4 rows x 4 consecutive columns of X, as stored
    XR xn[4][4];                                     // ... of the NEXT step: requested at the start of a step, so that
                                                     // nothing issued late in a step is waited for before its barrier
    auto fetch = [&](i64 c0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int l = tr + WAVES * i;
            stage[i] = (l < k && c0 + tc < d) ? T[(i64)l * ldt + c0 + tc] : 0.0;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (tr + WAVES * i < kp) tsh[((size_t)buf * kp + tr + WAVES * i) * RESID_TS + slot] = stage[i];
    };
    // X: rows row0 + lk + 4r, columns c0 + 4 lr .. + 3 (a 16-byte vector when the storage type is fp32 and the
    // columns are all there; elementwise at the ragged edge and for fp64 storage)
    auto fetch_x = [&](i64 c0, XR (&dst)[4][4]) {
        const i64 j = c0 + 4 * lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const i64 i = row0 + lk + 4 * r;
            bool done = false;
            if constexpr (sizeof(SX) == 4) {
                if (i < n && j + 3 < d) {
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + i * ldx + j));
                    dst[r][0] = v[0]; dst[r][1] = v[1]; dst[r][2] = v[2]; dst[r][3] = v[3];
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int c = 0; c < 4; ++c) dst[r][c] = (i < n && j + c < d) ? (XR)X[i * ldx + j + c] : XR(0);
            }
        }
    };
    // the lane's nibble rows of the bit-packed mask, requested with the X tile of the same step (round 4: loaded where they
    // were used, each of the four words was waited for on its own -- s_waitcnt vmcnt(0) four times per step and wave, with the
    // next step's T and X tiles in the same queue: the matrix cores were busy 38 % of the rebuild, SQ_VALU_MFMA_BUSY_CYCLES)
    unsigned mq[4] = {0xFu, 0xFu, 0xFu, 0xFu}, mn[4] = {0xFu, 0xFu, 0xFu, 0xFu};
    auto fetch_m = [&](i64 c0, unsigned (&dst)[4]) {
        if (!(MASKED && Mb)) return;
        const i64 j = c0 + 4 * lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const i64 i = row0 + lk + 4 * r;
            dst[r] = (i < n && j < d) ? Mb[(i >> 3) * ldb + (j >> 2)] : 0u;      // the raw word: shifted where it is used (a
                                                                                   // shift here is a wait for the load here)
        }
    };
    double so[4] = {0, 0, 0, 0}, sp[4] = {0, 0, 0, 0};
    const i64 cb = (i64)blockIdx.y * dchunk;
    const i64 ce = cb + dchunk < (i64)d ? cb + dchunk : (i64)d;
    fetch(cb);
    fetch_x(cb, xq);
    fetch_m(cb, mq);
    park(0);
    __syncthreads();
    int buf = 0;
    for (i64 c0 = cb; c0 < ce; c0 += 64, buf ^= 1) {
        const bool more = c0 + 64 < ce;
        if (more) {                                  // the next T tile AND the next X tile: in flight during the MFMAs
            fetch(c0 + 64);                          // and the epilogue below (round 1 asked for X after the epilogue and
            fetch_x(c0 + 64, xn);                    // then waited for it -- vmcnt(0) before the park -- once per step)
            fetch_m(c0 + 64, mn);
        }
        f64x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
        const double* tb = tsh + (size_t)buf * kp * RESID_TS;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double b = tb[(4 * s + lk) * RESID_TS + 16 * c + lr];
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[s], b, acc[c], 0, 0, 0);
            }
        }
        // the next T tile goes to LDS BEFORE the epilogue's stores are issued: the wait for its loads is a wait for everything this
        // wave has in the vector-memory queue, and after the epilogue that includes the four 16-byte stores of E -- their
        // acknowledgement was waited for in every step (round 4, from the ISA: s_waitcnt vmcnt(0) between the last store and the
        // first ds_write).  Here the stores of a step have the matrix-core phase of the next one to land.
        if (more) park(buf ^ 1);
        // epilogue: e = x - (W T), masked; sums per row; tile c, register r -> row lk + 4r, column 4 lr + c
        const i64 j0 = c0 + 4 * lr;
        // a tile with all its rows and columns there, fp32 storage, no array mask (wave-uniform): the same arithmetic without the
        // per-element edge tests -- those were a branch per element and row in the ISA, hundreds per step
        // (the builds without row sums only: with them the second epilogue takes the k = 52 masked build past 256 registers)
        const bool interior = !SUMS && sizeof(SX) == 4 && row0 + 15 < n && c0 + 64 <= (i64)d && (!MASKED || Mb);
        if (interior) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i64 i = row0 + lk + 4 * r;
                const unsigned bits = MASKED ? mq[r] >> ((int)(i & 7) << 2) : 0xFu;
                double ev[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double e = (double)xq[r][c] - acc[c][r];
                    const double m = MASKED ? (double)((bits >> c) & 1u) : 1.0;
                    if constexpr (SUMS) {
                        so[r] += m * e * e;
                        const double ep = e > 0.0 ? e : 0.0;
                        sp[r] += ep * ep;
                    }
                    ev[c] = MASKED ? m * e : e;
                }
                if constexpr (WRITE_E && sizeof(SX) == 4) {
                    const f32x4 o = f32x4{(float)ev[0], (float)ev[1], (float)ev[2], (float)ev[3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(E + i * lde + j0));
                }
            }
        } else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const i64 i = row0 + lk + 4 * r;
            if (i >= n) continue;
            unsigned bits = 0xFu;
            if (MASKED && Mb) bits = mq[r] >> ((int)(i & 7) << 2);
            double ev[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const i64 j = j0 + c;
                double e = 0.0, m = 0.0;
                if (j < d) {
                    e = (double)xq[r][c] - acc[c][r];
                    m = !MASKED ? 1.0 : (Mb ? (double)((bits >> c) & 1u) : (double)M[i * ldm + j]);
                    if constexpr (SUMS) {
                        so[r] += m * e * e;
                        const double ep = e > 0.0 ? e : 0.0;
                        sp[r] += ep * ep;
                    }
                }
                ev[c] = MASKED ? m * e : e;
            }
            if (WRITE_E) {
                bool stored = false;
                if constexpr (sizeof(SX) == 4) {
                    if (j0 + 3 < d) {
                        const f32x4 o = f32x4{(float)ev[0], (float)ev[1], (float)ev[2], (float)ev[3]};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(E + i * lde + j0));
                        stored = true;
                    }
                }
                if (!stored) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (j0 + c < d) E[i * lde + j0 + c] = (SX)ev[c];
                }
            }
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) xq[r][c] = xn[r][c];
#pragma unroll
            for (int r = 0; r < 4; ++r) mq[r] = mn[r];
        }
        __syncthreads();
    }
    if constexpr (!SUMS) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            so[r] += __shfl_xor(so[r], off, 64);
            sp[r] += __shfl_xor(sp[r], off, 64);
        }
        const i64 i = row0 + lk + 4 * r;
        if (lr == 0 && i < n) {
            if (rowobj) rowobj[i] = so[r];
            if (rowpos) rowpos[i] = sp[r];
        }
    }
}

// =========================================================================================
// k_xtt: Qt = (X T^T)^T, k x n k-major: the row products X T[l,:]^T of ALL topics in one pass over X.
// Used when T is fixed (fold-in / transform, nmf.py:417 `fix_T`): X T^T does not change between
// sweeps, so every later W-column update reads its X t_t from Qt instead of streaming X again.
// Block = 64 rows; threads = 16 row groups (4 rows) x 16 topic lanes (topics tx, tx+16, tx+32, tx+48);
// topics beyond 64 are done in further rounds.  A dense k-panel GEMM on the f64 vector ALU.
// =========================================================================================
template <typename SX>
__global__ __launch_bounds__(256) void k_xtt(const SX* __restrict__ X, i64 ldx, const double* __restrict__ T, i64 ldt,
                                             int n, int d, int k, double* __restrict__ Qt, i64 ldq) {
    constexpr int CT = 32;   // columns per LDS tile
    __shared__ double Xsh[64][CT + 1];
    __shared__ double Tsh[64][CT + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const i64 row0 = (i64)blockIdx.x * 64;
    for (int l0 = 0; l0 < k; l0 += 64) {
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
        for (int c0 = 0; c0 < d; c0 += CT) {
            __syncthreads();
            for (int idx = tid; idx < 64 * CT; idx += 256) {
                const int r = idx / CT, c = idx - r * CT;
                Xsh[r][c] = (row0 + r < n && c0 + c < d) ? (double)X[(row0 + r) * ldx + c0 + c] : 0.0;
                Tsh[r][c] = (l0 + r < k && c0 + c < d) ? T[(i64)(l0 + r) * ldt + c0 + c] : 0.0;
            }
            __syncthreads();
#pragma unroll 8
            for (int c = 0; c < CT; ++c) {
                double xv[4], tv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) xv[a] = Xsh[ty * 4 + a][c];
#pragma unroll
                for (int b = 0; b < 4; ++b) tv[b] = Tsh[tx + 16 * b][c];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = fma(xv[a], tv[b], acc[a][b]);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int l = l0 + tx + 16 * b;
            if (l >= k) continue;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const i64 i = row0 + ty * 4 + a;
                if (i < n) Qt[(i64)l * ldq + i] = acc[a][b];
            }
        }
    }
}

// =========================================================================================
// k_xtt_mfma: Qt = (X Tm^T)^T for up to 64 rows of Tm (topics, or the columns of a dense operand) on the matrix
// cores: D(16 rows of X x 16 topics) += A(16 rows x 4 columns of X) * B(4 columns x 16 topics), f64 MFMA 16x16x4.
// Wave w of the block owns rows 16w .. 16w+15.  Per step of 64 columns:
//   * the wave's X tile (16 x 64, storage type) is read coalesced -- 4 rows x 256 bytes per load -- and parked in a
//     wave-private LDS tile; a lane (row i = l&15, group g = l>>4) then picks its A values as 4 vectors
//     xs[i][16m + 4g .. +3]: MFMA number (m, e) contracts the four columns {16m + 4g + e}, g = 0..3 -- any order
//     of the columns is as good as another for a sum, and this one costs no lane movement;
//   * the T tile (64 columns x 16 NT topics) is shared by the 4 waves through LDS, stored [column][topic] (stride
//     16 NT + 1: the transposing stores hit different banks), so the B fragment of (m, e), topic tile c is 16
//     consecutive doubles; double-buffered: the next tile's loads are in flight during the MFMAs.
// Result layout (f64 MFMA): lane l, register r = X row (l>>4) + 4r, topic 16c + (l&15).
// =========================================================================================
template <typename SX, int NT>
__global__ __launch_bounds__(256) void k_xtt_mfma(const SX* __restrict__ X, i64 ldx, const double* __restrict__ T,
                                                  i64 ldt, int n, int d, int m, double* __restrict__ Qt, i64 ldq) {
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int VN = XV::N;                 // elements per 16-byte vector: 4 (fp32) or 2 (fp64)
    constexpr int TS = 16 * NT + 1;           // row stride of the T tile [column][topic]
    constexpr int XS = 64 + VN;               // row stride of the X tile (elements)
    constexpr int NI = (16 * NT) / 4;         // T rows per thread and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tsh = reinterpret_cast<double*>(smem);                         // [2][64][TS]
    SX* xsh = reinterpret_cast<SX*>(tsh + 2 * 64 * TS);                    // [4 waves][16][XS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 row0 = (i64)blockIdx.x * 64 + wave * 16;
    const int li = lane & 15, lg = lane >> 4;
    SX* xs = xsh + (size_t)wave * 16 * XS;
    const int tc = tid & 63, tr = tid >> 6;
    double stage[NI];
    auto fetch = [&](i64 c0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int l = tr + 4 * i;
            stage[i] = (l < m && c0 + tc < d) ? T[(i64)l * ldt + c0 + tc] : 0.0;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) tsh[((size_t)buf * 64 + tc) * TS + tr + 4 * i] = stage[i];
    };
    // X tile: VN-element vectors, 64 / VN per row, 64 lanes cover (64 * VN / 64) = VN rows per load
    constexpr int VPR = 64 / VN, RPL = 64 / VPR, NL = 16 / RPL;
    V xv[NL];
    auto fetch_x = [&](i64 c0) {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int r = q * RPL + lane / VPR, cv = lane % VPR;
            const i64 i = row0 + r, j = c0 + (i64)cv * VN;
            if (i < n && j + VN - 1 < d) {
                xv[q] = __builtin_nontemporal_load(reinterpret_cast<const V*>(X + i * ldx + j));
            } else {
                double e[VN];
#pragma unroll
                for (int c = 0; c < VN; ++c) e[c] = (i < n && j + c < d) ? (double)X[i * ldx + j + c] : 0.0;
                xv[q] = XV::pack(e);
            }
        }
    };
    auto park_x = [&]() {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int r = q * RPL + lane / VPR, cv = lane % VPR;
            *reinterpret_cast<V*>(xs + r * XS + cv * VN) = xv[q];
        }
    };
    f64x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
    fetch(0);
    fetch_x(0);
    park(0);
    park_x();
    __syncthreads();
    int buf = 0;
    for (i64 c0 = 0; c0 < d; c0 += 64, buf ^= 1) {
        const bool more = c0 + 64 < d;
        if (more) { fetch(c0 + 64); fetch_x(c0 + 64); }
        const double* tb = tsh + (size_t)buf * 64 * TS;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            double a4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) a4[e] = (double)xs[li * XS + 16 * mm + 4 * lg + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double* brow = tb + (size_t)(16 * mm + 4 * lg + e) * TS + li;
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[e], brow[16 * c], acc[c], 0, 0, 0);
            }
        }
        __syncthreads();              // every wave is done with this step's tiles (its own X tile included)
        if (more) { park(buf ^ 1); park_x(); }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int l = 16 * c + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const i64 i = row0 + lg + 4 * r;
            if (l < m && i < n) Qt[(i64)l * ldq + i] = acc[c][r];
        }
    }
}

// partial sums of v, v^2 and |v| over a strided matrix: out[b] = {sum, sumsq, sumabs}
__global__ __launch_bounds__(256) void k_norms(const double* __restrict__ A, i64 rows, i64 cols, i64 ld,
                                               double* __restrict__ out) {
    __shared__ double scratch[40];
    double s1 = 0, s2 = 0, s3 = 0;
    const i64 total = rows * cols;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 r = idx / cols, c = idx - r * cols;
        const double v = A[r * ld + c];
        s1 += v; s2 += v * v; s3 += fabs(v);
    }
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    s3 = block_sum(s3, scratch);
    if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = s1; out[blockIdx.x * 3 + 1] = s2; out[blockIdx.x * 3 + 2] = s3; }
}

// sum of squares of the stored X (float64 accumulation): out[b] partial of block b
template <typename SX>
__global__ __launch_bounds__(256) void k_sqsum(const SX* __restrict__ X, i64 ldx, i64 n, i64 d, double* __restrict__ out) {
    __shared__ double scratch[40];
    double s = 0.0;
    const i64 total = n * d;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 r = idx / d, c = idx - r * d;
        const double v = (double)X[r * ldx + c];
        s = fma(v, v, s);
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// Gram matrix of the rows of a k x len matrix: G[a*k + b] = <A[a,:], A[b,:]>, one workgroup per (a, b >= a)
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ A, i64 ld, i64 len, int k, double* __restrict__ G) {
    __shared__ double scratch[40];
    const int a = blockIdx.x, b = blockIdx.y;
    if (b < a) return;
    double s = 0.0;
    for (i64 i = threadIdx.x; i < len; i += 256) s = fma(A[(i64)a * ld + i], A[(i64)b * ld + i], s);
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) { G[a * k + b] = s; G[b * k + a] = s; }
}

// A (m x len, row stride ld) <- L^-1 A for a lower-triangular L (m x m, row-major, m <= 64): forward substitution down every
// column, one thread per column, the column in registers.  With L = chol(A A^T) the rows of the result are orthonormal:
// the Cholesky-QR step of the range finder behind the NNDSVD start (rri_range_finder; initialization.py:105).
__global__ __launch_bounds__(256) void k_lsolve_rows(double* __restrict__ A, i64 ld, i64 len, int m, const double* __restrict__ L) {
    __shared__ double Lsh[64 * 64];
    for (int e = threadIdx.x; e < m * m; e += 256) Lsh[e] = L[e];
    __syncthreads();
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    if (r >= len) return;
    double v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = i < m ? A[(i64)i * ld + r] : 0.0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (i < m) {
            double acc = v[i];
#pragma unroll
            for (int j = 0; j < i; ++j) acc = fma(-Lsh[i * m + j], v[j], acc);
            v[i] = acc / Lsh[i * m + i];
        }
    }
#pragma unroll
    for (int i = 0; i < 64; ++i)
        if (i < m) A[(i64)i * ld + r] = v[i];
}

// out[t] = sum_{b < nb} part[t * stride + b], fixed order, one workgroup per t
__global__ __launch_bounds__(256) void k_rows_sum(const double* __restrict__ part, int nb, int stride, double* __restrict__ out) {
    __shared__ double scratch[40];
    double s = 0.0;
    s = ordered_sum<8>(part + (i64)blockIdx.x * stride, 1, threadIdx.x, nb, 256);
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// sum of a double vector -> out[0]; argmax (first index) -> out_idx[0].  One workgroup.
__global__ __launch_bounds__(1024) void k_vec_sum_argmax(const double* __restrict__ v, i64 n, double* out_sum,
                                                         i64* out_idx) {
    __shared__ double scratch[40];
    __shared__ double wm[16];
    __shared__ i64 wi[16];
    double s = 0.0, mx = -1.0e300;
    i64 mi = (i64)0x7fffffffffffffffLL;
    for (i64 i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = v[i];
        s += x;
        if (x > mx) { mx = x; mi = i; }   // ascending i per thread: keeps the first index
    }
    s = block_sum(s, scratch);
    wave_argmax(mx, mi);
    if ((threadIdx.x & 63) == 0) { wm[threadIdx.x >> 6] = mx; wi[threadIdx.x >> 6] = mi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (wm[w] > wm[0] || (wm[w] == wm[0] && wi[w] < wi[0])) { wm[0] = wm[w]; wi[0] = wi[w]; }
        if (out_sum) out_sum[0] = s;
        if (out_idx) out_idx[0] = wi[0];
    }
}

// 'max_resid_document' reset, step 1: row = max(X[mi,:] - W[mi,:] T, 0) (nmf.py:771,774)
template <typename SX>
__global__ __launch_bounds__(256) void k_reset_row(const SX* __restrict__ X, i64 ldx, const double* __restrict__ Wt,
                                                   i64 ldw, const double* __restrict__ T, i64 ldt, int d, int k,
                                                   const i64* __restrict__ mi_ptr, double* __restrict__ rowout) {
    const i64 mi = *mi_ptr;
    const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    double acc = 0.0;
    for (int l = 0; l < k; ++l) acc = fma(Wt[(i64)l * ldw + mi], T[(i64)l * ldt + j], acc);
    rowout[j] = fmax((double)X[mi * ldx + j] - acc, 0.0);
}
// step 2: T[t,:] = row ; W[:,t] = e_mi (nmf.py:774-776)
__global__ __launch_bounds__(256) void k_reset_commit(double* __restrict__ Wt, i64 ldw, double* __restrict__ T,
                                                      i64 ldt, int n, int d, int t, const i64* __restrict__ mi_ptr,
                                                      const double* __restrict__ rowin) {
    const i64 mi = *mi_ptr;
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (idx < d) T[(i64)t * ldt + idx] = rowin[idx];
    if (idx < n) Wt[(i64)t * ldw + idx] = (idx == mi) ? 1.0 : 0.0;
}
// explicit reset vectors ('random', nmf.py:778-783): T[t,:] and W[:,t] from double buffers
__global__ __launch_bounds__(256) void k_set_row_col(double* __restrict__ Wt, i64 ldw, double* __restrict__ T,
                                                     i64 ldt, int n, int d, int t, const double* __restrict__ trow,
                                                     const double* __restrict__ wcolv) {
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (trow && idx < d) T[(i64)t * ldt + idx] = trow[idx];
    if (wcolv && idx < n) Wt[(i64)t * ldw + idx] = wcolv[idx];
}

__global__ __launch_bounds__(256) void k_argmax_rows(const double* __restrict__ Wt, i64 ldw, int n, int k,
                                                     int* __restrict__ out) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double best = Wt[i];
    int bi = 0;
    for (int l = 1; l < k; ++l) {
        const double v = Wt[(i64)l * ldw + i];
        if (v > best) { best = v; bi = l; }
    }
    out[i] = bi;
}

// sum over listed entries of (clip((W T)_ij) - val)^2  (sklearn_interface.py:85-91,172-182)
__global__ __launch_bounds__(256) void k_masked_sqerr(const double* __restrict__ Wt, i64 ldw,
                                                      const double* __restrict__ T, i64 ldt, int k,
                                                      const i64* __restrict__ ij, const double* __restrict__ vals,
                                                      i64 count, double lo, double hi, double* __restrict__ out) {
    __shared__ double scratch[40];
    double s = 0.0;
    for (i64 e = (i64)blockIdx.x * 256 + threadIdx.x; e < count; e += (i64)gridDim.x * 256) {
        const i64 i = ij[2 * e], j = ij[2 * e + 1];
        double acc = 0.0;
        for (int l = 0; l < k; ++l) acc = fma(Wt[(i64)l * ldw + i], T[(i64)l * ldt + j], acc);
        double pr = acc < lo ? lo : (acc > hi ? hi : acc);
        const double df = pr - vals[e];
        s += df * df;
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// 2-D copy with type conversion (host staging -> padded device layout); TRANSPOSE: dst[c][r] = src[r][c]
template <typename Src, typename Dst, bool TRANSPOSE>
__global__ __launch_bounds__(256) void k_convert2d(const Src* __restrict__ src, i64 lds_, Dst* __restrict__ dst,
                                                   i64 ldd, i64 rows, i64 cols) {
    const i64 total = rows * cols;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        if (TRANSPOSE) {   // idx runs along the DESTINATION rows (coalesced writes)
            const i64 c = idx / rows, r = idx - c * rows;
            dst[c * ldd + r] = (Dst)src[r * lds_ + c];
        } else {
            const i64 r = idx / cols, c = idx - r * cols;
            dst[r * ldd + c] = (Dst)src[r * lds_ + c];
        }
    }
}

// plain 16-byte streaming copy: the achievable-HBM yardstick measured beside the pass kernel
// diagnostics: which XCD (accelerator complex die) each workgroup of a launch on this stream lands on -- out[blockIdx] = XCC_ID
__global__ __launch_bounds__(64) void k_xcc_probe(int* __restrict__ out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 0xfu);
}

__global__ __launch_bounds__(256) void k_stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, i64 nvec) {
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (i64)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace rri
