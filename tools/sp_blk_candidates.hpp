// sp_blk_candidates.hpp -- kernels of tools/sp_blk_probe.hip that are NOT in the library: round 2's pass (k_sp_blk_r2) and the
// round-3 candidates that lost or were folded into rri_sparse_kernels.hpp's k_sp_blk (k_sp_blk2: software pipeline across
// segments; k_sp_blk3: the shipped structure with switches for each change that was tried).  Kept so that
// profiles/r03_sp_blk_probe*.log can be reproduced.
#pragma once
#include "rri_sparse_kernels.hpp"

namespace rri {

//   e' = e - (A1[s] * B1[g] + [UPD2] A2[s] * B2[g])            g = block offset + idx[p]
//   WRITE: val[p] = e' (rounded to the storage type; the sums then use the stored value, as the dense pass does)
//   DO_S : S1[blk][s] = sum e' * V[g] ,  S2[blk][s] = sum V[g]^2
// Every segment is padded to a multiple of 4 entries (pad offset SP_PAD), so a lane moves 4 consecutive entries
// per load: 8 bytes of offsets + 16 (fp32) / 32 (fp64) bytes of values, 8 such quads in flight per lane -- about
// 12 KB of reads in flight per wave; with one entry per load the passes ran at a third of the bandwidth, bound by
// the round trips of too few bytes in flight.  One segment per group of LPS lanes; 1024 threads share the tables.
constexpr unsigned short SP_PAD = 0xFFFF;

template <typename SX, bool DO_S, bool UPD2, bool WRITE, int LPS>
__global__ __launch_bounds__(1024) void k_sp_blk_r2(const SpWork* __restrict__ work, const i64* __restrict__ segptr,
                                                 i64 nseg, const unsigned short* __restrict__ idx,
                                                 SX* __restrict__ val, int bw, i64 gdim,
                                                 const double* __restrict__ B1, const double* __restrict__ B2,
                                                 const double* __restrict__ V, const double* __restrict__ A1,
                                                 const double* __restrict__ A2, double* __restrict__ S1,
                                                 double* __restrict__ S2, i64 lds, const DevState* __restrict__ st) {
    typedef typename SpTab<SX>::type TF;
    typedef typename SpQuad<SX>::type V4;
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TF* tb1 = reinterpret_cast<TF*>(smem);   // [bw]
    TF* tb2 = tb1 + bw;                      // [bw]
    TF* tv = tb2 + bw;                       // [bw]
    const SpWork w = work[blockIdx.x];
    const i64 g0 = (i64)w.blk * bw;
    for (int g = threadIdx.x; g < bw; g += 1024) {
        const bool in = g0 + g < gdim;
        tb1[g] = in ? (TF)B1[g0 + g] : TF(0);
        if (UPD2) tb2[g] = in ? (TF)B2[g0 + g] : TF(0);
        if (DO_S) tv[g] = in ? (TF)V[g0 + g] : TF(0);
    }
    __syncthreads();
    constexpr int UNR = 8;
    constexpr int GROUPS = 1024 / LPS;
    const int sub = threadIdx.x % LPS, grp = threadIdx.x / LPS;
    const i64* sp = segptr + (i64)w.blk * (nseg + 1);
    const sp_us4* idx4 = reinterpret_cast<const sp_us4*>(idx);
    V4* val4 = reinterpret_cast<V4*>(val);
    for (int s = w.s0 + grp; s < w.s1; s += GROUPS) {
        const i64 q0 = sp[s] >> 2, q1 = sp[s + 1] >> 2;     // in quads: segment bounds are multiples of 4
        const double c1 = (double)(TF)A1[s];
        const double c2 = UPD2 ? (double)(TF)A2[s] : 0.0;
        double s1 = 0.0, s2 = 0.0;
        for (i64 q = q0 + sub; q < q1; q += (i64)LPS * UNR) {
            sp_us4 g[UNR];
            V4 e[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const i64 qq = q + (i64)u * LPS;
                if (qq < q1) {
                    g[u] = __builtin_nontemporal_load(idx4 + qq);
                    e[u] = __builtin_nontemporal_load(val4 + qq);
                } else {
                    g[u] = sp_us4{SP_PAD, SP_PAD, SP_PAD, SP_PAD};
                    e[u] = V4{SX(0), SX(0), SX(0), SX(0)};
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                V4 out = e[u];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int gi = g[u][m];
                    if (gi == SP_PAD) continue;
                    double corr = c1 * (double)tb1[gi];
                    if (UPD2) corr = fma(c2, (double)tb2[gi], corr);
                    double x = (double)e[u][m] - corr;
                    if (WRITE) {
                        const SX r = (SX)x;
                        out[m] = r;
                        x = (double)r;
                    }
                    if (DO_S) {
                        const double v = (double)tv[gi];
                        s1 = fma(x, v, s1);
                        s2 = fma(v, v, s2);
                    }
                }
                if (WRITE) {
                    const i64 qq = q + (i64)u * LPS;
                    if (qq < q1) __builtin_nontemporal_store(out, val4 + qq);
                }
            }
        }
        if (DO_S) {
            s1 = group_sum<LPS>(s1);
            s2 = group_sum<LPS>(s2);
            if (sub == 0) { S1[(i64)w.blk * lds + s] = s1; S2[(i64)w.blk * lds + s] = s2; }
        }
    }
}

// ---- k_sp_blk2: the same pass, software-pipelined ACROSS segments (round 3) ---------------------------------------------
// What the counters and the ISA of k_sp_blk said (profiles/r02_pmc_sq_c5s.txt): its waves sit on s_waitcnt 67 % of
// their cycles.  The ISA shows why: (i) `if (gi == SP_PAD) continue` became control flow per entry, so the three LDS gathers of
// an entry are issued, waited for (lgkmcnt 2, 1, 0) and consumed before the next entry's are issued -- 32 dependent LDS round
// trips per lane and batch; (ii) a segment of BASELINE's pattern (500 entries = 125 quads) is ONE batch of a 32-lane group, so
// the "two batches in flight" rewrite of round 1 never had a second batch to overlap with: per segment a wave paid, in series,
// the segment bounds (a dependent global load), the quad loads (HBM), the gathers, the arithmetic, and the stores of the batch
// before -- which share the in-order vmcnt counter with the next batch's loads.
// Here: * pads point at a ZERO SLOT of the tables (offset bw; the host builds the copies that way), so the entry loop is
//         branch-free: all gathers of a batch are issued back to back, then the arithmetic;
//       * the bounds of the work item's segments (as 32-bit quad offsets) and their factors are staged in LDS with the
//         tables (<= SP2_MAX_SEGS segments per work item), so stepping to the next segment costs LDS reads, not a global
//         round trip;
//       * a lane group walks its segments as a stream of (segment, chunk) items -- a chunk is LPS x Q quads -- and the quads
//         of item i + 1 are requested BEFORE item i is worked on: the stream stays in flight during the gathers, the float64
//         arithmetic and the write-back of the item before;
//       * no vector-memory instruction of the loop sits under divergent control flow (the compiler jumps over such blocks
//         when no lane is active, and its s_waitcnt placement must then assume the fewest younger operations: the first
//         version waited for a load it had just issued).  Lanes with no quad in a chunk -- the tail of a segment, a group that
//         has run out of items -- load and store a DUMP quad of their own behind the copy (SP2_DUMP_QUADS quads of pads:
//         zero values, offsets at the zero slot; a set per work item), which takes them through the same instructions with no effect.
// Same arithmetic per entry, same order of a segment's partial sums per lane (chunks in order, quads u = 0 .. Q-1, entries
// m = 0 .. 3), same group_sum: bit-identical values and sums to k_sp_blk with LPS and the quads per lane equal.
constexpr int SP2_MAX_SEGS = 1024;
constexpr int SP2_DUMP_QUADS = 1024;         // per work item, one per thread: appended to idx / val of a copy
template <typename SX>
__host__ __device__ constexpr size_t sp2_lds_bytes(int bw) {
    typedef typename SpTab<SX>::type TF;
    return (((size_t)3 * (bw + 1) * sizeof(TF) + 15) / 16) * 16 + (size_t)(SP2_MAX_SEGS + 4) * 4 + (size_t)2 * SP2_MAX_SEGS * sizeof(TF);
}

template <typename SX, bool DO_S, bool UPD2, bool WRITE, int LPS, int Q, int MEM = 0>
__global__ __launch_bounds__(1024) void k_sp_blk2(const SpWork* __restrict__ work, const i64* __restrict__ segptr,
                                                  i64 nseg, const unsigned short* __restrict__ idx,
                                                  SX* __restrict__ val, i64 count, int bw, i64 gdim,
                                                  const double* __restrict__ B1, const double* __restrict__ B2,
                                                  const double* __restrict__ V, const double* __restrict__ A1,
                                                  const double* __restrict__ A2, double* __restrict__ S1,
                                                  double* __restrict__ S2, i64 lds, const DevState* __restrict__ st) {
    typedef typename SpTab<SX>::type TF;
    typedef typename SpQuad<SX>::type V4;
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bw1 = bw + 1;                  // slot bw: the zero entry the pads point at
    TF* tb1 = reinterpret_cast<TF*>(smem);
    TF* tb2 = tb1 + bw1;
    TF* tv = tb2 + bw1;
    unsigned* qrel = reinterpret_cast<unsigned*>(smem + (((size_t)3 * bw1 * sizeof(TF) + 15) / 16) * 16);   // [nsg + 1]
    TF* a1s = reinterpret_cast<TF*>(qrel + SP2_MAX_SEGS + 4);
    TF* a2s = a1s + SP2_MAX_SEGS;
    const SpWork w = work[blockIdx.x];
    const i64 g0 = (i64)w.blk * bw;
    const int nsg = w.s1 - w.s0;
    const i64* sp = segptr + (i64)w.blk * (nseg + 1) + w.s0;
    const i64 p0 = sp[0];
    double* S1w = S1 + (i64)w.blk * lds + w.s0;
    double* S2w = S2 + (i64)w.blk * lds + w.s0;
    for (int g = threadIdx.x; g < bw1; g += 1024) {
        const bool in = g < bw && g0 + g < gdim;
        tb1[g] = in ? (TF)B1[g0 + g] : TF(0);
        if (UPD2) tb2[g] = in ? (TF)B2[g0 + g] : TF(0);
        if (DO_S) tv[g] = in ? (TF)V[g0 + g] : TF(0);
    }
    for (int i = threadIdx.x; i <= nsg; i += 1024) qrel[i] = (unsigned)((sp[i] - p0) >> 2);
    for (int i = threadIdx.x; i < nsg; i += 1024) {
        a1s[i] = (TF)A1[w.s0 + i];
        if (UPD2) a2s[i] = (TF)A2[w.s0 + i];
        if (DO_S && sp[i + 1] == sp[i]) { S1w[i] = 0.0; S2w[i] = 0.0; }       // an empty segment still reports its (zero) sums
    }
    __syncthreads();
    constexpr int GROUPS = 1024 / LPS;
    constexpr unsigned CHUNK = (unsigned)LPS * Q;
    const int sub = threadIdx.x % LPS, grp = threadIdx.x / LPS;
    const sp_us4* idx4 = reinterpret_cast<const sp_us4*>(idx) + (p0 >> 2);
    V4* val4 = reinterpret_cast<V4*>(val) + (p0 >> 2);
    // this thread's dump quad (one set per WORK ITEM: 775 workgroups storing to one shared set ran 3.7x slower than the shipped
    // kernel -- same-line write contention at the memory side), relative like the rest
    const unsigned dump = (unsigned)((count - p0) >> 2) + blockIdx.x * (unsigned)SP2_DUMP_QUADS + threadIdx.x;

    // the group's stream of items: (segment i, chunk [c, min(c + CHUNK, e)) of its quads), empty segments skipped
    int i = grp;
    unsigned c = 0, e = 0;
    auto open_segment = [&]() {              // first non-empty segment at or after i (stride GROUPS); none: c = e = 0
        c = 0; e = 0;
        while (i < nsg) {
            c = qrel[i]; e = qrel[i + 1];
            if (e > c) return;
            i += GROUPS;
        }
        c = 0; e = 0;
    };
    open_segment();
    auto request = [&](unsigned cc, unsigned ee, sp_us4 (&gg)[Q], V4 (&vv)[Q]) {
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const unsigned q = cc + (unsigned)(u * LPS + sub);
            const unsigned qq = q < ee ? q : dump;
            if (MEM & 2) { gg[u] = idx4[qq]; vv[u] = val4[qq]; }
            else { gg[u] = __builtin_nontemporal_load(idx4 + qq); vv[u] = __builtin_nontemporal_load(val4 + qq); }
        }
    };
    double s1 = 0.0, s2 = 0.0;
    // one item: request the NEXT item's quads into (gn, evn), then work on this item's (g, ev), requested one step
    // earlier.  Two register sets that swap roles from step to step (the loop below is unrolled by two): no copies, so the
    // only waits are for the set that is about to be used, with the younger requests and the stores still in flight.
    auto step = [&](sp_us4 (&g)[Q], V4 (&ev)[Q], sp_us4 (&gn)[Q], V4 (&evn)[Q]) {
        const int ic = i;
        const unsigned cc = c, ec = e;
        const bool last = cc + CHUNK >= ec;  // this chunk ends its segment
        if (last) {
            i += GROUPS;
            open_segment();
        } else {
            c += CHUNK;
        }
        request(c, e, gn, evn);              // a group past its last item: c = e = 0, every lane takes its dump quad
        __builtin_amdgcn_sched_barrier(0);   // the requests leave BEFORE the work on this item (the scheduler would sink them)
        const double c1 = (double)a1s[ic];
        const double c2 = UPD2 ? (double)a2s[ic] : 0.0;
        // The gathers of quad u + 1 are issued before the arithmetic of quad u: the LDS (12 gathers per quad, ~2.4-way
        // conflicts on random offsets) and the float64 pipe work at the same time inside ONE wave.  With all 12 Q gathers
        // of an item issued first and the arithmetic after them, every wave alternated between an LDS-only and an ALU-only
        // phase and the two units' times added up (LDS ~45 us + ALU ~45 us + stream per launch; profiles/r03_sp_blk_probe.log).
        TF f1[2][4], f2[2][4], fv[2][4];
        auto gather = [&](int u, int b) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int gi = (int)g[u][m];
                f1[b][m] = tb1[gi];
                if (UPD2) f2[b][m] = tb2[gi];
                if (DO_S) fv[b][m] = tv[gi];
            }
        };
        gather(0, 0);
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const int b = u & 1;
            if (u + 1 < Q) gather(u + 1, b ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            V4 out = ev[u];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                double corr = c1 * (double)f1[b][m];
                if (UPD2) corr = fma(c2, (double)f2[b][m], corr);
                double x = (double)ev[u][m] - corr;
                if (WRITE) {
                    const SX r = (SX)x;
                    out[m] = r;
                    x = (double)r;
                }
                if (DO_S) {
                    const double v = (double)fv[b][m];
                    s1 = fma(x, v, s1);
                    s2 = fma(v, v, s2);
                }
            }
            if (WRITE && !(MEM & 4)) {
                const unsigned q = cc + (unsigned)(u * LPS + sub);
                if (MEM & 1) val4[q < ec ? q : dump] = out;
                else __builtin_nontemporal_store(out, val4 + (q < ec ? q : dump));
            }
            if (MEM & 4) ev[u] = out;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (WRITE && (MEM & 4)) {
#pragma unroll
            for (int u = 0; u < Q; ++u) {
                const unsigned q = cc + (unsigned)(u * LPS + sub);
                if (MEM & 1) val4[q < ec ? q : dump] = ev[u];
                else __builtin_nontemporal_store(ev[u], val4 + (q < ec ? q : dump));
            }
        }
        if (last && DO_S) {
            const double t1 = group_sum<LPS>(s1), t2 = group_sum<LPS>(s2);
            if (sub == 0) { S1w[ic] = t1; S2w[ic] = t2; }
            s1 = 0.0; s2 = 0.0;
        }
    };
    sp_us4 gA[Q], gB[Q];
    V4 eA[Q], eB[Q];
    request(c, e, gA, eA);
    if (WRITE) {
        // Q stores of zeros to the own dump quad: harmless, and they make the loop's entry look like its back edge to the
        // compiler's s_waitcnt placement (vector-memory operations complete in order: with Q stores younger than the first
        // requests on BOTH ways into the loop the first wait becomes vmcnt(Q + ...) instead of vmcnt(0), which on the back
        // edge would also wait for the write-back of the item before)
#pragma unroll
        for (int u = 0; u < Q; ++u) __builtin_nontemporal_store(V4{SX(0), SX(0), SX(0), SX(0)}, val4 + dump);
    }
    while (i < nsg) {
        step(gA, eA, gB, eB);
        if (!(i < nsg)) break;
        step(gB, eB, gA, eA);
    }
}

// ---- k_sp_blk3: k_sp_blk with less work per entry (round 3) -----------------------------------------------------------
// The probe (tools/sp_blk_probe.hip, profiles/r03_sp_blk_probe.log) shows the pass's time to be the SUM of what its units
// need -- stream ~45 us + LDS gathers ~40 us + vector ALU ~45 us at 5e7 entries -- for the shipped kernel and for the
// cross-segment pipeline above alike, so what shortens it is less work per entry:
//   * pads point at a zero slot (offset bw) instead of being branched around: no control flow per entry;
//   * {b1, b2} of an offset sit side by side in ONE table (PK): one 8-byte gather (ds_read_b64: same bank conflicts as a
//     4-byte one) instead of two -- two LDS instructions per entry instead of three;
//   * fp32 storage (F32C): the two corrections as two fp32 fused multiply-adds on the fp32 factors, e' = fma(-c2, b2,
//     fma(-c1, b1, e)) -- instead of three conversions, a float64 multiply, a float64 fma, a float64 subtraction and the
//     rounding back (each float64-rate instruction costs ~5 cycles per wave here, profiles/r01_alu_probe.txt).  The stored
//     value can differ from the float64 form's by one fp32 ulp of the LARGEST of the three terms (two roundings instead of
//     one); the sums are still float64.
template <typename SX, bool DO_S, bool UPD2, bool WRITE, int LPS, bool PK, bool F32C, int MEM = 0>
__global__ __launch_bounds__(1024) void k_sp_blk3(const SpWork* __restrict__ work, const i64* __restrict__ segptr,
                                                  i64 nseg, const unsigned short* __restrict__ idx,
                                                  SX* __restrict__ val, int bw, i64 gdim,
                                                  const double* __restrict__ B1, const double* __restrict__ B2,
                                                  const double* __restrict__ V, const double* __restrict__ A1,
                                                  const double* __restrict__ A2, double* __restrict__ S1,
                                                  double* __restrict__ S2, i64 lds, const DevState* __restrict__ st) {
    typedef typename SpTab<SX>::type TF;
    typedef typename SpQuad<SX>::type V4;
    static_assert(!F32C || sizeof(SX) == 4, "fp32 corrections are for fp32 storage");
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bw1 = bw + 1;                  // slot bw: the zero entry the pads point at
    TF* tb1 = reinterpret_cast<TF*>(smem);   // PK: [bw1][2] = {b1, b2}; else [bw1] b1, then [bw1] b2
    TF* tb2 = tb1 + bw1;
    TF* tv = tb1 + 2 * (size_t)bw1;          // [bw1]
    const SpWork w = work[blockIdx.x];
    const i64 g0 = (i64)w.blk * bw;
    for (int g = threadIdx.x; g < bw1; g += 1024) {
        const bool in = g < bw && g0 + g < gdim;
        const TF b1v = in ? (TF)B1[g0 + g] : TF(0);
        const TF b2v = (UPD2 && in) ? (TF)B2[g0 + g] : TF(0);
        if (PK) { tb1[2 * g] = b1v; tb1[2 * g + 1] = b2v; }
        else { tb1[g] = b1v; if (UPD2) tb2[g] = b2v; }
        if (DO_S) tv[g] = in ? (TF)V[g0 + g] : TF(0);
    }
    __syncthreads();
    // MEM & 24: waves 8..15 start 1 / 2 / 3 x ~1.7 us late (diagnostics: the 16 waves of the workgroup otherwise run their
    // wait-for-the-stream / gather / arithmetic phases in step with each other)
    if ((MEM & 24) && threadIdx.x >= 512) {
        for (int r = 0; r < ((MEM >> 3) & 3); ++r) __builtin_amdgcn_s_sleep(64);
    }
    constexpr int UNR = 8;
    constexpr int GROUPS = 1024 / LPS;
    const int sub = threadIdx.x % LPS, grp = threadIdx.x / LPS;
    const i64* sp = segptr + (i64)w.blk * (nseg + 1);
    const sp_us4* idx4 = reinterpret_cast<const sp_us4*>(idx);
    V4* val4 = reinterpret_cast<V4*>(val);
    typedef TF TF2 __attribute__((ext_vector_type(2)));
    const TF2* tb12 = reinterpret_cast<const TF2*>(tb1);
    // MEM & 32: the bounds and factors of the group's NEXT segment are requested while this one is worked on (they are a
    // dependent round trip to L2 in front of every segment's quads otherwise)
    i64 pq0 = 0, pq1 = 0;
    double pa1 = 0.0, pa2 = 0.0;
    if ((MEM & 32) && w.s0 + grp < w.s1) { pq0 = sp[w.s0 + grp]; pq1 = sp[w.s0 + grp + 1]; pa1 = A1[w.s0 + grp]; pa2 = UPD2 ? A2[w.s0 + grp] : 0.0; }
    for (int s = w.s0 + grp; s < w.s1; s += GROUPS) {
        i64 q0, q1;
        TF c1f, c2f;
        if (MEM & 32) {
            q0 = pq0 >> 2; q1 = pq1 >> 2; c1f = (TF)pa1; c2f = (TF)pa2;
            const int sn = s + GROUPS;
            if (sn < w.s1) { pq0 = sp[sn]; pq1 = sp[sn + 1]; pa1 = A1[sn]; pa2 = UPD2 ? A2[sn] : 0.0; }
        } else {
            q0 = sp[s] >> 2; q1 = sp[s + 1] >> 2;
            c1f = (TF)A1[s]; c2f = UPD2 ? (TF)A2[s] : TF(0);
        }
        const double c1 = (double)c1f, c2 = (double)c2f;
        double s1 = 0.0, s2 = 0.0;
        for (i64 q = q0 + sub; q < q1; q += (i64)LPS * UNR) {
            sp_us4 g[UNR];
            V4 e[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const i64 qq = q + (i64)u * LPS;
                if (qq < q1) {
                    if (MEM & 2) { g[u] = idx4[qq]; e[u] = val4[qq]; }
                    else { g[u] = __builtin_nontemporal_load(idx4 + qq); e[u] = __builtin_nontemporal_load(val4 + qq); }
                } else {
                    const unsigned short z = (unsigned short)bw;
                    g[u] = sp_us4{z, z, z, z};
                    e[u] = V4{SX(0), SX(0), SX(0), SX(0)};
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const i64 qq = q + (i64)u * LPS;
                if (qq >= q1) break;                         // group-uniform up to the segment's last batch
                V4 out = e[u];
                TF f1[4], f2[4], fv[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int gi = g[u][m];
                    if (PK) { const TF2 p = tb12[gi]; f1[m] = p[0]; f2[m] = p[1]; }
                    else { f1[m] = tb1[gi]; f2[m] = UPD2 ? tb2[gi] : TF(0); }
                    fv[m] = DO_S ? tv[gi] : TF(0);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    double x;
                    if constexpr (F32C) {
                        float xf = __builtin_fmaf(-c1f, f1[m], e[u][m]);
                        if (UPD2) xf = __builtin_fmaf(-c2f, f2[m], xf);
                        out[m] = xf;
                        x = (double)xf;
                    } else {
                        double corr = c1 * (double)f1[m];
                        if (UPD2) corr = fma(c2, (double)f2[m], corr);
                        x = (double)e[u][m] - corr;
                        if (WRITE) {
                            const SX r = (SX)x;
                            out[m] = r;
                            x = (double)r;
                        }
                    }
                    if (DO_S) {
                        const double v = (double)fv[m];
                        s1 = fma(x, v, s1);
                        s2 = fma(v, v, s2);
                    }
                }
                if (WRITE) {
                    if (MEM & 1) val4[qq] = out;
                    else __builtin_nontemporal_store(out, val4 + qq);
                }
            }
        }
        if (DO_S) {
            s1 = group_sum<LPS>(s1);
            s2 = group_sum<LPS>(s2);
            if (sub == 0) { S1[(i64)w.blk * lds + s] = s1; S2[(i64)w.blk * lds + s] = s2; }
        }
    }
}


}  // namespace rri
