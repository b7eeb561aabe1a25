// rri_sparse_kernels.hpp -- the weighted flavour (WRRI, nmf.py:687-701, :735-746) when W_mat is a 0/1
// observation pattern given as CSR: the recommender case (sklearn_interface.py:78-102), where only the
// observed entries of X - W T ever enter a sum.  Same algebra and the same schedule as rri_wrri_kernels.hpp
// (maintained masked residual E, pass B / pass C per topic step), but E lives on the pattern only:
//
//   CSR copy  (rowptr, col, e )   row products      b_i = sum_j e'_ij t_j ,  nt_i = sum_j t_j^2     (j in row i)
//   CSC copy  (colptr, row, ec)   column sums       a_j = sum_i w_i e'_ij ,  nw_j = sum_i w_i^2     (i in column j)
//
// Both copies take every rank-one correction e' = e - (a1_i b1_j + a2_i b2_j) with the same operations in
// the same order, so they stay bit-identical and every sum has a fixed order (no atomics).  One kernel serves
// both orientations: a "segment" is a row of the CSR copy or a column of the CSC copy, the per-segment factors
// come from one vector pair, the per-entry factors from LDS tables (k_sp_blk below; gathering them from L2
// instead ran at 1.3-1.7 TB/s of index / value traffic, bound by L2 sector bandwidth).
//
// Bytes per observed entry and topic step (fp32 values, uint16 block offsets): pass B 6 (CSR read), pass C
// 10 + 10 (read + write of both copies) = 26 B, against 12.25 B per DENSE entry of the bit-packed dense
// schedule: the sparse formulation moves fewer bytes below ~47 % density (at the 5 % of BASELINE config 5: 9x).
#pragma once
#include "rri_kernels.hpp"

namespace rri {

// Blocked segment store.  A copy of the pattern is cut into blocks along its GATHER dimension (column blocks
// of the CSR copy, row blocks of the CSC copy), at most SP_BLOCK_BYTES / (3 * sizeof(TF)) wide, so that the three
// factor vectors an entry needs {b1, b2, v}[gather index] sit in LDS for the whole block (10240 entries of fp32
// factors for an fp32 handle, 5120 of fp64 for an fp64 handle) and the per-entry index is a uint16 offset into
// the block.  Entries are ordered (block, segment, offset); segptr[block][segment] delimits them.  The sums of a
// segment come out per block -- S[block][segment] -- and the consumers that already add row-dot panels
// (k_wwcol) or row-block partials (k_reduce) add the blocks in a fixed order.
//
// Factors are rounded to the table type TF on BOTH sides (the per-segment scalars too), so the CSR and the CSC
// copy apply bit-identical corrections; for an fp32 handle that is the rounding the stored residual has anyway.
constexpr int SP_BLOCK_BYTES = 120 * 1024;
template <typename SX> struct SpTab { typedef float type; };
template <> struct SpTab<double> { typedef double type; };

struct SpWork { int blk, s0, s1, pad; };   // one workgroup: segments [s0, s1) of block blk

template <int LPS>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = LPS >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

//   e' = e - (A1[s] * B1[g] + [UPD2] A2[s] * B2[g])            g = block offset + idx[p]
//   WRITE: val[p] = e' (rounded to the storage type; the sums then use the stored value, as the dense pass does)
//   DO_S : S1[blk][s] = sum e' * V[g] ,  S2[blk][s] = sum V[g]^2
// Every segment is padded to a multiple of 4 entries, so a lane moves 4 consecutive entries per load: 8 bytes of offsets +
// 16 (fp32) / 32 (fp64) bytes of values, up to 8 such quads in flight per lane; with one entry per load the passes ran at
// a third of the bandwidth.  One segment per group of LPS lanes; 1024 threads share the tables.
//
// Round 3 (tools/sp_blk_probe.hip: this kernel, round 2's and the candidates that lost, outside the library on BASELINE
// config 5's pattern; profiles/r03_sp_blk_probe*.log): 141 -> 100 us per pass at 5e7 entries, 0.44 -> 0.63 of 8 TB/s.
//   * ONE ROUND OF WORKGROUPS.  The work list had ~3 x 256 items "of equal entry count" -- 774 to 780 in fact: three full
//     rounds of the 256 CUs (one 1024-thread workgroup each, the LDS tables) and a fourth for the last 6 to 12 items, a
//     quarter of the launch with 250 CUs idle.  The host now cuts at most n_cu items (rri_upload_observed_csr): 138 -> 115 us.
//   * PLAIN loads and stores instead of non-temporal ones: 115 -> 100 us (both copies; the read-modify-write of a line it
//     has just loaded is what the L2 is for).
//   * pads point at a ZERO SLOT of the tables (offset bw, written by the host when it builds the copies) instead of being
//     branched around: `if (gi == SP_PAD) continue` had become control flow per ENTRY, each entry's gathers issued, waited
//     for and consumed before the next entry's were issued (the ISA: lgkmcnt 2, 1, 0 per entry).  -8 % at 774 items.
//   * {b1, b2} of an offset side by side in one table: one 8-byte gather (ds_read_b64, the bank behaviour of a 4-byte one)
//     instead of two; SQ_LDS_BANK_CONFLICT 11.9e6 -> 8.0e6, SQ_LDS_IDX_ACTIVE 22.1e6 -> 16.8e6 cycles per launch, time
//     unchanged within the noise -- the LDS was not what the pass waited for.
// What did NOT help, measured on the same copies: a software pipeline across segments (the next segment's quads requested
// before this one is worked on, bounds and factors staged in LDS, dump quads for the idle lanes so that no memory
// instruction sits under divergent control flow: waves parked on s_waitcnt 67 % -> 20 %, but stalled at ISSUE 66 %, 150-175
// us); fp32 fused multiply-adds for the two corrections (-25 % vector-ALU instructions, 129.7 against 130.8 us, and a
// quarter of the stored values an ulp away); requesting the next segment's bounds ahead; starting half of the waves late.
template <typename SX> struct SpPair;
template <> struct SpPair<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct SpPair<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename SX> struct SpQuad;
template <> struct SpQuad<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct SpQuad<double> { typedef double type __attribute__((ext_vector_type(4))); };
typedef unsigned short sp_us4 __attribute__((ext_vector_type(4)));
// dynamic LDS of k_sp_blk: {b1, b2} pairs and v for bw + 1 offsets (the last one: the zero slot of the pads)
template <typename SX>
__host__ __device__ constexpr size_t sp_lds_bytes(int bw) { return (size_t)3 * (bw + 1) * sizeof(typename SpTab<SX>::type); }

template <typename SX, bool DO_S, bool UPD2, bool WRITE, int LPS>
__global__ __launch_bounds__(1024) void k_sp_blk(const SpWork* __restrict__ work, const i64* __restrict__ segptr,
                                                 i64 nseg, const unsigned short* __restrict__ idx,
                                                 SX* __restrict__ val, int bw, i64 gdim,
                                                 const double* __restrict__ B1, const double* __restrict__ B2,
                                                 const double* __restrict__ V, const double* __restrict__ A1,
                                                 const double* __restrict__ A2, double* __restrict__ S1,
                                                 double* __restrict__ S2, i64 lds, const DevState* __restrict__ st) {
    typedef typename SpTab<SX>::type TF;
    typedef typename SpPair<TF>::type TF2;
    typedef typename SpQuad<SX>::type V4;
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bw1 = bw + 1;                        // slot bw: the zero entry the pads point at
    TF2* tb12 = reinterpret_cast<TF2*>(smem);      // [bw1] {b1, b2}
    TF* tv = reinterpret_cast<TF*>(tb12 + bw1);    // [bw1]
    const SpWork w = work[blockIdx.x];
    const i64 g0 = (i64)w.blk * bw;
    for (int g = threadIdx.x; g < bw1; g += 1024) {
        const bool in = g < bw && g0 + g < gdim;
        TF2 p;
        p[0] = in ? (TF)B1[g0 + g] : TF(0);
        p[1] = (UPD2 && in) ? (TF)B2[g0 + g] : TF(0);
        tb12[g] = p;
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
#ifdef RRI_SP_MEM      /* experiments (tools/sp_mem_ab.sh): bit 0 non-temporal value loads, bit 1 offset loads, bit 2 stores */
                    g[u] = (RRI_SP_MEM & 2) ? __builtin_nontemporal_load(idx4 + qq) : idx4[qq];
                    e[u] = (RRI_SP_MEM & 1) ? __builtin_nontemporal_load(val4 + qq) : val4[qq];
#else
                    g[u] = idx4[qq];
                    e[u] = val4[qq];
#endif
                } else {
                    const unsigned short z = (unsigned short)bw;
                    g[u] = sp_us4{z, z, z, z};
                    e[u] = V4{SX(0), SX(0), SX(0), SX(0)};
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const i64 qq = q + (i64)u * LPS;
                if (qq >= q1) break;
                V4 out = e[u];
                TF2 f12[4];
                TF fv[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {                // the gathers of a quad first, then its arithmetic
                    const int gi = g[u][m];
                    f12[m] = tb12[gi];
                    fv[m] = DO_S ? tv[gi] : TF(0);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    double corr = c1 * (double)f12[m][0];
                    if (UPD2) corr = fma(c2, (double)f12[m][1], corr);
                    double x = (double)e[u][m] - corr;
                    if (WRITE) {
                        const SX r = (SX)x;
                        out[m] = r;
                        x = (double)r;
                    }
                    if (DO_S) {
                        const double v = (double)fv[m];
                        s1 = fma(x, v, s1);
                        s2 = fma(v, v, s2);
                    }
                }
#ifdef RRI_SP_MEM
                if (WRITE) { if (RRI_SP_MEM & 4) __builtin_nontemporal_store(out, val4 + qq); else val4[qq] = out; }
#else
                if (WRITE) val4[qq] = out;
#endif
            }
        }
        if (DO_S) {
            s1 = group_sum<LPS>(s1);
            s2 = group_sum<LPS>(s2);
            if (sub == 0) { S1[(i64)w.blk * lds + s] = s1; S2[(i64)w.blk * lds + s] = s2; }
        }
    }
}

// Residual on the pattern (the sparse counterpart of k_resid): r_ij = x_ij - W[i,:] . T[:,j] for the entries of
// row i.  One wave per row, 8 lanes per entry (lane q of a group takes the topics q, q+8, ...: one 64-byte
// read of the TRANSPOSED T per step), 8 entries per wave in flight.  Tt: d x kp (kp = k rounded up to 8, pad 0).
//   E      != NULL: E[p] = r                                   (refresh of the maintained residual)
//   rowobj != NULL: rowobj[i] = sum_j r^2                      (true_objective, nmf.py:71-94, on the mask)
//   rowpos != NULL: rowpos[i] = sum_j max(r, 0)^2              (reset search, nmf.py:770-773: outside the
//                                                               pattern x = 0 and max(0 - WT, 0) = 0)
template <typename SX>
__global__ __launch_bounds__(256) void k_sp_resid(const i64* __restrict__ rowptr, const int* __restrict__ col,
                                                  const SX* __restrict__ xval, i64 n, const double* __restrict__ Wt,
                                                  i64 ldw, const double* __restrict__ Tt, int k, int kp,
                                                  SX* __restrict__ E, double* __restrict__ rowobj,
                                                  double* __restrict__ rowpos) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* wsh = reinterpret_cast<double*>(smem);   // [4][kp]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 i = (i64)blockIdx.x * 4 + wave;
    const bool live = i < n;
    for (int l = lane; l < kp; l += 64) wsh[wave * kp + l] = (live && l < k) ? Wt[(i64)l * ldw + i] : 0.0;
    __syncthreads();
    const int q = lane & 7, slot = lane >> 3;
    const double* wrow = wsh + wave * kp;
    double obj = 0.0, pos = 0.0;
    const i64 p0 = live ? rowptr[i] : 0, p1 = live ? rowptr[i + 1] : 0;
    for (i64 pb = p0; pb < p1; pb += 8) {
        const i64 p = pb + slot;
        const bool ok = p < p1;
        const int j = ok ? col[p] : 0;
        const double* trow = Tt + (i64)j * kp;
        double a0 = 0.0, a1 = 0.0;
        int l = q;
        for (; l + 8 < kp; l += 16) {
            a0 = fma(wrow[l], trow[l], a0);
            a1 = fma(wrow[l + 8], trow[l + 8], a1);
        }
        if (l < kp) a0 = fma(wrow[l], trow[l], a0);
        double acc = a0 + a1;
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (ok && q == 0) {
            const double r = (double)xval[p] - acc;
            if (E) E[p] = (SX)r;
            obj = fma(r, r, obj);
            const double rp = fmax(r, 0.0);
            pos = fma(rp, rp, pos);
        }
    }
    if (rowobj || rowpos) {
        obj = wave_sum<double>(obj);
        pos = wave_sum<double>(pos);
        if (lane == 0 && live) {
            if (rowobj) rowobj[i] = obj;
            if (rowpos) rowpos[i] = pos;
        }
    }
}

// a blocked copy of the residual after a refresh: out[q] = e[perm[q]] (perm < 0: padding of a segment, kept 0)
template <typename SX>
__global__ __launch_bounds__(256) void k_sp_permute(const SX* __restrict__ e, const int* __restrict__ perm, i64 count,
                                                    SX* __restrict__ out) {
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < count; q += (i64)gridDim.x * 256) {
        const int p = perm[q];
        out[q] = p >= 0 ? e[p] : SX(0);
    }
}

// Sparse-times-dense products with the X on the pattern, for the randomized SVD behind the NNDSVD start of the
// recommender flavour (initialization.py:104-105 on W_mat .* X; 16 such products, 0.5 s each in scipy at 5e7 entries):
//   out[blk][seg][v] = sum over the entries p of segment (blk, seg) of x_p * B[blk * bw + idx[p]][v]      v < m <= 64
// on a blocked copy of the pattern (rows as segments: X B; columns as segments: X^T Q).  One wave per segment, lane =
// column v of B: every entry costs one coalesced row of B (L2-resident); x_p comes through the copy's permutation
// from the canonical CSR values.  The blocks of a segment are added by k_sp_sum_blocks.
template <typename SX>
__global__ __launch_bounds__(256) void k_sp_spmm(const i64* __restrict__ segptr, i64 nseg, int nblk,
                                                 const unsigned short* __restrict__ idx, const int* __restrict__ perm,
                                                 const SX* __restrict__ xcanon, int bw, const double* __restrict__ B,
                                                 int m, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const i64 wid = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (i64)nblk * nseg) return;
    const int blk = (int)(wid / nseg);
    const i64 seg = wid - (i64)blk * nseg;
    const i64 p0 = segptr[(i64)blk * (nseg + 1) + seg], p1 = segptr[(i64)blk * (nseg + 1) + seg + 1];
    const double* Bb = B + (i64)blk * bw * m;
    double acc = 0.0;
    for (i64 p = p0; p < p1; p += 4) {                 // segments are padded to multiples of 4 (SP_PAD, perm < 0)
        int off[4], pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { off[u] = idx[p + u]; pp[u] = perm[p + u]; }
        double xv[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = pp[u] >= 0;
            xv[u] = ok ? (double)xcanon[pp[u]] : 0.0;
            bv[u] = (ok && lane < m) ? Bb[(i64)off[u] * m + lane] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = fma(xv[u], bv[u], acc);
    }
    if (lane < m) out[wid * m + lane] = acc;
}

// res[seg][v] = sum_blk part[blk][seg][v], blocks in order
__global__ __launch_bounds__(256) void k_sp_sum_blocks(const double* __restrict__ part, i64 count, int nblk,
                                                       double* __restrict__ res) {
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    res[e] = ordered_sum<8>(part + e, count, 0, nblk, 1);
}

// the reset row max(X[mi,:] - W[mi,:] T, 0) (nmf.py:770-775) from the pattern of row mi = *row_idx; out has d
// entries and was zeroed by the caller
template <typename SX>
__global__ __launch_bounds__(256) void k_sp_reset_row(const i64* __restrict__ rowptr, const int* __restrict__ col,
                                                      const SX* __restrict__ xval, const double* __restrict__ Wt,
                                                      i64 ldw, const double* __restrict__ T, i64 ldt, int k,
                                                      const i64* __restrict__ row_idx, double* __restrict__ out) {
    const i64 mi = *row_idx;
    const i64 p = rowptr[mi] + (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= rowptr[mi + 1]) return;
    const int j = col[p];
    double acc = 0.0;
    for (int l = 0; l < k; ++l) acc = fma(Wt[(i64)l * ldw + mi], T[(i64)l * ldt + j], acc);
    out[j] = fmax((double)xval[p] - acc, 0.0);
}

}  // namespace rri
