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
// come from one vector pair, the per-entry factors are gathered through the index array from a packed table
// {B1, B2, V, -} (one 32-byte gather per entry instead of three 8-byte ones).
//
// Bytes per observed entry and topic step (fp32 values, int32 indices): pass B 8 (CSR read), pass C 12 + 12
// (read + write of both copies) = 32 B, against 12.25 B per DENSE entry of the bit-packed dense schedule:
// the sparse formulation moves fewer bytes below ~38 % density (at the 5 % of BASELINE config 5: 7.6x fewer).
#pragma once
#include "rri_kernels.hpp"

namespace rri {

struct __attribute__((aligned(32))) SpGather { double b1, b2, v, pad; };

// G[g] = {B1[g], B2[g], V[g]}; NULL vectors read as 0
__global__ __launch_bounds__(256) void k_sp_pack(const double* __restrict__ B1, const double* __restrict__ B2,
                                                 const double* __restrict__ V, i64 m, SpGather* __restrict__ G,
                                                 const DevState* __restrict__ st) {
    if (st->halt) return;
    const i64 g = (i64)blockIdx.x * 256 + threadIdx.x;
    if (g >= m) return;
    SpGather o;
    o.b1 = B1 ? B1[g] : 0.0;
    o.b2 = B2 ? B2[g] : 0.0;
    o.v = V ? V[g] : 0.0;
    o.pad = 0.0;
    G[g] = o;
}

// sum over the LPS lanes (a power of two <= 64) of a segment group; every lane of the group gets the total
template <int LPS>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = LPS >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One segment (row of the CSR copy / column of the CSC copy) per group of LPS lanes, 4 entries per lane in flight.
//   e' = e - (A1[s] * G[g].b1 + [UPD2] A2[s] * G[g].b2)        g = idx[p]
//   WRITE: val[p] = e' (rounded to the storage type; the sums then use the stored value, as the dense pass does)
//   DO_S : S1[s] = sum e' * G[g].v ,  S2[s] = sum G[g].v^2
template <typename SX, bool DO_S, bool UPD2, bool WRITE, int LPS>
__global__ __launch_bounds__(256) void k_sp_seg(const i64* __restrict__ ptr, const int* __restrict__ idx,
                                                SX* __restrict__ val, i64 nseg, const double* __restrict__ A1,
                                                const double* __restrict__ A2, const SpGather* __restrict__ G,
                                                double* __restrict__ S1, double* __restrict__ S2,
                                                const DevState* __restrict__ st) {
    if (st->halt) return;
    constexpr int UNR = 4;
    const int sub = threadIdx.x % LPS;
    const i64 seg = ((i64)blockIdx.x * 256 + threadIdx.x) / LPS;
    i64 p0 = 0, p1 = 0;
    double c1 = 0.0, c2 = 0.0;
    if (seg < nseg) {
        p0 = ptr[seg];
        p1 = ptr[seg + 1];
        c1 = A1[seg];
        if (UPD2) c2 = A2[seg];
    }
    double s1 = 0.0, s2 = 0.0;
    for (i64 p = p0 + sub; p < p1; p += (i64)LPS * UNR) {
        int g[UNR];
        SX e[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const i64 q = p + (i64)u * LPS;
            const bool ok = q < p1;
            g[u] = ok ? __builtin_nontemporal_load(idx + q) : -1;
            e[u] = ok ? __builtin_nontemporal_load(val + q) : SX(0);
        }
        SpGather t[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (g[u] >= 0) t[u] = G[g[u]];
            else { t[u].b1 = 0.0; t[u].b2 = 0.0; t[u].v = 0.0; t[u].pad = 0.0; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            double corr = c1 * t[u].b1;
            if (UPD2) corr = fma(c2, t[u].b2, corr);
            double x = (double)e[u] - corr;
            if (WRITE) {
                const SX r = (SX)x;
                if (g[u] >= 0) __builtin_nontemporal_store(r, val + p + (i64)u * LPS);
                x = (double)r;
            }
            if (DO_S && g[u] >= 0) {
                s1 = fma(x, t[u].v, s1);
                s2 = fma(t[u].v, t[u].v, s2);
            }
        }
    }
    if (DO_S) {
        s1 = group_sum<LPS>(s1);
        s2 = group_sum<LPS>(s2);
        if (sub == 0 && seg < nseg) { S1[seg] = s1; S2[seg] = s2; }
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

// the CSC copy of the residual after a refresh: ec[p] = e[perm[p]]
template <typename SX>
__global__ __launch_bounds__(256) void k_sp_permute(const SX* __restrict__ e, const int* __restrict__ perm, i64 nnz,
                                                    SX* __restrict__ ec) {
    for (i64 p = (i64)blockIdx.x * 256 + threadIdx.x; p < nnz; p += (i64)gridDim.x * 256) ec[p] = e[perm[p]];
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
