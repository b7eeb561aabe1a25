// rri_wrri_kernels.hpp -- elementwise-weighted RRI (the reference's W_mat path, "Algorithm 10 / WRRI":
// nmf.py:687-701 T side, :735-746 W side; qf_min's vector-c branch optimization.py:75-87).
//
// The reference rebuilds Rt = W_mat .* (X - W_{-t} T) with a full GEMM twice per topic.  Here the masked
// residual E = M .* (X - W T) lives in HBM and is corrected by rank-one terms:
//     numer_T[j] = sum_i w_i M_ij (X - W_{-t}T)_ij = a_j + t_j nw_j ,  a = w^T E ,  nw = (w^2)^T M
//     numer_W[i] = b_i + w_i nt_i ,  b = E' t' ,  nt = M (t'^2) ,  E' = E - M .* (w dt^T)
//     E''        = E' - M .* (dw t'^T)
// Dense handles, round 4: ONE read-modify-write pass over (E, M) per topic step (k_wpass<Y, Z, UPD2, WRITE>) -- it folds the
// W-column term of the step before and this step's T-row term into E, writes E, and takes the row products b, nt AND the
// column sums of the next topic; the term the W update then leaves pending reaches those sums through a pass over the mask
// alone (k_wmcorr; see "one read-modify-write pass" below).  (2 + 2/32) n d s bytes per topic step with a bit-packed mask.
// Rounds 1-3 (RRI_WPASS_ONE=0): pass B reads E, M and takes the row products, pass C applies both corrections, writes E and
// takes the next column sums: (3 + 2/32) n d s.  E is refreshed from X, W, T (k_resid) once per sweep either way, so the
// storage rounding of E never accumulates over more than k updates.
#pragma once
#include "rri_kernels.hpp"

namespace rri {

// The weights W_mat reach the kernels either as an SX array or, when every entry is 0 or 1 (the recommender
// case: observed / not observed), bit-packed.  Layout of the packed mask: one uint32 per (8 rows x 4 columns):
//     Mb[(row >> 3) * ldb + (col >> 2)]   bit ((row & 7) * 4 + (col & 3))
// so the word a lane needs for its 4 (or 2) columns covers the 8 rows it keeps in flight, and the 64 lanes of a
// wave read 64 consecutive words: ONE coalesced 256-byte load per 8 rows instead of eight 32-byte ones.
__device__ __forceinline__ unsigned mask_bit(const unsigned* __restrict__ Mb, i64 ldb, i64 row, i64 col) {
    return (Mb[(row >> 3) * ldb + (col >> 2)] >> (((int)(row & 7) << 2) + (int)(col & 3))) & 1u;
}

template <typename SX, bool MBITS>
struct MaskLoad {
    typedef typename XVec<SX>::type V;
    static constexpr int VN = XVec<SX>::N;
    // what stays in registers while U rows are in flight: one word (bits) or one 16-byte vector
    typedef typename std::conditional<MBITS, unsigned, V>::type Raw;
    static __device__ __forceinline__ Raw zero() {
        if constexpr (MBITS) return 0u;
        else return XVec<SX>::zero();
    }
    static __device__ __forceinline__ Raw load(const SX* __restrict__ M, i64 ldm, const unsigned* __restrict__ Mb,
                                               i64 ldb, i64 row, int col) {
        // bits: the lane's nibble (or half nibble) of row `row`, shifted down to bit 0
        if constexpr (MBITS) return Mb[(row >> 3) * ldb + (col >> 2)] >> (((int)(row & 7) << 2) + (col & 3));
        else return stream_load<true>(reinterpret_cast<const V*>(M + row * ldm + col));
    }
    static __device__ __forceinline__ void expand(const Raw& r, double (&me)[VN]) {
        if constexpr (MBITS) {
#pragma unroll
            for (int e = 0; e < VN; ++e) me[e] = (double)((r >> e) & 1u);
        } else {
            XVec<SX>::unpack(r, me);
        }
    }
};

// out[0] != 0 when some entry of M is neither 0 nor 1
template <typename SX>
__global__ __launch_bounds__(256) void k_mask_nonbinary(const SX* __restrict__ M, i64 ldm, i64 n, i64 d,
                                                        int* __restrict__ out) {
    int bad = 0;
    const i64 total = n * d;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 r = idx / d, c = idx - r * d;
        const SX v = M[r * ldm + c];
        if (!(v == SX(0) || v == SX(1))) bad = 1;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(out, 1);
}

// packs M into the (8 rows x 4 columns)-per-word layout; rows beyond n and columns beyond d are 0
template <typename SX>
__global__ __launch_bounds__(256) void k_mask_pack(const SX* __restrict__ M, i64 ldm, i64 n, i64 d,
                                                   unsigned* __restrict__ Mb, i64 ldb) {
    const i64 ngroups = (n + 7) >> 3;
    const i64 total = ngroups * ldb;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 rg = idx / ldb, cg = idx - rg * ldb;
        unsigned bits = 0;
        for (int rr = 0; rr < 8; ++rr) {
            const i64 r = rg * 8 + rr;
            if (r >= n) break;
            for (int cc = 0; cc < 4; ++cc) {
                const i64 c = cg * 4 + cc;
                if (c < d && M[r * ldm + c] != SX(0)) bits |= (1u << (rr * 4 + cc));
            }
        }
        Mb[idx] = bits;
    }
}

// CSR row -> dense row (one wave per row): X[r, indices[p]] = data[p]
template <typename DT, typename SX>
__global__ __launch_bounds__(256) void k_csr_scatter(const i64* __restrict__ indptr, const int* __restrict__ indices,
                                                     const DT* __restrict__ data, i64 n, SX* __restrict__ X, i64 ldx) {
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    for (i64 p = indptr[r] + (threadIdx.x & 63); p < indptr[r + 1]; p += 64) X[r * ldx + indices[p]] = (SX)data[p];
}
// observation bits of a CSR matrix: bit (r, indices[p]) = 1 where data[p] != 0
template <typename DT>
__global__ __launch_bounds__(256) void k_csr_pattern_bits(const i64* __restrict__ indptr, const int* __restrict__ indices,
                                                          const DT* __restrict__ data, i64 n, unsigned* __restrict__ Mb,
                                                          i64 ldb) {
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    for (i64 p = indptr[r] + (threadIdx.x & 63); p < indptr[r + 1]; p += 64) {
        if (data[p] != DT(0)) {
            const int c = indices[p];
            atomicOr(&Mb[(r >> 3) * ldb + (c >> 2)], 1u << (((int)(r & 7) << 2) + (c & 3)));
        }
    }
}

// Same block geometry as k_pass: 4 waves = 4 adjacent 1 KiB-wide panels x one row block.
//   e' = e - m (a1_i b1_j + [UPD2] a2_i b2_j)      (a1,a2: per row, from LDS; b1,b2: per column, registers)
//   DO_Y: Ypart = sum_j e' t_j , Y2part = sum_j m t_j^2     DO_Z: Zpart = sum_i w_i e' , [DO_Z2] Z2part = sum_i w_i^2 m
//   (DO_Z2 = false: a sparse 0/1 mask, whose second sum the mask-only kernel k_wmcorr_cols takes beside its own)
template <typename SX, bool DO_Y, bool DO_Z, bool UPD2, bool WRITE, int U, bool NT, bool MBITS, bool RS, bool DO_Z2>
__device__ __forceinline__ void wpass_body(SX* __restrict__ E, const SX* __restrict__ M, i64 ldx, i64 ldm,
                                           const unsigned* __restrict__ Mb, i64 ldb, int n, int ncols,
                                           const double* __restrict__ trow, const double* __restrict__ wcol,
                                           const double* __restrict__ a1v, const double* __restrict__ b1v,
                                           const double* __restrict__ a2v, const double* __restrict__ b2v,
                                           double* __restrict__ Ypart, double* __restrict__ Y2part,
                                           double* __restrict__ Zpart, double* __restrict__ Z2part, i64 ldz,
                                           int rpb, int npg, const DevState* __restrict__ st, int nrb_il_rot) {
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int VN = XV::N;
    constexpr int PW = 64 * VN;
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ysh = reinterpret_cast<double*>(smem);   // [4][rpb]
    double* y2sh = ysh + 4 * rpb;                    // [4][rpb]
    double* wsh = y2sh + 4 * rpb;                    // [rpb]
    double* a1sh = wsh + rpb;                        // [rpb]
    double* a2sh = a1sh + rpb;                       // [rpb]
    static_assert(!RS || U == 8, "the LDS row-sum path reduces 8 rows at a time");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* tile = a2sh + rpb + wave * (8 * 72);     // [4][8*72] private row-sum tiles (RS)
    // (top bits of the last argument: the tile a workgroup takes inside its group of 8, rotated -- another XCD for every tile,
    // nothing else changed; the handle's calibrated value, see calibrate_rot in rri_hip.hip)
    const int rot = (int)(((unsigned)nrb_il_rot >> 27) & 7u);
    const int nrb_il = nrb_il_rot & 0x07ffffff;
    int bid = (int)blockIdx.x;
    if (rot != 0 && (bid | 7) < (int)gridDim.x) bid = (bid & ~7) | ((bid + rot) & 7);
    const int pg = bid % npg, rb = bid / npg;
    // rows of this block, local index li -> global row: contiguous (rb rpb + li), or -- nrb_il > 0 -- chunk q of U rows is
    // chunk q nrb + rb of the matrix, so the workgroups running at one time walk ONE window of E as a linear stream does
    // (what made the read-modify-write pass of the unweighted residual schedule 4-9 % faster, k_pass)
    auto grow = [&](int li) -> int {
        return nrb_il > 0 ? ((li / U) * nrb_il + rb) * U + (li % U) : rb * rpb + li;
    };
    const int col = (pg * 4 + wave) * PW + lane * VN;
    for (int i = threadIdx.x; i < rpb; i += 256) {
        const int g = grow(i);
        if (DO_Z) wsh[i] = g < n ? wcol[g] : 0.0;
        a1sh[i] = g < n ? a1v[g] : 0.0;
        if (UPD2) a2sh[i] = g < n ? a2v[g] : 0.0;
    }
    __syncthreads();
    const bool ok = col < ncols;
    const bool wave_has_cols = (pg * 4 + wave) * PW < ncols;
    double tv[VN], b1[VN], b2[VN], zacc[VN], z2acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) {
        zacc[e] = z2acc[e] = 0.0;
        tv[e] = (DO_Y && ok) ? trow[col + e] : 0.0;
        b1[e] = ok ? b1v[col + e] : 0.0;
        b2[e] = (UPD2 && ok) ? b2v[col + e] : 0.0;
    }
    if (wave_has_cols) {
        for (int l0 = 0; l0 < rpb; l0 += U) {
            const int r = grow(l0);                  // the U rows of a chunk are consecutive, r a multiple of U
            if (r >= n) break;
            const int row1 = n;
            const int row0 = r - l0;                 // so that (rr - row0) below is the local index l0 + u
            V x[U];
            typename MaskLoad<SX, MBITS>::Raw mk[U];
            // bit-packed mask: the U rows of a chunk (U = 4 or 8, r a multiple of U) lie in ONE word per lane -- one
            // load per chunk.  (Loaded row by row, as the array form below must be, the same word was requested U times:
            // as many vector-memory instructions again as the residual itself takes, PMC SQ_INSTS_VMEM_RD 8.1e6 against
            // 4.1e6 for k_pass on the same bytes -- profiles/r02_pmc_sq_c5_before_mask_word_fix.txt.)
            constexpr int NW = (U + 7) / 8;          // mask words per chunk: 8 rows each
            unsigned mword[NW];
            if constexpr (MBITS) {
                static_assert(U == 4 || U % 8 == 0, "a chunk of rows must be whole 8-row mask words (or half of one)");
#pragma unroll
                for (int q = 0; q < NW; ++q)
                    mword[q] = (ok && r + 8 * q < row1) ? Mb[(i64)((r >> 3) + q) * ldb + (col >> 2)] >> (col & 3) : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u;
                x[u] = XV::zero();
                mk[u] = MaskLoad<SX, MBITS>::zero();
                if (rr < row1 && ok) {
                    x[u] = stream_load<NT>(reinterpret_cast<const V*>(E + (i64)rr * ldx + col));
                    if constexpr (MBITS) mk[u] = mword[u >> 3] >> ((rr & 7) << 2);
                    else mk[u] = MaskLoad<SX, MBITS>::load(M, ldm, Mb, ldb, rr, col);
                }
            }
            double ys[U], y2s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u;
                double wv = 0.0, c1 = 0.0, c2 = 0.0;
                if (rr < row1) {
                    if (DO_Z) wv = wsh[rr - row0];
                    c1 = a1sh[rr - row0];
                    if (UPD2) c2 = a2sh[rr - row0];
                }
                double xe[VN], me[VN];
                XV::unpack(x[u], xe);
                MaskLoad<SX, MBITS>::expand(mk[u], me);
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    double corr = c1 * b1[e];
                    if (UPD2) corr = fma(c2, b2[e], corr);
                    xe[e] = fma(-me[e], corr, xe[e]);
                }
                if constexpr (WRITE) {
                    const V rounded = XV::pack(xe);
                    if (rr < row1 && ok) stream_store<NT>(reinterpret_cast<V*>(E + (i64)rr * ldx + col), rounded);
                    if constexpr (sizeof(SX) == 4) XV::unpack(rounded, xe);
                }
                double yp = 0.0, y2p = 0.0;
                const double wv2 = wv * wv;
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    if (DO_Y) { yp = fma(xe[e], tv[e], yp); y2p = fma(me[e] * tv[e], tv[e], y2p); }   // (m t) t: t^2 is not kept in registers
                    if (DO_Z) zacc[e] = fma(wv, xe[e], zacc[e]);
                    if (DO_Z && DO_Z2) z2acc[e] = fma(wv2, me[e], z2acc[e]);
                }
                if (DO_Y) {
                    ys[u] = RS ? yp : wave_sum<double>(yp);
                    y2s[u] = RS ? y2p : wave_sum<double>(y2p);
                }
            }
            if constexpr (DO_Y && RS) {
                const double t1 = wave_rowsum8(reinterpret_cast<const double (&)[8]>(ys), tile, lane);
                const double t2 = wave_rowsum8(reinterpret_cast<const double (&)[8]>(y2s), tile, lane);
                const int rr = r + (lane >> 3);
                if ((lane & 7) == 0) { ysh[wave * rpb + rr - row0] = (rr < row1) ? t1 : 0.0; y2sh[wave * rpb + rr - row0] = (rr < row1) ? t2 : 0.0; }
            } else if (DO_Y) {
                double yv = ys[0], y2v = y2s[0];
#pragma unroll
                for (int u = 1; u < U; ++u)
                    if (lane == u) { yv = ys[u]; y2v = y2s[u]; }
                const int rr = r + lane;
                if (lane < U) { ysh[wave * rpb + rr - row0] = (rr < row1) ? yv : 0.0; y2sh[wave * rpb + rr - row0] = (rr < row1) ? y2v : 0.0; }
            }
        }
        if (DO_Z && ok) {
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                Zpart[(i64)rb * ldz + col + e] = zacc[e];
                if (DO_Z2) Z2part[(i64)rb * ldz + col + e] = z2acc[e];
            }
        }
    } else if (DO_Y) {
        for (int i = lane; i < rpb; i += 64) { ysh[wave * rpb + i] = 0.0; y2sh[wave * rpb + i] = 0.0; }
    }
    if (DO_Y) {
        __syncthreads();
        for (int i = threadIdx.x; i < rpb; i += 256) {
            const int g = grow(i);
            if (g < n) {
                Ypart[(i64)pg * n + g] = (ysh[i] + ysh[rpb + i]) + (ysh[2 * rpb + i] + ysh[3 * rpb + i]);
                Y2part[(i64)pg * n + g] = (y2sh[i] + y2sh[rpb + i]) + (y2sh[2 * rpb + i] + y2sh[3 * rpb + i]);
            }
        }
    }
}

#define RRI_WPASS_ARGS                                                                                                     \
    SX *__restrict__ E, const SX *__restrict__ M, i64 ldx, i64 ldm, const unsigned *__restrict__ Mb, i64 ldb, int n, int ncols, \
        const double *__restrict__ trow, const double *__restrict__ wcol, const double *__restrict__ a1v,                  \
        const double *__restrict__ b1v, const double *__restrict__ a2v, const double *__restrict__ b2v,                   \
        double *__restrict__ Ypart, double *__restrict__ Y2part, double *__restrict__ Zpart, double *__restrict__ Z2part, \
        i64 ldz, int rpb, int npg, const DevState *__restrict__ st, int nrb_il
#define RRI_WPASS_PASS E, M, ldx, ldm, Mb, ldb, n, ncols, trow, wcol, a1v, b1v, a2v, b2v, Ypart, Y2part, Zpart, Z2part, ldz, rpb, npg, st, nrb_il
template <typename SX, bool DO_Y, bool DO_Z, bool UPD2, bool WRITE, int U, bool NT, bool MBITS, bool RS, bool DO_Z2 = DO_Z>
__global__ __launch_bounds__(256) void k_wpass(RRI_WPASS_ARGS) {
    wpass_body<SX, DO_Y, DO_Z, UPD2, WRITE, U, NT, MBITS, RS, DO_Z2>(RRI_WPASS_PASS);
}
// The same pass compiled for four waves per SIMD (128 registers): the one-pass step with 4 rows in flight needs 130 as the
// compiler allocates it freely and fits 128 without a spill when told to (8 rows in flight would spill 110).  The step that
// leaves nw to the mask-only kernel (DO_Z2 = false) needs 98 either way and runs as k_wpass (1.353 against 1.366 ms in one
// process; built for FIVE waves, 96 registers and 2 spilled, 1.50 ms -- profiles/r04_wpass_one_variants.log).
template <typename SX, bool DO_Y, bool DO_Z, bool UPD2, bool WRITE, int U, bool NT, bool MBITS, bool RS, bool DO_Z2 = DO_Z>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wpass_occ4(RRI_WPASS_ARGS) {
    wpass_body<SX, DO_Y, DO_Z, UPD2, WRITE, U, NT, MBITS, RS, DO_Z2>(RRI_WPASS_PASS);
}
#undef RRI_WPASS_ARGS
#undef RRI_WPASS_PASS

// ---- one read-modify-write pass per topic step (round 4) -------------------------------------------------------------
// The pass of step t (k_wpass<DO_Y, DO_Z, UPD2, WRITE>) folds the two pending rank-one terms (dw_{t-1} t'_{t-1}^T and
// w_t dt_t^T) into E, writes it, and takes BOTH the row products of the W update of topic t and the column sums
// z = w_{t+1}^T E, nw = (w_{t+1}^2)^T M of the next T row -- column t+1 of W is untouched by step t.  What z lacks is the term
// the W update of step t leaves pending, M .* (dw_t t'_t^T):
//     a_j = z_j - t'_{t,j} c_j ,      c_j = sum_i M_ij w_{t+1,i} dw_{t,i}
// c needs no residual: it is a pass over the MASK alone (bit-packed: n d / 8 bytes), k_wmcorr below.  nw does not change.
// Per topic step (2 + 1/32 + 1/32) n d s bytes instead of (3 + 2/32) (nmf.py:687-701, 735-746 are what both compute).
//
// k_wmcorr: Cpart[rb][j] = sum over the rows i of block rb of M_ij u_i, u_i = wn_i dw_i.  4 waves = 4 adjacent column panels
// x one row block (rpb rows, a multiple of 8); u goes through LDS (every lane of a wave reads the same row: broadcast).
// Bit-packed: a lane owns ONE word column (4 columns) whatever the storage type, 8 words (64 rows) in flight; per mask bit
// one bit-field extract, one conversion and one float64 fused multiply-add -- the kernel is bound by those (10^9 bits at
// BASELINE config 5), not by the 125 MB it reads.  Array masks (weights that are not all 0 / 1): the geometry of k_pass.
template <typename SX, bool MBITS, bool SKIP>
__global__ __launch_bounds__(256) void k_wmcorr(const SX* __restrict__ M, i64 ldm, const unsigned* __restrict__ Mb, i64 ldb,
                                                int n, int ncols, const double* __restrict__ wn,
                                                const double* __restrict__ dw, double* __restrict__ Cpart, i64 ldz, int rpb,
                                                int npg, const DevState* __restrict__ st) {
    typedef XVec<SX> XV;
    typedef typename XV::type V;
    constexpr int CN = MBITS ? 4 : XV::N;            // columns per lane
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ush = reinterpret_cast<double*>(smem);   // [rpb]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pg = blockIdx.x % npg, rb = blockIdx.x / npg;
    const int row0 = rb * rpb;
    for (int i = threadIdx.x; i < rpb; i += 256) {
        const int g = row0 + i;
        ush[i] = g < n ? wn[g] * dw[g] : 0.0;
    }
    __syncthreads();
    const int col = ((pg * 4 + wave) * 64 + lane) * CN;
    if ((pg * 4 + wave) * 64 * CN >= ncols) return;  // wave-uniform
    const bool ok = col < ncols;
    double acc[CN];
#pragma unroll
    for (int e = 0; e < CN; ++e) acc[e] = 0.0;
    if constexpr (MBITS) {
        constexpr int UG = 8;                        // row groups (words) in flight per lane
        const int ngroups = (n + 7) >> 3;
        const int g0 = row0 >> 3, g1 = min(ngroups, (row0 + rpb) >> 3);
        const i64 wc = col >> 2;
        const i64 wcc = ok ? wc : 0;                 // lanes beyond the last word column read (and drop) word column 0
        for (int g = g0; g < g1; g += UG) {
            unsigned w[UG];
            // unconditional loads (a group beyond the block: its last group again, unused), so that all UG are in flight at once
#pragma unroll
            for (int q = 0; q < UG; ++q) w[q] = Mb[(i64)min(g + q, g1 - 1) * ldb + wcc];
#pragma unroll
            for (int q = 0; q < UG; ++q) {
                const double* up = ush + ((g + q - g0) << 3);
                if (g + q >= g1) break;              // wave-uniform
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const double u = up[r];
                    // a row whose factor is zero (w_{t+1,i} = 0 or an unchanged entry of column t: both common, the columns are
                    // clipped at zero) adds nothing: one compare instead of twelve instructions.  Every lane reads the same u.
                    if (SKIP && u == 0.0) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[e] = fma((double)((w[q] >> (r * 4 + e)) & 1u), u, acc[e]);
                }
            }
        }
    } else {
        constexpr int U = 8;
        const int row1 = min(n, row0 + rpb);
        for (int r = row0; r < row1; r += U) {
            V x[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                x[u] = (r + u < row1 && ok) ? stream_load<true>(reinterpret_cast<const V*>(M + (i64)(r + u) * ldm + col)) : XV::zero();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u >= row1) break;
                double me[CN];
                XV::unpack(x[u], me);
                const double uv = ush[r + u - row0];
#pragma unroll
                for (int e = 0; e < CN; ++e) acc[e] = fma(me[e], uv, acc[e]);
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int e = 0; e < CN; ++e)
            if (col + e < ldz) Cpart[(i64)rb * ldz + col + e] = acc[e];
    }
}

// ---- the same correction on a SPARSE 0/1 mask: walk the set bits only -------------------------------------------------------
// k_wmcorr spends three vector instructions on every mask bit, set or not: 10^9 of them at BASELINE config 5, where 5 % are set.
// A second packed copy of the mask with the ROWS in the bits -- Mc[(row >> 5) * ldc + col], bit (row & 31): a lane owns one column
// and a word holds 32 rows of it -- lets a lane find its set bits with count-trailing-zeros and add u[row] for those alone
// (~7 instructions per SET bit; a wave runs as long as its fullest word).  Same sums in the same row order; used below 12 %
// density (k_mask_cols_from_bits counts the bits), the dense-bit kernel above otherwise.  n d / 8 bytes more device memory.
__global__ __launch_bounds__(256) void k_mask_cols_from_bits(const unsigned* __restrict__ Mb, i64 ldb, i64 n, unsigned* __restrict__ Mc,
                                                             i64 ldc, unsigned long long* __restrict__ nnz) {
    const i64 ngroups = (n + 31) >> 5;
    const i64 total = ngroups * ldc;
    unsigned long long cnt = 0;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 g = idx / ldc, col = idx - g * ldc;
        unsigned out = 0u;
        for (int q = 0; q < 4; ++q) {
            const i64 rg = g * 4 + q;                    // 8-row group of the (8 rows x 4 columns)-per-word layout
            if (rg * 8 >= n) break;
            const unsigned word = Mb[rg * ldb + (col >> 2)] >> (int)(col & 3);
#pragma unroll
            for (int r = 0; r < 8; ++r) out |= ((word >> (r * 4)) & 1u) << (q * 8 + r);
        }
        Mc[idx] = out;
        cnt += (unsigned)__popc(out);
    }
    cnt = (unsigned long long)wave_sum_i32((int)cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(nnz, cnt);
}

// HAS_DW: the correction c (Cpart).  NW: also nw_j = sum_i M_ij wn_i^2 (N2part) -- the second column sum of the T-row step is a
// sum over the mask alone as well, and taking it here (one more LDS read and add per SET bit) spares the read-modify-write pass
// its second set of accumulators and n_row_blocks x d doubles written there and read back by k_wreduce (125 MB each way at
// BASELINE config 5).  Same sum in another (still fixed) order.
template <bool HAS_DW, bool NW>
__global__ __launch_bounds__(256) void k_wmcorr_cols(const unsigned* __restrict__ Mc, i64 ldc, int n, int ncols,
                                                     const double* __restrict__ wn, const double* __restrict__ dw,
                                                     double* __restrict__ Cpart, double* __restrict__ N2part, i64 ldz, int rpb,
                                                     int npg, const DevState* __restrict__ st) {
    static_assert(HAS_DW || NW, "nothing to take");
    if (st->halt) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PS = (HAS_DW && NW) ? 2 : 1;       // doubles per row in LDS: {u, wn^2} | u | wn^2
    double* ush = reinterpret_cast<double*>(smem);   // [rpb][PS], rpb a multiple of 32
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pg = blockIdx.x % npg, rb = blockIdx.x / npg;
    const int row0 = rb * rpb;
    for (int i = threadIdx.x; i < rpb; i += 256) {
        const int g = row0 + i;
        const double wv = g < n ? wn[g] : 0.0;
        if (HAS_DW) ush[i * PS] = g < n ? wv * dw[g] : 0.0;
        if (NW) ush[i * PS + (PS - 1)] = wv * wv;
    }
    __syncthreads();
    const int col = (pg * 4 + wave) * 64 + lane;
    if ((pg * 4 + wave) * 64 >= ncols) return;        // wave-uniform
    const bool ok = col < ncols;
    const i64 colc = ok ? col : 0;
    const int g0 = row0 >> 5, g1 = min((n + 31) >> 5, (row0 + rpb) >> 5);
    double acc = 0.0, acc2 = 0.0;
    constexpr int UG = 8;
    for (int g = g0; g < g1; g += UG) {
        unsigned w[UG];
#pragma unroll
        for (int q = 0; q < UG; ++q) w[q] = Mc[(i64)min(g + q, g1 - 1) * ldc + colc];
#pragma unroll
        for (int q = 0; q < UG; ++q) {
            if (g + q >= g1) break;                   // wave-uniform
            const double* up = ush + ((g + q - g0) << 5) * PS;
            unsigned wq = w[q];
            while (wq) {                              // rows in ascending order, as the dense-bit kernel adds them
                const int b = __builtin_ctz(wq);      // (four words walked at once into four sums: 83 us against 60 -- the predicated
                wq &= wq - 1u;                        //  selects cost more than the shared waits save)
                if constexpr (PS == 2) {
                    const f64x2 v = *reinterpret_cast<const f64x2*>(up + 2 * b);
                    acc += v[0];
                    acc2 += v[1];
                } else if (HAS_DW) acc += up[b];
                else acc2 += up[b];
            }
        }
    }
    if (ok && col < ldz) {
        if (HAS_DW) Cpart[(i64)rb * ldz + col] = acc;
        if (NW) N2part[(i64)rb * ldz + col] = acc2;
    }
}

// Both fixed-order reductions of a weighted T-row step in one launch (they were two launches of k_reduce), with the
// correction above:  red[j] = sum_b Zpart[b][j] - tprow[j] sum_b Cpart[b][j] ,  red[ldz + j] = sum_b Z2part[b][j].
// Cpart == NULL: no term pending.  Row-sharded runs all-reduce red afterwards: the correction is a sum over rows like z.
// check_prev: the column verdict of the last W update (nmf.py:471-476, 793-816; k_wcheck_wcol's) is taken here, by workgroup 0,
// at the position of THIS step -- the T-row kernel that follows returns at once when it halts the run: one launch less per step.
__device__ __forceinline__ void wcol_verdict(double a, double f, int tprev, int sweep, int pos, const KParams& p, DevState* st);
__global__ __launch_bounds__(1024) void k_wreduce(const double* __restrict__ Zpart, const double* __restrict__ Z2part, i64 ldz,
                                                  int nrb, int nrb2, const double* __restrict__ Cpart, int nrbc,
                                                  const double* __restrict__ tprow, double* __restrict__ red,
                                                  const double* __restrict__ Gpart, int nwb, int k, int check_prev, int tprev,
                                                  int sweep, int pos, KParams p, DevState* st) {
    if (st->halt) return;
    __shared__ double sh[40];
    const int tid = threadIdx.x;
    if (check_prev && blockIdx.x == 0) {
        double a = ordered_sum<8>(Gpart + k + 1, k + 2, tid, nwb, 1024);
        double f = ordered_sum<8>(Gpart + k, k + 2, tid, nwb, 1024);
        a = block_sum(a, sh);
        f = block_sum(f, sh);
        if (tid == 0) wcol_verdict(a, f, tprev, sweep, pos, p, st);
        __syncthreads();
    }
    // a thread takes TWO adjacent columns (16-byte loads: 512 contiguous bytes per partial row and half-wave instead of 256 -- the
    // partial arrays are read once, nrb x LD each, and with the small row blocks the writing passes like there are a thousand
    // rows of them); every column is still summed over the rows in the same fixed order: g, g + 32, ... then the 32 groups
    const int c = tid & 31, g = tid >> 5;
    const i64 j = (i64)blockIdx.x * 64 + 2 * c;
    double a[2] = {0.0, 0.0}, b[2] = {0.0, 0.0}, cc[2] = {0.0, 0.0};
    auto sum2 = [&](const double* __restrict__ P, int rows, double (&out)[2]) {
        for (int i = g; i < rows; i += 32 * 8) {
            f64x2 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ii = i + q * 32;
                v[q] = ii < rows ? *reinterpret_cast<const f64x2*>(P + (i64)ii * ldz + j) : f64x2{0.0, 0.0};
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { out[0] += v[q][0]; out[1] += v[q][1]; }
        }
    };
    if (j < ldz) {      // ldz is even and j is even: the pair is inside the row
        sum2(Zpart, nrb, a);
        sum2(Z2part, nrb2, b);      // the pass's row blocks, or the mask-only kernel's (k_wmcorr_cols<.., NW>)
        if (Cpart) sum2(Cpart, nrbc, cc);
    }
    __shared__ double sh2[3][32 * 33];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        sh2[0][g * 33 + c] = a[h];
        sh2[1][g * 33 + c] = b[h];
        sh2[2][g * 33 + c] = cc[h];
        __syncthreads();
        if (tid < 32 && j + h < ldz) {   // here c == tid
            double sa = 0.0, sb = 0.0, sc = 0.0;
            for (int q = 0; q < 32; ++q) { sa += sh2[0][q * 33 + tid]; sb += sh2[1][q * 33 + tid]; sc += sh2[2][q * 33 + tid]; }
            red[j + h] = Cpart ? fma(-tprow[j + h], sc, sa) : sa;
            red[ldz + j + h] = sb;
        }
    }
}

// T row, vector-c qf_min (optimization.py:75-87), before the optional rescale.  red = [a | nw] (2 x ldz).
// flags[b] = 1 when the block saw a negative denominator (-> "unbounded" unless s or ub is given).
__global__ __launch_bounds__(128) void k_wtrow(const double* __restrict__ T, i64 ldt, int d, int t,
                                               const double* __restrict__ red, i64 ldz, double* __restrict__ xraw,
                                               double* __restrict__ tpart, i64* __restrict__ flags, KParams p,
                                               const DevState* __restrict__ st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    const int tid = threadIdx.x;
    const i64 j = (i64)blockIdx.x * 128 + tid;
    double x = 0.0, neg = 0.0;
    if (j < d) {
        const double a = red[j], nw = red[ldz + j], tj = T[(i64)t * ldt + j];
        const double numer = fma(tj, nw, a) - p.reg_t_l1;      // w^T Rt - reg_t_l1 (nmf.py:437)
        const double c = nw + p.reg_t_l2;                       // nmf.py:438
        if (c < 0.0) neg = 1.0;
        if (c > 0.0) x = fmax(numer, 0.0) / (c + p.eps);
        if (p.has_trs) x = fmin(x, p.t_row_sum);                // ub = t_row_sum (min(ub, s) when s is given)
        xraw[j] = x;
    }
    const double s = block_sum(x, scratch);
    const double ng = block_sum(neg, scratch);
    if (tid == 0) { tpart[blockIdx.x] = s; flags[blockIdx.x] = ng > 0.0 ? 1 : 0; }
}

__device__ __forceinline__ void wcol_verdict(double a, double f, int tprev, int sweep, int pos, const KParams& p, DevState* st);

// Where few row blocks leave partial column sums (pattern-only handles: one per 10240 rows; small dense problems), the column
// verdict of the last W update (k_wcheck_wcol), the two reductions (k_reduce on Zpart and Z2part) and k_wtrow are ONE launch:
// four dependent 5-us kernels per topic step become one (round 3: the six small kernels between two passes were 34 us of a
// 260-us topic step at BASELINE config 5 on the observed pattern).  Every workgroup takes the verdict itself from the same
// sums -- the same decision everywhere, nothing is updated when the step halts -- and leaves red = [a | nw] as k_reduce does.
__global__ __launch_bounds__(128) void k_wtrow_small(const double* __restrict__ T, i64 ldt, int d, int t,
                                                     const double* __restrict__ Zpart, const double* __restrict__ Z2part,
                                                     i64 ldz, int nrb, int nrb2, const double* __restrict__ Cpart, int nrbc,
                                                     const double* __restrict__ tprow,
                                                     const double* __restrict__ Gpart, int nwb, int k,
                                                     int check_prev, int tprev, int sweep, double* __restrict__ red,
                                                     double* __restrict__ xraw, double* __restrict__ tpart,
                                                     i64* __restrict__ flags, KParams p, DevState* st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    const int tid = threadIdx.x;
    if (check_prev) {          // nmf.py:471-476 / 793-816 for the column updated last, at the position of THIS step
        double a = ordered_sum<8>(Gpart + k + 1, k + 2, tid, nwb, 128);
        double f = ordered_sum<8>(Gpart + k, k + 2, tid, nwb, 128);
        a = block_sum(a, scratch);
        f = block_sum(f, scratch);
        const bool unb = f > 0.0 && !p.has_wrs;
        const bool ev = (a <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
        const bool err = !ev && !(a > 0.0);
        if (unb || ev || err) {
            if (blockIdx.x == 0 && tid == 0) wcol_verdict(a, f, tprev, sweep, t, p, st);
            return;
        }
    }
    const i64 j = (i64)blockIdx.x * 128 + tid;
    double x = 0.0, neg = 0.0;
    if (j < ldz) {
        double a = ordered_sum<8>(Zpart + j, ldz, 0, nrb, 1);
        const double nw = ordered_sum<8>(Z2part + j, ldz, 0, nrb2, 1);
        // the rank-one term the last W update left pending in E (k_wmcorr): a_j = z_j - t'_j c_j
        if (Cpart) a = fma(-tprow[j], ordered_sum<8>(Cpart + j, ldz, 0, nrbc, 1), a);
        red[j] = a;
        red[ldz + j] = nw;
        if (j < d) {
            const double tj = T[(i64)t * ldt + j];
            const double numer = fma(tj, nw, a) - p.reg_t_l1;      // w^T Rt - reg_t_l1 (nmf.py:437)
            const double c = nw + p.reg_t_l2;                       // nmf.py:438
            if (c < 0.0) neg = 1.0;
            if (c > 0.0) x = fmax(numer, 0.0) / (c + p.eps);
            if (p.has_trs) x = fmin(x, p.t_row_sum);                // ub = t_row_sum (min(ub, s) when s is given)
            xraw[j] = x;
        }
    }
    const double s = block_sum(x, scratch);
    const double ng = block_sum(neg, scratch);
    if (tid == 0) { tpart[blockIdx.x] = s; flags[blockIdx.x] = ng > 0.0 ? 1 : 0; }
}

// finishes the weighted T row: unbounded check, rescale to sum s (optimization.py:85-87), the row checks of
// _project_and_check_reset_t, T[t,:] and dt = scale_w * t_new - t_old (scale_w = nt1 when fix_W keeps and
// rescales the column, nmf.py:450-452; 1 otherwise).
__global__ __launch_bounds__(1024) void k_wtrow_final(double* __restrict__ T, i64 ldt, int d, int t,
                                                      double* __restrict__ xraw, const double* __restrict__ tpart,
                                                      const i64* __restrict__ flags, int nblk,
                                                      double* __restrict__ dt, int scale_w, int sweep, KParams p,
                                                      DevState* st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    const int tid = threadIdx.x;
    double ps = 0.0, pf = 0.0;
    for (int b = tid; b < nblk; b += blockDim.x) { ps += tpart[b]; pf += (double)flags[b]; }
    const double nx = block_sum(ps, scratch);
    const double anyneg = block_sum(pf, scratch);
    const bool project = p.project_T && p.has_trs;
    if (anyneg > 0.0 && !project && !p.has_trs) {     // any(c<0) and s is None and ub is None
        if (tid == 0) { st->halt = HALT_ERR_UNBOUNDED; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t; }
        return;
    }
    double sumT = nx;
    if (project) {
        const double f = p.t_row_sum / nx;            // x = s * x / x.sum()
        for (int j = tid; j < d; j += blockDim.x) xraw[j] = p.t_row_sum * xraw[j] / nx;
        (void)f;
        __syncthreads();
        double s2 = 0.0;
        for (int j = tid; j < d; j += blockDim.x) s2 += xraw[j];
        sumT = block_sum(s2, scratch);
    }
    bool event = false;
    if (sumT > 1e-10 || p.reset_method == RESET_NONE) {
        if (p.has_trs && p.t_row_sum != 0.0 && p.project_T && fabs(sumT - p.t_row_sum) > 1e-15) {
            int it2 = 0;
            const double th = simplex_theta(xraw, d, p.t_row_sum, scratch, &it2);
            for (int j = tid; j < d; j += blockDim.x) xraw[j] = fmax(xraw[j] - th, 0.0);
        }
    } else if (p.resets_left > 0) {
        event = true;
    }
    __syncthreads();
    const double sw = scale_w ? nx : 1.0;
    for (int j = tid; j < d; j += blockDim.x) {
        const double told = T[(i64)t * ldt + j], tnew = xraw[j];
        dt[j] = sw * tnew - told;
        T[(i64)t * ldt + j] = tnew;
    }
    if (tid == 0) {
        st->nt1 = nx;
        st->sumT = sumT;
        if (event) { st->halt = HALT_EVENT_RESET_T; st->halt_topic = t; st->halt_sweep = sweep; st->halt_pos = t; }
    }
}

// W column, vector-c qf_min: numer_i = b_i + w_i nt_i - reg_w_l1, c_i = nt_i + reg_w_l2 (nmf.py:464-469).
// Writes W[:,t], wold = previous column, dw = new - old; Gpart[b][k+1] = sum of the new column,
// Gpart[b][k] = 1 when a negative denominator was seen.
__global__ __launch_bounds__(256) void k_wwcol(double* __restrict__ Wt, i64 ldw, int n, int k, int t,
                                               const double* __restrict__ Ypart, const double* __restrict__ Y2part,
                                               int npg, double* __restrict__ wold, double* __restrict__ dw,
                                               double* __restrict__ Gpart, KParams p, const DevState* __restrict__ st) {
    if (st->halt) return;
    __shared__ double scratch[40];
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    double wnew = 0.0, neg = 0.0;
    if (i < n) {
        const double b = ordered_sum<8>(Ypart + i, n, 0, npg, 1), nt = ordered_sum<8>(Y2part + i, n, 0, npg, 1);
        const double w0 = Wt[(i64)t * ldw + i];
        const double numer = fma(w0, nt, b) - p.reg_w_l1;
        const double c = nt + p.reg_w_l2;
        if (c < 0.0) neg = 1.0;
        if (c > 0.0) wnew = fmax(numer, 0.0) / (c + p.eps);
        if (p.has_wrs) wnew = fmin(wnew, p.w_row_sum);
        Wt[(i64)t * ldw + i] = wnew;
        wold[i] = w0;
        dw[i] = wnew - w0;
    }
    const double sw = block_sum(wnew, scratch);
    const double ng = block_sum(neg, scratch);
    if (threadIdx.x == 0) {
        double* gp = Gpart + (i64)blockIdx.x * (k + 2);
        gp[k] = ng;
        gp[k + 1] = sw;
    }
}

// column check for the weighted flavour: unbounded (negative denominator without ub) first, then the
// reset / assert logic of k_check_wcol.  (a, f) = (sum of the new column, 1 when a negative denominator was seen).
__device__ __forceinline__ void wcol_verdict(double a, double f, int tprev, int sweep, int pos, const KParams& p,
                                             DevState* st) {
    if (f > 0.0 && !p.has_wrs) {
        st->halt = HALT_ERR_UNBOUNDED; st->halt_topic = tprev; st->halt_sweep = sweep; st->halt_pos = pos;
        return;
    }
    const bool ev = (a <= 1e-10) && p.reset_method != RESET_NONE && p.resets_left > 0;
    const bool err = !ev && !(a > 0.0);
    if (ev || err) {
        st->halt = ev ? HALT_EVENT_RESET_W : HALT_ERR_W_COL_ZERO;
        st->halt_topic = tprev; st->halt_sweep = sweep; st->halt_pos = pos;
    }
}

// tail == NULL: decide from this handle's rows.  tail != NULL (row-sharded): only park the local (a, f) in
// tail[0..1]; the verdict is k_wcheck_tail's, on the all-reduced pair.
__global__ __launch_bounds__(256) void k_wcheck_wcol(const double* __restrict__ Gpart, int nwb, int k, int tprev,
                                                     int sweep, int pos, KParams p, DevState* st,
                                                     double* __restrict__ tail) {
    if (st->halt) return;
    __shared__ double scratch[40];
    double a = 0.0, f = 0.0;
    a = ordered_sum<8>(Gpart + k + 1, k + 2, threadIdx.x, nwb, (int)blockDim.x);
    f = ordered_sum<8>(Gpart + k, k + 2, threadIdx.x, nwb, (int)blockDim.x);
    a = block_sum(a, scratch);
    f = block_sum(f, scratch);
    if (threadIdx.x == 0) {
        if (tail) { tail[0] = a; tail[1] = f; }
        else wcol_verdict(a, f, tprev, sweep, pos, p, st);
    }
}

__global__ __launch_bounds__(64) void k_wcheck_tail(const double* __restrict__ tail, int tprev, int sweep, int pos,
                                                    KParams p, DevState* st) {
    if (st->halt) return;
    if (threadIdx.x == 0) wcol_verdict(tail[0], tail[1], tprev, sweep, pos, p, st);
}

}  // namespace rri
