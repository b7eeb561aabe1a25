// rri_device.hpp -- device-side helpers shared by the RRI kernels (gfx950 / CDNA4 only).
//
// Wave = 64 lanes.  Cross-lane sums use DPP modifiers (one VALU op per step, no LDS crossbar
// traffic) or, for 8 rows at once, a wave-private LDS tile; workgroup sums go wave -> LDS -> wave.
// All reductions are in float64 and in a fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rri {

typedef long long i64;

// ---- DPP lane moves -------------------------------------------------------------------
// dpp_ctrl: quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141,
//           row_bcast:15 0x142, row_bcast:31 0x143 (GFX9 wave64 forms).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp(double v) {
    int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(v));
    int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_get(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int lane_get(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// Sum over the 64 lanes of a FULLY ACTIVE wave; result returned wave-uniform.
// Sum of p[i * stride] for i = i0, i0 + step, ... < i1, added IN THAT ORDER (bit-identical to the plain loop), with
// B loads in flight: the plain loop compiles to one load and a full wait per iteration -- a dependent round trip to
// L2 / HBM per term, which is what the small per-topic kernels spent their time on (k_trow_small 9 us at 10000 x 1000).
template <int B>
__device__ __forceinline__ double ordered_sum(const double* __restrict__ p, long long stride, int i0, int i1, int step) {
    double a = 0.0;
    for (int i = i0; i < i1; i += step * B) {
        double v[B];
#pragma unroll
        for (int q = 0; q < B; ++q) {
            const int ii = i + q * step;
            v[q] = ii < i1 ? p[(long long)ii * stride] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < B; ++q) a += v[q];
    }
    return a;
}

// the same, continuing a sum already begun (a0): the terms are still added one by one in index order
template <int B>
__device__ __forceinline__ double ordered_sum_acc(double a0, const double* __restrict__ p, long long stride, int i0, int i1, int step) {
    double a = a0;
    for (int i = i0; i < i1; i += step * B) {
        double v[B];
#pragma unroll
        for (int q = 0; q < B; ++q) {
            const int ii = i + q * step;
            v[q] = ii < i1 ? p[(long long)ii * stride] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < B; ++q) a += v[q];
    }
    return a;
}

// the same sum left in lane 63 only (no readlane: nothing goes through scalar registers -- k_pass keeps 16 row dots per chunk,
// which as wave-uniform values cost 32 SGPRs and spilled 102 of them in the read-modify-write instantiations)
template <typename S>
__device__ __forceinline__ S wave_sum_lane63(S v) {
    v += dpp<0xB1, 0xf>(v);
    v += dpp<0x4E, 0xf>(v);
    v += dpp<0x141, 0xf>(v);
    v += dpp<0x140, 0xf>(v);
    v += dpp<0x142, 0xa>(v);
    v += dpp<0x143, 0xc>(v);
    return v;
}
template <typename S>
__device__ __forceinline__ S wave_sum(S v) {
    v += dpp<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp<0x141, 0xf>(v);   // row_half_mirror
    v += dpp<0x140, 0xf>(v);   // row_mirror: every lane of a 16-lane row holds the row total
    v += dpp<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v += dpp<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave total
    return lane_get(v, 63);
}
// Integer sum and float64 maximum over a fully active wave, wave-uniform, by the same six DPP steps (a __shfl_xor butterfly is
// six ds_bpermute round trips through the LDS crossbar, ~0.3 us of dependent latency: too much for a reduction that sits in an
// iteration on one wave).  For the maximum the lanes a step does not write keep their own value.
// sum over the aligned group of 8 (4) lanes a lane belongs to, in every lane of the group: the butterfly of __shfl_xor 1, 2, 4
// with the same additions
__device__ __forceinline__ double group4_sum(double v) {
    v += dpp<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += dpp<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ double group8_sum(double v) {
    v = group4_sum(v);
    v += dpp<0x141, 0xf>(v);     // row_half_mirror: the other quad of the 8 lanes
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += dpp_i32<0xB1, 0xf>(v);
    v += dpp_i32<0x4E, 0xf>(v);
    v += dpp_i32<0x141, 0xf>(v);
    v += dpp_i32<0x140, 0xf>(v);
    v += dpp_i32<0x142, 0xa>(v);
    v += dpp_i32<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_keep(double v) {
    const int l = __double2loint(v), h = __double2hiint(v);
    const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_keep<0xB1, 0xf>(v));
    v = fmax(v, dpp_keep<0x4E, 0xf>(v));
    v = fmax(v, dpp_keep<0x141, 0xf>(v));
    v = fmax(v, dpp_keep<0x140, 0xf>(v));
    v = fmax(v, dpp_keep<0x142, 0xa>(v));
    v = fmax(v, dpp_keep<0x143, 0xc>(v));
    return lane_get(v, 63);
}
// Sums of 8 rows at once: v[u] is this lane's partial of row u (u < 8).  The wave parks the 8 x 64 partials in
// a PRIVATE LDS tile (8 rows x 72 doubles, padded), re-reads them transposed -- lane = (row r = lane>>3,
// part p = lane&7) adds the 8 partials p*8..p*8+7 of row r -- and three DPP steps add the 8 parts.  Returns
// the total of row (lane>>3), identical in the 8 lanes of that row group.  About 4 ops per row instead of the
// 18 of six f64 DPP steps.  LDS instructions of one wave execute in issue order, so only the compiler needs
// a fence between the stores and the loads.
__device__ __forceinline__ double wave_rowsum8(const double (&v)[8], double* tile, int lane) {
    const int wpos = lane + (lane >> 3);              // column `lane` of a row, padded by one per 8
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[u * 72 + wpos] = v[u];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int r = lane >> 3, p = lane & 7;
    const double* src = tile + r * 72 + p * 9;        // columns p*8 .. p*8+7 of row r
    double s = ((src[0] + src[1]) + (src[2] + src[3])) + ((src[4] + src[5]) + (src[6] + src[7]));
    s += dpp<0xB1, 0xf>(s);      // quad_perm [1,0,3,2]
    s += dpp<0x4E, 0xf>(s);      // quad_perm [2,3,0,1]
    s += dpp<0x141, 0xf>(s);     // row_half_mirror: the other quad of the 8-lane group
    __builtin_amdgcn_wave_barrier();
    return s;
}

// The same reduction in two steps, so that the 8 partials need not be live in registers at once: row u's partial is
// parked as soon as it is complete (park), the transposed read and the adds follow after the 8th (finish).  Same
// operations in the same order as wave_rowsum8: bit-identical sums.
__device__ __forceinline__ void wave_rowsum8_park(double* tile, int u, int lane, double v) {
    tile[u * 72 + lane + (lane >> 3)] = v;
}
__device__ __forceinline__ double wave_rowsum8_finish(double* tile, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int r = lane >> 3, p = lane & 7;
    const double* src = tile + r * 72 + p * 9;
    double s = ((src[0] + src[1]) + (src[2] + src[3])) + ((src[4] + src[5]) + (src[6] + src[7]));
    s += dpp<0xB1, 0xf>(s);
    s += dpp<0x4E, 0xf>(s);
    s += dpp<0x141, 0xf>(s);
    __builtin_amdgcn_wave_barrier();
    return s;
}

// max with first-index tie-break over a fully active wave (butterfly through LDS-free shuffles)
__device__ __forceinline__ void wave_argmax(double& v, i64& idx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ov = __shfl_xor(v, off, 64);
        i64 oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// Workgroup sum of doubles; every thread gets the total.  `scratch` holds >= 33 doubles.
// All threads of the block must call (blockDim.x a multiple of 64).
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double w = wave_sum<double>(v);
    __syncthreads();  // protect scratch from a previous use
    if (lane == 0) scratch[wave] = w;
    __syncthreads();
    double tot = 0.0;
    for (int i = 0; i < nw; ++i) tot += scratch[i];  // fixed order: deterministic
    return tot;
}

}  // namespace rri
