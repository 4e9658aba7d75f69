// Row plan of the fused grouped MLP (shared by mlp.hip and mlp_rowwave.hip).
//
// The ball query pads a ball that holds cnt < nsample points with copies of its FIRST hit
// (lib/utils/tf_ops/grouping/tf_grouping_g.cu:245-248, :339-343), so rows cnt .. nsample-1 of the grouped tensor
// are identical to row 0, the MLP (row-wise) maps them to identical outputs, and the max over nsample
// (lib/utils/layers_util.py:178) equals the max over the first max(cnt,1) rows -- exactly, bit for bit.  On
// KITTI-like clouds most balls are far from full (layer1/layer2 bands: 1-8 points of 32/64), so the kernels
// evaluate only the distinct rows, in GRANULES of 8 rows (one quarter of a 32-row MFMA tile; the D layout of
// v_mfma_f32_32x32x16 puts rows 8q..8q+7 into registers 4q..4q+3 of the two lane halves, so a granule's maximum is
// three v_max + one v_permlane32_swap whatever the ball boundaries are):
//   mlp_plan_kernel   ball i -> g_i = ceil(clamp(cnt_i, 1, ns) / 8) granules, placed "next fit" into 32-row tiles (a
//                     ball of <= 4 granules never straddles a tile boundary; the rest of a tile is padded with
//                     invalid entries) by a scan over phase -> (phase, advance) functions, one workgroup per
//                     scale, no atomics; list gran[], entry = ball << 7 | ordinal << 1 | split.  `split` marks a
//                     ball of more than 32 distinct rows: its partial maxima meet through an atomic max on the
//                     output, which the plan kernel has zeroed (relu(max + bias) >= 0, so the bit patterns order
//                     like the floats and max commutes with the monotone relu(. + bias)).
//   MLP kernels       tile t = granules 4t .. 4t+3, read the list instead of computing (ball, sample) from the tile
//                     index; after the last layer the granule maxima of a ball's run inside the tile are combined
//                     with wave-uniform branches and written (plain store, or atomic max for split balls).
// `dense` plans (every ball gets ceil(ns/8) granules) reproduce the old behaviour for A/B measurements.
//
// Round 6: TIGHT packing is the default.  Next fit padded the open tile whenever the next ball did not fit -- a ball of
// three granules followed by one of two wastes a quarter of a tile -- which cost 5-22 % of the evaluated rows of the wide
// scales on the generator's frames (layer 4 scale 1: 11 296 rows per two frames instead of 9 280; on ring-structured frames
// 10-20 % in every layer: tools/plan_padding_sim.py, profiles/r06_plan_padding_sim.txt).  Now granules follow one another
// without padding; a ball whose granules lie in two tiles carries the `split` bit like a ball of more than 32 rows, and
// mlp_plan_zero_kernel zeroes the rows of split balls over the whole chip.  Flags bit 7 of the plan call keeps next fit.
//
// Round 5: granules of FOUR rows (GR = 4, eight per tile) for the scales the row-wave kernels of mlp_rowwave.hip take.
// On KITTI-like frames the inner bands of layer 1 / layer 2 hold 1.0-1.7 points per ball: eight-row granules evaluate
// 4.7-8x the distinct rows there, four-row granules half of that.  A four-row granule is one lane half of a register
// quad of the D layout (rows 8q .. 8q+3 in the lower half, 8q+4 .. 8q+7 in the upper one), so its maximum is the same
// three v_max and the same v_permlane32_swap -- without the final max across the halves.  hdr[3] of a plan says which
// granule it was built with; the kernels of mlp.hip / mlp_wide128.hip / mlp_gemm.hip read eight-row plans only.
#pragma once
#include "sa_common.h"

namespace sa {

constexpr int kPlanHeaderInts = 4;     // [0] granules, [1] split balls, [2] distinct rows, [3] rows per granule (8, 4 or 2)
constexpr int kPlanMaxOrd = 64;        // ordinal field: 6 bits -> nsample <= 512 (8-row granules), <= 256 (4-row)

__device__ __forceinline__ int plan_entry(const int *gran, int ngran, int G) { return G < ngran ? gran[G] : -1; }
__device__ __forceinline__ int plan_ball(int e) { return e >> 7; }
// sample index of row j (0..7) of the granule; rows past nsample repeat sample 0 (idx rows are already padded with
// the first hit up to nsample by the ball query)
template <int GR = 8>
__device__ __forceinline__ int plan_sample(int e, int j, int ns) {
    const int s = ((e >> 1) & (kPlanMaxOrd - 1)) * GR + j;
    return (e < 0 || s >= ns) ? 0 : s;
}

// per-granule maxima of one 32-row tile held in the D layout (lane = (column, half); reg r = row (r&3)+8*(r>>2)+4*half)
typedef float plan_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void granule_max(const plan_f32x16 &a, float (&qm)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a0 = fmax_nn(a[4 * q], a[4 * q + 1]);
        const float a1 = fmax_nn(a[4 * q + 2], a[4 * q + 3]);
        const float x = fmax_nn(a0, a1);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        qm[q] = fmax_nn(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
}

// the 4-row form: qm[2q + h] = maximum over rows 8q + 4h .. 8q + 4h + 3 (register quad q of lane half h), in EVERY lane
__device__ __forceinline__ void granule_max4(const plan_f32x16 &a, float (&qm)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a0 = fmax_nn(a[4 * q], a[4 * q + 1]);
        const float a1 = fmax_nn(a[4 * q + 2], a[4 * q + 3]);
        const float x = fmax_nn(a0, a1);
        // v_permlane32_swap(x, x): [0] = the lower half's value in both halves, [1] = the upper half's
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        qm[2 * q] = __uint_as_float(sw[0]);
        qm[2 * q + 1] = __uint_as_float(sw[1]);
    }
}

// the 2-row form (round 6): qm[4q + 2h + j] = maximum over rows 8q + 4h + 2j, + 1 (registers 4q + 2j, + 1 of lane half h), in
// EVERY lane -- sixteen granules per tile for the scales whose balls mostly hold a single point
__device__ __forceinline__ void granule_max2(const plan_f32x16 &a, float (&qm)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a0 = fmax_nn(a[4 * q], a[4 * q + 1]);
        const float a1 = fmax_nn(a[4 * q + 2], a[4 * q + 3]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a0), false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1), __float_as_uint(a1), false, false);
        qm[4 * q + 0] = __uint_as_float(s0[0]);
        qm[4 * q + 1] = __uint_as_float(s1[0]);
        qm[4 * q + 2] = __uint_as_float(s0[1]);
        qm[4 * q + 3] = __uint_as_float(s1[1]);
    }
}

// ent[], cn[]: the tile's NG plan entries (4 of 8 rows or 8 of 4 rows) and their balls' counts, WAVE-UNIFORM (SGPRs): every
// branch below is scalar.  Lanes 0..31 hold output channel c of the tile; writes relu(max + bias) (0 for empty balls,
// layers_util.py:178-181).
template <int NG>
__device__ __forceinline__ void pool_write_tile(float (&qm)[NG], const int (&ent)[NG], const int (&cn)[NG], float bias_c,
                                                int c, int N, float *out, int out_stride, int out_off, int lane) {
#pragma unroll
    for (int g = 1; g < NG; ++g)
        if (ent[g] >= 0 && ent[g - 1] >= 0 && plan_ball(ent[g]) == plan_ball(ent[g - 1])) qm[g] = fmax_nn(qm[g], qm[g - 1]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (ent[g] < 0) continue;
        const bool last = g == NG - 1 || ent[g < NG - 1 ? g + 1 : NG - 1] < 0 ||
                          plan_ball(ent[g < NG - 1 ? g + 1 : NG - 1]) != plan_ball(ent[g]);
        if (!last) continue;
        if (lane < 32 && c < N) {
            float v = qm[g] + bias_c;
            v = v > 0.0f ? v : 0.0f;
            if (cn[g] <= 0) v = 0.0f;
            float *p = out + (unsigned)(plan_ball(ent[g]) * out_stride + out_off + c);
            if (ent[g] & 1) atomicMax((unsigned *)p, __float_as_uint(v));
            else *p = v;
        }
    }
}

}  // namespace sa
