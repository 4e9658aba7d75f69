// fp32 -> fp16 operand conversion of the one-pass grouped-MLP form, with the range guard.
//
// v_cvt_pk_f16_f32 rounds to nearest even and does NOT saturate: an activation above 65504 becomes +inf, inf * 0
// is NaN, and the library is built -fno-honor-nans.  BN-normalised activations are O(1-10), but that is a property of
// the checkpoint, not of the code (tf_util.py:424-444 folds whatever statistics were saved).  So every converted
// fragment is also tested for inf / NaN halves: a max tree over its packed registers (v_pk_max_f16; v_pk_max_u16 on
// the magnitude bits for the signed input fragments, so that NaN inputs are caught too), one add + and +
// compare, and the resulting lane mask is OR-ed into a 64-bit mask that lives in SCALAR registers.  (A per-lane VGPR
// carried through the whole persistent loop was tried first: the loop-carried vector register made hipcc's allocator
// spill 26-190 registers in the streamed kernels; the scalar mask leaves their allocation unchanged.)  When a wave is
// done and the mask is non-zero it raises the caller's overflow word with one relaxed atomic.  The result of an
// overflowed call is unspecified; the flag is what makes it loud (3dssd_amd/backbone.py raise_if_overflow,
// pipeline.Ticket.result).  The ReLU of a hidden layer is applied to the PACKED pair after the conversion
// (v_pk_max_f16 with 0; rounding is monotone and odd, so relu-then-convert == convert-then-relu): cvt + packed relu
// = 2 VALU per pair where max, max, cvt was 3, which pays for the guard.
#pragma once
#include "sa_common.h"

namespace sa {

typedef _Float16 act_h2 __attribute__((ext_vector_type(2)));
typedef float act_f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long f16_guard_t;      // wave-uniform: lanes that saw an out-of-range value

__device__ __forceinline__ unsigned cvt2_f16(float a, float b) {
    const act_f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, act_h2));
}
__device__ __forceinline__ unsigned cvt2_f16_relu(float a, float b) {
    const act_f2 v = {a, b};
    const act_h2 z = {(_Float16)0.0f, (_Float16)0.0f};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_convertvector(v, act_h2), z));
}
__device__ __forceinline__ unsigned pk_max_f16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(act_h2, a), __builtin_bit_cast(act_h2, b)));
}
// t: two NON-NEGATIVE fp16; is either >= 0x7C00 (inf / NaN)?  0x7C00 + 0x0400 = 0x8000, no carry between the halves
__device__ __forceinline__ bool f16_pair_bad(unsigned t) { return ((t + 0x04000400u) & 0x80008000u) != 0u; }

// fragments of non-negative values (after the packed ReLU)
__device__ __forceinline__ void f16_guard(const uint4 &a, f16_guard_t &g) {
    g |= __builtin_amdgcn_ballot_w64(f16_pair_bad(pk_max_f16(pk_max_f16(a.x, a.y), pk_max_f16(a.z, a.w))));
}
__device__ __forceinline__ void f16_guard(const uint2 &a, f16_guard_t &g) {
    g |= __builtin_amdgcn_ballot_w64(f16_pair_bad(pk_max_f16(a.x, a.y)));
}
// fragments of signed values (the gathered inputs): magnitude BITS compared as unsigned 16-bit integers
// (v_pk_max_u16).  The float maximum has maxNum semantics -- a NaN half next to a finite one would be dropped (ADVICE
// r3) -- while as integers NaN (0x7C01..0x7FFF) > inf (0x7C00) > every finite magnitude.  NaN inputs therefore raise
// the flag like out-of-range ones.  Hidden activations need no such test: a NaN there has a NaN or inf operand behind
// it, which this guard (inputs) or the inf test of an earlier layer has already flagged.
typedef unsigned short act_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(act_u2, a), __builtin_bit_cast(act_u2, b)));
}
__device__ __forceinline__ void f16_guard_signed(const uint4 &a, f16_guard_t &g) {
    const unsigned m = 0x7FFF7FFFu;
    g |= __builtin_amdgcn_ballot_w64(f16_pair_bad(pk_max_u16(pk_max_u16(a.x & m, a.y & m), pk_max_u16(a.z & m, a.w & m))));
}
// end of a wave's work (all lanes active): a non-zero mask -> *flag |= 1
__device__ __forceinline__ void f16_overflow_report(f16_guard_t g, int *flag, int lane) {
    if (g != 0ull && flag != nullptr && lane == 0) atomicOr(flag, 1);
}

}  // namespace sa
