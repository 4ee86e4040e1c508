// Scalar pieces of the closure quantities shared by bh_closure.hip (one workgroup per chain) and
// the fused epilogue / prologue of bh_dft_sep3.hip (blackhole.py:332-391, 448-491, 665-669).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cdir(float2 v, int conj) {   // blackhole.py:332-334
    return conj ? make_float2(v.x, -v.y) : v;
}
__device__ __forceinline__ float cabs(float2 v) { return hypotf(v.x, v.y); }

// 1 - cos(d), bh_weights_t::cphase_form.  Form 0, 2 sin^2(d/2): the reference's fp32 (1 - cos)
// cancels catastrophically once the chain fits the data (d ~ 1e-3 leaves ~3 significant bits);
// this form stays accurate to fp32 rounding and is what the fp64 evaluation of the reference
// formula converges to.  Form 1: the reference's literal expression (blackhole.py:473).
__device__ __forceinline__ float one_minus_cos(float d, int form) {
    if (form == 1) return 1.0f - cosf(d);
    const float s = sinf(0.5f * d);
    return 2.0f * s * s;
}

// e(chi) and e'(chi) of normalize_chisq (no gradient through the comparison)
__device__ __forceinline__ void fold(float chi, int normalize, float& e, float& de) {
    if (normalize && chi < 1.0f) { e = 1.0f / chi; de = -1.0f / (chi * chi); }
    else { e = chi; de = 1.0f; }
}

