// The Langevin / damped-HMC / gradient-descent update of one state element and the in-kernel noise
// (Philox4x32-10 + Box-Muller), shared by k_mcmc_update (mcmc_update.hip) and the black-hole
// adjoint that applies the update where it would have stored the gradient (bh_dft_sep3.hip).
#pragma once
#include "common.h"

// One element of the update; `v_old` is only read when momentum != 0 (the shipped blackhole
// configuration runs hmc with momentum 0, where the old velocity has weight zero).
// mcmc_elem_a<ALGO>: the form is known at compile time (the adjoint's instantiations: a run-time
// p.algo there becomes scalar branches per element that cut the tile loop into basic blocks).
template <int ALGO>
__device__ __forceinline__ float mcmc_elem_a(float xi, float g, float xti, float pr, float e, float v_old,
                                             const mcmc_params_t& p, float& v_new) {
    if (ALGO == 2) {     // MCMCSampler.optimize, daps.py:210-223: x -= lr (g + (x - x0hat) / sigma^2)
        v_new = 0.f;
        return xi - p.lr * (g + (xi - xti) * p.inv_sigma2);
    }
    const float s = -g * p.inv_tau2 + (xti - xi) * p.inv_sigma2 + pr;
    if (ALGO == 0) {
        v_new = 0.f;
        return xi + p.lr * s + p.sqrt_2lr * e;
    }
    v_new = p.momentum * v_old + p.sqrt_lr * s + p.noise_scale * e;
    return xi + v_new * p.sqrt_lr;
}

__device__ __forceinline__ float mcmc_elem(float xi, float g, float xti, float pr, float e, float v_old,
                                           const mcmc_params_t& p, float& v_new) {
    if (p.algo == 2) return mcmc_elem_a<2>(xi, g, xti, pr, e, v_old, p, v_new);
    if (p.algo == 0) return mcmc_elem_a<0>(xi, g, xti, pr, e, v_old, p, v_new);
    return mcmc_elem_a<1>(xi, g, xti, pr, e, v_old, p, v_new);
}

// ---- in-kernel noise: Philox4x32-10 (Salmon et al., SC'11), counter = (float4 index within the
// chain, 0, step lo, step hi), key = the chain's 64-bit seed; four uniforms -> four normals by
// Box-Muller.  A chain's noise depends only on (its seed, the step, the element): not on the
// batch it is run in, the rank, or the launch geometry.  oracle/philox_oracle.py restates it.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const uint32_t lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// (u32 a, u32 b) -> two independent N(0,1): radius from a (uniform in (0,1]), angle from b.
// Hardware log2 / sqrt / sin / cos (v_log_f32, v_sqrt_f32: 1 ulp; v_sin_f32 / v_cos_f32 take
// the angle in revolutions, absolute error ~1e-6): no branches and ~12 instructions -- the
// library logf / sincospif cost ~10 x that and their range checks split the adjoint's tile loop
// into dozens of basic blocks (spills).  Against the fp64 oracle the normals agree to 2e-6 * max.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1], 24 bits: exact
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));   // sqrt(-2 ln u)
    const float ph = (float)(b >> 8) * (1.0f / 16777216.0f);             // angle / 2 pi in [0, 1)
    n0 = r * __builtin_amdgcn_cosf(ph);
    n1 = r * __builtin_amdgcn_sinf(ph);
}

// The noise of one Philox call (four normals).  Two layouts of the four over a chain's elements:
// linear -- call j covers elements 4 j .. 4 j + 3 (mcmc_update_philox, quad_w = 0);
// column quad, for states made of rows of quad_w elements -- call j = (row / 4) * quad_w + column
// covers rows 4 (row / 4) .. + 3 of one column, which is what one lane of the adjoint's
// accumulator tile holds (mcmc_update_philox with quad_w > 0 draws the same numbers).
__device__ __forceinline__ void philox_normals4(uint64_t j, uint64_t step, uint64_t seed, float (&n)[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)step, (uint32_t)(step >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    box_muller(r[0], r[1], n[0], n[1]);
    box_muller(r[2], r[3], n[2], n[3]);
}
