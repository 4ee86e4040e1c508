// Fused Langevin / damped-HMC state update of the DAPS inner loop (HBM-bound, elementwise).
//
// Replaces the ~10 elementwise torch launches of score_fn (sampler/daps.py:93-104) and
// mc_update (:135-144) and the per-step host synchronisation of the NaN guard (:202-203):
//   score = -grad / tau^2 + (xt - x) / sigma^2 + prior
//   langevin: x += lr * score + sqrt(2 lr) * eps
//   hmc:      v  = momentum * v + sqrt(lr) * score + sqrt(2 (1 - momentum)) * eps ;  x += sqrt(lr) * v
//   optimize (algo 2, daps.py:210-223):  x -= lr * (grad + (x - xt) / sigma^2)   with xt = x0hat
// The python scalars of the reference (np.sqrt results, float64) enter fp32 tensor ops as fp32
// scalars; here they arrive already rounded to fp32 in mcmc_params_t.
#include "common.h"

// One element of the update; `v_old` is only read when momentum != 0 (the shipped blackhole
// configuration runs hmc with momentum 0, where the old velocity has weight zero).
__device__ __forceinline__ float mcmc_elem(float xi, float g, float xti, float pr, float e, float v_old,
                                           const mcmc_params_t& p, float& v_new) {
    if (p.algo == 2) {   // MCMCSampler.optimize, daps.py:210-223: x -= lr (g + (x - x0hat) / sigma^2)
        v_new = 0.f;
        return xi - p.lr * (g + (xi - xti) * p.inv_sigma2);
    }
    const float s = -g * p.inv_tau2 + (xti - xi) * p.inv_sigma2 + pr;
    if (p.algo == 0) {
        v_new = 0.f;
        return xi + p.lr * s + p.sqrt_2lr * e;
    }
    v_new = p.momentum * v_old + p.sqrt_lr * s + p.noise_scale * e;
    return xi + v_new * p.sqrt_lr;
}

#define MCMC_CHUNK_F4 16384      // float4 per array and chunk: 256 KB

// VEC = 4: every pointer 16-byte aligned and n % 4 == 0 (float4 accesses); VEC = 1 otherwise.
// grad / prior / eps are read once and never again: non-temporal, so that x and the velocity
// (read and written) keep the cache.
template <int VEC>
__global__ __launch_bounds__(256) void k_mcmc_update(float* __restrict__ x, float* __restrict__ vel,
                                                     const float* __restrict__ grad,
                                                     const float* __restrict__ xt,
                                                     const float* __restrict__ prior,
                                                     const float* __restrict__ eps, int64_t n,
                                                     mcmc_params_t p, int32_t* __restrict__ nan_flag) {
    bool bad = false;
    // momentum 0: the old velocity has weight zero in every later step, so it is neither read nor
    // stored (1 GB less traffic per step on a 64-chain black-hole state)
    const bool rd_vel = p.algo == 1 && p.momentum != 0.f;
    const bool wr_vel = rd_vel;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int64_t n4 = n >> 2;
        auto body = [&](int64_t i) {
            const f4 xi = ((const f4*)x)[i];
            const f4 g = __builtin_nontemporal_load((const f4*)grad + i);
            f4 e = {0.f, 0.f, 0.f, 0.f};
            if (eps) e = __builtin_nontemporal_load((const f4*)eps + i);
            const f4 xti = ((const f4*)xt)[i];
            f4 pr = {0.f, 0.f, 0.f, 0.f}, vo = {0.f, 0.f, 0.f, 0.f}, xn, vn;
            if (prior) pr = __builtin_nontemporal_load((const f4*)prior + i);
            if (rd_vel) vo = ((const f4*)vel)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v;
                xn[k] = mcmc_elem(xi[k], g[k], xti[k], pr[k], e[k], vo[k], p, v);
                vn[k] = v;
                bad |= (xn[k] != xn[k]);
            }
            ((f4*)x)[i] = xn;
            if (wr_vel) ((f4*)vel)[i] = vn;
        };
        // Workgroup ids go round the 8 XCDs.  Large states are dealt to the XCDs in chunks of
        // 256 KB per array (XCD c takes the chunks q with q % 8 == c) instead of 4 KB: 9-14 %
        // faster on a 1 GB state (scripts/probes/xcd_probe.*, scripts/kbench_mcmc.py).  The
        // remainder past the last full round of 8 chunks goes to the workgroups of XCD 0.
        const int64_t CH = MCMC_CHUNK_F4, nfull = n4 / (8 * CH) * (8 * CH);
        if ((gridDim.x & 7) == 0 && nfull > 0) {
            const int c8 = blockIdx.x & 7;
            const int64_t per_xcd = nfull / 8, nu = per_xcd + (c8 == 0 ? n4 - nfull : 0);
            const int64_t ustride = (int64_t)(gridDim.x >> 3) * blockDim.x;
            for (int64_t u = (int64_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x; u < nu; u += ustride)
                body(u < per_xcd ? ((u / CH) * 8 + c8) * CH + (u % CH) : nfull + (u - per_xcd));
        } else {
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) body(i);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float v;
            const float xn = mcmc_elem(x[i], grad[i], xt[i], prior ? prior[i] : 0.f, eps ? eps[i] : 0.f,
                                       rd_vel ? vel[i] : 0.f, p, v);
            if (wr_vel) vel[i] = v;
            x[i] = xn;
            bad |= (xn != xn);
        }
    }
    if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
}

extern "C" int mcmc_update(float* x, float* velocity, const float* grad, const float* xt,
                           const float* prior, const float* eps, int64_t n, const mcmc_params_t* p,
                           int32_t* nan_flag, void* stream) {
    STEP_REQUIRE(x && grad && xt && p, "mcmc_update: null pointer");
    STEP_REQUIRE(n > 0, "mcmc_update: n = %lld", (long long)n);
    STEP_REQUIRE(p->algo >= 0 && p->algo <= 2, "mcmc_update: unknown algo %d", p->algo);
    STEP_REQUIRE(p->algo != 1 || velocity, "mcmc_update: hmc needs a velocity buffer");
    STEP_REQUIRE(p->algo == 2 || eps, "mcmc_update: langevin / hmc need the noise tensor");
    const uintptr_t al = (uintptr_t)x | (uintptr_t)velocity | (uintptr_t)grad | (uintptr_t)xt |
                         (uintptr_t)prior | (uintptr_t)eps;
    const bool vec = (n % 4 == 0) && (al % 16 == 0);
    int64_t blocks = ((vec ? n / 4 : n) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (vec)
        hipLaunchKernelGGL(k_mcmc_update<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           x, velocity, grad, xt, prior, eps, n, *p, nan_flag);
    else
        hipLaunchKernelGGL(k_mcmc_update<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           x, velocity, grad, xt, prior, eps, n, *p, nan_flag);
    STEP_CHECK_LAUNCH("mcmc_update");
    return STEP_OK;
}
