// Fused Langevin / damped-HMC state update of the DAPS inner loop (HBM-bound, elementwise).
//
// Replaces the ~10 elementwise torch launches of score_fn (sampler/daps.py:93-104) and
// mc_update (:135-144) and the per-step host synchronisation of the NaN guard (:202-203):
//   score = -grad / tau^2 + (xt - x) / sigma^2 + prior
//   langevin: x += lr * score + sqrt(2 lr) * eps
//   hmc:      v  = momentum * v + sqrt(lr) * score + sqrt(2 (1 - momentum)) * eps ;  x += sqrt(lr) * v
// The python scalars of the reference (np.sqrt results, float64) enter fp32 tensor ops as fp32
// scalars; here they arrive already rounded to fp32 in mcmc_params_t.
#include "common.h"

__global__ __launch_bounds__(256) void k_mcmc_update(float* __restrict__ x, float* __restrict__ vel,
                                                     const float* __restrict__ grad,
                                                     const float* __restrict__ xt,
                                                     const float* __restrict__ prior,
                                                     const float* __restrict__ eps, int64_t n,
                                                     mcmc_params_t p, int32_t* __restrict__ nan_flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float xi = x[i];
        float s = -grad[i] * p.inv_tau2 + (xt[i] - xi) * p.inv_sigma2;
        if (prior) s += prior[i];
        float xn;
        if (p.algo == 0) {
            xn = xi + p.lr * s + p.sqrt_2lr * eps[i];
        } else {
            const float v = p.momentum * vel[i] + p.sqrt_lr * s + p.noise_scale * eps[i];
            vel[i] = v;
            xn = xi + v * p.sqrt_lr;
        }
        x[i] = xn;
        bad |= (xn != xn);
    }
    if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
}

extern "C" int mcmc_update(float* x, float* velocity, const float* grad, const float* xt,
                           const float* prior, const float* eps, int64_t n, const mcmc_params_t* p,
                           int32_t* nan_flag, void* stream) {
    STEP_REQUIRE(x && grad && xt && eps && p, "mcmc_update: null pointer");
    STEP_REQUIRE(n > 0, "mcmc_update: n = %lld", (long long)n);
    STEP_REQUIRE(p->algo == 0 || p->algo == 1, "mcmc_update: unknown algo %d", p->algo);
    STEP_REQUIRE(p->algo == 0 || velocity, "mcmc_update: hmc needs a velocity buffer");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mcmc_update, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                       velocity, grad, xt, prior, eps, n, *p, nan_flag);
    STEP_CHECK_LAUNCH("mcmc_update");
    return STEP_OK;
}
