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

// (u32 a, u32 b) -> two independent N(0,1): radius from a (uniform in (0,1]), angle from b
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1], 24 bits: exact
    const float r = sqrtf(-2.0f * logf(u));
    float sn, cs;
    sincospif((float)(b >> 8) * (2.0f / 16777216.0f), &sn, &cs);          // angle 2 pi b / 2^32
    n0 = r * cs;
    n1 = r * sn;
}

// VEC = 4: every pointer 16-byte aligned and n % 4 == 0 (float4 accesses); VEC = 1 otherwise.
// grad / prior / eps are read once and never again: non-temporal, so that x and the velocity
// (read and written) keep the cache.
struct McmcNoise {               // PHILOX instantiation only
    const int64_t* seeds;        // [n / chain_len] device
    int64_t chain_f4;            // float4 per chain
    int64_t step;                // used when step_dev is null
    const int64_t* step_dev;     // device counter (graph replay): the step is read from here
    float* eps_out;              // optional: the noise that was used (recording, tests)
};

template <int VEC, bool PHILOX>
__global__ __launch_bounds__(256) void k_mcmc_update(float* __restrict__ x, float* __restrict__ vel,
                                                     const float* __restrict__ grad,
                                                     const float* __restrict__ xt,
                                                     const float* __restrict__ prior,
                                                     const float* __restrict__ eps, int64_t n,
                                                     mcmc_params_t p, int32_t* __restrict__ nan_flag,
                                                     McmcNoise nz) {
    bool bad = false;
    // momentum 0: the old velocity has weight zero in every later step, so it is neither read nor
    // stored (1 GB less traffic per step on a 64-chain black-hole state)
    const bool rd_vel = p.algo == 1 && p.momentum != 0.f;
    const bool wr_vel = rd_vel;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int64_t n4 = n >> 2;
        const uint64_t step = PHILOX ? (uint64_t)(nz.step_dev ? *nz.step_dev : nz.step) : 0;
        auto body = [&](int64_t i) {
            const f4 xi = ((const f4*)x)[i];
            const f4 g = __builtin_nontemporal_load((const f4*)grad + i);
            f4 e = {0.f, 0.f, 0.f, 0.f};
            if (PHILOX) {
                const int64_t b = i / nz.chain_f4, j = i - b * nz.chain_f4;
                const uint64_t seed = (uint64_t)nz.seeds[b];
                uint32_t r[4];
                philox4x32_10((uint32_t)j, (uint32_t)((uint64_t)j >> 32), (uint32_t)step, (uint32_t)(step >> 32),
                              (uint32_t)seed, (uint32_t)(seed >> 32), r);
                float n0, n1, n2, n3;
                box_muller(r[0], r[1], n0, n1);
                box_muller(r[2], r[3], n2, n3);
                e = f4{n0, n1, n2, n3};
                if (nz.eps_out) ((f4*)nz.eps_out)[i] = e;
            } else if (eps) e = __builtin_nontemporal_load((const f4*)eps + i);
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
    // one atomic per wave that saw a NaN, and none once the flag is up: on a diverged state every
    // wave of the grid would otherwise queue at one address (measured 16 -> 165 us on a 17 MB state)
    if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0 &&
        __hip_atomic_load(nan_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicOr(nan_flag, 1);
}

static int mcmc_blocks(int64_t n, bool vec) {
    int64_t blocks = ((vec ? n / 4 : n) + 255) / 256;
    return (int)(blocks > 4096 ? 4096 : blocks);
}

extern "C" int mcmc_update_philox(float* x, float* velocity, const float* grad, const float* xt,
                                  const float* prior, int64_t n, const mcmc_params_t* p,
                                  const int64_t* seeds, int64_t chain_len, int64_t step,
                                  const int64_t* step_dev, float* eps_out, int32_t* nan_flag,
                                  void* stream) {
    STEP_REQUIRE(x && grad && xt && p && seeds, "mcmc_update_philox: null pointer");
    STEP_REQUIRE(p->algo == 0 || p->algo == 1, "mcmc_update_philox: algo %d draws no noise", p->algo);
    STEP_REQUIRE(p->algo != 1 || velocity, "mcmc_update_philox: hmc needs a velocity buffer");
    STEP_REQUIRE(n > 0 && chain_len > 0 && n % chain_len == 0 && chain_len % 4 == 0,
                 "mcmc_update_philox: n = %lld must be whole chains of chain_len = %lld (a multiple of 4)",
                 (long long)n, (long long)chain_len);
    STEP_REQUIRE(step >= 0, "mcmc_update_philox: negative step");
    const uintptr_t al = (uintptr_t)x | (uintptr_t)velocity | (uintptr_t)grad | (uintptr_t)xt |
                         (uintptr_t)prior | (uintptr_t)eps_out;
    STEP_REQUIRE(al % 16 == 0, "mcmc_update_philox: buffers must be 16-byte aligned");
    McmcNoise nz{seeds, chain_len / 4, step, step_dev, eps_out};
    hipLaunchKernelGGL((k_mcmc_update<4, true>), dim3((unsigned)mcmc_blocks(n, true)), dim3(256), 0,
                       (hipStream_t)stream, x, velocity, grad, xt, prior, nullptr, n, *p, nan_flag, nz);
    STEP_CHECK_LAUNCH("mcmc_update_philox");
    return STEP_OK;
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
    const int blocks = mcmc_blocks(n, vec);
    if (vec)
        hipLaunchKernelGGL((k_mcmc_update<4, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           x, velocity, grad, xt, prior, eps, n, *p, nan_flag, McmcNoise{});
    else
        hipLaunchKernelGGL((k_mcmc_update<1, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           x, velocity, grad, xt, prior, eps, n, *p, nan_flag, McmcNoise{});
    STEP_CHECK_LAUNCH("mcmc_update");
    return STEP_OK;
}
