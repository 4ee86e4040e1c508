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
#include "mcmc_elem.h"

#define MCMC_CHUNK_F4 16384      // float4 per array and chunk: 256 KB

// VEC = 4: every pointer 16-byte aligned and n % 4 == 0 (float4 accesses); VEC = 1 otherwise.
// grad / prior / eps are read once and never again: non-temporal, so that x and the velocity
// (read and written) keep the cache.
struct McmcNoise {               // PHILOX instantiation only
    const int64_t* seeds;        // [n / chain_len] device
    int64_t chain_f4;            // float4 per chain (VEC = 4)
    int64_t chain_len;           // elements per chain
    int64_t step;                // used when step_dev is null
    const int64_t* step_dev;     // device counter (graph replay): the step is read from here
    float* eps_out;              // optional: the noise that was used (recording, tests)
    int64_t quad_w;              // 0: linear layout; > 0: column quads over rows of quad_w (mcmc_elem.h)
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
                float nn[4];
                if (nz.quad_w == 0) {
                    philox_normals4((uint64_t)j, step, seed, nn);
                    e = f4{nn[0], nn[1], nn[2], nn[3]};
                } else {
                    // this float4 is one row of four neighbouring column quads: a call per column,
                    // one normal of each (4 x the generator work; the path that needs this layout
                    // at speed is the adjoint, whose lanes own whole quads)
                    const int64_t e0 = 4 * j, row = e0 / nz.quad_w, col = e0 - row * nz.quad_w;
                    const uint64_t q0 = (uint64_t)((row >> 2) * nz.quad_w + col);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        philox_normals4(q0 + k, step, seed, nn);
                        const int c = (int)(row & 3);
                        e[k] = c == 0 ? nn[0] : c == 1 ? nn[1] : c == 2 ? nn[2] : nn[3];
                    }
                }
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
        const uint64_t step = PHILOX ? (uint64_t)(nz.step_dev ? *nz.step_dev : nz.step) : 0;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float v, e = (!PHILOX && eps) ? eps[i] : 0.f;
            if (PHILOX) {
                // the numbers of the float4 path, one element at a time (chains whose length is not a
                // multiple of 4, or buffers that are not 16-byte aligned): element j of a chain is
                // normal j & 3 of call j >> 2; in the quad layout normal row & 3 of the column's call
                const int64_t b = i / nz.chain_len, j = i - b * nz.chain_len;
                uint64_t call = (uint64_t)(j >> 2);
                int k = (int)(j & 3);
                if (nz.quad_w > 0) {
                    const int64_t row = j / nz.quad_w, col = j - row * nz.quad_w;
                    call = (uint64_t)((row >> 2) * nz.quad_w + col);
                    k = (int)(row & 3);
                }
                float nn[4];
                philox_normals4(call, step, (uint64_t)nz.seeds[b], nn);
                e = k == 0 ? nn[0] : k == 1 ? nn[1] : k == 2 ? nn[2] : nn[3];
                if (nz.eps_out) nz.eps_out[i] = e;
            }
            const float xn = mcmc_elem(x[i], grad[i], xt[i], prior ? prior[i] : 0.f, e,
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
                                  const int64_t* step_dev, int64_t quad_w, float* eps_out,
                                  int32_t* nan_flag, void* stream) {
    STEP_REQUIRE(x && grad && xt && p && seeds, "mcmc_update_philox: null pointer");
    STEP_REQUIRE(p->algo == 0 || p->algo == 1, "mcmc_update_philox: algo %d draws no noise", p->algo);
    STEP_REQUIRE(p->algo != 1 || velocity, "mcmc_update_philox: hmc needs a velocity buffer");
    STEP_REQUIRE(n > 0 && chain_len > 0 && n % chain_len == 0,
                 "mcmc_update_philox: n = %lld must be whole chains of chain_len = %lld",
                 (long long)n, (long long)chain_len);
    STEP_REQUIRE(step >= 0, "mcmc_update_philox: negative step");
    STEP_REQUIRE(quad_w >= 0 && (quad_w == 0 || chain_len % quad_w == 0),
                 "mcmc_update_philox: quad_w = %lld: chains must be whole rows of quad_w elements", (long long)quad_w);
    const uintptr_t al = (uintptr_t)x | (uintptr_t)velocity | (uintptr_t)grad | (uintptr_t)xt |
                         (uintptr_t)prior | (uintptr_t)eps_out;
    // float4 path: aligned buffers, chains of whole float4s and (quad layout) whole groups of four
    // rows of whole float4s; anything else takes the same numbers one element at a time
    const bool vec = al % 16 == 0 && chain_len % 4 == 0 &&
                     (quad_w == 0 || (quad_w % 4 == 0 && chain_len % (4 * quad_w) == 0));
    McmcNoise nz{seeds, chain_len / 4, chain_len, step, step_dev, eps_out, quad_w};
    if (vec)
        hipLaunchKernelGGL((k_mcmc_update<4, true>), dim3((unsigned)mcmc_blocks(n, true)), dim3(256), 0,
                           (hipStream_t)stream, x, velocity, grad, xt, prior, nullptr, n, *p, nan_flag, nz);
    else
        hipLaunchKernelGGL((k_mcmc_update<1, true>), dim3((unsigned)mcmc_blocks(n, false)), dim3(256), 0,
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
