// Closure quantities over baseline triangles / quadrangles, their chi^2 terms, the
// normalised weighted loss and its gradient w.r.t. the visibilities -- one workgroup per
// chain, everything staged in LDS (a chain's visibilities are 14 KB at 64 x 28 baselines).
//
// Replaces, from forward_operator/blackhole.py: correct_vis_direction :332-334,
// forward_bisepectra_from_vis :341-353, forward_cp_from_vis :355-363,
// forward_logcamp_bispectra_from_vis :365-380, forward_logcamp_from_vis :382-391,
// forward_amp_from_vis :336-339, chi2_* :448-491, normalize_chisq :665-669, the weighted
// sum of loss :736-771, and the autograd backward of all of it.
// The backward scatter over triangles/quadrangles is a gather through the transposed
// incidence tables (bh_build_incidence), so results are bitwise reproducible.
#include "common.h"
#include "closure_math.h"

#define CL_THREADS 256
#define CLL_THREADS 1024     // k_closure_loss: latency-bound, one workgroup per chain

struct ClosureArgs {
    bh_tables_t tab;
    bh_meas_t y;
    bh_weights_t w;
};

// LDS: sv float2[n_vis] | dcp float[n_tri] | dca float[n_quad] | red float[96]
//      | (STAGED) cp_ptr int[n_vis+1] | ca_ptr int[n_vis+1] | cp_ent int[3 n_tri] | ca_ent int[4 n_quad]
// STAGED: the incidence tables of the backward gather are copied into LDS first (coalesced), so
// the per-baseline walk ptr -> ent -> d is LDS latency instead of a chain of dependent L2 reads.
template <bool STAGED>
__global__ __launch_bounds__(CLL_THREADS) void k_closure_loss(
    const float2* __restrict__ vis, const float* __restrict__ flux, ClosureArgs a,
    float* __restrict__ loss, float* __restrict__ chi2, float2* __restrict__ gvis,
    float* __restrict__ gflux) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int M = a.tab.n_vis, Kcp = a.tab.n_tri, Kca = a.tab.n_quad, T = a.tab.T;
    float2* sv = (float2*)lds;
    float* dcp = lds + 2 * (size_t)M;
    float* dca = dcp + Kcp;
    float* red = dca + Kca;
    const int* cp_ptr = a.tab.cp_ptr;
    const int* ca_ptr = a.tab.ca_ptr;
    const int* cp_ent = a.tab.cp_ent;
    const int* ca_ent = a.tab.ca_ent;
    // The kernel is a chain of latencies, so every global load that depends on nothing is issued
    // here, together: this chain's visibilities (the first two per thread; the rest, if any,
    // below), and the table entries and measurement of the thread's first triangle and first
    // quadrangle.  Issued one phase at a time they cost an L2 / HBM round trip each.
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t yb = a.y.batch > 1 ? (size_t)b : 0;
    const float2 v_pre0 = vis[(size_t)b * M + min(tid, M - 1)];
    const float2 v_pre1 = vis[(size_t)b * M + min(tid + CLL_THREADS, M - 1)];
    int t_idx[3] = {0, 0, 0}, t_cj[3] = {0, 0, 0}, q_idx[4] = {0, 0, 0, 0};
    float t_y = 0.f, t_s = 1.f, q_y = 0.f, q_s = 1.f;
    if (a.w.w_cp > 0.f && tid < Kcp) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { t_idx[j] = a.tab.cp_index[3 * tid + j]; t_cj[j] = a.tab.cp_conj[3 * tid + j]; }
        t_y = (a.y.cphase + yb * Kcp)[tid];
        t_s = (a.y.sigma_cphase + yb * Kcp)[tid];
    }
    if (a.w.w_camp > 0.f && tid < Kca) {
#pragma unroll
        for (int j = 0; j < 4; ++j) q_idx[j] = a.tab.camp_index[4 * tid + j];
        q_y = (a.y.logcamp + yb * Kca)[tid];
        q_s = (a.y.sigma_logcamp + yb * Kca)[tid];
    }
    if (STAGED && gvis) {
        // only this workgroup's slice of the baselines: pointers [s0, s1] and the entries between
        // ptr[s0] and ptr[s1], kept at their global positions (the unused part of the LDS image
        // is never touched)
        const int s0 = (int)((long long)M * blockIdx.y / gridDim.y), s1 = (int)((long long)M * (blockIdx.y + 1) / gridDim.y);
        int* l_cp_ptr = (int*)(red + 96);
        int* l_ca_ptr = l_cp_ptr + M + 1;
        int* l_cp_ent = l_ca_ptr + M + 1;
        int* l_ca_ent = l_cp_ent + 3 * Kcp;
        if (a.w.w_cp > 0.f) {
            for (int i = s0 + threadIdx.x; i <= s1; i += CLL_THREADS) l_cp_ptr[i] = cp_ptr[i];
            for (int i = cp_ptr[s0] + threadIdx.x; i < cp_ptr[s1]; i += CLL_THREADS) l_cp_ent[i] = cp_ent[i];
        }
        if (a.w.w_camp > 0.f) {
            for (int i = s0 + threadIdx.x; i <= s1; i += CLL_THREADS) l_ca_ptr[i] = ca_ptr[i];
            for (int i = ca_ptr[s0] + threadIdx.x; i < ca_ptr[s1]; i += CLL_THREADS) l_ca_ent[i] = ca_ent[i];
        }
        cp_ptr = l_cp_ptr; ca_ptr = l_ca_ptr; cp_ent = l_cp_ent; ca_ent = l_ca_ent;
    }
    // gridDim.y workgroups share a chain: every one of them evaluates the chi^2 sums (they need
    // all closure quantities), each takes a slice [m0, m1) of the baselines in the backward gather
    const int m0 = (int)((long long)M * blockIdx.y / gridDim.y), m1 = (int)((long long)M * (blockIdx.y + 1) / gridDim.y);

    if (tid < M) sv[tid] = v_pre0;
    if (tid + CLL_THREADS < M) sv[tid + CLL_THREADS] = v_pre1;
    for (int m = tid + 2 * CLL_THREADS; m < M; m += CLL_THREADS) sv[m] = vis[(size_t)b * M + m];
    __syncthreads();

    // --- closure phase: chi = 2 mean((1 - cos(y - phi)) / s^2), blackhole.py:472-474
    float s_cp = 0.f;
    if (a.w.w_cp > 0.f) {
        const float* yv = a.y.cphase + yb * Kcp;
        const float* ys = a.y.sigma_cphase + yb * Kcp;
        for (int k = tid; k < Kcp; k += CLL_THREADS) {
            if (k != tid) {                                // the first one came with the prologue
#pragma unroll
                for (int j = 0; j < 3; ++j) { t_idx[j] = a.tab.cp_index[3 * k + j]; t_cj[j] = a.tab.cp_conj[3 * k + j]; }
                t_y = yv[k]; t_s = ys[k];
            }
            const float2 i1 = cdir(sv[t_idx[0]], t_cj[0]);
            const float2 i2 = cdir(sv[t_idx[1]], t_cj[1]);
            const float2 i3 = cdir(sv[t_idx[2]], t_cj[2]);
            const float2 p = cmul(cmul(i1, i2), i3);
            const float phi = atan2f(p.y, p.x);
            const float d = t_y - phi;
            const float is2 = 1.0f / (t_s * t_s);
            s_cp += one_minus_cos(d, a.w.cphase_form) * is2;
            dcp[k] = -sinf(d) * is2;                       // d/dphi of (1 - cos(y - phi))/s^2
        }
    }
    // --- log closure amplitude: chi = mean(((l - y)/s)^2), blackhole.py:482-483
    float s_ca = 0.f;
    if (a.w.w_camp > 0.f) {
        const float* yv = a.y.logcamp + yb * Kca;
        const float* ys = a.y.sigma_logcamp + yb * Kca;
        for (int k = tid; k < Kca; k += CLL_THREADS) {
            if (k != tid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) q_idx[j] = a.tab.camp_index[4 * k + j];
                q_y = yv[k]; q_s = ys[k];
            }
            const float a1 = cabs(sv[q_idx[0]]);
            const float a2 = cabs(sv[q_idx[1]]);
            const float a3 = cabs(sv[q_idx[2]]);
            const float a4 = cabs(sv[q_idx[3]]);
            const float l = logf(a1) + logf(a2) - logf(a3) - logf(a4);
            const float r = (l - q_y) / q_s;
            s_ca += r * r;
            dca[k] = 2.0f * r / q_s;                       // d/dl of ((l - y)/s)^2
        }
    }
    // --- visibility / amplitude terms (zero weight in the shipped config), :453-464
    float s_vis = 0.f, s_amp = 0.f;
    if (a.w.w_vis > 0.f) {
        const float2* yv = (const float2*)a.y.vis_c64 + yb * M;
        const float* ys = a.y.sigma_vis + yb * M;
        for (int m = tid; m < M; m += CLL_THREADS) {
            const float r = cabs(make_float2(sv[m].x - yv[m].x, sv[m].y - yv[m].y)) / ys[m];
            s_vis += r * r;
        }
    }
    if (a.w.w_amp > 0.f) {
        const float* yv = a.y.amp + yb * M;
        const float* ys = a.y.sigma_amp + yb * M;
        for (int m = tid; m < M; m += CLL_THREADS) {
            const float r = (cabs(sv[m]) - yv[m]) / ys[m];
            s_amp += r * r;
        }
    }
    // --- flux: mean(((f - y)/2)^2), :490-491
    float s_fl = 0.f;
    if (a.w.w_flux > 0.f) {
        const float* yv = a.y.flux + yb * T;
        for (int t = tid; t < T; t += CLL_THREADS) {
            const float r = (flux[(size_t)b * T + t] - yv[t]) * 0.5f;
            s_fl += r * r;
        }
    }
    // the five partial sums in one pass: wave sums, one LDS exchange, every thread adds the
    // per-wave values in wave order (two barriers instead of ten)
    float tot[5] = {wave_sum(s_cp), wave_sum(s_ca), wave_sum(s_vis), wave_sum(s_amp), wave_sum(s_fl)};
    {
        const int w = tid >> 6, nw = CLL_THREADS >> 6;
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) red[q * nw + w] = tot[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            float t = 0.f;
            for (int i = 0; i < nw; ++i) t += red[q * nw + i];
            tot[q] = t;
        }
    }
    const float chi_cp = a.w.w_cp > 0.f ? 2.0f * tot[0] / Kcp : 0.f;
    const float chi_ca = a.w.w_camp > 0.f ? tot[1] / Kca : 0.f;
    const float chi_vis = a.w.w_vis > 0.f ? tot[2] / M * 0.5f : 0.f;
    const float chi_amp = a.w.w_amp > 0.f ? tot[3] / M : 0.f;
    const float chi_fl = a.w.w_flux > 0.f ? tot[4] / T : 0.f;

    float e_cp = 0.f, d_cp = 0.f, e_ca = 0.f, d_ca = 0.f, e_vis = 0.f, d_vis = 0.f, e_amp = 0.f,
          d_amp = 0.f;
    if (a.w.w_cp > 0.f) fold(chi_cp, a.w.normalize, e_cp, d_cp);
    if (a.w.w_camp > 0.f) fold(chi_ca, a.w.normalize, e_ca, d_ca);
    if (a.w.w_vis > 0.f) fold(chi_vis, a.w.normalize, e_vis, d_vis);
    if (a.w.w_amp > 0.f) fold(chi_amp, a.w.normalize, e_amp, d_amp);
    if (tid == 0 && blockIdx.y == 0) {
        // same accumulation order as loss(): vis, amp, cp, camp, flux (blackhole.py:742-769)
        float L = 0.f;
        if (a.w.w_vis > 0.f) L += a.w.w_vis * e_vis;
        if (a.w.w_amp > 0.f) L += a.w.w_amp * e_amp;
        if (a.w.w_cp > 0.f) L += a.w.w_cp * e_cp;
        if (a.w.w_camp > 0.f) L += a.w.w_camp * e_ca;
        if (a.w.w_flux > 0.f) L += a.w.w_flux * chi_fl;
        loss[b] = L;
        if (chi2) {
            chi2[b * 5 + 0] = chi_vis; chi2[b * 5 + 1] = chi_amp; chi2[b * 5 + 2] = chi_cp;
            chi2[b * 5 + 3] = chi_ca;  chi2[b * 5 + 4] = chi_fl;
        }
    }
    if (!gvis) return;

    // --- backward: dL/dchi factors, then gather per baseline through the incidence tables
    const float f_cp = a.w.w_cp * d_cp * 2.0f / Kcp;
    const float f_ca = a.w.w_camp * d_ca / Kca;
    const float f_vis = a.w.w_vis * d_vis / M;          // d/dv of |v-y|^2/(2 s^2 M) = (v-y)/(s^2 M)
    const float f_amp = a.w.w_amp * d_amp * 2.0f / M;
    __syncthreads();                                     // dcp / dca complete
    for (int m = m0 + tid; m < m1; m += CLL_THREADS) {
        const float2 v = sv[m];
        const float n2 = v.x * v.x + v.y * v.y;
        float gp = 0.f, gl = 0.f;                        // coefficients of i v/|v|^2 and v/|v|^2
        if (a.w.w_cp > 0.f) {
            for (int e = cp_ptr[m]; e < cp_ptr[m + 1]; ++e) {
                const int ent = cp_ent[e];
                const float d = dcp[ent >> 3];
                gp += (ent & 1) ? -d : d;                // conj leg: arg(conj v) = -arg v
            }
            gp *= f_cp;
        }
        if (a.w.w_camp > 0.f) {
            for (int e = ca_ptr[m]; e < ca_ptr[m + 1]; ++e) {
                const int ent = ca_ent[e];
                const float d = dca[ent >> 3];
                gl += (((ent >> 1) & 3) < 2) ? d : -d;   // legs 0,1 numerator, 2,3 denominator
            }
            gl *= f_ca;
        }
        float gx = 0.f, gy = 0.f;
        if (a.w.w_cp > 0.f || a.w.w_camp > 0.f) {
            gx = (gl * v.x - gp * v.y) / n2;
            gy = (gl * v.y + gp * v.x) / n2;
        }
        if (a.w.w_vis > 0.f) {
            const float2 yv = ((const float2*)a.y.vis_c64 + yb * M)[m];
            const float s = (a.y.sigma_vis + yb * M)[m];
            gx += f_vis * (v.x - yv.x) / (s * s);
            gy += f_vis * (v.y - yv.y) / (s * s);
        }
        if (a.w.w_amp > 0.f) {
            const float s = (a.y.sigma_amp + yb * M)[m];
            const float am = sqrtf(n2);
            const float c = f_amp * (am - (a.y.amp + yb * M)[m]) / (s * s) / am;
            gx += c * v.x;
            gy += c * v.y;
        }
        gvis[(size_t)b * M + m] = make_float2(gx, gy);
    }
    if (blockIdx.y == 0)
        for (int t = tid; t < T; t += CLL_THREADS) {
            float g = 0.f;
            if (a.w.w_flux > 0.f)
                g = a.w.w_flux * (flux[(size_t)b * T + t] - (a.y.flux + yb * T)[t]) * 0.5f / T;
            gflux[(size_t)b * T + t] = g;
        }
}

static int check_tables(const char* who, const bh_tables_t* tab) {
    STEP_REQUIRE(tab, "%s: null tables", who);
    STEP_REQUIRE(tab->n_vis > 0 && tab->n_tri >= 0 && tab->n_quad >= 0 && tab->T > 0,
                 "%s: bad table sizes", who);
    STEP_REQUIRE(tab->n_tri == 0 || (tab->cp_index && tab->cp_conj), "%s: null cp tables", who);
    STEP_REQUIRE(tab->n_quad == 0 || tab->camp_index, "%s: null camp tables", who);
    const size_t lds = (2 * (size_t)tab->n_vis + tab->n_tri + tab->n_quad + 96) * sizeof(float);
    STEP_REQUIRE(lds <= 150 * 1024, "%s: tables need %zu B of LDS (max 150 KiB)", who, lds);
    return STEP_OK;
}

extern "C" int bh_closure_loss(const float* vis_c64, const float* flux, int B,
                               const bh_tables_t* tab, const bh_meas_t* y, const bh_weights_t* w,
                               float* loss, float* chi2, float* gvis_c64, float* gflux,
                               void* stream) {
    STEP_REQUIRE(vis_c64 && flux && y && w && loss && B > 0, "bh_closure_loss: null pointer / B");
    if (int rc = check_tables("bh_closure_loss", tab)) return rc;
    STEP_REQUIRE((gvis_c64 == nullptr) == (gflux == nullptr), "bh_closure_loss: gvis and gflux go together");
    STEP_REQUIRE(y->batch == 1 || y->batch == B, "bh_closure_loss: measurement batch %d vs B=%d", y->batch, B);
    STEP_REQUIRE(w->cphase_form == 0 || w->cphase_form == 1, "bh_closure_loss: cphase_form %d", w->cphase_form);
    STEP_REQUIRE(!(w->w_cp > 0) || (tab->n_tri > 0 && y->cphase && y->sigma_cphase), "bh_closure_loss: cphase data missing");
    STEP_REQUIRE(!(w->w_camp > 0) || (tab->n_quad > 0 && y->logcamp && y->sigma_logcamp), "bh_closure_loss: logcamp data missing");
    STEP_REQUIRE(!(w->w_vis > 0) || (y->vis_c64 && y->sigma_vis), "bh_closure_loss: vis data missing");
    STEP_REQUIRE(!(w->w_amp > 0) || (y->amp && y->sigma_amp), "bh_closure_loss: amp data missing");
    STEP_REQUIRE(!(w->w_flux > 0) || y->flux, "bh_closure_loss: flux data missing");
    STEP_REQUIRE(!gvis_c64 || ((!(w->w_cp > 0) || (tab->cp_ptr && tab->cp_ent)) &&
                               (!(w->w_camp > 0) || (tab->ca_ptr && tab->ca_ent))),
                 "bh_closure_loss: incidence tables missing");
    ClosureArgs a{*tab, *y, *w};
    const size_t base = (2 * (size_t)tab->n_vis + tab->n_tri + tab->n_quad + 96) * sizeof(float);
    const size_t staged = base + (2 * ((size_t)tab->n_vis + 1) + 3 * (size_t)tab->n_tri +
                                  4 * (size_t)tab->n_quad) * sizeof(int);
    // the kernel is latency-bound and B workgroups leave most CUs idle: split the backward gather
    // of a chain over several workgroups while the grid stays within one round
    int parts = 1;
    constexpr int CLL_PARTS = 4;
    if (gvis_c64) while (parts < CLL_PARTS && B * parts * 2 <= 256) parts *= 2;
    if (staged <= 150 * 1024) {
        step_allow_lds((const void*)k_closure_loss<true>, staged);
        hipLaunchKernelGGL(k_closure_loss<true>, dim3(B, parts), dim3(CLL_THREADS), staged, (hipStream_t)stream,
                           (const float2*)vis_c64, flux, a, loss, chi2, (float2*)gvis_c64, gflux);
    } else {
        step_allow_lds((const void*)k_closure_loss<false>, base);
        hipLaunchKernelGGL(k_closure_loss<false>, dim3(B, parts), dim3(CLL_THREADS), base, (hipStream_t)stream,
                           (const float2*)vis_c64, flux, a, loss, chi2, (float2*)gvis_c64, gflux);
    }
    STEP_CHECK_LAUNCH("bh_closure_loss");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- evaluator chi^2
__global__ __launch_bounds__(CL_THREADS) void k_closure_chisq(
    const float2* __restrict__ vis, const float* __restrict__ vis_scale, int F, ClosureArgs a,
    float* __restrict__ chi2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int M = a.tab.n_vis, Kcp = a.tab.n_tri, Kca = a.tab.n_quad, T = a.tab.T;
    float2* sv = (float2*)lds;
    float* red = lds + 2 * (size_t)M;
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t yb = a.y.batch > 1 ? (size_t)b : 0;
    for (int m = tid; m < M; m += CL_THREADS) {
        float2 v = vis[(size_t)b * M + m];
        if (vis_scale) {
            const float s = vis_scale[(size_t)b * T + m / F];
            v.x *= s; v.y *= s;
        }
        sv[m] = v;
    }
    __syncthreads();
    float s_vis = 0.f, s_amp = 0.f, s_cp = 0.f, s_ca = 0.f;
    if (a.y.vis_c64 && a.y.sigma_vis) {
        const float2* yv = (const float2*)a.y.vis_c64 + yb * M;
        const float* ys = a.y.sigma_vis + yb * M;
        for (int m = tid; m < M; m += CL_THREADS) {
            const float r = cabs(make_float2(sv[m].x - yv[m].x, sv[m].y - yv[m].y)) / ys[m];
            s_vis += r * r;
        }
    }
    if (a.y.amp && a.y.sigma_amp) {
        const float* yv = a.y.amp + yb * M;
        const float* ys = a.y.sigma_amp + yb * M;
        for (int m = tid; m < M; m += CL_THREADS) {
            const float r = (cabs(sv[m]) - yv[m]) / ys[m];
            s_amp += r * r;
        }
    }
    if (a.y.cphase && a.y.sigma_cphase) {
        const float* yv = a.y.cphase + yb * Kcp;
        const float* ys = a.y.sigma_cphase + yb * Kcp;
        for (int k = tid; k < Kcp; k += CL_THREADS) {
            const float2 i1 = cdir(sv[a.tab.cp_index[3 * k + 0]], a.tab.cp_conj[3 * k + 0]);
            const float2 i2 = cdir(sv[a.tab.cp_index[3 * k + 1]], a.tab.cp_conj[3 * k + 1]);
            const float2 i3 = cdir(sv[a.tab.cp_index[3 * k + 2]], a.tab.cp_conj[3 * k + 2]);
            const float2 p = cmul(cmul(i1, i2), i3);
            s_cp += one_minus_cos(yv[k] - atan2f(p.y, p.x), a.w.cphase_form) / (ys[k] * ys[k]);
        }
    }
    if (a.y.logcamp && a.y.sigma_logcamp) {
        const float* yv = a.y.logcamp + yb * Kca;
        const float* ys = a.y.sigma_logcamp + yb * Kca;
        for (int k = tid; k < Kca; k += CL_THREADS) {
            const float l = logf(cabs(sv[a.tab.camp_index[4 * k + 0]])) +
                            logf(cabs(sv[a.tab.camp_index[4 * k + 1]])) -
                            logf(cabs(sv[a.tab.camp_index[4 * k + 2]])) -
                            logf(cabs(sv[a.tab.camp_index[4 * k + 3]]));
            const float r = (l - yv[k]) / ys[k];
            s_ca += r * r;
        }
    }
    const float c_vis = block_sum(s_vis, red) / M * 0.5f;
    const float c_amp = block_sum(s_amp, red) / M;
    const float c_cp = Kcp ? 2.0f * block_sum(s_cp, red) / Kcp : 0.f;
    const float c_ca = Kca ? block_sum(s_ca, red) / Kca : 0.f;
    if (tid == 0) {
        chi2[b * 4 + 0] = c_vis; chi2[b * 4 + 1] = c_amp; chi2[b * 4 + 2] = c_cp; chi2[b * 4 + 3] = c_ca;
    }
}

extern "C" int bh_closure_chisq(const float* vis_c64, const float* vis_scale, int B, int F,
                                const bh_tables_t* tab, const bh_meas_t* y, int cphase_form,
                                float* chi2, void* stream) {
    STEP_REQUIRE(vis_c64 && y && chi2 && B > 0 && F > 0, "bh_closure_chisq: null pointer / sizes");
    if (int rc = check_tables("bh_closure_chisq", tab)) return rc;
    STEP_REQUIRE(tab->n_vis == tab->T * F, "bh_closure_chisq: n_vis != T*F");
    STEP_REQUIRE(y->batch == 1 || y->batch == B, "bh_closure_chisq: measurement batch %d vs B=%d", y->batch, B);
    ClosureArgs a{*tab, *y, bh_weights_t{0, 0, 0, 0, 0, 0, cphase_form}};
    const size_t lds = (2 * (size_t)tab->n_vis + 16) * sizeof(float);
    step_allow_lds((const void*)k_closure_chisq, lds);
    hipLaunchKernelGGL(k_closure_chisq, dim3(B), dim3(CL_THREADS), lds, (hipStream_t)stream,
                       (const float2*)vis_c64, vis_scale, F, a, chi2);
    STEP_CHECK_LAUNCH("bh_closure_chisq");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- observation
__global__ __launch_bounds__(CL_THREADS) void k_closure_observe(
    const float2* __restrict__ vis, const float* __restrict__ sigma, bh_tables_t tab,
    float* __restrict__ amp, float* __restrict__ sigma_amp, float* __restrict__ cphase,
    float* __restrict__ sigma_cp, float* __restrict__ logcamp, float* __restrict__ sigma_ca) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int M = tab.n_vis, Kcp = tab.n_tri, Kca = tab.n_quad;
    float2* sv = (float2*)lds;
    float* ss = lds + 2 * (size_t)M;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int m = tid; m < M; m += CL_THREADS) {
        const float2 v = vis[(size_t)b * M + m];
        sv[m] = v;
        ss[m] = sigma[m];
        amp[(size_t)b * M + m] = cabs(v);
        sigma_amp[(size_t)b * M + m] = sigma[m] + 0.f * cabs(v);   // blackhole.py:338 (NaN/inf propagate)
    }
    __syncthreads();
    for (int k = tid; k < Kcp; k += CL_THREADS) {
        float s2 = 0.f;
        float2 p = make_float2(1.f, 0.f);
        float2 leg[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int m = tab.cp_index[3 * k + j];
            leg[j] = cdir(sv[m], tab.cp_conj[3 * k + j]);
            const float am = cabs(leg[j]);
            s2 += ss[m] * ss[m] / (am * am);
        }
        p = cmul(cmul(leg[0], leg[1]), leg[2]);
        cphase[(size_t)b * Kcp + k] = atan2f(p.y, p.x);
        sigma_cp[(size_t)b * Kcp + k] = sqrtf(s2);
    }
    for (int k = tid; k < Kca; k += CL_THREADS) {
        float s2 = 0.f, l = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = tab.camp_index[4 * k + j];
            const float am = cabs(sv[m]);
            s2 += ss[m] * ss[m] / (am * am);
            l += (j < 2) ? logf(am) : -logf(am);
        }
        logcamp[(size_t)b * Kca + k] = l;
        sigma_ca[(size_t)b * Kca + k] = sqrtf(s2);
    }
}

extern "C" int bh_closure_observe(const float* vis_c64, const float* sigma, int B,
                                  const bh_tables_t* tab, float* amp, float* sigma_amp,
                                  float* cphase, float* sigma_cphase, float* logcamp,
                                  float* sigma_logcamp, void* stream) {
    STEP_REQUIRE(vis_c64 && sigma && amp && sigma_amp && B > 0, "bh_closure_observe: null pointer / B");
    if (int rc = check_tables("bh_closure_observe", tab)) return rc;
    STEP_REQUIRE(tab->n_tri == 0 || (cphase && sigma_cphase), "bh_closure_observe: null cphase outputs");
    STEP_REQUIRE(tab->n_quad == 0 || (logcamp && sigma_logcamp), "bh_closure_observe: null logcamp outputs");
    const size_t lds = (3 * (size_t)tab->n_vis + 16) * sizeof(float);
    step_allow_lds((const void*)k_closure_observe, lds);
    hipLaunchKernelGGL(k_closure_observe, dim3(B), dim3(CL_THREADS), lds, (hipStream_t)stream,
                       (const float2*)vis_c64, sigma, *tab, amp, sigma_amp, cphase, sigma_cphase,
                       logcamp, sigma_logcamp);
    STEP_CHECK_LAUNCH("bh_closure_observe");
    return STEP_OK;
}
