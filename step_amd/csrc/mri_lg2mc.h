// Multi-coil dynamic-MRI Langevin step in ONE kernel on the pipelined transforms of mri_lg2.h:
//   y_c = M FFT2(fftshift(S_c x)),  loss = sum_c |y_c - meas_c|^2,  grad = sum_c conj(S_c) A^H(2 r_c).
// NO REFERENCE COUNTERPART (the reference operator is single-coil, dynamic_mri.py:42-48; BASELINE.json
// names the coil-sensitivity multiply).
//
// Round 2 ran this as three kernels -- z_c = S_c x, the single-coil step on B x (T C) frames,
// sum_c conj(S_c) g_c -- whose two [B][T][C][2][R][R] buffers crossed HBM twice each: 3.3 x the
// algorithmic traffic, 0.80 of 1.74 ms in the two streaming kernels.  Here a persistent workgroup
// owns a frame for all coils:
//   coil loop:   forward column phase with the multiply folded into the strip loads (x and S_c
//                arrive in rotating register sets, z = S_c x is formed in registers: z never
//                exists in memory), row phase against meas_c in LDS, then the coil's processed
//                kept rows (nk x 192 complex, 80 KB) go to the workgroup's own scratch slot;
//   strip loop:  strip-OUTER, coil-INNER inverse column phase: a thread's twelve output pixels
//                accumulate conj(S_c) * (inverse column transform of coil c's kept rows) over the
//                coils in registers and are stored ONCE -- no per-coil gradient in memory; the
//                pass-A inputs come straight from the scratch slot, one (strip, coil) ahead.
// Per frame and coil the memory system sees x again (cache-resident after the first coil), S_c
// twice (static, shared by every frame), the measurement's kept rows, and 2 x 80 KB of scratch.
// Same transforms as mri_lg2.h on the same values: C = 1 with S = 1 + 0i gives the single-coil
// kernel's gradient bit for bit (1 * x, 0 + u are exact).
#pragma once
#include "mri_lg2.h"

// x, grad: [nframes][2][R][R]; sens: [C][2][R][R]; y: [y_batch * T][C][2][R][R];
// zs_all: [gridDim.x][C][nkc][R] complex (this workgroup's slot: any contents);
// loss_part[frame]: kept rows (all coils), plus the masked-out rows when y_batch > 1;
// unk_part[frame] (y_batch == 1, b < min(B, 16)): the b-th share of the masked-out rows of the
// measurement frame, all coils (every chain's loss gets every share: the finish kernel adds them).
template <class Ctx>
__device__ __forceinline__ void lg2mc_body(const Ctx& ctx, const float* __restrict__ x,
                                           const float* __restrict__ sens, int C,
                                           const float* __restrict__ y, int y_batch,
                                           const int* __restrict__ rows, const int* __restrict__ nrows,
                                           int mask_frames, int T, int nframes, int nkc,
                                           const float* __restrict__ chain_scale,
                                           float* __restrict__ loss_part, float* __restrict__ unk_part,
                                           float* __restrict__ grad, cf32* zs_all) {
    constexpr int R = LG2_R, SW = LG2_SW, NS = LG2_NS, RS = LG2_RS, Q = LG2_R / 4;
    const Lg2Lds L = lg2_carve(ctx.lds(), nkc);
    const int tid = ctx.tid();
    const size_t fsz = (size_t)2 * R * R;
#define LG2_IDS(t_)                                                                    \
    int t_ = tid;                                                                      \
    LG2_OPAQUE(t_);                                                                    \
    const int col = t_ & 31, g = t_ >> 5;                                              \
    const bool a_task = g < 12;                                                        \
    const int bb = g < 12 ? g : 0;                                                     \
    const unsigned offA = (unsigned)(bb * R + col) * 4u;                               \
    const unsigned offB = (unsigned)(g * R + col) * 4u;                                \
    (void)offA; (void)offB; (void)a_task;
#define LG2_RP(tab, i) ((int)(signed char)((tab)[(i) >> 2] >> (8 * ((i) & 3))))

    int frame = ctx.block();
    if (frame >= nframes) return;
    const int nblk = ctx.nblocks();
    cf32* zs = zs_all + (size_t)ctx.block() * C * nkc * R;
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;

    // register sets of the forward column phase: ZA / ZB alternate between "z of the strip being
    // transformed" and "x of the next strip, in flight"; SN holds the next strip's sensitivities
    cf32 za[16], zb[16], sn[16];
    int my_row = rows[(size_t)(mask_frames == 1 ? 0 : frame % T) * R + (tid < R ? tid : 0)];
    {
        LG2_IDS(t0)
        const float* xf = x + (size_t)frame * fsz;
        lg2_load_strip(xf, xf + R * R, offA, za);
        lg2_load_strip(sens, sens + R * R, offA, sn);
    }
    int nk = 0, mf_built = -1;
    unsigned rpB[3], rpA[4];

#pragma unroll 1
    for (;;) {
        const int b = frame / T, t = frame - b * T;
        const int mf = mask_frames == 1 ? 0 : t;
        if (mf != mf_built) {
            if (mf_built >= 0) my_row = rows[(size_t)mf * R + (tid < R ? tid : 0)];
            nk = nrows[mf];
            lg2_tables(ctx, L, my_row, nk);
            LG2_IDS(t1)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rpA[i] = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    rpA[i] |= (unsigned)((a_task ? L.rowpos[12 * (4 * i + k) + bb] : -1) & 0xff) << (8 * k);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rpB[i] = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    rpB[i] |= (unsigned)(L.rowpos[g + 16 * (4 * i + k)] & 0xff) << (8 * k);
            }
            mf_built = mf;
        }
        const float* xr = x + (size_t)frame * fsz;
        const float* xi = xr + R * R;
        const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
        float acc = 0.f, acc_u = 0.f;
        const bool share = y_batch > 1 || b < nparts;        // uniform over the workgroup
        const int fnext = frame + nblk;
        const bool has_next = fnext < nframes;

#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const float* sr = sens + (size_t)c * fsz;
            const float* si = sr + R * R;
            const float* yr = y + (((size_t)(y_batch > 1 ? b : 0) * T + t) * C + c) * fsz;
            const float* yi = yr + R * R;
            // ------------------------------------------------------------ forward column phase
            {
                LG2_IDS(t2)
                const float sgB = lg2_sgn(g);
                cf32 u[12];
                // za = x of strip 0, sn = S_c of strip 0 (requested under the previous row phase)
#pragma unroll
                for (int a = 0; a < 16; ++a) za[a] = c_mul(za[a], sn[a]);
                {
                    unsigned o = offA + SW * 4u;
                    LG2_OPAQUE(o);
                    lg2_load_strip(xr, xi, o, zb);
                    lg2_load_strip(sr, si, o, sn);
                }
                // one stage: pass B of strip s - 1 (from u), pass A of strip s (set Z), the A -> B
                // exchange; then the next strip's x (set XN) is multiplied by its sensitivities (SN)
                // and the strip after that is requested into Z and SN
#define LG2MC_STAGE(s, Z, XN)                                                          \
                {                                                                      \
                    if ((s) > 0) {                                                     \
                        fft192_pass_b<-1>(u);                                          \
                        const int n = lg2_phys(((s) - 1) * SW + col);                  \
                        unsigned rq[3] = {rpB[0], rpB[1], rpB[2]};                     \
                        LG2_OPAQUE(rq[0]); LG2_OPAQUE(rq[1]); LG2_OPAQUE(rq[2]);       \
                        _Pragma("unroll") for (int d = 0; d < 12; ++d) {               \
                            const int j = LG2_RP(rq, d);                               \
                            if (j >= 0) L.Rk[j * RS + n] = c_make(sgB * u[d].x, sgB * u[d].y); \
                        }                                                              \
                    }                                                                  \
                    LG2_FENCE();                                                       \
                    if (a_task) {                                                      \
                        fft16<-1>(Z);                                                  \
                        _Pragma("unroll") for (int cc = 1; cc < 16; ++cc) {            \
                            const cf32 w = L.tw[bb * cc];                              \
                            Z[cc] = c_rot<-1>(Z[cc], w.x, w.y);                        \
                        }                                                              \
                    }                                                                  \
                    ctx.sync();                                                        \
                    if (a_task) {                                                      \
                        _Pragma("unroll") for (int cc = 0; cc < 16; ++cc) L.S[(12 * cc + bb) * SW + col] = Z[cc]; \
                    }                                                                  \
                    if ((s) + 1 < NS) {                                                \
                        _Pragma("unroll") for (int a = 0; a < 16; ++a) XN[a] = c_mul(XN[a], sn[a]); \
                    }                                                                  \
                    if ((s) + 2 < NS) {                                                \
                        unsigned o = offA + (unsigned)((s) + 2) * (SW * 4u);           \
                        LG2_OPAQUE(o);                                                 \
                        lg2_load_strip(xr, xi, o, Z);                                  \
                        lg2_load_strip(sr, si, o, sn);                                 \
                    }                                                                  \
                    ctx.sync();                                                        \
                    _Pragma("unroll") for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col]; \
                }
                static_assert(NS == 6, "stages are written out");
                LG2MC_STAGE(0, za, zb) LG2MC_STAGE(1, zb, za) LG2MC_STAGE(2, za, zb)
                LG2MC_STAGE(3, zb, za) LG2MC_STAGE(4, za, zb) LG2MC_STAGE(5, zb, za)
#undef LG2MC_STAGE
                {
                    fft192_pass_b<-1>(u);
                    const int n = lg2_phys((NS - 1) * SW + col);
#pragma unroll
                    for (int d = 0; d < 12; ++d) {
                        const int j = LG2_RP(rpB, d);
                        if (j >= 0) L.Rk[j * RS + n] = c_make(sgB * u[d].x, sgB * u[d].y);
                    }
                }
            }
            // the masked-out rows of this coil's measurement (A(x) = 0 there: the residual is -y):
            // the chain's own rows with a measurement per chain, a share of them with a shared one
            if (share) {
                const int r0 = y_batch > 1 ? 0 : b * R / nparts, r1 = y_batch > 1 ? R : (b + 1) * R / nparts;
                const int ne = (r1 - r0) * 2 * Q;
                float su = 0.f;
#pragma unroll 1
                for (int e0 = tid; e0 < ne; e0 += 4 * LG2_THREADS) {
                    float4 v[4];
                    bool m[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = e0 + i * LG2_THREADS, ec = e < ne ? e : 0;
                        const int row = r0 + ec / (2 * Q), r = ec % (2 * Q), pl = r >= Q;
                        m[i] = e < ne && L.rowpos[row] < 0;
                        v[i] = *(const float4*)(yr + (size_t)pl * R * R + row * R + 4 * (r - pl * Q));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float qv = v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
                        su += m[i] ? qv : 0.f;
                    }
                }
                if (y_batch > 1) acc += su; else acc_u += su;
            }
            // the next coil's (or the next frame's first coil's) strip 0 travels under the row phase
            if (c + 1 < C || has_next) {
                LG2_IDS(tn)
                const float* xf = c + 1 < C ? xr : x + (size_t)fnext * fsz;
                const float* sf = c + 1 < C ? sr + fsz : sens;
                lg2_load_strip(xf, xf + R * R, offA, za);
                lg2_load_strip(sf, sf + R * R, offA, sn);
            }
            lg2_row_phase<LG2_THREADS>(ctx, L.Rk, L.tw, L.rowl, rows + (size_t)mf * R, nk, yr, yi, cs2,
                                       grad != nullptr, frame, acc);
            ctx.sync();
            if (grad) {
                // the coil's processed kept rows -> this workgroup's scratch slot
                cf32* zc = zs + (size_t)c * nkc * R;
                for (int e = tid; e < nk * R; e += LG2_THREADS) {
                    const int j = e / R, n = e - j * R;
                    zc[(size_t)j * R + n] = L.Rk[j * RS + lg2_phys(n)];
                }
                ctx.sync();                              // the row buffer is free for the next coil
            }
        }
        if (grad) {
            // ------------------------------------------------------------ inverse column phase
            // iteration it = strip * C + coil; av = the pass-A inputs of the iteration, read straight
            // from the scratch slot one iteration ahead; sv = S_c at this thread's 12 output pixels
            float* dr = grad + (size_t)frame * fsz;
            float* di = dr + R * R;
            LG2_IDS(t3)
            cf32 av[16], u[12], sv[12], ga[12];
            const int nit = NS * C;
#define LG2MC_LOAD_A(s_, c_)                                                           \
            if (a_task) {                                                              \
                const cf32* zc_ = zs + (size_t)(c_) * nkc * R + (s_) * SW + col;       \
                unsigned rq_[4] = {rpA[0], rpA[1], rpA[2], rpA[3]};                    \
                LG2_OPAQUE(rq_[0]); LG2_OPAQUE(rq_[1]); LG2_OPAQUE(rq_[2]); LG2_OPAQUE(rq_[3]); \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) {                       \
                    const int j_ = LG2_RP(rq_, i);                                     \
                    av[i] = j_ >= 0 ? zc_[j_ * R] : c_make(0.f, 0.f);                  \
                }                                                                      \
            }
            LG2MC_LOAD_A(0, 0)
#pragma unroll
            for (int d = 0; d < 12; ++d) ga[d] = c_make(0.f, 0.f);
            int s = 0, c = 0;
#pragma unroll 1
            for (int it = 0; it < nit; ++it) {
                const float* sr = sens + (size_t)c * fsz;
                unsigned ob = offB + (unsigned)s * (SW * 4u);
                LG2_OPAQUE(ob);
#pragma unroll
                for (int d = 0; d < 12; ++d) {
                    const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                    sv[d] = c_make(lg2_ld(sr, o), lg2_ld(sr + R * R, o));
                }
                if (a_task) {
                    fft16<1>(av);
#pragma unroll
                    for (int cc = 1; cc < 16; ++cc) {
                        const cf32 w_ = L.tw[bb * cc];
                        av[cc] = c_rot<1>(av[cc], w_.x, w_.y);
                    }
                }
                ctx.sync();                              // the previous iteration's u loads are done
                if (a_task) {
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc) L.S[(12 * cc + bb) * SW + col] = av[cc];
                }
                int s2 = s, c2 = c + 1;
                if (c2 == C) { c2 = 0; ++s2; }
                if (it + 1 < nit) { LG2MC_LOAD_A(s2, c2) }
                ctx.sync();
#pragma unroll
                for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
                LG2_FENCE();
                fft192_pass_b<1>(u);
                // conj(S_c) * u, summed over the coils in coil order
#pragma unroll
                for (int d = 0; d < 12; ++d) {
                    ga[d].x += fmaf(u[d].x, sv[d].x, u[d].y * sv[d].y);
                    ga[d].y += fmaf(u[d].y, sv[d].x, -(u[d].x * sv[d].y));
                }
                if (c + 1 == C) {
#pragma unroll
                    for (int d = 0; d < 12; ++d) {
                        const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                        lg2_st(dr, o, ga[d].x);
                        lg2_st(di, o, ga[d].y);
                        ga[d] = c_make(0.f, 0.f);
                    }
                }
                s = s2; c = c2;
            }
#undef LG2MC_LOAD_A
        }
        const float tot = lg2_block_sum(ctx, acc, L.red);
        if (tid == 0) loss_part[frame] = tot;
        if (y_batch == 1 && b < nparts) {
            ctx.sync();
            const float tot_u = lg2_block_sum(ctx, acc_u, L.red);
            if (tid == 0) unk_part[frame] = tot_u;
        }
        if (!has_next) break;
        frame = fnext;
        ctx.sync();
    }
#undef LG2_RP
#undef LG2_IDS
}
