// Dynamic-MRI Langevin step with wave roles: the column passes of mri_lg2.h split over the 16
// waves of a 1024-thread workgroup instead of running back to back in every thread
// (forward_operator/dynamic_mri.py:42-48 with the loss of base.py:40-41 and its adjoint, SURVEY
// Appendix B "MRI").
//   waves 0-5   pass A of the column transforms: 12 tasks per column (2 per wave) x 32 columns;
//               forward inputs come straight from global memory, two register sets alternate so
//               that a strip's loads are in flight for one and a half strips;
//   waves 6-13  pass B: 16 tasks per column (2 per wave) x 32 columns; forward: kept k_y rows into
//               the row buffer; inverse: image rows to global memory;
//   waves 14-15 idle during the column passes (they work in the row phase and the sums).
// Pass A of strip s+1 and pass B of strip s run at the same time on different waves: a strip
// costs max(A, B) instead of A + B, and four waves per SIMD (<= 128 registers) hide the LDS and
// barrier latencies that two could not.  Two barriers per strip, as in mri_lg2.h; the row phase
// is lg2_row_phase on 1024 threads (every row task in one round).
//
// EXPERIMENT, not built into libstep_amd.so (round 1): parity green on the host emulation and on
// the GPU; 0.097 ms at B = 32 like mri_lg2.h, 0.598 ms at B = 256 against 0.568 ms.  The forward
// column phase is bound by the x loads in flight (63 outstanding loads per wave, six loading
// waves), not by the passes that the roles overlap: without its loads it takes 12 us, with
// them 22 us in both kernels.
#pragma once
#include "../../step_amd/csrc/mri_lg2.h"

#define LGW_THREADS 1024

FFT_HD size_t lgw_lds_bytes(int nkc) {
    return (size_t)(LG2_R + LG2_R * LG2_SW + (size_t)nkc * LG2_RS) * sizeof(cf32) +
           2 * LG2_R * sizeof(int) + LGW_THREADS * sizeof(float);
}

// Arguments as lg2_body.
template <bool YQ, class Ctx>
LG2_DEV void lgw_body(const Ctx& ctx, const float* __restrict__ x, const float* __restrict__ y,
                      int y_batch, const int* __restrict__ rows, const int* __restrict__ nrows,
                      int mask_frames, int T, int nframes, int nkc,
                      const float* __restrict__ chain_scale, float* __restrict__ loss_part,
                      float* __restrict__ unk_part, float* __restrict__ grad) {
    constexpr int R = LG2_R, SW = LG2_SW, NS = LG2_NS, RS = LG2_RS, Q = LG2_R / 4, NT = LGW_THREADS;
    const Lg2Lds L = lg2_carve(ctx.lds(), nkc);
    const int tid = ctx.tid();
    const size_t fsz = (size_t)2 * R * R;
    const int frame = ctx.block();
    if (frame >= nframes) return;
    const int b = frame / T, t = frame - b * T;
    const int mf = mask_frames == 1 ? 0 : t;
    const float* xr = x + (size_t)frame * fsz;
    const float* xi = xr + R * R;
    const float* yr = y + ((size_t)(y_batch > 1 ? b : 0) * T + t) * fsz;
    const float* yi = yr + R * R;
    const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);

    // roles (uniform per wave): half-wave index hw = tid >> 5 picks the task, tid & 31 the column
    const int col = tid & 31, hw = tid >> 5;
    const bool roleA = hw < 12, roleB = hw >= 12 && hw < 28;
    const int bb = roleA ? hw : 0;                   // pass A task: rows 12 a + bb
    const int g = roleB ? hw - 12 : 0;               // pass B task: rows g + 16 d
    const unsigned offA = (unsigned)(bb * R + col) * 4u;
    const unsigned offB = (unsigned)(g * R + col) * 4u;
    const float sgB = lg2_sgn(g);

    // pass A register sets (strips s, s + 1).  A thread has one role, so pass B keeps its 12
    // values in the first set: the roles share registers instead of adding up.
    cf32 a0[16], a1[16];
    cf32 (&u)[16] = a0;
    if (roleA) {
        lg2_load_strip(xr, xi, offA, a0);
        lg2_load_strip(xr, xi, offA + SW * 4u, a1);
    }

    // tables of the mask frame
    const int nk = nrows[mf];
    for (int e = tid; e < R; e += NT) {
        L.tw[e] = ctx.cos_sin_2pi_over_192(e);
        L.rowpos[e] = -1;
    }
    ctx.sync();
    for (int e = tid; e < nk; e += NT) {
        const int rr = rows[(size_t)mf * R + e];
        L.rowpos[rr] = e;
        L.rowl[e] = rr;
    }
    ctx.sync();
#define LGW_RP(tab, i) ((int)(signed char)((tab)[(i) >> 2] >> (8 * ((i) & 3))))
    unsigned rpA[4], rpB[3];                         // row-buffer positions, one byte each (0xff: none)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rpA[i] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            rpA[i] |= (unsigned)(L.rowpos[12 * (4 * i + k) + bb] & 0xff) << (8 * k);
        LG2_OPAQUE(rpA[i]);                          // keep the four positions packed
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rpB[i] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            rpB[i] |= (unsigned)(L.rowpos[g + 16 * (4 * i + k)] & 0xff) << (8 * k);
        LG2_OPAQUE(rpB[i]);
    }

    // the x-independent part of the loss, |y|^2 of the masked-out rows: all of them for a
    // measurement per chain (into acc), this workgroup's share for a batch-1 measurement (acc_u)
    float acc = 0.f, acc_u = 0.f;
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;
    const bool has_share = !YQ && b < nparts;        // uniform over the workgroup
    if (YQ || has_share) {
        const int r0 = YQ ? 0 : b * R / nparts, r1 = YQ ? R : (b + 1) * R / nparts;
        const int total = (r1 - r0) * 2 * Q;
        float su = 0.f;
#pragma unroll 1
        for (int e0 = 0; e0 < total; e0 += 4 * NT) {
            float4 w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + tid + i * NT;
                w[i].x = w[i].y = w[i].z = w[i].w = 0.f;
                if (e < total) {
                    const int row = r0 + e / (2 * Q), rq = e % (2 * Q), pl = rq >= Q;
                    if (L.rowpos[row] < 0)
                        w[i] = *(const float4*)(yr + (size_t)pl * R * R + row * R + 4 * (rq - pl * Q));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) su += w[i].x * w[i].x + w[i].y * w[i].y + w[i].z * w[i].z + w[i].w * w[i].w;
        }
        if (YQ) acc = su; else acc_u = su;
    }
    ctx.tick(0, frame);

    // ---------------------------------------------------------------- forward column phase
    // one stage: pass A of strip s from register set `cur`, then `cur` is refilled with strip s + 2
#define LGW_FWD_STAGE(s, cur)                                                          \
    {                                                                                  \
        if (roleA && (s) < NS) {                                                       \
            fft16<-1>(cur);                                                            \
            _Pragma("unroll") for (int c = 1; c < 16; ++c) {                           \
                const cf32 w = L.tw[bb * c];                                           \
                cur[c] = c_rot<-1>(cur[c], w.x, w.y);                                  \
            }                                                                          \
        }                                                                              \
        if (roleB && (s) > 0) {      /* pass B of strip s - 1: kept rows -> row buffer */ \
            fft12<-1>(reinterpret_cast<cf32 (&)[12]>(u));                              \
            const int n = lg2_phys(((s) - 1) * SW + col);                              \
            _Pragma("unroll") for (int d = 0; d < 12; ++d) {                           \
                const int j = LGW_RP(rpB, d);                                          \
                if (j >= 0) L.Rk[j * RS + n] = c_make(sgB * u[d].x, sgB * u[d].y);     \
            }                                                                          \
        }                                                                              \
        ctx.sync();                                                                    \
        if (roleA && (s) < NS) {                                                       \
            _Pragma("unroll") for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = cur[c]; \
            if ((s) + 2 < NS) lg2_load_strip(xr, xi, offA + (unsigned)((s) + 2) * (SW * 4u), cur); \
        }                                                                              \
        ctx.sync();                                                                    \
        if (roleB && (s) < NS) {                                                       \
            _Pragma("unroll") for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col]; \
        }                                                                              \
    }
#pragma unroll 1
    for (int s = 0; s < NS; s += 2) {
        LGW_FWD_STAGE(s, a0)
        LGW_FWD_STAGE(s + 1, a1)
    }
    LGW_FWD_STAGE(NS, a0)                            // drains pass B of the last strip
#undef LGW_FWD_STAGE
    ctx.tick(1, frame);

    // ---------------------------------------------------------------- row phase
    lg2_row_phase<NT>(ctx, L.Rk, L.tw, L.rowl, rows + (size_t)mf * R, nk, yr, yi, cs2, grad != nullptr,
                      frame, acc);
    if (grad) {
        ctx.sync();
        ctx.tick(5, frame);
        // ------------------------------------------------------------ inverse column phase
        float* dr = grad + (size_t)frame * fsz;
        float* di = dr + R * R;
#define LGW_INV_A(strip)                                                               \
        if (roleA) {                                                                   \
            const int n_ = lg2_phys((strip) * SW + col);                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) {                           \
                const int j_ = LGW_RP(rpA, i);                                         \
                a0[i] = j_ >= 0 ? L.Rk[j_ * RS + n_] : c_make(0.f, 0.f);               \
            }                                                                          \
            fft16<1>(a0);                                                              \
            _Pragma("unroll") for (int c = 1; c < 16; ++c) {                           \
                const cf32 w_ = L.tw[bb * c];                                          \
                a0[c] = c_rot<1>(a0[c], w_.x, w_.y);                                   \
            }                                                                          \
        }
        LGW_INV_A(0)
        if (roleA) {
#pragma unroll
            for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = a0[c];
        }
        ctx.sync();
        if (roleB) {
#pragma unroll
            for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
        }
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) { LGW_INV_A(s + 1) }
            if (roleB) {
                fft12<1>(reinterpret_cast<cf32 (&)[12]>(u));
                const unsigned ob = offB + (unsigned)s * (SW * 4u);
#pragma unroll
                for (int d = 0; d < 12; ++d) {
                    const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                    lg2_st(dr, o, u[d].x);
                    lg2_st(di, o, u[d].y);
                }
            }
            if (s + 1 < NS) {
                ctx.sync();
                if (roleA) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = a0[c];
                }
                ctx.sync();
                if (roleB) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
                }
            }
        }
#undef LGW_INV_A
    }
#undef LGW_RP
    ctx.tick(6, frame);
    ctx.sync();
    const float tot = lg2_block_sum<NT>(ctx, acc, L.red);
    if (tid == 0) loss_part[frame] = tot;
    if (has_share) {
        ctx.sync();
        const float tot_u = lg2_block_sum<NT>(ctx, acc_u, L.red);
        if (tid == 0) unk_part[frame] = tot_u;
    }
    ctx.tick(7, frame);
}
