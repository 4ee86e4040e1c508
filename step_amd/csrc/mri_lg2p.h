// Dynamic-MRI Langevin step, persistent form of mri_lg2.h for one measurement shared by all
// chains (y_batch = 1, the sampler's case): a workgroup walks the frames blk, blk + nblocks, ...
// and requests the first three strips of its NEXT frame as soon as the forward column phase of the
// current one has released the three register sets -- they travel under the row phase and the
// inverse column phase, so the next forward phase starts on data that is already there, the
// twiddle / row tables are built once per workgroup (once per frame for per-frame masks), and a
// frame's gradient stores drain under the next frame's transforms.
// Same arithmetic, same order of operations as lg2_body<false>: results are bit-identical.
#pragma once
#include "mri_lg2.h"

template <class Ctx>
LG2_DEV void lg2p_body(const Ctx& ctx, const float* __restrict__ x, const float* __restrict__ y,
                       const int* __restrict__ rows, const int* __restrict__ nrows,
                       int mask_frames, int T, int nframes, int nkc,
                       const float* __restrict__ chain_scale, float* __restrict__ loss_part,
                       float* __restrict__ unk_part, float* __restrict__ grad) {
    constexpr bool YQ = false;
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
    const unsigned offA = (unsigned)(bb * R + col) * 4u; /* (row bb, column col), strip 0 */ \
    const unsigned offB = (unsigned)(g * R + col) * 4u;  /* (row g, column col) */     \
    (void)offA; (void)offB; (void)a_task;
#define LG2_RP(tab, i) ((int)(signed char)((tab)[(i) >> 2] >> (8 * ((i) & 3))))

    int frame = ctx.block();
    if (frame >= nframes) return;
    const int nblk = ctx.nblocks();
    cf32 a0[16], a1[16], a2[16];
    int my_row = rows[(size_t)(mask_frames == 1 ? 0 : frame % T) * R + (tid < R ? tid : 0)];
    {
        LG2_IDS(t0)
        const float* xf = x + (size_t)frame * fsz;
        lg2_load_strip(xf, xf + R * R, offA, a0);
        lg2_load_strip(xf, xf + R * R, offA + SW * 4u, a1);
        lg2_load_strip(xf, xf + R * R, offA + 2 * SW * 4u, a2);
    }
    int nk = 0, mf_built = -1;
    unsigned rpB[3], rpA[4];
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;

#pragma unroll 1
    for (;;) {
        const int b = frame / T, t = frame - b * T;
        const int mf = mask_frames == 1 ? 0 : t;
        if (mf != mf_built) {                        // uniform: per-frame masks rebuild per frame
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
        const float* yr = y + (size_t)t * fsz;
        const float* yi = yr + R * R;
        const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
        float acc = 0.f, acc_u = 0.f;
        const bool has_share = b < nparts;               // uniform over the workgroup
        const int fnext = frame + nblk;
        const bool has_next = fnext < nframes;           // uniform
        // ---------------------------------------------------------------- forward column phase
        {
        LG2_IDS(t2)
        const float sgB = lg2_sgn(g);                // (-1)^ky for ky = g + 16 d
        cf32 u[12];
        constexpr int NSETS = YQ ? 2 : 3;
        float4 yq[LG2_YQ];
        if (YQ) {
#pragma unroll
            for (int i = 0; i < LG2_YQ; ++i) yq[i].x = yq[i].y = yq[i].z = yq[i].w = 0.f;
        }
        // one stage: pass B of strip s - 1 (from u), pass A of strip s (register set `cur`), the
        // A -> B exchange through the strip buffer, then `cur` is refilled with strip s + NSETS
#define LG2_FWD_STAGE(s, cur, prev)                                                          \
        {                                                                              \
            if (YQ) {                                                                  \
                /* masked-out rows 32 s .. 32 s + 31 of the measurement (both planes), fetched a */ \
                /* strip ahead of their use; kept rows stay zero here (the row phase has them) */  \
                _Pragma("unroll") for (int i = 0; i < LG2_YQ; ++i) {                   \
                    acc += yq[i].x * yq[i].x + yq[i].y * yq[i].y + yq[i].z * yq[i].z + yq[i].w * yq[i].w; \
                    const int e = tid + i * LG2_THREADS;     /* < 32 rows * 2 planes * Q */ \
                    const int row = 32 * (s) + e / (2 * Q), r = e % (2 * Q), pl = r >= Q; \
                    yq[i].x = yq[i].y = yq[i].z = yq[i].w = 0.f;                       \
                    if (L.rowpos[row] < 0)                                             \
                        yq[i] = *(const float4*)((const char*)yr +                     \
                                                 (unsigned)(pl * R * R + row * R + 4 * (r - pl * Q)) * 4u); \
                }                                                                      \
            }                                                                          \
            LG2_FENCE();                                                               \
            if ((s) > 0) {                   /* pass B of strip s-1, kept rows -> Rk */ \
                fft192_pass_b<-1>(u);                                                  \
                const int n = lg2_phys(((s) - 1) * SW + col);                          \
                unsigned rq[3] = {rpB[0], rpB[1], rpB[2]};                             \
                LG2_OPAQUE(rq[0]); LG2_OPAQUE(rq[1]); LG2_OPAQUE(rq[2]);               \
                /* strip s + 2 goes into the set that stage s - 1 has just handed to the strip buffer, */ \
                /* requested HERE, a few loads behind each of pass B's stores, not as a burst of 32 */     \
                /* (which holds every wave at the issue until the address unit has worked it off) */       \
                const bool refill = (s) + 2 < NS && !YQ;                               \
                unsigned o = offA + (unsigned)((s) + 2) * (SW * 4u);                   \
                LG2_OPAQUE(o);                                                         \
                _Pragma("unroll") for (int d = 0; d < 12; ++d) {                       \
                    const int j = LG2_RP(rq, d);                                       \
                    if (j >= 0) L.Rk[j * RS + n] = c_make(sgB * u[d].x, sgB * u[d].y); \
                    if (refill) {                                                      \
                        lg2_load_pair(xr, xi, o, d, prev);                             \
                        if (d < 4) lg2_load_pair(xr, xi, o, 12 + d, prev);             \
                        LG2_FENCE();                                                   \
                    }                                                                  \
                }                                                                      \
            }                                                                          \
            LG2_FENCE();                                                               \
            if (a_task) {                    /* pass A of strip s */                   \
                fft16<-1>(cur);                                                        \
                _Pragma("unroll") for (int c = 1; c < 16; ++c) {                       \
                    const cf32 w = L.tw[bb * c];                                       \
                    cur[c] = c_rot<-1>(cur[c], w.x, w.y);                              \
                }                                                                      \
            }                                                                          \
            ctx.sync();                                                                \
            if (a_task) {                                                              \
                _Pragma("unroll") for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = cur[c]; \
            }                                                                          \
            if ((s) + NSETS < NS && YQ && a_task) {                                    \
                unsigned o = offA + (unsigned)((s) + NSETS) * (SW * 4u);               \
                LG2_OPAQUE(o);                                                         \
                lg2_load_strip(xr, xi, o, cur);                                        \
            }                                                                          \
            ctx.sync();                                                                \
            _Pragma("unroll") for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col]; \
        }
        // Straight-line, not a loop: the compiler derives each wait count from every path that
        // reaches the instruction, and in a loop the last round -- nothing refilled, the set in
        // use holds the youngest loads -- forces "wait for everything" on all rounds, which
        // drains the sets that should stay in flight.
        static_assert(NS == 6, "stages are written out");
        if (YQ) {
#pragma unroll 1
            for (int s = 0; s < NS; s += 2) {
                LG2_FWD_STAGE(s, a0, a1)
                LG2_FWD_STAGE(s + 1, a1, a0)
            }
        } else {
            LG2_FWD_STAGE(0, a0, a2) LG2_FWD_STAGE(1, a1, a0) LG2_FWD_STAGE(2, a2, a1)
            LG2_FWD_STAGE(3, a0, a2) LG2_FWD_STAGE(4, a1, a0) LG2_FWD_STAGE(5, a2, a1)
        }
#undef LG2_FWD_STAGE
        ctx.tick(1, frame);
        if (YQ) {
#pragma unroll
            for (int i = 0; i < LG2_YQ; ++i)
                acc += yq[i].x * yq[i].x + yq[i].y * yq[i].y + yq[i].z * yq[i].z + yq[i].w * yq[i].w;
        }
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
        // this chain's share of the masked-out measurement rows: here, where no strip is in
        // flight (the wait for these loads is a wait for everything older), four loads at a
        // time and none of them under a branch (a kept row is fetched and not added)
        if (has_share) {
            const int r0 = b * R / nparts, r1 = (b + 1) * R / nparts, ne = (r1 - r0) * 2 * Q;
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
                    const float q = v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
                    acc_u += m[i] ? q : 0.f;             // a select: a NaN in a kept row stays out
                }
            }
        }
        // the three register sets are free: the next frame's first strips travel under the row
        // phase and the inverse column phase.  Requested one strip at a time from inside the row phase
        // (behind its measurement loads, between its rounds): all 96 loads at once ran every wave into
        // the 63-load limit of the counter, and the whole workgroup stood at the issue for the 5 us
        // the first of them took to come back -- before a single row had been touched.
        const float* xf = x + (size_t)(has_next ? fnext : frame) * fsz;
        auto next_strips = [&](int i) {
            if (!has_next) return;                       // uniform
            LG2_IDS(tn)
            if (i == 0) lg2_load_strip(xf, xf + R * R, offA, a0);
            else if (i == 1) lg2_load_strip(xf, xf + R * R, offA + SW * 4u, a1);
            else lg2_load_strip(xf, xf + R * R, offA + 2 * SW * 4u, a2);
        };
        lg2_row_phase<LG2_THREADS>(ctx, L.Rk, L.tw, L.rowl, rows + (size_t)mf * R, nk, yr, yi, cs2,
                                   grad != nullptr, frame, acc, next_strips);
        if (grad) {
            ctx.sync();
            ctx.tick(5, frame);

            // ------------------------------------------------------------ inverse column phase
            float* dr = grad + (size_t)frame * fsz;
            float* di = dr + R * R;
            LG2_IDS(t3)
            cf32 a[16], u[12];
#define LG2_INV_A(strip)                                                              \
            if (a_task) {                                                             \
                const int n_ = lg2_phys((strip) * SW + col);                          \
                unsigned rq_[4] = {rpA[0], rpA[1], rpA[2], rpA[3]};                   \
                LG2_OPAQUE(rq_[0]); LG2_OPAQUE(rq_[1]); LG2_OPAQUE(rq_[2]); LG2_OPAQUE(rq_[3]); \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) {                      \
                    const int j_ = LG2_RP(rq_, i);                                    \
                    a[i] = j_ >= 0 ? L.Rk[j_ * RS + n_] : c_make(0.f, 0.f);           \
                }                                                                     \
                fft16<1>(a);                                                          \
                _Pragma("unroll") for (int c = 1; c < 16; ++c) {                      \
                    const cf32 w_ = L.tw[bb * c];                                     \
                    a[c] = c_rot<1>(a[c], w_.x, w_.y);                                \
                }                                                                     \
            }
            LG2_INV_A(0)
            if (a_task) {
#pragma unroll
                for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = a[c];
            }
            ctx.sync();
#pragma unroll
            for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) { LG2_INV_A(s + 1) }
                LG2_FENCE();
                fft192_pass_b<1>(u);
                unsigned ob = offB + (unsigned)s * (SW * 4u);
                LG2_OPAQUE(ob);
#pragma unroll
                for (int d = 0; d < 12; ++d) {
                    const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                    lg2_st(dr, o, u[d].x);
                    lg2_st(di, o, u[d].y);
                }
                if (s + 1 < NS) {
                    ctx.sync();
                    if (a_task) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = a[c];
                    }
                    ctx.sync();
#pragma unroll
                    for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
                }
            }
#undef LG2_INV_A
        }
        ctx.tick(6, frame);
        const float tot = lg2_block_sum(ctx, acc, L.red);
        if (tid == 0) loss_part[frame] = tot;
        if (has_share) {
            ctx.sync();
            const float tot_u = lg2_block_sum(ctx, acc_u, L.red);
            if (tid == 0) unk_part[frame] = tot_u;
        }
        if (!has_next) break;
        frame = fnext;
        ctx.sync();                                  // red, S and the row buffer are free again
    }
#undef LG2_RP
#undef LG2_IDS
}
