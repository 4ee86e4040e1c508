// Dynamic-MRI Langevin step with TWO workgroups per CU (round 3).
//
// mri_lg2.h / mri_lg2p.h keep a frame's kept k_y rows in an 86 KB LDS row buffer next to the 49 KB
// strip buffer: one workgroup per CU, two waves per SIMD, and a frame's phases run in series -- the
// vector ALU is busy 46 % of the step, 41 % of the wave time is waiting.  Here the kept rows live in
// the workgroup's own slot of a global scratch (80 KB, written and read back within microseconds:
// L2 / Infinity Cache traffic, not HBM) and LDS holds either the strip buffer (column phases) or
// one GROUP of <= 32 kept rows (row phase), never both: 54-58 KB and <= 128 registers, so two
// 512-thread workgroups share a CU -- four waves per SIMD -- and one frame's barriers, waits and
// its row phase (no memory traffic) hide under the other frame's column phases.
//   forward column phase: mri_lg2.h's stages with ONE register set (the next strip is requested as soon
//              as the set is in the strip buffer; the other workgroup covers the wait); pass-B outputs -> scratch
//   row phase: per group of kept rows: scratch -> LDS, the three in-place passes of lg2_row_phase
//              (same arithmetic, one round), LDS -> scratch
//   inverse column phase: pass-A inputs straight from the scratch, one strip ahead
// Persistent: 2 x CUs workgroups walk the frames; a workgroup's scratch slot is reused per frame.
// Same transforms on the same values as mri_lg2.h: bit-identical gradient.
#pragma once
#include "mri_lg2.h"

#define LG2X_GROUP 32             // kept rows per row-phase group (16 lanes per row: one round of 512 threads)

struct Lg2xLds {
    cf32* tw;      // [192]
    cf32* S;       // [192][32] strip buffer (column phases) ...
    cf32* Rg;      // ... aliased by one group of kept rows [LG2X_GROUP][208] (row phase)
    int* rowpos;   // [192]
    int* rowl;     // [192]
    float* red;    // [512]
};
__host__ __device__ static inline size_t lg2x_lds_bytes() {
    const size_t strip = (size_t)LG2_R * LG2_SW * sizeof(cf32), grp = (size_t)LG2X_GROUP * LG2_RS * sizeof(cf32);
    return LG2_R * sizeof(cf32) + (strip > grp ? strip : grp) + 2 * LG2_R * sizeof(int) + LG2_THREADS * sizeof(float);
}
__device__ __forceinline__ Lg2xLds lg2x_carve(float* lds) {
    const size_t strip = (size_t)LG2_R * LG2_SW * sizeof(cf32), grp = (size_t)LG2X_GROUP * LG2_RS * sizeof(cf32);
    Lg2xLds L;
    L.tw = (cf32*)lds;
    L.S = L.tw + LG2_R;
    L.Rg = L.S;
    L.rowpos = (int*)((char*)L.S + (strip > grp ? strip : grp));
    L.rowl = L.rowpos + LG2_R;
    L.red = (float*)(L.rowl + LG2_R);
    return L;
}

// The row phase of mri_lg2.h (lg2_row_phase) on ONE group of ng <= 32 kept rows held in Rg[ng][208]:
// one round (row j = tid / 16), the same operations in the same order.  rl: the group's k_y rows (LDS).
// Begins with a sync (Rg complete), ends NOT synced.
__device__ __forceinline__ void lg2x_row_group(cf32* Rg, const cf32* tw, const int* rl, int ng,
                                               const float* __restrict__ yr, const float* __restrict__ yi,
                                               float cs2, bool with_grad, float& acc) {
    constexpr int R = LG2_R, RS = LG2_RS;
    float ya[12], yb[12];
    int tr = threadIdx.x;
    LG2_OPAQUE(tr);
    const int c16 = tr & 15, j = tr >> 4;
    const bool mine = j < ng;
    if (mine) {
        const unsigned o_ = (unsigned)(rl[j] * R + c16) * 4u;
#pragma unroll
        for (int d = 0; d < 12; ++d) {
            ya[d] = lg2_ld(yr, o_ + 64u * d);
            yb[d] = lg2_ld(yi, o_ + 64u * d);
        }
    }
    __syncthreads();
    if (mine && c16 < 12) {
        cf32* row = Rg + j * RS + c16;
        cf32 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = row[13 * i];
        fft192_pass_a<-1>(v, c16, tw);
#pragma unroll
        for (int i = 0; i < 16; ++i) row[13 * i] = v[i];
    }
    __syncthreads();
    if (mine) {
        cf32* row = Rg + j * RS + 13 * c16;
        cf32 v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = row[i];
        fft192_pass_b<-1>(v);
        const float sx = lg2_sgn(c16), sc = cs2 * sx * lg2_sgn(rl[j]);
#pragma unroll
        for (int d = 0; d < 12; ++d) {
            const float rr = sx * v[d].x - ya[d], ri = sx * v[d].y - yb[d];
            acc += rr * rr + ri * ri;
            v[d] = c_make(sc * rr, sc * ri);
        }
        if (with_grad) {
            fft12<1>(v);
#pragma unroll
            for (int q = 1; q < 12; ++q) {
                const cf32 w = tw[q * c16];
                v[q] = c_rot<1>(v[q], w.x, w.y);
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) row[i] = v[i];
        }
    }
    if (with_grad) {
        __syncthreads();
        if (mine && c16 < 12) {
            cf32* row = Rg + j * RS + c16;
            cf32 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = row[13 * i];
            fft16<1>(v);
#pragma unroll
            for (int i = 0; i < 16; ++i) row[13 * i] = v[i];
        }
    }
}

// x, grad: [nframes][2][R][R]; y: [T][2][R][R] (one measurement for all chains);
// zs_all: [gridDim.x][nkc][R] complex (this workgroup's slot: any contents);
// loss_part / unk_part as for lg2p_body.
__device__ __forceinline__ void lg2x_body(float* lds, const float* __restrict__ x, const float* __restrict__ y,
                                          const int* __restrict__ rows, const int* __restrict__ nrows,
                                          int mask_frames, int T, int nframes, int nkc,
                                          const float* __restrict__ chain_scale, float* __restrict__ loss_part,
                                          float* __restrict__ unk_part, float* __restrict__ grad, cf32* zs_all) {
    constexpr int R = LG2_R, SW = LG2_SW, NS = LG2_NS, RS = LG2_RS, Q = LG2_R / 4;
    const Lg2xLds L = lg2x_carve(lds);
    const int tid = threadIdx.x;
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

    int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int nblk = gridDim.x;
    cf32* zs = zs_all + (size_t)blockIdx.x * nkc * R;
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;
    int nk = 0, mf_built = -1;
    unsigned rpB[3], rpA[4];

#pragma unroll 1
    for (;;) {
        const int b = frame / T, t = frame - b * T;
        const int mf = mask_frames == 1 ? 0 : t;
        const float* xr = x + (size_t)frame * fsz;
        const float* xi = xr + R * R;
        cf32 a0[16];
        {
            LG2_IDS(t0)
            lg2_load_strip(xr, xi, offA, a0);
        }
        if (mf != mf_built) {
            const int my_row = rows[(size_t)mf * R + (tid < R ? tid : 0)];
            nk = nrows[mf];
            __syncthreads();                             // the previous frame is done with the tables
            for (int e = tid; e < R; e += LG2_THREADS) {
                float sn, cn;
                sincospif((float)e * (1.0f / 96.0f), &sn, &cn);
                L.tw[e] = c_make(cn, sn);
                L.rowpos[e] = -1;
            }
            __syncthreads();
            if (tid < nk) {
                L.rowpos[my_row] = tid;
                L.rowl[tid] = my_row;
            }
            __syncthreads();
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
        const float* yr = y + (size_t)t * fsz;
        const float* yi = yr + R * R;
        const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
        float acc = 0.f, acc_u = 0.f;
        const bool has_share = b < nparts;

        // ---------------------------------------------------------------- forward column phase
        {
            LG2_IDS(t2)
            const float sgB = lg2_sgn(g);
            cf32 u[12];
#define LG2X_PASS_B(strip)                                                             \
            {                                                                          \
                fft192_pass_b<-1>(u);                                                  \
                cf32* zc_ = zs + (strip) * SW + col;                                   \
                unsigned rq[3] = {rpB[0], rpB[1], rpB[2]};                             \
                LG2_OPAQUE(rq[0]); LG2_OPAQUE(rq[1]); LG2_OPAQUE(rq[2]);               \
                _Pragma("unroll") for (int d = 0; d < 12; ++d) {                       \
                    const int j = LG2_RP(rq, d);                                       \
                    if (j >= 0) zc_[j * R] = c_make(sgB * u[d].x, sgB * u[d].y);       \
                }                                                                      \
            }
#define LG2X_STAGE(s)                                                                  \
            {                                                                          \
                if ((s) > 0) LG2X_PASS_B((s) - 1)                                      \
                LG2_FENCE();                                                           \
                if (a_task) {                                                          \
                    fft16<-1>(a0);                                                     \
                    _Pragma("unroll") for (int c = 1; c < 16; ++c) {                   \
                        const cf32 w = L.tw[bb * c];                                   \
                        a0[c] = c_rot<-1>(a0[c], w.x, w.y);                            \
                    }                                                                  \
                }                                                                      \
                __syncthreads();                                                       \
                if (a_task) {                                                          \
                    _Pragma("unroll") for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = a0[c]; \
                }                                                                      \
                if ((s) + 1 < NS) {          /* the set is free: the next strip travels under pass B */ \
                    unsigned o = offA + (unsigned)((s) + 1) * (SW * 4u);               \
                    LG2_OPAQUE(o);                                                     \
                    lg2_load_strip(xr, xi, o, a0);                                     \
                }                                                                      \
                __syncthreads();                                                       \
                _Pragma("unroll") for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col]; \
            }
            static_assert(NS == 6, "stages are written out");
            LG2X_STAGE(0) LG2X_STAGE(1) LG2X_STAGE(2) LG2X_STAGE(3) LG2X_STAGE(4) LG2X_STAGE(5)
            LG2X_PASS_B(NS - 1)
#undef LG2X_STAGE
#undef LG2X_PASS_B
        }
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
                    const float qv = v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
                    acc_u += m[i] ? qv : 0.f;
                }
            }
        }
        // ---------------------------------------------------------------- row phase, group by group
        {
            const int ngroups = (nk + LG2X_GROUP - 1) / LG2X_GROUP;
            const int per = ngroups ? (nk + ngroups - 1) / ngroups : 0;
#pragma unroll 1
            for (int g0 = 0; g0 < nk; g0 += per) {
                const int ng = nk - g0 < per ? nk - g0 : per;
                __syncthreads();                         // the scratch rows are written / Rg is free
                for (int e = tid; e < ng * R; e += LG2_THREADS) {
                    const int jl = e / R, n = e - jl * R;
                    L.Rg[jl * RS + lg2_phys(n)] = zs[(size_t)(g0 + jl) * R + n];
                }
                lg2x_row_group(L.Rg, L.tw, L.rowl + g0, ng, yr, yi, cs2, grad != nullptr, acc);
                if (grad) {
                    __syncthreads();
                    for (int e = tid; e < ng * R; e += LG2_THREADS) {
                        const int jl = e / R, n = e - jl * R;
                        zs[(size_t)(g0 + jl) * R + n] = L.Rg[jl * RS + lg2_phys(n)];
                    }
                }
            }
        }
        if (grad) {
            __syncthreads();                             // every group's rows are in the scratch, Rg -> S
            // ------------------------------------------------------------ inverse column phase
            float* dr = grad + (size_t)frame * fsz;
            float* di = dr + R * R;
            LG2_IDS(t3)
            cf32 av[16], u[12];
#define LG2X_LOAD_A(s_)                                                                \
            if (a_task) {                                                              \
                const cf32* zc_ = zs + (s_) * SW + col;                                \
                unsigned rq_[4] = {rpA[0], rpA[1], rpA[2], rpA[3]};                    \
                LG2_OPAQUE(rq_[0]); LG2_OPAQUE(rq_[1]); LG2_OPAQUE(rq_[2]); LG2_OPAQUE(rq_[3]); \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) {                       \
                    const int j_ = LG2_RP(rq_, i);                                     \
                    av[i] = j_ >= 0 ? zc_[j_ * R] : c_make(0.f, 0.f);                  \
                }                                                                      \
            }
            LG2X_LOAD_A(0)
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
                if (a_task) {
                    fft16<1>(av);
#pragma unroll
                    for (int c = 1; c < 16; ++c) {
                        const cf32 w_ = L.tw[bb * c];
                        av[c] = c_rot<1>(av[c], w_.x, w_.y);
                    }
                }
                if (s > 0) __syncthreads();              // the previous strip's u loads are done
                if (a_task) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = av[c];
                }
                if (s + 1 < NS) { LG2X_LOAD_A(s + 1) }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
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
            }
#undef LG2X_LOAD_A
        }
        // ---------------------------------------------------------------- losses
        __syncthreads();
        {
            float a = wave_sum(acc), bsum = wave_sum(acc_u);
            const int w = tid >> 6;
            if ((tid & 63) == 0) { L.red[2 * w] = a; L.red[2 * w + 1] = bsum; }
            __syncthreads();
            if (tid == 0) {
                float ta = 0.f, tb = 0.f;
#pragma unroll
                for (int k = 0; k < LG2_THREADS / 64; ++k) { ta += L.red[2 * k]; tb += L.red[2 * k + 1]; }
                loss_part[frame] = ta;
                if (has_share) unk_part[frame] = tb;
            }
        }
        frame += nblk;
        if (frame >= nframes) break;
        __syncthreads();                                 // red, S and the scratch slot are free again
    }
#undef LG2_RP
#undef LG2_IDS
}
