// Dynamic-MRI Langevin step with the COLUMN transforms on the bf16 matrix cores.
//
// The column passes of mri_lg2.h are 192-point FFTs on the vector ALU: 5600 vector instructions
// per wave and frame, the vector ALU busy 46 % of a step that moves its bytes at 0.48 of the HBM
// peak.  But the mask keeps only nk <= 54 of the 192 k_y rows (52 at the shipped acc 6 / 24 ACS),
// so the forward column pass is a PRUNED DFT, a [nk x 192] complex matrix applied to the 192 x 192
// complex image, and the inverse column pass its adjoint: GEMM-shaped work of the size the
// black-hole kernels (bh_dft_sep3.hip) already hide under their HBM stream.  Same two ideas:
//   * exact 3 x bf16 operand split, six plane products accumulated in fp32 by
//     v_mfma_f32_32x32x16_bf16 (error <= one fp32 rounding per product);
//   * fold about y -> 192 - y: W[k][192 - y] = conj(W[k][y]), so with E = X[y] + X[192 - y],
//     O = X[y] - X[192 - y] (y = 1..95; y = 0 and y = 96 pair with themselves, weight 1/2)
//         Re Z = C Er + S Oi,   Im Z = C Ei - S Or,      C = cos, S = sin of 2 pi ky y / 192
//     half the matrix work; the sign (-1)^ky that folds the fftshift is part of the tables.
// forward:  wave w < 6 owns the 32-column strip w: 7 blocks of 16 row pairs stream through a
//           register ring straight into the B operand, the A operand (the planes of C and S, built
//           on the host: mri_build_dft_planes) comes from LDS; accumulators -> the LDS row buffer.
// rows:     lg2_row_phase of mri_lg2.h (row FFT, residual, loss, inverse row FFT), unchanged.
// inverse:  P = C^T R, Q = S^T R per strip with R (the kept rows of the strip's columns) as the
//           register B operand and the transposed planes from LDS;
//           out[y] = (Pr - Qr, Pi + Qi), out[192 - y] = (Pr + Qr, Pi - Qi); row 96 on the vector ALU.
#pragma once
#include "mri_lg2.h"

typedef __bf16 lg2m_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int lg2m_u32x4 __attribute__((ext_vector_type(4)));

#define LG2M_NKB 7                // forward k-blocks of 16 folded rows (kk = 0..111; 0..96 carry weight)
#define LG2M_FWD_UNITS (LG2M_NKB * 2 * 3 * 2)   // 16-byte units per kept row in the forward planes
#define LG2M_NK_MAX 54            // planes + row buffer fit 160 KB of LDS up to here

__device__ __forceinline__ lg2m_bf16x8 lg2m_bf(lg2m_u32x4 v) { return __builtin_bit_cast(lg2m_bf16x8, v); }
#define LG2M_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(lg2m_bf(a), lg2m_bf(b), c, 0, 0, 0)
#define LG2M_SIX(ah, am, al, bh, bm, bl, acc)                    \
    acc = LG2M_MFMA(al, bh, acc); acc = LG2M_MFMA(ah, bl, acc);  \
    acc = LG2M_MFMA(am, bm, acc); acc = LG2M_MFMA(am, bh, acc);  \
    acc = LG2M_MFMA(ah, bm, acc); acc = LG2M_MFMA(ah, bh, acc);

// v[0..7] -> three packed bf16x8 planes with v[j] == hi[j] + mid[j] + lo[j] exactly (truncation)
__device__ __forceinline__ void lg2m_split3(const float (&v)[8], lg2m_u32x4& hi, lg2m_u32x4& mid, lg2m_u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[2 * j], b = v[2 * j + 1];
        const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        hi[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
        const float ra = a - __uint_as_float(ua & 0xffff0000u);
        const float rb = b - __uint_as_float(ub & 0xffff0000u);
        const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
        mid[j] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
        const float la = ra - __uint_as_float(va & 0xffff0000u);
        const float lb = rb - __uint_as_float(vb & 0xffff0000u);
        lo[j] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
    }
}

struct Lg2mLds {
    cf32* tw;          // [192]
    lg2m_u32x4* P;     // the planes (forward: [kb][part][plane][h][nr]); aliased by the strip buffer S
    cf32* S;           // = (cf32*)P: [192][32], inverse FFT column passes
    cf32* Rk;          // [nk][208]
    int* rowpos;       // [192]
    int* rowl;         // [192]
    float* red;        // [512]
};

__host__ __device__ static inline size_t lg2m_plane_bytes(int nk) {
    const size_t fwd = (size_t)LG2M_FWD_UNITS * (nk < 1 ? 1 : nk) * 16, strip = (size_t)LG2_R * LG2_SW * sizeof(cf32);
    return fwd > strip ? fwd : strip;
}
__host__ __device__ static inline size_t lg2m_lds_bytes(int nk) {
    return LG2_R * sizeof(cf32) + lg2m_plane_bytes(nk) + (size_t)(nk < 1 ? 1 : nk) * LG2_RS * sizeof(cf32) +
           2 * LG2_R * sizeof(int) + LG2_THREADS * sizeof(float);
}
__device__ __forceinline__ Lg2mLds lg2m_carve(float* lds, int nk) {
    Lg2mLds L;
    L.tw = (cf32*)lds;
    L.P = (lg2m_u32x4*)(L.tw + LG2_R);
    L.S = (cf32*)L.P;
    L.Rk = (cf32*)((char*)L.P + lg2m_plane_bytes(nk));
    L.rowpos = (int*)(L.Rk + (size_t)(nk < 1 ? 1 : nk) * LG2_RS);
    L.rowl = L.rowpos + LG2_R;
    L.red = (float*)(L.rowl + LG2_R);
    return L;
}

// Forward column pass of strip `strip` (this wave), all 192 rows, kept rows -> Rk.
// xr, xi: the frame's planes; P: forward planes of the frame's mask in LDS, nr = rows per unit block.
__device__ __forceinline__ void lg2m_fwd_strip(const float* __restrict__ xr, const float* __restrict__ xi,
                                               const lg2m_u32x4* P, int nr, int nk, int strip, int lane,
                                               cf32* Rk) {
    constexpr int R = LG2_R, RS = LG2_RS;
    const int li = lane & 31, h = lane >> 5;
    const int xcol = strip * 32 + li;
    f32x16 zr0, zi0, zr1, zi1;
#pragma unroll
    for (int r = 0; r < 16; ++r) zr0[r] = zi0[r] = zr1[r] = zi1[r] = 0.f;
    // block kb: folded rows kk = 16 kb + 8 h + j: top = row kk, mirror = row (192 - kk) mod 192
    float xq[3][32];
#define LG2M_LOAD(dst, kb)                                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                 \
        const int kk = (kb) * 16 + 8 * h + j;                                       \
        const int mr = kk == 0 ? 0 : R - kk;                                        \
        dst[j] = xr[kk * R + xcol];                                                 \
        dst[8 + j] = xr[mr * R + xcol];                                             \
        dst[16 + j] = xi[kk * R + xcol];                                            \
        dst[24 + j] = xi[mr * R + xcol];                                            \
    }
    const int r0 = min(li, nr - 1), r1 = min(32 + li, nr - 1);       // clamped: rows >= nk are never kept
#define LG2M_COMPUTE(src, kb)                                                       \
    {                                                                               \
        float er[8], ei[8], orn[8], oi[8];                                          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                             \
            er[j] = src[j] + src[8 + j];                                            \
            orn[j] = src[8 + j] - src[j];            /* -(top - mirror) */          \
            ei[j] = src[16 + j] + src[24 + j];                                      \
            oi[j] = src[16 + j] - src[24 + j];                                      \
        }                                                                           \
        lg2m_u32x4 erh, erm, erl, eih, eim, eil;                                    \
        lg2m_split3(er, erh, erm, erl);                                             \
        lg2m_split3(ei, eih, eim, eil);                                             \
        const lg2m_u32x4* a = P + (size_t)(((kb) * 2 + 0) * 3) * 2 * nr + h * nr;   \
        {                                                                           \
            const lg2m_u32x4 ah = a[r0], am = a[2 * nr + r0], al = a[4 * nr + r0];  \
            LG2M_SIX(ah, am, al, erh, erm, erl, zr0)                                \
            LG2M_SIX(ah, am, al, eih, eim, eil, zi0)                                \
        }                                                                           \
        {                                                                           \
            const lg2m_u32x4 ah = a[r1], am = a[2 * nr + r1], al = a[4 * nr + r1];  \
            LG2M_SIX(ah, am, al, erh, erm, erl, zr1)                                \
            LG2M_SIX(ah, am, al, eih, eim, eil, zi1)                                \
        }                                                                           \
        if ((kb) < LG2M_NKB - 1) {                   /* sin(pi ky) = 0: no sine part in the last block */ \
            lg2m_u32x4 oih, oim, oil, orh, orm, orl;                                \
            lg2m_split3(oi, oih, oim, oil);                                         \
            lg2m_split3(orn, orh, orm, orl);                                        \
            const lg2m_u32x4* s = a + (size_t)6 * nr;                               \
            {                                                                       \
                const lg2m_u32x4 ah = s[r0], am = s[2 * nr + r0], al = s[4 * nr + r0]; \
                LG2M_SIX(ah, am, al, oih, oim, oil, zr0)                            \
                LG2M_SIX(ah, am, al, orh, orm, orl, zi0)                            \
            }                                                                       \
            {                                                                       \
                const lg2m_u32x4 ah = s[r1], am = s[2 * nr + r1], al = s[4 * nr + r1]; \
                LG2M_SIX(ah, am, al, oih, oim, oil, zr1)                            \
                LG2M_SIX(ah, am, al, orh, orm, orl, zi1)                            \
            }                                                                       \
        }                                                                           \
    }
    LG2M_LOAD(xq[0], 0)
    __builtin_amdgcn_sched_barrier(0);
    LG2M_LOAD(xq[1], 1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < LG2M_NKB; ++kb) {
        if (kb + 2 < LG2M_NKB) {
            LG2M_LOAD(xq[(kb + 2) % 3], kb + 2)
            __builtin_amdgcn_sched_barrier(0);
        }
        LG2M_COMPUTE(xq[kb % 3], kb)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef LG2M_LOAD
#undef LG2M_COMPUTE
    // accumulators -> row buffer: row j = 32 mt + mfma32_row(r, h), column xcol
    const int np = lg2_phys(xcol);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j0 = mfma32_row(r, h), j1 = 32 + j0;
        if (j0 < nk) Rk[j0 * RS + np] = c_make(zr0[r], zi0[r]);
        if (j1 < nk) Rk[j1 * RS + np] = c_make(zr1[r], zi1[r]);
    }
}

// One workgroup per frame (stage A of the matrix-core kernel: forward column pass on the MFMA, the
// inverse column pass still the FFT of mri_lg2.h).  planes: [mask_frames][LG2M_FWD_UNITS][64] u32x4.
template <class Ctx>
__device__ __forceinline__ void lg2m_body(const Ctx& ctx, const float* __restrict__ x, const float* __restrict__ y,
                                          const int* __restrict__ rows, const int* __restrict__ nrows,
                                          int mask_frames, int T, int nframes, int nkc,
                                          const lg2m_u32x4* __restrict__ planes,
                                          const float* __restrict__ chain_scale, float* __restrict__ loss_part,
                                          float* __restrict__ unk_part, float* __restrict__ grad) {
    constexpr int R = LG2_R, SW = LG2_SW, NS = LG2_NS, RS = LG2_RS, Q = LG2_R / 4;
    const Lg2mLds L = lg2m_carve(ctx.lds(), nkc);
    const int tid = ctx.tid(), lane = tid & 63, wave = tid >> 6;
    const size_t fsz = (size_t)2 * R * R;
#define LG2_IDS(t_)                                                                    \
    int t_ = tid;                                                                      \
    LG2_OPAQUE(t_);                                                                    \
    const int col = t_ & 31, g = t_ >> 5;                                              \
    const bool a_task = g < 12;                                                        \
    const int bb = g < 12 ? g : 0;                                                     \
    const unsigned offB = (unsigned)(g * R + col) * 4u;                                \
    (void)offB; (void)a_task;
#define LG2_RP(tab, i) ((int)(signed char)((tab)[(i) >> 2] >> (8 * ((i) & 3))))
    const int frame = ctx.block();
    if (frame >= nframes) return;
    const int b = frame / T, t = frame - b * T;
    const int mf = mask_frames == 1 ? 0 : t;
    const int my_row = rows[(size_t)mf * R + (tid < R ? tid : 0)];
    const int nk = nrows[mf];
    const int nr = nk < 1 ? 1 : nk;
    // the mask frame's forward planes -> LDS, rows compacted from 64 to nr
#ifndef LG2M_EXP_SKIP_STAGE
    {
        const lg2m_u32x4* src = planes + (size_t)mf * LG2M_FWD_UNITS * 64;
        for (int e = tid; e < LG2M_FWD_UNITS * nr; e += LG2_THREADS) {
            const int blk = e / nr, row = e - blk * nr;
            L.P[e] = src[blk * 64 + row];
        }
    }
#endif
    {
        // tw, rowpos, rowl (lg2_tables, with its two syncs: the planes are complete after them too)
        Lg2Lds L2;
        L2.tw = L.tw; L2.S = L.S; L2.Rk = L.Rk; L2.rowpos = L.rowpos; L2.rowl = L.rowl; L2.red = L.red;
        lg2_tables(ctx, L2, my_row, nk);
    }
    unsigned rpA[4];
    {
        LG2_IDS(t1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rpA[i] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rpA[i] |= (unsigned)((a_task ? L.rowpos[12 * (4 * i + k) + bb] : -1) & 0xff) << (8 * k);
        }
    }
    const float* xr = x + (size_t)frame * fsz;
    const float* xi = xr + R * R;
    const float* yr = y + (size_t)t * fsz;
    const float* yi = yr + R * R;
    const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
    float acc = 0.f, acc_u = 0.f;
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;
    const bool has_share = b < nparts;

    // ---------------------------------------------------------------- forward column pass (MFMA)
#ifndef LG2M_EXP_SKIP_FWD
    if (wave < NS) lg2m_fwd_strip(xr, xi, L.P, nr, nk, wave, lane, L.Rk);
#endif
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
    lg2_row_phase<LG2_THREADS>(ctx, L.Rk, L.tw, L.rowl, rows + (size_t)mf * R, nk, yr, yi, cs2,
                               grad != nullptr, frame, acc);
    if (grad) {
        ctx.sync();
        // ------------------------------------------------------------ inverse column phase (FFT, mri_lg2.h)
        float* dr = grad + (size_t)frame * fsz;
        float* di = dr + R * R;
        LG2_IDS(t3)
        cf32 a[16], u[12];
#define LG2_INV_A(strip)                                                              \
        if (a_task) {                                                                 \
            const int n_ = lg2_phys((strip) * SW + col);                              \
            unsigned rq_[4] = {rpA[0], rpA[1], rpA[2], rpA[3]};                       \
            LG2_OPAQUE(rq_[0]); LG2_OPAQUE(rq_[1]); LG2_OPAQUE(rq_[2]); LG2_OPAQUE(rq_[3]); \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) {                          \
                const int j_ = LG2_RP(rq_, i);                                        \
                a[i] = j_ >= 0 ? L.Rk[j_ * RS + n_] : c_make(0.f, 0.f);               \
            }                                                                         \
            fft16<1>(a);                                                              \
            _Pragma("unroll") for (int c = 1; c < 16; ++c) {                          \
                const cf32 w_ = L.tw[bb * c];                                         \
                a[c] = c_rot<1>(a[c], w_.x, w_.y);                                    \
            }                                                                         \
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
    const float tot = lg2_block_sum(ctx, acc, L.red);
    if (tid == 0) loss_part[frame] = tot;
    if (has_share) {
        ctx.sync();
        const float tot_u = lg2_block_sum(ctx, acc_u, L.red);
        if (tid == 0) unk_part[frame] = tot_u;
    }
#undef LG2_RP
#undef LG2_IDS
}
