// Dynamic-MRI Langevin step, third-generation body: the column transforms run entirely in
// registers.  Same operator as mri_lg2.h (forward_operator/dynamic_mri.py:42-48 with the loss of
// base.py:40-41 and its adjoint, SURVEY Appendix B "MRI"), same row phase, same row buffer.
//
// Column FFT of 192 points = 4 lanes x 48 points:
//   * a wave owns a strip of 16 columns for the whole frame; the four lanes of a quad (lane & 3 = r)
//     hold the rows n = 4 n' + r of one column, 48 complex values = 96 registers per lane, loaded
//     straight from global memory (every load instruction: 4 rows x 64 contiguous bytes);
//   * each lane runs a 48-point FFT on its registers (48 = 12 x 4, compile-time twiddles),
//     multiplies by W192^(r k') and the quad finishes with a 4-point DFT across its lanes --
//     two butterfly stages on DPP quad permutes, no LDS, no barrier;
//   * the forward transform keeps only the masked k_y rows (into the LDS row buffer), the inverse
//     gathers them back, runs the mirror image of the same network and stores image rows.
// So the strip buffer of mri_lg2.h, its 24 barriers per frame and ~330 LDS accesses per thread
// are gone, the 12 waves of a workgroup (768 threads, <= 168 registers) walk their strips
// independently, and a frame's whole image is requested from HBM at kernel start.
// Plain C++ over a context like mri_lg2.h; quad exchanges go through ctx.quad_xor<M>().
//
// EXPERIMENT, not built into libstep_amd.so (round 1): parity green on the host emulation and on
// the GPU, forward column phase 9 us per frame against 25 us for mri_lg2.h, but the frame is
// loaded, transformed and stored in three separate phases with one workgroup per CU (94 KB of
// LDS), so HBM idles during the transforms and the step is slower overall (0.125 ms vs 0.095 ms
// at B = 32).  Needs two workgroups per CU or a cross-frame pipeline.
#pragma once
#include "../../step_amd/csrc/mri_lg2.h"

#define LG3_THREADS 768
#define LG3_NP 48                 // points per lane

struct Lg3Lds {
    cf32* tw;      // [192]
    cf32* Rk;      // [nkc + 1][208]: the kept rows and one spare row that takes the masked-out ones
    int* rowpos;   // [192]
    int* rowl;     // [192]
    float* red;    // [768]
};

FFT_HD size_t lg3_lds_bytes(int nkc) {
    return (size_t)(LG2_R + (size_t)(nkc + 1) * LG2_RS) * sizeof(cf32) + 2 * LG2_R * sizeof(int) +
           LG3_THREADS * sizeof(float);
}

FFT_HD Lg3Lds lg3_carve(float* lds, int nkc) {
    Lg3Lds L;
    L.tw = (cf32*)lds;
    L.Rk = L.tw + LG2_R;
    L.rowpos = (int*)(L.Rk + (size_t)(nkc + 1) * LG2_RS);
    L.rowl = L.rowpos + LG2_R;
    L.red = (float*)(L.rowl + LG2_R);
    return L;
}

// (cos, sin)(2 pi k / 48), k = b c <= 33
FFT_HD constexpr float lg3_c48(int k) {
    constexpr float t[34] = {1.f, 0.991444861f, 0.965925826f, 0.923879533f, 0.866025404f, 0.79335334f,
        0.707106781f, 0.608761429f, 0.5f, 0.382683432f, 0.258819045f, 0.130526192f, 0.f, -0.130526192f,
        -0.258819045f, -0.382683432f, -0.5f, -0.608761429f, -0.707106781f, -0.79335334f, -0.866025404f,
        -0.923879533f, -0.965925826f, -0.991444861f, -1.f, -0.991444861f, -0.965925826f, -0.923879533f,
        -0.866025404f, -0.79335334f, -0.707106781f, -0.608761429f, -0.5f, -0.382683432f};
    return t[k];
}
FFT_HD constexpr float lg3_s48(int k) {
    constexpr float t[34] = {0.f, 0.130526192f, 0.258819045f, 0.382683432f, 0.5f, 0.608761429f,
        0.707106781f, 0.79335334f, 0.866025404f, 0.923879533f, 0.965925826f, 0.991444861f, 1.f, 0.991444861f,
        0.965925826f, 0.923879533f, 0.866025404f, 0.79335334f, 0.707106781f, 0.608761429f, 0.5f, 0.382683432f,
        0.258819045f, 0.130526192f, 0.f, -0.130526192f, -0.258819045f, -0.382683432f, -0.5f, -0.608761429f,
        -0.707106781f, -0.79335334f, -0.866025404f, -0.923879533f};
    return t[k];
}

// 48-point FFT in registers, input in natural order v[n'], n' = 4 a + b; output F[c + 12 d] at
// v[4 c + d]:  12-point over a for each b, twiddle W48^(b c), 4-point over b for each c.
template <int SIGN>
LG2_DEV void lg3_fft48_dit(cf32 (&v)[LG3_NP]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cf32 t[12];
#pragma unroll
        for (int a = 0; a < 12; ++a) t[a] = v[4 * a + b];
        fft12<SIGN>(t);
#pragma unroll
        for (int c = 0; c < 12; ++c) v[4 * c + b] = t[c];
        LG2_FENCE();
    }
#pragma unroll
    for (int c = 1; c < 12; ++c)
#pragma unroll
        for (int b = 1; b < 4; ++b) v[4 * c + b] = c_rot<SIGN>(v[4 * c + b], lg3_c48(b * c), lg3_s48(b * c));
#pragma unroll
    for (int c = 0; c < 12; ++c) fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

// The mirror image: input H[c + 12 d] at v[4 c + d]; output in natural order v[n'], n' = 4 a + b:
// 4-point over d for each c, twiddle W48^(b c), 12-point over c for each b.
template <int SIGN>
LG2_DEV void lg3_fft48_dif(cf32 (&v)[LG3_NP]) {
#pragma unroll
    for (int c = 0; c < 12; ++c) fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
#pragma unroll
    for (int c = 1; c < 12; ++c)
#pragma unroll
        for (int b = 1; b < 4; ++b) v[4 * c + b] = c_rot<SIGN>(v[4 * c + b], lg3_c48(b * c), lg3_s48(b * c));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cf32 t[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) t[c] = v[4 * c + b];
        fft12<SIGN>(t);
#pragma unroll
        for (int a = 0; a < 12; ++a) v[4 * a + b] = t[a];
        LG2_FENCE();
    }
}

// index k' = c + 12 d of the value held in v[4 c + d]
FFT_HD constexpr int lg3_kq(int idx) { return (idx >> 2) + 12 * (idx & 3); }

// One workgroup (768 threads) per frame.  Arguments as lg2_body<false>/<true>: YQ = a measurement
// per chain (every workgroup sums the masked-out rows of its own measurement frame), !YQ = one
// measurement for all chains (the first 16 chains' workgroups each sum a 16th, into unk_part).
template <bool YQ, class Ctx>
LG2_DEV void lg3_body(const Ctx& ctx, const float* __restrict__ x, const float* __restrict__ y,
                      int y_batch, const int* __restrict__ rows, const int* __restrict__ nrows,
                      int mask_frames, int T, int nframes, int nkc,
                      const float* __restrict__ chain_scale, float* __restrict__ loss_part,
                      float* __restrict__ unk_part, float* __restrict__ grad) {
    constexpr int R = LG2_R, RS = LG2_RS, Q = LG2_R / 4, NT = LG3_THREADS, NP = LG3_NP;
    const Lg3Lds L = lg3_carve(ctx.lds(), nkc);
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

    // column roles: wave = strip of 16 columns, quad = one column, lane & 3 = r
    const int r = tid & 3, colq = (tid & 63) >> 2, strip = tid >> 6;
    const int column = 16 * strip + colq;
    const unsigned off0 = (unsigned)(r * R + column) * 4u;       // (row r, column), bytes
    const int mq = (r >> 1) + 2 * (r & 1);           // this lane ends up with k = k' + 48 mq
    const float s_r0 = (r & 1) ? -1.f : 1.f, s_r1 = (r & 2) ? -1.f : 1.f;
    const bool lane3 = r == 3;

    cf32 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {                   // rows 4 i + r: the whole frame is in flight
        const unsigned o = off0 + (unsigned)i * (4u * R * 4u);
        v[i] = c_make(lg2_ld(xr, o), lg2_ld(xi, o));
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

    // the x-independent part of the loss: |y|^2 of the masked-out rows
    float acc = 0.f, acc_u = 0.f;
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;
    const bool has_share = !YQ && b < nparts;        // uniform over the workgroup
    if (YQ || has_share) {
        const int r0 = YQ ? 0 : b * R / nparts, r1 = YQ ? R : (b + 1) * R / nparts;
        float s = 0.f;
#pragma unroll 4
        for (int e = tid; e < (r1 - r0) * 2 * Q; e += NT) {
            const int row = r0 + e / (2 * Q), rq = e % (2 * Q), pl = rq >= Q;
            if (L.rowpos[row] < 0) {
                const float4 w = *(const float4*)(yr + (size_t)pl * R * R + row * R + 4 * (rq - pl * Q));
                s += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
            }
        }
        if (YQ) acc = s; else acc_u = s;
    }
    ctx.tick(0, frame);

    // ---------------------------------------------------------------- forward columns
    lg3_fft48_dit<-1>(v);
    LG2_FENCE();
#pragma unroll
    for (int i0 = 0; i0 < NP; i0 += 8) {             // W192^(r k'), k' = lg3_kq(i); 8 at a time
        cf32 w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = L.tw[r * lg3_kq(i0 + i)];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i0 + i] = c_rot<-1>(v[i0 + i], w[i].x, w[i].y);
        LG2_FENCE();
    }
    // 4-point DFT across the quad: partner r ^ 2, lane 3 times -i, partner r ^ 1;
    // lane r then holds X[k' + 48 mq]
    {
        const int n = lg2_phys(column);
#pragma unroll
        for (int i0 = 0; i0 < NP; i0 += 8) {
            int jj[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) jj[i] = L.rowpos[lg3_kq(i0 + i) + 48 * mq];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cf32 p = ctx.template quad_xor<2>(v[i0 + i]);
                cf32 w = c_make(p.x + s_r1 * v[i0 + i].x, p.y + s_r1 * v[i0 + i].y);
                if (lane3) w = c_make(w.y, -w.x);
                p = ctx.template quad_xor<1>(w);
                w = c_make(p.x + s_r0 * w.x, p.y + s_r0 * w.y);
                const float sg = lg2_sgn(lg3_kq(i0 + i) + 48 * mq);
                const int j = jj[i] >= 0 ? jj[i] : nkc;  // masked-out rows land in the spare row
                L.Rk[j * RS + n] = c_make(sg * w.x, sg * w.y);
            }
            LG2_FENCE();
        }
    }
    ctx.tick(1, frame);

    // ---------------------------------------------------------------- row phase
    lg2_row_phase<NT>(ctx, L.Rk, L.tw, L.rowl, rows + (size_t)mf * R, nk, yr, yi, cs2, grad != nullptr,
                      frame, acc);
    if (grad) {
        ctx.sync();
        ctx.tick(5, frame);
        // ------------------------------------------------------------ inverse columns
        {
            const int n = lg2_phys(column);
#pragma unroll
            for (int i0 = 0; i0 < NP; i0 += 8) {
                int jj[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) jj[i] = L.rowpos[lg3_kq(i0 + i) + 48 * mq];
                cf32 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = jj[i] >= 0 ? L.Rk[jj[i] * RS + n] : c_make(0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    cf32 p = ctx.template quad_xor<1>(w[i]);
                    cf32 q = c_make(p.x + s_r0 * w[i].x, p.y + s_r0 * w[i].y);
                    if (lane3) q = c_make(-q.y, q.x);
                    p = ctx.template quad_xor<2>(q);
                    v[i0 + i] = c_make(p.x + s_r1 * q.x, p.y + s_r1 * q.y);
                }
                LG2_FENCE();
            }
        }
        LG2_FENCE();
#pragma unroll
        for (int i0 = 0; i0 < NP; i0 += 8) {
            cf32 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = L.tw[r * lg3_kq(i0 + i)];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i0 + i] = c_rot<1>(v[i0 + i], w[i].x, w[i].y);
            LG2_FENCE();
        }
        lg3_fft48_dif<1>(v);
        float* dr = grad + (size_t)frame * fsz;
        float* di = dr + R * R;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned o = off0 + (unsigned)i * (4u * R * 4u);
            lg2_st(dr, o, v[i].x);
            lg2_st(di, o, v[i].y);
        }
    }
    ctx.tick(6, frame);
    {
        ctx.sync();
        const float tot = lg2_block_sum<LG3_THREADS>(ctx, acc, L.red);
        if (tid == 0) loss_part[frame] = tot;
        if (has_share) {
            ctx.sync();
            const float tot_u = lg2_block_sum<LG3_THREADS>(ctx, acc_u, L.red);
            if (tid == 0) unk_part[frame] = tot_u;
        }
    }
    ctx.tick(7, frame);
}
