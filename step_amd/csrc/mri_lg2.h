// Dynamic-MRI Langevin step, second-generation body: forward FFT2 + mask + residual + loss +
// adjoint of forward_operator/dynamic_mri.py:42-48 with base.py:40-41, fused per frame
// (SURVEY Appendix B "MRI"), for masks whose kept rows fit one LDS row buffer (<= 64 rows).
//
// Same arithmetic as k_mri_loss_grad (mri_fft.hip) -- 192 = 16 x 12 in-register passes over LDS,
// pruned to the kept k_y rows, fftshift folded into signs -- but scheduled differently:
//   * software-pipelined column passes: while a thread transforms pass A of strip s (inputs
//     loaded from global memory straight into the pass-A register layout, up to three strips
//     ahead, in register sets that rotate by name) it
//     also finishes pass B of strip s-1 from registers; the strip buffer only carries the
//     A -> B exchange: two barriers and 28 LDS accesses per thread and strip instead of three
//     barriers and 83;
//   * this thread's row-buffer positions live in registers, one byte each (constant per mask);
//   * the row phase is three in-place passes: pass A, then pass B + residual + loss + the first
//     pass of the inverse in the same registers (the inverse runs the transposed decomposition
//     k = c + 16 d -> n = q + 12 p), then the inverse's 16-point pass;
//   * row buffer padded 12 -> 13 per group (stride 208): both row-pass access patterns are
//     conflict-free;
//   * the masked-out k_y rows, where A(x) = 0 and the residual is -y, contribute sum |y|^2 to the
//     loss: a constant of the measurement.  With one measurement for all chains (y_batch = 1,
//     YQ = false) the workgroups of the first 16 chains each sum a 16th of those rows once --
//     a quarter less traffic per chain; with a measurement per chain (YQ = true) every
//     workgroup sums its own, 32 rows per strip, loaded one strip ahead of their use;
//   * one workgroup per frame (a loop over frames inside the kernel carries the prefetched
//     strip around the back edge and the register allocator then spills ~170 values, whose
//     reloads wait behind every store in flight -- measured 1.5x slower than no loop).
// The body is plain C++ over a small context (thread id, barrier, LDS base), so the same code
// runs on the host under pthreads (tests/test_mri_lg2_host.py) and on the GPU.
#pragma once
#include "fft_small.h"

#define LG2_R 192
#define LG2_SW 32
#define LG2_NS 6                  // strips per frame
#define LG2_THREADS 512
#define LG2_RS 208                // row-buffer stride in cf32: 16 groups of 12, each padded to 13
#define LG2_NKC_MAX 64
#define LG2_YQ 6                  // float4 loads per thread and strip for the masked-out rows
#define LG2_UNK_PARTS 16          // a batch-1 measurement: its masked-out rows are shared out

#if defined(__HIPCC__)
#define LG2_DEV __device__ __forceinline__
// The value is declared modified: address arithmetic derived from it stays inside the loop
// instead of being hoisted into dozens of long-lived (and then spilled) registers.
#define LG2_OPAQUE(v) asm volatile("" : "+v"(v))
// The instruction scheduler does not move anything across this point: the two transforms of a
// pipeline stage run one after the other instead of interleaved, which is what keeps their
// temporaries (and with them the kernel) within the register file.
#define LG2_FENCE() __builtin_amdgcn_sched_barrier(0)
// a group of `n` instructions of the classes in `mask` (0x002 vector ALU, 0x020 vector memory read) at this
// point of the scheduling region's order: alternating groups spread loads between arithmetic
#define LG2_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#else
#define LG2_SGB(mask, n)
#define LG2_FENCE()
#define LG2_DEV inline
#define LG2_OPAQUE(v) asm volatile("" : "+r"(v))
struct float4 { float x, y, z, w; };
#endif

// element at byte offset `off` (32 bit, unsigned) from a workgroup-uniform base: on the GPU
// this is the scalar-base + vector-offset form of global_load / global_store
LG2_DEV float lg2_ld(const float* base, unsigned off) { return *(const float*)((const char*)base + off); }
// gradient stores are streamed out (non-temporal on the GPU): the gradient is not read again by this
// kernel, and keeping it out of the caches leaves them to x and the shared measurement
// (B = 64: 0.150 -> 0.128 ms, B = 32 and B = 256 unchanged; round 2)
#if defined(__HIPCC__)
LG2_DEV void lg2_st(float* base, unsigned off, float v) { __builtin_nontemporal_store(v, (float*)((char*)base + off)); }
#else
LG2_DEV void lg2_st(float* base, unsigned off, float v) { *(float*)((char*)base + off) = v; }
#endif

struct Lg2Lds {
    cf32* tw;      // [192]   (cos, sin)(2 pi m / 192)
    cf32* S;       // [192][32] pass A -> pass B exchange of one strip
    cf32* Rk;      // [nkc][208] kept k_y rows
    int* rowpos;   // [192] position of a k_y row in the kept-row list, -1 if masked out
    int* rowl;     // [192] the kept-row list
    float* red;    // [512] loss reduction
};

FFT_HD size_t lg2_lds_bytes(int nkc) {
    return (size_t)(LG2_R + LG2_R * LG2_SW + (size_t)nkc * LG2_RS) * sizeof(cf32) +
           2 * LG2_R * sizeof(int) + LG2_THREADS * sizeof(float);
}

FFT_HD Lg2Lds lg2_carve(float* lds, int nkc) {
    Lg2Lds L;
    L.tw = (cf32*)lds;
    L.S = L.tw + LG2_R;
    L.Rk = L.S + LG2_R * LG2_SW;
    L.rowpos = (int*)(L.Rk + (size_t)nkc * LG2_RS);
    L.rowl = L.rowpos + LG2_R;
    L.red = (float*)(L.rowl + LG2_R);
    return L;
}

// slot n of a row (natural order) -> position in the padded row
FFT_HD int lg2_phys(int n) { return n + (int)((unsigned)n / 12u); }
FFT_HD float lg2_sgn(int k) { return (k & 1) ? -1.f : 1.f; }

// pass-A inputs for task (bb, col): image rows 12 a + bb; off = byte offset of (row bb, column)
LG2_DEV void lg2_load_strip(const float* xr, const float* xi, unsigned off, cf32 (&v)[16]) {
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const unsigned o = off + (unsigned)a * (12u * LG2_R * 4u);
        v[a] = c_make(lg2_ld(xr, o), lg2_ld(xi, o));
    }
}

// one of the 16 (re, im) pairs of lg2_load_strip
LG2_DEV void lg2_load_pair(const float* xr, const float* xi, unsigned off, int a, cf32 (&v)[16]) {
    const unsigned o = off + (unsigned)a * (12u * LG2_R * 4u);
    v[a] = c_make(lg2_ld(xr, o), lg2_ld(xi, o));
}

// Tables of one mask frame.  The caller syncs before (LDS reuse) -- this function ends synced.
// `my_row` = rowlist[tid] for tid < nk (nk <= LG2_NKC_MAX <= LG2_THREADS), fetched by the caller
// ahead of the first strip: fetched here, behind the strip's 32 loads, the wait for it would be
// a wait for the strip.
template <class Ctx>
LG2_DEV void lg2_tables(const Ctx& ctx, const Lg2Lds& L, int my_row, int nk) {
    const int tid = ctx.tid();
    for (int e = tid; e < LG2_R; e += LG2_THREADS) {
        L.tw[e] = ctx.cos_sin_2pi_over_192(e);
        L.rowpos[e] = -1;
    }
    ctx.sync();
    if (tid < nk) {
        L.rowpos[my_row] = tid;
        L.rowl[tid] = my_row;
    }
    ctx.sync();
}

template <int NT = LG2_THREADS, class Ctx>
LG2_DEV float lg2_block_sum(const Ctx& ctx, float v, float* red) {
    const int tid = ctx.tid();
    red[tid] = v;
    ctx.sync();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) s += red[tid + 64 * k];
        red[tid] = s;
    }
    ctx.sync();
    float t = 0.f;
    if (tid == 0)
        for (int k = 0; k < 64; ++k) t += red[k];
    return t;                                        // valid in thread 0
}

// Row phase on the kept rows Rk[nk][208] (slot n of a row at n + n / 12), in place, NT threads:
// forward pass A (16-point, twiddle), then pass B (12-point) + residual against the measurement
// rows + loss + the first pass of the inverse in the same registers (the inverse runs the
// transposed decomposition k = c + 16 d -> n = q + 12 p), then the inverse's 16-point pass.
// In: Rk[j][kx'] = (-1)^ky * column FFT; out: the adjoint's input rows in natural order.
// rowlist: the kept rows of this mask frame (global memory); rowl: the same list in LDS.
// Row j of round r is handled by lanes 16 j .. 16 j + 15 of thread index e = tid + r NT; 2 NT / 16
// must cover nk.  Ends with the row buffer complete but NOT synced when with_grad is false.
// `hook(i)`, i = 0, 1, 2: called once after the first round's measurement loads are out, once between the two
// pass-B rounds and once after them -- the persistent kernel requests the three strips of its next frame
// there, one at a time, so that no wave ever has more loads in flight than the counter holds (63).
struct Lg2NoHook {
    LG2_DEV void operator()(int) const {}
};

template <int NT, class Ctx, class Hook = Lg2NoHook>
LG2_DEV void lg2_row_phase(const Ctx& ctx, cf32* Rk, const cf32* tw, const int* rowl,
                           const int* __restrict__ rowlist, int nk, const float* __restrict__ yr,
                           const float* __restrict__ yi, float cs2, bool with_grad, int frame,
                           float& acc, const Hook& hook = Hook()) {
    constexpr int R = LG2_R, RS = LG2_RS;
    // measurement rows of this thread's pass-B tasks (row j, group c): kx = c + 16 d; the
    // first round's are fetched here, the second round's under the first round's transform
    float ya[12], yb[12], yc[12], yd[12];
    int tr = ctx.tid();
    LG2_OPAQUE(tr);
    const int c16 = tr & 15, j0 = tr >> 4, j1 = (tr + NT) >> 4;
#define LG2_YLOAD(j_, re_, im_)                                                        \
    if ((j_) < nk) {                                                               \
        const unsigned o_ = (unsigned)(rowlist[(j_)] * R + c16) * 4u; \
        _Pragma("unroll") for (int d = 0; d < 12; ++d) {                           \
            re_[d] = lg2_ld(yr, o_ + 64u * d);                                     \
            im_[d] = lg2_ld(yi, o_ + 64u * d);                                     \
        }                                                                          \
    }
    LG2_YLOAD(j0, ya, yb)
    hook(0);
    ctx.sync();                                  // Rk complete
    ctx.tick(2, frame);

    // ---------------------------------------------------------------- row phase
    // pass A (forward along x): task (row j, bb), 16 lanes per row of which 12 work
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        const int j = r ? j1 : j0;
        if (j < nk && c16 < 12) {
            cf32* row = Rk + j * RS + c16;
            cf32 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = row[13 * i];
            fft192_pass_a<-1>(v, c16, tw);
#pragma unroll
            for (int i = 0; i < 16; ++i) row[13 * i] = v[i];
        }
    }
    ctx.sync();
    ctx.tick(3, frame);
    // pass B + residual + loss + first pass of the inverse (12-point over d, twiddle q c)
#define LG2_ROW_B(j_, re_, im_)                                                        \
    if ((j_) < nk) {                                                               \
        cf32* row = Rk + (j_) * RS + 13 * c16;                                   \
        cf32 v[12];                                                                \
        _Pragma("unroll") for (int i = 0; i < 12; ++i) v[i] = row[i];              \
        fft192_pass_b<-1>(v);            /* v[d] = (-1)^ky FFT2(z)[ky][c + 16 d] */ \
        const float sx = lg2_sgn(c16), sc = cs2 * sx * lg2_sgn(rowl[j_]);        \
        _Pragma("unroll") for (int d = 0; d < 12; ++d) {                           \
            const float rr = sx * v[d].x - re_[d], ri = sx * v[d].y - im_[d];      \
            acc += rr * rr + ri * ri;                                              \
            v[d] = c_make(sc * rr, sc * ri);                                       \
        }                                                                          \
        if (with_grad) {                                                           \
            fft12<1>(v);                                                           \
            _Pragma("unroll") for (int q = 1; q < 12; ++q) {                       \
                const cf32 w = tw[q * c16];                                      \
                v[q] = c_rot<1>(v[q], w.x, w.y);                                   \
            }                                                                      \
            _Pragma("unroll") for (int i = 0; i < 12; ++i) row[i] = v[i];          \
        }                                                                          \
    }
    LG2_YLOAD(j1, yc, yd)
    LG2_FENCE();
    LG2_ROW_B(j0, ya, yb)
    LG2_FENCE();
    hook(1);
    LG2_FENCE();
    LG2_ROW_B(j1, yc, yd)
    LG2_FENCE();
    hook(2);
#undef LG2_ROW_B
#undef LG2_YLOAD
    if (with_grad) {
        ctx.sync();
        ctx.tick(4, frame);
        // second pass of the inverse: task (row j, q), 16-point over c -> n = q + 12 p
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const int j = r ? j1 : j0;
            if (j < nk && c16 < 12) {
                cf32* row = Rk + j * RS + c16;
                cf32 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = row[13 * i];
                fft16<1>(v);
#pragma unroll
                for (int i = 0; i < 16; ++i) row[13 * i] = v[i];
            }
        }
    }
}

// One workgroup per frame (frame = block index).  x, grad: [nframes][2][R][R];
// y: [y_batch*T][2][R][R]; rows [mask_frames][R], nrows [mask_frames]; every mask frame has at
// most nkc <= 64 kept rows (the caller checked).  grad may be null (loss only).
// YQ: loss_part[frame] = sum |A(x) - y|^2 over all rows.  !YQ (y_batch == 1): loss_part[frame]
// covers the kept rows, and unk_part[b * T + t], b < min(B, 16), the b-th share of the masked-out
// rows of measurement frame t (every chain's loss gets all shares: the caller adds them up).
// The state update applied where the inverse column phase would have stored the gradient (UPD > 0:
// `grad` is not written): x (the very state the step was evaluated at) and x0hat are read at the
// thread's twelve output pixels while its last pass runs, x is stored.
//   UPD 1: MCMCSampler.optimize, daps.py:210-223:  x -= lr (g + (x - x0hat) / sigma^2)
//   UPD 2 / 3: the Langevin / momentum-free HMC step of MCMCSampler.sample (daps.py:93-104, :135-144) for
//   the 'gaussian' prior solver, score = -g / tau^2 + (x0hat - x) / sigma^2, with the noise READ from
//   `eps` (laid out like x): the arithmetic of mcmc_elem_a<0 / 1> (mcmc_elem.h) with xt = x0hat, no prior.
struct Lg2Upd {
    float* x;            // [nframes][2][R][R]: the state, updated in place
    const float* x0;     // x0hat
    const float* eps;    // UPD >= 2
    float lr, inv_sigma2, inv_tau2, sqrt_lr, sqrt_2lr, noise_scale;
    int* nan_flag;       // UPD >= 2, may be null: set to 1 when an updated element is NaN
};

template <int UPD>
LG2_DEV float lg2_upd_elem(float xi, float g, float xh, float e, const Lg2Upd& p) {
    if (UPD == 1) return xi - p.lr * (g + (xi - xh) * p.inv_sigma2);
    const float s = -g * p.inv_tau2 + (xh - xi) * p.inv_sigma2 + 0.f;
    if (UPD == 2) return xi + p.lr * s + p.sqrt_2lr * e;
    const float v = 0.f * 0.f + p.sqrt_lr * s + p.noise_scale * e;
    return xi + v * p.sqrt_lr;
}

template <bool YQ, int UPD = 0, class Ctx = void>
LG2_DEV void lg2_body(const Ctx& ctx, const float* __restrict__ x, const float* __restrict__ y,
                      int y_batch, const int* __restrict__ rows, const int* __restrict__ nrows,
                      int mask_frames, int T, int nframes, int nkc,
                      const float* __restrict__ chain_scale, float* __restrict__ loss_part,
                      float* __restrict__ unk_part, float* __restrict__ grad, Lg2Upd up = Lg2Upd()) {
    constexpr int R = LG2_R, SW = LG2_SW, NS = LG2_NS, RS = LG2_RS, Q = LG2_R / 4;
    const Lg2Lds L = lg2_carve(ctx.lds(), nkc);
    const int tid = ctx.tid();
    const size_t fsz = (size_t)2 * R * R;
    // thread roles in the column passes: column col of the strip; pass A task bb = g (12 per
    // column, waves 6 and 7 have none), pass B task c = g (16 per column).  Each phase derives
    // them again from an opaque copy of the thread id, so that its address arithmetic is not
    // hoisted out of the frame loop into long-lived registers.
#define LG2_IDS(t_)                                                                    \
    int t_ = tid;                                                                      \
    LG2_OPAQUE(t_);                                                                    \
    const int col = t_ & 31, g = t_ >> 5;                                              \
    const bool a_task = g < 12;                                                        \
    const int bb = g < 12 ? g : 0;                                                     \
    const unsigned offA = (unsigned)(bb * R + col) * 4u; /* (row bb, column col), strip 0 */ \
    const unsigned offB = (unsigned)(g * R + col) * 4u;  /* (row g, column col) */     \
    (void)offA; (void)offB; (void)a_task;

    const int frame = ctx.block();
    if (frame >= nframes) return;
    // Pass-A inputs in flight: register sets that ROTATE by name (the strip loop is unrolled over
    // them) -- copying a set into the next would wait for loads that are not needed yet.  Set k
    // holds strip s = k mod NSETS; it is refilled with strip s + NSETS as soon as its pass-A
    // results are in the strip buffer.  Three sets, or two when the measurement prefetch (YQ)
    // needs the registers.
    cf32 a0[16], a1[16], a2[16];
    int my_row;
    {
        const int t_ = frame % T;
        my_row = rows[(size_t)(mask_frames == 1 ? 0 : t_) * R + (tid < R ? tid : 0)];
        LG2_IDS(t0)
        const float* xf = x + (size_t)frame * fsz;
        // only the first strip before the set-up: a wave may have 63 loads outstanding and the
        // 64th would hold it at the issue until a load returns, with the tables still to build
        // (the loads are not under `if (a_task)`: waves 6 and 7 fetch row 0 again and drop it.
        // A load inside a divergent branch gives the compiler a path on which it was skipped;
        // there an older set holds the youngest loads, and the wait counts, which must be right
        // on every path, become "wait for everything" on all of them)
        // YQ keeps the branch: its measurement prefetch is conditional anyway, and the extra
        // loads only cost there.
        if (!YQ || a_task) lg2_load_strip(xf, xf + R * R, offA, a0);
    }

    // per-thread constants of a mask frame: the row-buffer
    // positions (one byte each, 0xff = masked out) of this thread's pass-B outputs ky = g + 16 d
    // and inverse pass-A inputs ky = 12 a + bb
    int nk = 0;
    unsigned rpB[3], rpA[4];
#define LG2_RP(tab, i) ((int)(signed char)((tab)[(i) >> 2] >> (8 * ((i) & 3))))

    {
        const int b = frame / T, t = frame - b * T;
        const int mf = mask_frames == 1 ? 0 : t;
        {
            nk = nrows[mf];
            lg2_tables(ctx, L, my_row, nk);
            LG2_IDS(t1)
            if (!YQ || a_task) {
                const float* xf = x + (size_t)frame * fsz;
                lg2_load_strip(xf, xf + R * R, offA + SW * 4u, a1);
            }
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
        }
        const float* xr = x + (size_t)frame * fsz;
        const float* xi = xr + R * R;
        const float* yr = y + ((size_t)(y_batch > 1 ? b : 0) * T + t) * fsz;
        const float* yi = yr + R * R;
        const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
        float acc = 0.f, acc_u = 0.f;
        const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;
        const bool has_share = !YQ && b < nparts;        // uniform over the workgroup
        ctx.tick(0, frame);

        // ---------------------------------------------------------------- forward column phase
        {
        LG2_IDS(t2)
        const float sgB = lg2_sgn(g);                // (-1)^ky for ky = g + 16 d
        cf32 u[12];
        constexpr int NSETS = YQ ? 2 : 3;
        if (!YQ) lg2_load_strip(xr, xi, offA + 2 * SW * 4u, a2);
        float4 yq[LG2_YQ];
        if (YQ) {
#pragma unroll
            for (int i = 0; i < LG2_YQ; ++i) yq[i].x = yq[i].y = yq[i].z = yq[i].w = 0.f;
        }
        // one stage: pass B of strip s - 1 (from u), pass A of strip s (register set `cur`), the
        // A -> B exchange through the strip buffer, then `cur` is refilled with strip s + NSETS
#define LG2_FWD_STAGE(s, cur)                                                          \
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
                _Pragma("unroll") for (int d = 0; d < 12; ++d) {                       \
                    const int j = LG2_RP(rq, d);                                       \
                    if (j >= 0) L.Rk[j * RS + n] = c_make(sgB * u[d].x, sgB * u[d].y); \
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
            if ((s) + NSETS < NS && (!YQ || a_task)) {                                 \
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
                LG2_FWD_STAGE(s, a0)
                LG2_FWD_STAGE(s + 1, a1)
            }
        } else {
            LG2_FWD_STAGE(0, a0) LG2_FWD_STAGE(1, a1) LG2_FWD_STAGE(2, a2)
            LG2_FWD_STAGE(3, a0) LG2_FWD_STAGE(4, a1) LG2_FWD_STAGE(5, a2)
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
        lg2_row_phase<LG2_THREADS>(ctx, L.Rk, L.tw, L.rowl, rows + (size_t)mf * R, nk, yr, yi, cs2,
                                   UPD != 0 || grad != nullptr, frame, acc);
        bool bad = false;
        if (UPD != 0 || grad) {
            ctx.sync();
            ctx.tick(5, frame);

            // ------------------------------------------------------------ inverse column phase
            float* dr = (UPD ? up.x : grad) + (size_t)frame * fsz;
            float* di = dr + R * R;
            const float* hr = UPD ? up.x0 + (size_t)frame * fsz : nullptr;
            const float* er = UPD >= 2 ? up.eps + (size_t)frame * fsz : nullptr;
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
                unsigned ob = offB + (unsigned)s * (SW * 4u);
                LG2_OPAQUE(ob);
                if (UPD) {
                    // state, x0hat (and noise) of the twelve output pixels travel under the last pass
                    cf32 xo[12], xh[12], ev[12];
#pragma unroll
                    for (int d = 0; d < 12; ++d) {
                        const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                        xo[d] = c_make(lg2_ld(dr, o), lg2_ld(di, o));
                        xh[d] = c_make(lg2_ld(hr, o), lg2_ld(hr + R * R, o));
                        ev[d] = UPD >= 2 ? c_make(lg2_ld(er, o), lg2_ld(er + R * R, o)) : c_make(0.f, 0.f);
                    }
                    fft192_pass_b<1>(u);
#pragma unroll
                    for (int d = 0; d < 12; ++d) {
                        const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                        const float nr_ = lg2_upd_elem<UPD>(xo[d].x, u[d].x, xh[d].x, ev[d].x, up);
                        const float ni_ = lg2_upd_elem<UPD>(xo[d].y, u[d].y, xh[d].y, ev[d].y, up);
                        bad |= (nr_ != nr_) | (ni_ != ni_);
                        lg2_st(dr, o, nr_);
                        lg2_st(di, o, ni_);
                    }
                } else {
                fft192_pass_b<1>(u);
#pragma unroll
                for (int d = 0; d < 12; ++d) {
                    const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                    lg2_st(dr, o, u[d].x);
                    lg2_st(di, o, u[d].y);
                }
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
        if (UPD >= 2 && up.nan_flag && bad) ctx.raise_flag(up.nan_flag);
        ctx.tick(6, frame);
        const float tot = lg2_block_sum(ctx, acc, L.red);
        if (tid == 0) loss_part[frame] = tot;
        if (has_share) {
            ctx.sync();
            const float tot_u = lg2_block_sum(ctx, acc_u, L.red);
            if (tid == 0) unk_part[frame] = tot_u;
        }
        ctx.tick(7, frame);
    }
#undef LG2_RP
#undef LG2_IDS
}
