// Mixed-radix Stockham passes for lines of ANY length N <= 1024 held in LDS: the transforms behind the
// any-resolution dynamic-MRI kernels (mri_any.hip; the reference's operator takes any `resolution`,
// forward_operator/dynamic_mri.py:9,42-48 -- the tuned kernels of mri_fft.hip are R = 192 only).
// Plain C++ like fft_small.h: tests/test_fft_any_host.py runs the same code on the host against numpy.
//
// Layout: `nl` lines side by side, element i of line c at buf[i * LD + c]; autosort passes (natural order in,
// natural order out, no bit reversal).  One pass of radix r at stride s (s = product of the radices already
// done, n = N / s the current sub-length, m = n / r):
//   out[q + s (r p + j)] = w_n^(p j) * sum_k in[q + s (p + k m)] w_r^(j k),   p < m, q < s, j < r,
// a butterfly is the pair (p, q) = one index t = p s + q < N / r, its inputs sit at t + k N / r.
// tw[e] = (cos, sin)(2 pi e / N); SIGN = -1 forward (torch.fft.fftn), +1 unnormalised inverse (the adjoint).
//
// Radices 16, 12, 8, 4, 3, 2, 5, 7 run IN PLACE through registers: every thread gathers the inputs of its butterflies
// (fft_any_gather), the workgroup meets at a barrier, every thread scatters its outputs (fft_any_scatter) --
// one buffer instead of two, so two or three workgroups share a CU.  A thread is one line c (of NLP, the
// line count rounded up to a power of two) and the butterflies t = trow + i TPR, i < MAXB: no division by the
// line count; p = t / s is a multiply by a constant of the plan.  Any other prime factor takes the out-of-place
// direct pass (fft_any_pass), which needs a second buffer.
#pragma once
#include "fft_small.h"

#define FFT_ANY_MAX_N 1024
#define FFT_ANY_MAX_PASS 12
#define FFT_ANY_REGS 16          // complex values a thread may hold between gather and scatter

struct FftAnyPlan {
    int npass;
    int direct;                       // some radix > 7: the out-of-place pass and a second buffer are needed
    int radix[FFT_ANY_MAX_PASS];
    unsigned magic[FFT_ANY_MAX_PASS]; // t / s = (t * magic) >> 21 for t < 1024, s = the pass's stride
};

// butterflies a thread may hold for radix r; the radices that run through registers
static inline int fft_any_maxb(int r) { return FFT_ANY_REGS / r; }
static inline bool fft_any_in_regs(int r) { return r <= 5 || r == 7 || r == 8 || r == 12 || r == 16; }

// fewest passes for 2^a 3^b out of the register radices 16, 12, 8, 4, 3, 2 (a <= 10, b <= 6: a tiny search)
static inline int fft_any_best(int a, int b, int* out) {
    static const int R[6] = {16, 12, 8, 4, 3, 2}, DA[6] = {4, 2, 3, 2, 0, 1}, DB[6] = {0, 1, 0, 0, 1, 0};
    if (a == 0 && b == 0) return 0;
    int best = 1 << 20, tmp[FFT_ANY_MAX_PASS];
    for (int i = 0; i < 6; ++i) {
        if (a < DA[i] || b < DB[i]) continue;
        const int n = 1 + fft_any_best(a - DA[i], b - DB[i], tmp);
        if (n < best) {
            best = n;
            out[0] = R[i];
            for (int k = 1; k < n && k < FFT_ANY_MAX_PASS; ++k) out[k] = tmp[k - 1];
        }
    }
    return best;
}

// 2s and 3s in as few register passes as possible (192 = 16 x 12, 256 = 16 x 16, ...), then 5s and 7s (register
// passes too), then any other prime factor as a direct r-point sum
static inline bool fft_any_plan(int n, FftAnyPlan* p) {
    p->npass = 0;
    p->direct = 0;
    if (n < 1 || n > FFT_ANY_MAX_N) return false;
    int a = 0, b = 0;
    while (n % 2 == 0) { n /= 2; ++a; }
    while (n % 3 == 0) { n /= 3; ++b; }
    p->npass = fft_any_best(a, b, p->radix);
    if (p->npass > FFT_ANY_MAX_PASS) return false;
    for (int f = 5; n > 1 && p->npass < FFT_ANY_MAX_PASS; f += 2)
        while (n % f == 0 && p->npass < FFT_ANY_MAX_PASS) { p->radix[p->npass++] = f; n /= f; }
    if (n != 1) return false;
    int s = 1;
    for (int i = 0; i < p->npass; ++i) {
        p->magic[i] = (1u << 21) / (unsigned)s + 1u;
        for (unsigned t = 0; t < FFT_ANY_MAX_N; ++t)
            if (((t * p->magic[i]) >> 21) != t / (unsigned)s) return false;   // never: checked, not assumed
        if (!fft_any_in_regs(p->radix[i])) p->direct = 1;
        s *= p->radix[i];
    }
    return true;
}

// the largest power-of-two line count NLP (<= 64) with which `nthreads` threads cover every pass of the plan
// in registers (N / r <= maxb(r) * nthreads / NLP) and the line buffer(s) stay within `lds_bytes`
static inline int fft_any_lines(int n, const FftAnyPlan* p, int nthreads, int pad, size_t lds_bytes) {
    for (int nlp = 64; nlp >= 1; nlp >>= 1) {
        bool ok = (size_t)(p->direct ? 2 : 1) * n * (nlp + pad) * sizeof(cf32) <= lds_bytes;
        for (int i = 0; ok && i < p->npass; ++i) {
            const int r = p->radix[i];
            if (fft_any_in_regs(r) && n / r > fft_any_maxb(r) * (nthreads / nlp)) ok = false;
        }
        if (ok) return nlp;
    }
    return 0;
}

template <int SIGN>
FFT_HD void fft8(cf32 (&u)[8]) {
    const float h = 0.70710678118654752f;
    fft4<SIGN>(u[0], u[2], u[4], u[6]);               // E[k] in u[0], u[2], u[4], u[6]
    fft4<SIGN>(u[1], u[3], u[5], u[7]);               // O[k] in u[1], u[3], u[5], u[7]
    const cf32 o1 = c_rot<SIGN>(u[3], h, h), o2 = c_muli<SIGN>(u[5]), o3 = c_rot<SIGN>(u[7], -h, h);
    const cf32 e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1];
    u[0] = c_add(e0, o0); u[4] = c_sub(e0, o0);
    u[1] = c_add(e1, o1); u[5] = c_sub(e1, o1);
    u[2] = c_add(e2, o2); u[6] = c_sub(e2, o2);
    u[3] = c_add(e3, o3); u[7] = c_sub(e3, o3);
}

template <int SIGN, int R>
FFT_HD void fft_any_butterfly(cf32 (&u)[R], const cf32* tw, int nb) {
    if constexpr (R == 16) {
        fft16<SIGN>(u);
    } else if constexpr (R == 12) {
        fft12<SIGN>(u);
    } else if constexpr (R == 8) {
        fft8<SIGN>(u);
    } else if constexpr (R == 4) {
        fft4<SIGN>(u[0], u[1], u[2], u[3]);
    } else if constexpr (R == 2) {
        const cf32 a = u[0], b = u[1];
        u[0] = c_add(a, b);
        u[1] = c_sub(a, b);
    } else if constexpr (R == 3) {
        fft3<SIGN>(u[0], u[1], u[2]);
    } else {                                          // 5, 7: the direct sum, roots from the table
        static_assert(R == 5 || R == 7, "no register butterfly of this radix");
        cf32 o[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            cf32 acc = u[0];
#pragma unroll
            for (int k = 1; k < R; ++k) {
                const cf32 w = tw[((j * k) % R) * nb];
                acc = c_add(acc, c_rot<SIGN>(u[k], w.x, w.y));
            }
            o[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) u[j] = o[j];
    }
}

// thread `tid` of `nthreads`: line c = tid & (NLP - 1), butterflies t = (tid / NLP) + i (nthreads / NLP)
template <int R, int MAXB>
FFT_HD void fft_any_gather(const cf32* buf, int N, int nl, int NLP, int LD, int tid, int nthreads, cf32 (&v)[MAXB][R]) {
    const int nb = N / R, is = nb * LD, c = tid & (NLP - 1), tpr = nthreads / NLP;
    int t = tid / NLP;
#pragma unroll
    for (int i = 0; i < MAXB; ++i, t += tpr) {
        if (t < nb && c < nl) {
            const cf32* in = buf + t * LD + c;
#pragma unroll
            for (int k = 0; k < R; ++k) v[i][k] = in[k * is];
        }
    }
}

template <int SIGN, int R, int MAXB>
FFT_HD void fft_any_scatter(cf32* buf, const cf32* tw, int N, int s, unsigned magic, int nl, int NLP, int LD, int tid,
                            int nthreads, cf32 (&v)[MAXB][R]) {
    const int nb = N / R, os = s * LD, c = tid & (NLP - 1), tpr = nthreads / NLP;
    int t = tid / NLP;
#pragma unroll
    for (int i = 0; i < MAXB; ++i, t += tpr) {
        if (t < nb && c < nl) {
            const int p = (int)(((unsigned)t * magic) >> 21), q = t - p * s;
            cf32* out = buf + (q + R * s * p) * LD + c;
            const int w1 = p * s;                     // w_n^p = tw[p s]
            fft_any_butterfly<SIGN, R>(v[i], tw, nb);
            out[0] = v[i][0];
#pragma unroll
            for (int j = 1; j < R; ++j) {
                const cf32 w = tw[w1 * j];
                out[j * os] = c_rot<SIGN>(v[i][j], w.x, w.y);
            }
        }
    }
}

// out-of-place pass of any radix (a -> b): the direct r-point sum, one OUTPUT per work item (a prime length is
// a single butterfly per line: its r outputs are what there is to share out); serves the prime factors above 7
template <int SIGN>
FFT_HD void fft_any_pass(const cf32* a, cf32* b, const cf32* tw, int N, int s, int r, int nl, int LD,
                         int tid, int nthreads) {
    const int nb = N / r, rs = r * s;
    const int is = nb * LD, os = s * LD;
    for (int e = tid; e < N * nl; e += nthreads) {
        const int tj = e / nl, c = e - tj * nl;
        const int t = tj / r, j = tj - t * r;
        const int p = t / s, q = t - p * s;
        const cf32* in = a + t * LD + c;
        cf32 acc = in[0];
        int jk = 0;                                   // (j k) mod r
        for (int k = 1; k < r; ++k) {
            jk += j;
            if (jk >= r) jk -= r;
            const cf32 w = tw[jk * nb], v = c_rot<SIGN>(in[k * is], w.x, w.y);
            acc = c_add(acc, v);
        }
        const cf32 w = tw[p * s * j];
        b[(q + rs * p + s * j) * LD + c] = c_rot<SIGN>(acc, w.x, w.y);
    }
}
