// In-register small FFTs (3, 4, 12, 16, 24 points), the 192 = 16 x 12 decomposition used by
// the fused MRI kernels and the 384 = 16 x 24 one of the radial-MRI grid transforms.  Plain C++ so that the same code is unit-tested on the host
// (tests/test_fft_small_host.py compiles it with g++ against a naive DFT).
//
// SIGN = -1: forward transform  X[k] = sum_n x[n] exp(-2 pi i n k / N)   (torch.fft.fftn)
// SIGN = +1: unnormalised inverse (the adjoint of the forward), exp(+2 pi i n k / N).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define FFT_HD __host__ __device__ __forceinline__
#else
#define FFT_HD inline
#endif

struct __attribute__((aligned(8))) cf32 {
    float x, y;
};

FFT_HD cf32 c_make(float x, float y) { cf32 r; r.x = x; r.y = y; return r; }
FFT_HD cf32 c_add(cf32 a, cf32 b) { return c_make(a.x + b.x, a.y + b.y); }
FFT_HD cf32 c_sub(cf32 a, cf32 b) { return c_make(a.x - b.x, a.y - b.y); }
FFT_HD cf32 c_mul(cf32 a, cf32 b) { return c_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// multiply by exp(SIGN * i * theta) given (cos theta, sin theta)
// Written with explicit fused multiply-adds: left to the compiler, which of the two products of
// each component is fused into the add differs from call site to call site, and two kernels that
// run the same transform on the same values would differ in the last bit.
template <int SIGN>
FFT_HD cf32 c_rot(cf32 a, float c, float s) {
    return SIGN < 0 ? c_make(fmaf(a.x, c, a.y * s), fmaf(a.y, c, -(a.x * s)))
                    : c_make(fmaf(a.x, c, -(a.y * s)), fmaf(a.y, c, a.x * s));
}
// multiply by SIGN * i
template <int SIGN>
FFT_HD cf32 c_muli(cf32 a) { return SIGN < 0 ? c_make(a.y, -a.x) : c_make(-a.y, a.x); }

template <int SIGN>
FFT_HD void fft4(cf32& a0, cf32& a1, cf32& a2, cf32& a3) {
    const cf32 t0 = c_add(a0, a2), t1 = c_sub(a0, a2), t2 = c_add(a1, a3);
    const cf32 t3 = c_muli<SIGN>(c_sub(a1, a3));
    a0 = c_add(t0, t2);
    a2 = c_sub(t0, t2);
    a1 = c_add(t1, t3);
    a3 = c_sub(t1, t3);
}

template <int SIGN>
FFT_HD void fft3(cf32& a0, cf32& a1, cf32& a2) {
    const float h = 0.8660254037844386f;            // sin(2 pi / 3)
    const cf32 s = c_add(a1, a2);
    const cf32 d = c_muli<SIGN>(c_make(h * (a1.x - a2.x), h * (a1.y - a2.y)));
    const cf32 m = c_make(a0.x - 0.5f * s.x, a0.y - 0.5f * s.y);
    a0 = c_add(a0, s);
    a1 = c_add(m, d);
    a2 = c_sub(m, d);
}

// 16-point FFT, natural order in and out:  n = 4a + b, k = c + 4d.
template <int SIGN>
FFT_HD void fft16(cf32 (&v)[16]) {
    const float c1 = 0.9238795325112867f, s1 = 0.3826834323650898f, r2 = 0.7071067811865476f;
    // step 1: for each b, 4-point FFT over a (elements 4a+b) -> Y_b[c] stored at v[4c+b]
#pragma unroll
    for (int b = 0; b < 4; ++b) fft4<SIGN>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // step 2: twiddle Y_b[c] by W16^(b c)
    v[4 * 1 + 1] = c_rot<SIGN>(v[4 * 1 + 1], c1, s1);        // bc = 1
    v[4 * 1 + 2] = c_rot<SIGN>(v[4 * 1 + 2], r2, r2);        // 2
    v[4 * 1 + 3] = c_rot<SIGN>(v[4 * 1 + 3], s1, c1);        // 3
    v[4 * 2 + 1] = c_rot<SIGN>(v[4 * 2 + 1], r2, r2);        // 2
    v[4 * 2 + 2] = c_muli<SIGN>(v[4 * 2 + 2]);               // 4
    v[4 * 2 + 3] = c_rot<SIGN>(v[4 * 2 + 3], -r2, r2);       // 6
    v[4 * 3 + 1] = c_rot<SIGN>(v[4 * 3 + 1], s1, c1);        // 3
    v[4 * 3 + 2] = c_rot<SIGN>(v[4 * 3 + 2], -r2, r2);       // 6
    v[4 * 3 + 3] = c_rot<SIGN>(v[4 * 3 + 3], -c1, -s1);      // 9
    // step 3: for each c, 4-point FFT over b (elements 4c+b) -> X[c + 4d] left at v[4c+d]
#pragma unroll
    for (int c = 0; c < 4; ++c) fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    // transpose so that v[k] = X[k]:  X[c + 4d] sits at 4c + d
    cf32 t;
#define FFT_SWAP(i, j) t = v[i]; v[i] = v[j]; v[j] = t;
    FFT_SWAP(1, 4) FFT_SWAP(2, 8) FFT_SWAP(3, 12) FFT_SWAP(6, 9) FFT_SWAP(7, 13) FFT_SWAP(11, 14)
#undef FFT_SWAP
}

// 12-point FFT, natural order in and out:  n = 4a + b (a < 3, b < 4), k = c + 3d (c < 3, d < 4).
template <int SIGN>
FFT_HD void fft12(cf32 (&v)[12]) {
    const float h = 0.8660254037844386f;
    // step 1: for each b, 3-point FFT over a (elements 4a+b) -> Y_b[c] at v[4c+b]
#pragma unroll
    for (int b = 0; b < 4; ++b) fft3<SIGN>(v[b], v[4 + b], v[8 + b]);
    // step 2: twiddle by W12^(b c)
    v[4 * 1 + 1] = c_rot<SIGN>(v[4 * 1 + 1], h, 0.5f);       // 1
    v[4 * 1 + 2] = c_rot<SIGN>(v[4 * 1 + 2], 0.5f, h);       // 2
    v[4 * 1 + 3] = c_muli<SIGN>(v[4 * 1 + 3]);               // 3
    v[4 * 2 + 1] = c_rot<SIGN>(v[4 * 2 + 1], 0.5f, h);       // 2
    v[4 * 2 + 2] = c_rot<SIGN>(v[4 * 2 + 2], -0.5f, h);      // 4
    v[4 * 2 + 3] = c_make(-v[4 * 2 + 3].x, -v[4 * 2 + 3].y); // 6
    // step 3: for each c, 4-point FFT over b -> X[c + 3d] at v[4c+d]
#pragma unroll
    for (int c = 0; c < 3; ++c) fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    // reorder to natural order: out[c + 3d] = v[4c + d]
    cf32 o[12];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) o[c + 3 * d] = v[4 * c + d];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = o[i];
}

// 192-point FFT as 16 x 12 (n = 12 a + b, k = c + 16 d):
//   pass A, task b (0..11):  Y_b[c] = W192^(b c) * FFT16_a( x[12 a + b] ),  stored at slot 12 c + b
//   pass B, task c (0..15):  X[c + 16 d] = FFT12_b( Y_b[c] ),  read from slots 12 c + b
// tw[m] = (cos, sin)(2 pi m / 192), m < 192.
template <int SIGN>
FFT_HD void fft192_pass_a(cf32 (&v)[16], int b, const cf32* tw) {
    fft16<SIGN>(v);
#pragma unroll
    for (int c = 1; c < 16; ++c) {
        const cf32 w = tw[b * c];
        v[c] = c_rot<SIGN>(v[c], w.x, w.y);
    }
}
template <int SIGN>
FFT_HD void fft192_pass_b(cf32 (&v)[12]) { fft12<SIGN>(v); }

// 24-point FFT, natural order in and out: radix 2 over two 12-point transforms
// (X[d] = E[d] + W24^d O[d], X[d + 12] = E[d] - W24^d O[d]).
template <int SIGN>
FFT_HD void fft24(cf32 (&v)[24]) {
    // (cos, sin)(2 pi d / 24), d < 12
    const float C24[12] = {1.f, 0.96592582628906829f, 0.86602540378443865f, 0.70710678118654752f, 0.5f,
                           0.25881904510252076f, 0.f, -0.25881904510252076f, -0.5f, -0.70710678118654752f,
                           -0.86602540378443865f, -0.96592582628906829f};
    const float S24[12] = {0.f, 0.25881904510252076f, 0.5f, 0.70710678118654752f, 0.86602540378443865f,
                           0.96592582628906829f, 1.f, 0.96592582628906829f, 0.86602540378443865f,
                           0.70710678118654752f, 0.5f, 0.25881904510252076f};
    cf32 e[12], o[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { e[j] = v[2 * j]; o[j] = v[2 * j + 1]; }
    fft12<SIGN>(e);
    fft12<SIGN>(o);
#pragma unroll
    for (int d = 0; d < 12; ++d) {
        const cf32 t = c_rot<SIGN>(o[d], C24[d], S24[d]);
        v[d] = c_add(e[d], t);
        v[d + 12] = c_sub(e[d], t);
    }
}

// 384-point FFT as 16 x 24 (n = 24 a + b, k = c + 16 d), the oversampled grid of the radial-MRI
// NUFFT (nufft_fft.hip):
//   pass A, task b (0..23):  Y_b[c] = W384^(b c) * FFT16_a( x[24 a + b] ),  stored at slot 24 c + b
//   pass B, task c (0..15):  X[c + 16 d] = FFT24_b( Y_b[c] ),  read from slots 24 c + b
// tw[m] = (cos, sin)(2 pi m / 384), m < 384 (b c <= 23 * 15 = 345).
template <int SIGN>
FFT_HD void fft384_pass_a(cf32 (&v)[16], int b, const cf32* tw) {
    fft16<SIGN>(v);
#pragma unroll
    for (int c = 1; c < 16; ++c) {
        const cf32 w = tw[b * c];
        v[c] = c_rot<SIGN>(v[c], w.x, w.y);
    }
}
template <int SIGN>
FFT_HD void fft384_pass_b(cf32 (&v)[24]) { fft24<SIGN>(v); }
