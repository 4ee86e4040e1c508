// Device side of fft_any.h for the gfx950 kernels: the workgroup-level driver of the mixed-radix passes over
// LDS lines (mri_any.hip: the any-size dynamic-MRI operator; nufft_any.hip: the oversampled-grid FFT2 of the
// radial operator at sizes the fused 384-point transforms are not built for).
#pragma once
#include "common.h"
#include "fft_any.h"

#ifndef ANY_THREADS
#define ANY_THREADS 512
#endif
#ifndef ANY_LDS_BUF
#define ANY_LDS_BUF (64 * 1024)
#endif
// (line buffer(s) of a workgroup: two workgroups per CU at 64 KB, three below 52 KB)

static size_t any_lds_bytes(int n, int ld, int nl, int direct) {
    return ((size_t)n + (size_t)(direct ? 2 : 1) * n * ld + nl) * sizeof(cf32) + 16 * sizeof(float) +
           (size_t)((nl + 1) & ~1) * sizeof(int);
}

__device__ __forceinline__ void any_twiddles(cf32* tw, int n) {
    for (int e = threadIdx.x; e < n; e += ANY_THREADS) {
        float sn, cn;
        sincospif((float)(2 * e) / (float)n, &sn, &cn);
        tw[e] = c_make(cn, sn);
    }
}

template <int SIGN, int R>
__device__ __forceinline__ void any_pass_reg(cf32* buf, const cf32* tw, int n, int s, unsigned magic, int nl,
                                             int nlp, int ld) {
    constexpr int MAXB = FFT_ANY_REGS / R;
    cf32 v[MAXB][R];
    fft_any_gather<R, MAXB>(buf, n, nl, nlp, ld, (int)threadIdx.x, ANY_THREADS, v);
    __syncthreads();
    fft_any_scatter<SIGN, R, MAXB>(buf, tw, n, s, magic, nl, nlp, ld, (int)threadIdx.x, ANY_THREADS, v);
    __syncthreads();
}

// transform of the nl lines of `a` in place; `b` (a second buffer) only when the plan has a direct pass
template <int SIGN>
__device__ __forceinline__ void any_fft(cf32* a, cf32* b, const cf32* tw, int n, int nl, int nlp, int ld,
                                        const FftAnyPlan& plan) {
    int s = 1;
    for (int i = 0; i < plan.npass; ++i) {
        const int r = plan.radix[i];
        const unsigned mg = plan.magic[i];
        switch (r) {
        case 16: any_pass_reg<SIGN, 16>(a, tw, n, s, mg, nl, nlp, ld); break;
        case 12: any_pass_reg<SIGN, 12>(a, tw, n, s, mg, nl, nlp, ld); break;
        case 8: any_pass_reg<SIGN, 8>(a, tw, n, s, mg, nl, nlp, ld); break;
        case 4: any_pass_reg<SIGN, 4>(a, tw, n, s, mg, nl, nlp, ld); break;
        case 2: any_pass_reg<SIGN, 2>(a, tw, n, s, mg, nl, nlp, ld); break;
        case 3: any_pass_reg<SIGN, 3>(a, tw, n, s, mg, nl, nlp, ld); break;
        case 5: any_pass_reg<SIGN, 5>(a, tw, n, s, mg, nl, nlp, ld); break;
        case 7: any_pass_reg<SIGN, 7>(a, tw, n, s, mg, nl, nlp, ld); break;
        default:
            fft_any_pass<SIGN>(a, b, tw, n, s, r, nl, ld, (int)threadIdx.x, ANY_THREADS);
            __syncthreads();
            for (int e = threadIdx.x; e < n * ld; e += ANY_THREADS) a[e] = b[e];
            __syncthreads();
        }
        s *= r;
    }
}

