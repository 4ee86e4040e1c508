// FFT2 of the radial operator's oversampled grids at ANY grid size G <= 1024, in place: what the gridding
// path (nufft_grid.hip) runs between k_nufft_pad_apod and the interpolation, and between the spread and
// k_nufft_crop_apod, when the fused, pruned 384-point transforms of nufft_fft.hip (R = 192) do not apply.
// NO REFERENCE COUNTERPART (see nufft_grid.hip); until round 3 these sizes went through a library FFT.
// Two kernels on the mixed-radix LDS passes of fft_any.h: rows (contiguous lines, transposed into LDS so that
// the passes see them side by side), then column strips.  sign -1: forward; +1: unnormalised inverse.
#include "fft_any_lds.h"

// NR (a power of two) rows of a frame per workgroup: element (row jj, column x) at a[x * LD + jj]
template <int SIGN>
__global__ __launch_bounds__(ANY_THREADS, 4) void k_fft2_any_rows(cf32* __restrict__ grid, int G, int NR, int ngroups,
                                                                  FftAnyPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = NR + 1;
    cf32* tw = (cf32*)lds;
    cf32* a = tw + G;
    cf32* b = a + (size_t)G * LD;                     // only with a direct pass in the plan
    const int frame = blockIdx.x / ngroups, grp = blockIdx.x - frame * ngroups, tid = threadIdx.x;
    const int j0 = grp * NR, nr = min(NR, G - j0);
    cf32* rows = grid + ((size_t)frame * G + j0) * G;
    any_twiddles(tw, G);
    for (int e0 = tid; e0 < nr * G; e0 += 4 * ANY_THREADS) {
        cf32 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ANY_THREADS;
            v[u] = rows[e < nr * G ? e : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ANY_THREADS, jj = e / G, xx = e - jj * G;
            if (e < nr * G) a[xx * LD + jj] = v[u];
        }
    }
    __syncthreads();
    any_fft<SIGN>(a, b, tw, G, nr, NR, LD, plan);
    for (int e = tid; e < nr * G; e += ANY_THREADS) {
        const int jj = e / G, xx = e - jj * G;
        rows[e] = a[xx * LD + jj];
    }
}

// SW (a power of two) columns of a frame per workgroup: element (row i, column c) at a[i * SW + c]
template <int SIGN>
__global__ __launch_bounds__(ANY_THREADS, 4) void k_fft2_any_cols(cf32* __restrict__ grid, int G, int SW, int lgSW,
                                                                  int nstrips, FftAnyPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    cf32* tw = (cf32*)lds;
    cf32* a = tw + G;
    cf32* b = a + (size_t)G * SW;
    const int frame = blockIdx.x / nstrips, strip = blockIdx.x - frame * nstrips, tid = threadIdx.x;
    const int c0 = strip * SW, nl = min(SW, G - c0);
    cf32* base = grid + (size_t)frame * G * G + c0;
    any_twiddles(tw, G);
    for (int e0 = tid; e0 < G * SW; e0 += 8 * ANY_THREADS) {
        cf32 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * ANY_THREADS, i = e >> lgSW, c = e & (SW - 1);
            v[u] = base[e < G * SW && c < nl ? (size_t)i * G + c : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * ANY_THREADS;
            if (e < G * SW) a[e] = v[u];
        }
    }
    __syncthreads();
    any_fft<SIGN>(a, b, tw, G, nl, SW, SW, plan);
    for (int e = tid; e < G * SW; e += ANY_THREADS) {
        const int i = e >> lgSW, c = e & (SW - 1);
        if (c < nl) base[(size_t)i * G + c] = a[e];
    }
}

extern "C" int nufft_fft2_any_supported(int G) {
    FftAnyPlan p;
    return G >= 2 && G <= FFT_ANY_MAX_N && fft_any_plan(G, &p) && fft_any_lines(G, &p, ANY_THREADS, 1, ANY_LDS_BUF) > 0;
}

extern "C" int nufft_fft2_any(float* grid_c64, int nframes, int G, int sign, void* stream) {
    const char* who = "nufft_fft2_any";
    STEP_REQUIRE(grid_c64 && nframes > 0, "%s: null pointer / no frames", who);
    STEP_REQUIRE(sign == 1 || sign == -1, "%s: sign %d (-1 forward, +1 unnormalised inverse)", who, sign);
    STEP_REQUIRE(((uintptr_t)grid_c64 & 7) == 0, "%s: grid must be 8-byte aligned", who);
    FftAnyPlan plan;
    STEP_REQUIRE(G >= 2 && G <= FFT_ANY_MAX_N && fft_any_plan(G, &plan), "%s: grid size %d (2 .. %d)", who, G,
                 FFT_ANY_MAX_N);
    const int SW = fft_any_lines(G, &plan, ANY_THREADS, 0, ANY_LDS_BUF);
    int NR = fft_any_lines(G, &plan, ANY_THREADS, 1, ANY_LDS_BUF);
    STEP_REQUIRE(SW > 0 && NR > 0, "%s: %d does not fit the line buffers", who, G);
    int lgSW = 0;
    while ((1 << lgSW) < SW) ++lgSW;
    const int nstrips = (G + SW - 1) / SW, ngroups = (G + NR - 1) / NR;
    STEP_REQUIRE((long long)nframes * (nstrips > ngroups ? nstrips : ngroups) < (1ll << 31), "%s: %d frames make too large a grid",
                 who, nframes);
    const size_t nbuf = plan.direct ? 2 : 1;
    const size_t lds_r = ((size_t)G + nbuf * G * (NR + 1)) * sizeof(cf32), lds_c = ((size_t)G + nbuf * G * SW) * sizeof(cf32);
    hipStream_t st = (hipStream_t)stream;
#define FFT2_ANY_LAUNCH(SG)                                                                                   \
    step_allow_lds((const void*)k_fft2_any_rows<SG>, lds_r);                                                  \
    hipLaunchKernelGGL(k_fft2_any_rows<SG>, dim3((unsigned)nframes * ngroups), dim3(ANY_THREADS), lds_r, st,  \
                       (cf32*)grid_c64, G, NR, ngroups, plan);                                                \
    STEP_CHECK_LAUNCH(who);                                                                                   \
    step_allow_lds((const void*)k_fft2_any_cols<SG>, lds_c);                                                  \
    hipLaunchKernelGGL(k_fft2_any_cols<SG>, dim3((unsigned)nframes * nstrips), dim3(ANY_THREADS), lds_c, st,  \
                       (cf32*)grid_c64, G, SW, lgSW, nstrips, plan);                                          \
    STEP_CHECK_LAUNCH(who)
    if (sign < 0) { FFT2_ANY_LAUNCH(-1); }
    else { FFT2_ANY_LAUNCH(1); }
#undef FFT2_ANY_LAUNCH
    return STEP_OK;
}
