// Black-hole interferometric DFT in separable form on the fp32 matrix cores of gfx950.
//
// ehtim's ftmatrix (called at blackhole.py:226) is rank-1 per baseline:
//   A[t][f][y*W+x] = ey[t][f][y] * ex[t][f][x],  ey = pulse * exp(2 pi i y_l v), ex = exp(2 pi i x_l u)
// so  vis[f] = sum_x ex[f][x] * ( sum_y ey[f][y] X[y][x] )         (forward, blackhole.py:315-321)
//     grad[y][x] = sum_f Re( conj(ex[f][x]) * conj(ey[f][y]) g[f] ) (its autograd adjoint)
// The inner sums are real GEMMs with the same 4*F*H*W flop count per frame as the dense
// contraction, but the only HBM stream is the image itself (16.8 MB per chain per direction
// at 64x256x256): no 0.94 GB matrix is ever read.  Flux rides along as baseline f = F with
// ey = ex = 1 (blackhole.py:328-329).
//
// forward, one workgroup (8 waves) per (chain, frame), one 32-column strip per wave:
//   Z[(f,c)][x] = sum_y EY[(f,c)][y] X[y][x]    M = 64 rows (Re f | Im f), N = x, K = y
//   A operand from an LDS image of ey ([y][64], conflict-free ds_read_b32), B operand straight
//   from global memory as coalesced dwords (one 128-byte row segment per half wave), two
//   v_mfma_f32_32x32x2_f32 per loaded dword.  Epilogue: complex multiply by ex in the
//   accumulator layout, column sums through an LDS transpose, fixed-order sum over strips.
// adjoint, one workgroup (8 waves) per (chain, frame), one 32-column strip per wave:
//   grad[y][x] = sum_k G[k][y] EXK[k][x]        M = y, N = x, K = 2F+1 rows
//   G[k][y] (g folded into ey) is built in LDS, EXK lives in registers per strip,
//   accumulator rows are image rows -> 128-byte row stores.
// Two workgroups per CU (LDS 2 x 66 KB, <= 128 registers per lane): four waves per SIMD hide
// the dword loads / row stores behind the other waves' MFMAs.
#include "common.h"

#define SEP_THREADS 512
#define SEP_WAVES 8
#define SEP_U 8           // k-steps per software-pipeline stage of the forward (16 image rows)
#define SEP_MAX_STRIPS 16 // W <= 512

// One strip of the forward: tot0[r] = Re Z[f][x], tot1[r] = Im Z[f][x], f = mfma32_row(r, h).
// acc0/acc1 collect one MFMA chain per pipeline stage (SEP_U k-steps = 16 image rows, started
// from C = 0) and are then added into tot0/tot1: short fma chains keep the fp32 rounding of
// the H-row sum at the level of torch's pairwise CPU sum.
__device__ __forceinline__ void fwd_strip(const float* __restrict__ xf, const float* eyl, int xo,
                                          int W, int nblk, int h, int li, float scale, float shift,
                                          float clip_min, f32x16& tot0, f32x16& tot1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tot0[r] = tot1[r] = 0.f;
    float xa[SEP_U], xb[SEP_U];
#define SEP_LOAD(dst, blk)                                                         \
    _Pragma("unroll") for (int u = 0; u < SEP_U; ++u)                              \
        dst[u] = __builtin_nontemporal_load(xf + xo + (2 * ((blk) * SEP_U + u)) * W);
#define SEP_FLUSH _Pragma("unroll") for (int r = 0; r < 16; ++r) { tot0[r] += acc0[r]; tot1[r] += acc1[r]; }
#define SEP_ACCINIT _Pragma("unroll") for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#define SEP_COMPUTE(src, blk)                                                      \
    {                                                                              \
        f32x16 acc0, acc1;                                                         \
        SEP_ACCINIT                                                                \
        _Pragma("unroll") for (int u = 0; u < SEP_U; ++u) {                        \
            const float xv = prep_pixel(src[u], scale, shift, clip_min);           \
            const float* ar = eyl + (2 * ((blk) * SEP_U + u) + h) * 64 + li;       \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[0], xv, acc0, 0, 0, 0); \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[32], xv, acc1, 0, 0, 0);\
        }                                                                          \
        SEP_FLUSH                                                                  \
    }
    SEP_LOAD(xa, 0);
#pragma unroll 1
    for (int blk = 0; blk < nblk; blk += 2) {
        if (blk + 1 < nblk) { SEP_LOAD(xb, blk + 1); }
        SEP_COMPUTE(xa, blk);
        if (blk + 2 < nblk) { SEP_LOAD(xa, blk + 2); }
        if (blk + 1 < nblk) { SEP_COMPUTE(xb, blk + 1); }
    }
#undef SEP_LOAD
#undef SEP_COMPUTE
}

// Epilogue of one strip: p[f] = sum_x ex[f][x] Z[f][x] over the strip's 32 columns.
// Each lane multiplies its 16 rows, the wave transposes through its private LDS scratch
// ([32 values][64 lanes]), lane (i, h) sums value i of half h over the 32 columns with a
// rotated start (bank-conflict free) in a fixed order, and stores into sred[strip].
__device__ __forceinline__ void fwd_reduce(const f32x16& tot0, const f32x16& tot1,
                                           const float* __restrict__ exr,
                                           const float* __restrict__ exi, int W, int xcol, int lane,
                                           float* scr, float* sred_strip) {
    const int li = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = mfma32_row(r, h);
        const float er = exr[f * W + xcol], ei = exi[f * W + xcol];
        scr[r * 64 + lane] = er * tot0[r] - ei * tot1[r];
        scr[(16 + r) * 64 + lane] = er * tot1[r] + ei * tot0[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // pairwise, not running: the sum over x is the level of the contraction whose fp32 rounding shows
    // most in an ill-conditioned visibility (scripts/probes/sep_sum_order.py: summation noise 3.6e-9 of
    // sum |a x| with running sums over the 32 columns and the strips, 2.0e-9 with trees -- what is
    // left is the rounding inside the 16-row MFMA chains)
    float v[32];
    const float* row = scr + li * 64 + h * 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = row[(j + li) & 31];
#pragma unroll
    for (int w = 16; w > 0; w >>= 1)
#pragma unroll
        for (int j = 0; j < w; ++j) v[j] += v[j + w];
    // lane (li, h): component c = li >> 4, baseline f = mfma32_row(li & 15, h)
    sred_strip[(li >> 4) * 32 + mfma32_row(li & 15, h)] = v[0];
}

// floats in front of sred: the ey image, or the aliased transpose scratch if that is larger
__host__ __device__ static inline int fwd_sred_offset(int H, int nstrip) {
    const int ey = H * 64, scr = SEP_WAVES * 2048;
    return (nstrip <= SEP_WAVES && ey < scr) ? scr : ey;
}

// ------------------------------------------------------------------------------- forward
// HC / WC: compile-time image height / width (0 = runtime values, at half the occupancy): the
// row loads then take scalar-base + immediate addressing instead of one address register each.
// LDS: ey image [H][64] | sred [nstrip][64] | (only when W > 256) scratch [8][32][64];
// with W <= 256 every wave owns at most one strip and the scratch aliases the ey image once
// all waves are past their MFMA loop.
template <int HC, int WC>
__global__ __launch_bounds__(SEP_THREADS, (HC ? 4 : 2)) void k_dft_fwd_sep(
    const float* __restrict__ ey_t, const float* __restrict__ ex_ri, const float* __restrict__ x,
    float* __restrict__ vis, float* __restrict__ flux, int T, int F, int Hrt, int Wrt, float scale,
    float shift, float clip_min) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = HC ? HC : Hrt, W = WC ? WC : Wrt;
    const int nstrip = W >> 5, nblk = (H / 2) / SEP_U;
    float* eyl = lds;                         // [H][64]
    float* sred = lds + fwd_sred_offset(H, nstrip);   // [nstrip][64]
    const int t = blockIdx.x, b = blockIdx.y;   // ids in memory order; XCD c gets t % 8 == c
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;

    {   // stage the frame's ey image (contiguous H*64 floats)
        const float4* src = (const float4*)(ey_t + (size_t)t * H * 64);
        float4* dst = (float4*)eyl;
        for (int e = threadIdx.x; e < H * 16; e += SEP_THREADS) dst[e] = src[e];
    }
    __syncthreads();

    const float* xf = x + ((size_t)b * T + t) * H * W;
    const float* exr = ex_ri + ((size_t)t * 2 + 0) * 32 * W;
    const float* exi = ex_ri + ((size_t)t * 2 + 1) * 32 * W;

    if (nstrip <= SEP_WAVES) {
        f32x16 tot0, tot1;
        const bool mine = wave < nstrip;
        const int xcol = wave * 32 + li;
        if (mine) fwd_strip(xf, eyl, h * W + xcol, W, nblk, h, li, scale, shift, clip_min, tot0, tot1);
        __syncthreads();                      // every wave is done reading the ey image
        if (mine) fwd_reduce(tot0, tot1, exr, exi, W, xcol, lane, lds + wave * 2048, sred + wave * 64);
    } else {
        float* scr = sred + nstrip * 64 + wave * 2048;
        for (int strip = wave; strip < nstrip; strip += SEP_WAVES) {
            f32x16 tot0, tot1;
            const int xcol = strip * 32 + li;
            fwd_strip(xf, eyl, h * W + xcol, W, nblk, h, li, scale, shift, clip_min, tot0, tot1);
            fwd_reduce(tot0, tot1, exr, exi, W, xcol, lane, scr, sred + strip * 64);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x >> 5, f = threadIdx.x & 31;
        float sv[SEP_MAX_STRIPS];                                // the strips: a tree as well
#pragma unroll
        for (int w = 0; w < SEP_MAX_STRIPS; ++w) sv[w] = w < nstrip ? sred[w * 64 + c * 32 + f] : 0.f;
#pragma unroll
        for (int w = SEP_MAX_STRIPS / 2; w > 0; w >>= 1)
#pragma unroll
            for (int j = 0; j < w; ++j) sv[j] += sv[j + w];
        const float s = sv[0];
        if (f < F) vis[(((size_t)b * T + t) * F + f) * 2 + c] = s;
        else if (f == F && c == 0) flux[(size_t)b * T + t] = s;
    }
}

extern "C" int bh_dft_fwd_sep(const float* ey_t, const float* ex_ri, int T, int F, int H, int W,
                              const float* x, int B, float scale, float shift, float clip_min,
                              float* vis_c64, float* flux, void* stream) {
    STEP_REQUIRE(ey_t && ex_ri && x && vis_c64 && flux, "bh_dft_fwd_sep: null pointer");
    STEP_REQUIRE(T > 0 && B > 0 && F > 0 && F <= 31, "bh_dft_fwd_sep: need 1 <= F <= 31 (F=%d)", F);
    STEP_REQUIRE(H > 0 && H % 32 == 0 && H <= 512, "bh_dft_fwd_sep: need H %% 32 == 0, H <= 512 (H=%d)", H);
    STEP_REQUIRE(W > 0 && W % 32 == 0 && W <= 32 * SEP_MAX_STRIPS, "bh_dft_fwd_sep: need W %% 32 == 0, W <= 512 (W=%d)", W);
    STEP_REQUIRE(T <= 65535, "bh_dft_fwd_sep: T too large");
    const int nstrip = W / 32;
    size_t floats = (size_t)fwd_sred_offset(H, nstrip) + nstrip * 64;
    if (nstrip > SEP_WAVES) floats += SEP_WAVES * 2048;          // separate transpose scratch
    const size_t lds = floats * sizeof(float);
    STEP_REQUIRE(lds <= 160 * 1024, "bh_dft_fwd_sep: %zu B of LDS needed (H=%d W=%d)", lds, H, W);
    if (H == 256 && W == 256) {
        step_allow_lds((const void*)k_dft_fwd_sep<256, 256>, lds);
        hipLaunchKernelGGL((k_dft_fwd_sep<256, 256>), dim3(T, B), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_t, ex_ri, x, vis_c64, flux, T, F, H, W, scale,
                           shift, clip_min);
    } else {
        step_allow_lds((const void*)k_dft_fwd_sep<0, 0>, lds);
        hipLaunchKernelGGL((k_dft_fwd_sep<0, 0>), dim3(T, B), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_t, ex_ri, x, vis_c64, flux, T, F, H, W, scale,
                           shift, clip_min);
    }
    STEP_CHECK_LAUNCH("bh_dft_fwd_sep");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- adjoint
// HC / WC: compile-time image height / width (0 = use the runtime values) so that the LDS reads
// of G and the row stores take immediate offsets instead of one address register each.
template <int HC, int WC, int YS>
__global__ __launch_bounds__(SEP_THREADS, (HC ? 8 : 2)) void k_dft_adj_sep(
    const float* __restrict__ ey_ri, const float* __restrict__ ex_ri,
    const float* __restrict__ gvis, const float* __restrict__ gflux,
    const float* __restrict__ chain_scale, float* __restrict__ grad, int T, int F, int Hrt, int Wrt,
    float out_scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = HC ? HC : Hrt, W = WC ? WC : Wrt;
    const int Hs = H / YS, y0 = blockIdx.z * Hs;   // this workgroup's image rows [y0, y0 + Hs)
    const int KP = (2 * F + 2) & ~1;          // k rows: Re f | Im f | flux | (zero pad)
    float* G = lds;                           // [KP][Hs]
    float* g2 = lds + KP * Hs;                // [2F + 2]: Re g | Im g | gflux | 0
    const int t = blockIdx.x, b = blockIdx.y;   // ids in memory order; XCD c gets t % 8 == c
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const float cs = out_scale * (chain_scale ? chain_scale[b] : 1.f);

    for (int e = threadIdx.x; e < 2 * F + 2; e += SEP_THREADS) {
        float v = 0.f;
        if (e < F) v = gvis[(((size_t)b * T + t) * F + e) * 2];
        else if (e < 2 * F) v = gvis[(((size_t)b * T + t) * F + (e - F)) * 2 + 1];
        else if (e == 2 * F) v = gflux[(size_t)b * T + t];
        g2[e] = cs * v;
    }
    __syncthreads();
    // G[k][y]: k < F: Re(conj(ey) g) = eyR gR + eyI gI;  F <= k < 2F: Im = eyR gI - eyI gR;
    // k = 2F: gflux (ey = 1);  branch-free so that the table loads of all iterations overlap
    const float* eyr = ey_ri + ((size_t)t * 2 + 0) * 32 * H + y0;
    const float* eyi = ey_ri + ((size_t)t * 2 + 1) * 32 * H + y0;
#pragma unroll 4
    for (int e = threadIdx.x; e < KP * Hs; e += SEP_THREADS) {
        const int k = e / Hs, y = e - k * Hs;
        const bool im = k >= F;
        const int f = min(im ? k - F : k, F);               // f = F: the flux row (ey = 1 + 0i)
        const float a = eyr[f * H + y], c = eyi[f * H + y];
        const float gr = g2[min(f, F - 1)], gi = g2[F + min(f, F - 1)];
        float v = im ? a * gi - c * gr : a * gr + c * gi;
        if (k >= 2 * F) v = (k == 2 * F) ? g2[2 * F] : 0.f;
        G[e] = v;
    }
    __syncthreads();

    const float* exr = ex_ri + ((size_t)t * 2 + 0) * 32 * W;
    const float* exi = ex_ri + ((size_t)t * 2 + 1) * 32 * W;
    float* gf = grad + ((size_t)b * T + t) * H * W + (size_t)y0 * W;
    const int ksteps = KP >> 1, nstrip = W >> 5, ntile = Hs >> 5;
    for (int strip = wave; strip < nstrip; strip += SEP_WAVES) {
        const int xcol = strip * 32 + li;
        float bx[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {                      // EXK[k][x]: Re ex | Im ex | 1 | 0
            const int k = 2 * s + h;
            const bool im = k >= F;
            const int f = min(im ? k - F : k, F);           // row F of ex_ri is 1 + 0i
            const float* tab = (im && k < 2 * F) ? exi : exr;
            const float v = tab[f * W + xcol];
            bx[s] = (k <= 2 * F) ? v : 0.f;
        }
#pragma unroll 1
        for (int tile = 0; tile < ntile; ++tile) {
            const float* gp = G + h * Hs + tile * 32 + li;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 32; ++s)
                if (s < ksteps)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[(2 * s) * Hs], bx[s], acc, 0, 0, 0);
            const int go = (tile * 32 + 4 * h) * W + xcol;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gf[go + ((r & 3) + 8 * (r >> 2)) * W] = acc[r];
        }
    }
}

extern "C" int bh_dft_adj_sep(const float* ey_ri, const float* ex_ri, int T, int F, int H, int W,
                              const float* gvis_c64, const float* gflux, const float* chain_scale,
                              int B, float out_scale, float* grad, void* stream) {
    STEP_REQUIRE(ey_ri && ex_ri && gvis_c64 && gflux && grad, "bh_dft_adj_sep: null pointer");
    STEP_REQUIRE(T > 0 && B > 0 && F > 0 && F <= 31, "bh_dft_adj_sep: need 1 <= F <= 31 (F=%d)", F);
    STEP_REQUIRE(H > 0 && H % 32 == 0 && H <= 512, "bh_dft_adj_sep: need H %% 32 == 0, H <= 512 (H=%d)", H);
    STEP_REQUIRE(W > 0 && W % 32 == 0 && W <= 32 * SEP_MAX_STRIPS, "bh_dft_adj_sep: need W %% 32 == 0, W <= 512 (W=%d)", W);
    STEP_REQUIRE(T <= 65535, "bh_dft_adj_sep: T too large");
    const int KP = (2 * F + 2) & ~1;
    if (H == 256 && W == 256) {
        // two workgroups per frame (128 image rows each): 30 KB of LDS and 56 registers per lane
        // let four workgroups share a CU, which keeps the matrix pipe fed while others build G
        const size_t lds = ((size_t)KP * 128 + 2 * F + 2) * sizeof(float);
        step_allow_lds((const void*)k_dft_adj_sep<256, 256, 2>, lds);
        hipLaunchKernelGGL((k_dft_adj_sep<256, 256, 2>), dim3(T, B, 2), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_ri, ex_ri, gvis_c64, gflux, chain_scale, grad, T,
                           F, H, W, out_scale);
    } else {
        const size_t lds = ((size_t)KP * H + 2 * F + 2) * sizeof(float);
        step_allow_lds((const void*)k_dft_adj_sep<0, 0, 1>, lds);
        hipLaunchKernelGGL((k_dft_adj_sep<0, 0, 1>), dim3(T, B), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_ri, ex_ri, gvis_c64, gflux, chain_scale, grad, T,
                           F, H, W, out_scale);
    }
    STEP_CHECK_LAUNCH("bh_dft_adj_sep");
    return STEP_OK;
}
