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
//   accumulator layout, a 31-shuffle reduce-scatter over the 32 columns, fixed-order sum over strips.
// adjoint, one workgroup (8 waves) per (chain, frame), one 32-column strip per wave:
//   grad[y][x] = sum_k G[k][y] EXK[k][x]        M = y, N = x, K = 2F+1 rows
//   G[k][y] (g folded into ey) is built in LDS, EXK lives in registers per strip,
//   accumulator rows are image rows -> 128-byte row stores.
// Two workgroups per CU (LDS 2 x 64 KB, <= 128 registers per lane): four waves per SIMD hide
// the dword loads / row stores behind the other waves' MFMAs.
#include "common.h"

#define SEP_THREADS 512
#define SEP_WAVES 8
#define SEP_U 8           // k-steps per software-pipeline stage of the forward (16 image rows)
#define SEP_MAX_STRIPS 16 // W <= 512

// ------------------------------------------------------------------------------- forward
// HC / WC: compile-time image height / width (0 = runtime values, at half the occupancy): the
// row loads then take scalar-base + immediate addressing instead of one address register each.
template <int HC, int WC>
__global__ __launch_bounds__(SEP_THREADS, (HC ? 4 : 2)) void k_dft_fwd_sep(
    const float* __restrict__ ey_t, const float* __restrict__ ex_ri, const float* __restrict__ x,
    float* __restrict__ vis, float* __restrict__ flux, int T, int F, int Hrt, int Wrt, float scale,
    float shift, float clip_min) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = HC ? HC : Hrt, W = WC ? WC : Wrt;
    float* eyl = lds;                         // [H][64]
    float* sred = lds + (size_t)H * 64;       // [nstrip][64]
    const int b = blockIdx.x, t = blockIdx.y;
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
    const int nstrip = W >> 5, nblk = (H / 2) / SEP_U;

    for (int strip = wave; strip < nstrip; strip += SEP_WAVES) {
        const int xcol = strip * 32 + li;
        const int xo = h * W + xcol;                          // row 2*ks + h, per-lane int offset
        // acc0/acc1 collect one MFMA chain per pipeline stage (SEP_U k-steps = 16 image rows,
        // started from C = 0) and are then added into tot0/tot1: short fma chains keep the fp32
        // rounding of the H-row sum at the level of torch's pairwise CPU sum.
        f32x16 tot0, tot1;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot0[r] = tot1[r] = 0.f;

        float xa[SEP_U], xb[SEP_U];
#define SEP_LOAD(dst, blk)                                                         \
    _Pragma("unroll") for (int u = 0; u < SEP_U; ++u)                              \
        dst[u] = xf[xo + (2 * ((blk) * SEP_U + u)) * W];
#define SEP_COMPUTE(src, blk)                                                      \
    {                                                                              \
        f32x16 acc0, acc1;                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;    \
        _Pragma("unroll") for (int u = 0; u < SEP_U; ++u) {                        \
            const float xv = prep_pixel(src[u], scale, shift, clip_min);           \
            const float* ar = eyl + (2 * ((blk) * SEP_U + u) + h) * 64 + li;       \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[0], xv, acc0, 0, 0, 0); \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[32], xv, acc1, 0, 0, 0);\
        }                                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                           \
            tot0[r] += acc0[r];                                                    \
            tot1[r] += acc1[r];                                                    \
        }                                                                          \
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
        // tot0[r] = Re Z[f][x], tot1[r] = Im Z[f][x] with f = mfma32_row(r, h), x = xcol;
        // v[r] / v[16+r] = Re / Im of ex[f][x] * Z[f][x]
        float v[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = mfma32_row(r, h);
            const float er = exr[f * W + xcol], ei = exi[f * W + xcol];
            v[r] = er * tot0[r] - ei * tot1[r];
            v[16 + r] = er * tot1[r] + ei * tot0[r];
        }
        // reduce-scatter over the 32 columns of each half wave: lane li ends with the total of v[li]
#pragma unroll
        for (int m = 16, n = 32; m >= 1; m >>= 1, n >>= 1) {
            const bool up = (li & m) != 0;
#pragma unroll
            for (int i = 0; i < n / 2; ++i) {
                const float keep = up ? v[i + n / 2] : v[i];
                const float send = up ? v[i] : v[i + n / 2];
                v[i] = keep + __shfl_xor(send, m);
            }
        }
        // lane li of half h: component c = li >> 4, baseline f = mfma32_row(li & 15, h)
        sred[strip * 64 + (li >> 4) * 32 + mfma32_row(li & 15, h)] = v[0];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x >> 5, f = threadIdx.x & 31;
        float s = 0.f;
        for (int w = 0; w < nstrip; ++w) s += sred[w * 64 + c * 32 + f];
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
    const size_t lds = ((size_t)H * 64 + (W / 32) * 64) * sizeof(float);
    if (H == 256 && W == 256) {
        step_allow_lds((const void*)k_dft_fwd_sep<256, 256>, lds);
        hipLaunchKernelGGL((k_dft_fwd_sep<256, 256>), dim3(B, T), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_t, ex_ri, x, vis_c64, flux, T, F, H, W, scale,
                           shift, clip_min);
    } else {
        step_allow_lds((const void*)k_dft_fwd_sep<0, 0>, lds);
        hipLaunchKernelGGL((k_dft_fwd_sep<0, 0>), dim3(B, T), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_t, ex_ri, x, vis_c64, flux, T, F, H, W, scale,
                           shift, clip_min);
    }
    STEP_CHECK_LAUNCH("bh_dft_fwd_sep");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- adjoint
// HC / WC: compile-time image height / width (0 = use the runtime values) so that the LDS reads
// of G and the row stores take immediate offsets instead of one address register each.
template <int HC, int WC>
__global__ __launch_bounds__(SEP_THREADS, (HC ? 4 : 2)) void k_dft_adj_sep(
    const float* __restrict__ ey_ri, const float* __restrict__ ex_ri,
    const float* __restrict__ gvis, const float* __restrict__ gflux,
    const float* __restrict__ chain_scale, float* __restrict__ grad, int T, int F, int Hrt, int Wrt,
    float out_scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = HC ? HC : Hrt, W = WC ? WC : Wrt;
    const int KP = (2 * F + 2) & ~1;          // k rows: Re f | Im f | flux | (zero pad)
    float* G = lds;                           // [KP][H]
    float* g2 = lds + (size_t)KP * H;         // [2F + 1]
    const int b = blockIdx.x, t = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const float cs = out_scale * (chain_scale ? chain_scale[b] : 1.f);

    for (int e = threadIdx.x; e < 2 * F + 1; e += SEP_THREADS) {
        float v;
        if (e < F) v = gvis[(((size_t)b * T + t) * F + e) * 2];
        else if (e < 2 * F) v = gvis[(((size_t)b * T + t) * F + (e - F)) * 2 + 1];
        else v = gflux[(size_t)b * T + t];
        g2[e] = cs * v;
    }
    __syncthreads();
    const float* eyr = ey_ri + ((size_t)t * 2 + 0) * 32 * H;
    const float* eyi = ey_ri + ((size_t)t * 2 + 1) * 32 * H;
    for (int e = threadIdx.x; e < KP * H; e += SEP_THREADS) {
        const int k = e / H, y = e - k * H;
        float v = 0.f;
        if (k < F) v = eyr[k * H + y] * g2[k] + eyi[k * H + y] * g2[F + k];
        else if (k < 2 * F) {
            const int f = k - F;
            v = eyr[f * H + y] * g2[F + f] - eyi[f * H + y] * g2[f];
        } else if (k == 2 * F) v = g2[2 * F];
        G[e] = v;
    }
    __syncthreads();

    const float* exr = ex_ri + ((size_t)t * 2 + 0) * 32 * W;
    const float* exi = ex_ri + ((size_t)t * 2 + 1) * 32 * W;
    float* gf = grad + ((size_t)b * T + t) * H * W;
    const int ksteps = KP >> 1, nstrip = W >> 5, ntile = H >> 5;
    for (int strip = wave; strip < nstrip; strip += SEP_WAVES) {
        const int xcol = strip * 32 + li;
        float bx[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int k = 2 * s + h;
            float v = 0.f;
            if (k < F) v = exr[k * W + xcol];
            else if (k < 2 * F) v = exi[(k - F) * W + xcol];
            else if (k == 2 * F) v = 1.f;
            bx[s] = v;
        }
#pragma unroll 1
        for (int tile = 0; tile < ntile; ++tile) {
            const float* gp = G + h * H + tile * 32 + li;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 32; ++s)
                if (s < ksteps)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[(2 * s) * H], bx[s], acc, 0, 0, 0);
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
    const size_t lds = ((size_t)KP * H + 2 * F + 1) * sizeof(float);
    if (H == 256 && W == 256) {
        step_allow_lds((const void*)k_dft_adj_sep<256, 256>, lds);
        hipLaunchKernelGGL((k_dft_adj_sep<256, 256>), dim3(B, T), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_ri, ex_ri, gvis_c64, gflux, chain_scale, grad, T,
                           F, H, W, out_scale);
    } else {
        step_allow_lds((const void*)k_dft_adj_sep<0, 0>, lds);
        hipLaunchKernelGGL((k_dft_adj_sep<0, 0>), dim3(B, T), dim3(SEP_THREADS), lds,
                           (hipStream_t)stream, ey_ri, ex_ri, gvis_c64, gflux, chain_scale, grad, T,
                           F, H, W, out_scale);
    }
    STEP_CHECK_LAUNCH("bh_dft_adj_sep");
    return STEP_OK;
}
