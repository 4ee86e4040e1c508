// Black-hole interferometric DFT with an arbitrary dense matrix A_vis (the reference's
// attribute, forward_operator/blackhole.py:155) on the fp32 matrix cores of gfx950.
//
//   forward  vis[b][t*F+f] = sum_n A[t][f][n] x'[b][t][n]      (blackhole.py:315-321)
//            flux[b][t]    = sum_n x'[b][t][n]                 (blackhole.py:328-329)
//   adjoint  grad[b][t][n] = s_b ( sum_f Re(conj A[t][f][n] g[b][t*F+f]) + gflux[b][t] )
//
// Both are real GEMMs per frame over the packed real matrix R_t [RP][HW] whose rows are
// Re A[f] (r < F), Im A[f] (F <= r < 2F) and a row of ones (r = 2F, the flux row):
//   forward  out[r][b] = sum_n R_t[r][n] x'[b][n]   M = r, N = chain, K = pixel (split-K)
//   adjoint  outT[b][n] = sum_r g[b][r] R_t[r][n]   M = chain, N = pixel, K = r
// v_mfma_f32_32x32x2_f32 is exact fp32 (an fma chain), so the 1e-4 parity budget is spent
// only on summation order.  With dense A the forward streams T*RP*HW*4 bytes of A per call,
// which bounds it by HBM until B is in the hundreds; the separable path (bh_dft_sep.hip)
// removes that traffic when A has ehtim's rank-1 structure.
#include "common.h"

static inline int rows_padded(int F) { return ((2 * F + 1) + 31) / 32 * 32; }

extern "C" int bh_dense_rows_padded(int F) { return rows_padded(F); }
extern "C" size_t bh_dense_packed_floats(int T, int F, int HW) {
    return (size_t)T * rows_padded(F) * HW;
}

static inline int fwd_ksplit(int HW) {
    // >= 8 chunks of 8 pixels per wave (4 waves per workgroup), at most 16 slices per frame
    int ks = HW / 8 / 32;
    if (ks < 1) ks = 1;
    if (ks > 16) ks = 16;
    return ks;
}
static inline int fwd_nt(int B) { return B > 32 ? 2 : 1; }
static inline int fwd_bpad(int B) { int g = 32 * fwd_nt(B); return (B + g - 1) / g * g; }

extern "C" size_t bh_dense_fwd_workspace_floats(int T, int F, int HW, int B) {
    return (size_t)T * fwd_ksplit(HW) * rows_padded(F) * fwd_bpad(B);
}

// ------------------------------------------------------------------------------- pack
__global__ void k_dense_pack(const float2* __restrict__ A, int T, int F, int HW, int RP,
                             float* __restrict__ a_frag, float* __restrict__ a_rows) {
    const size_t total = (size_t)T * RP * HW;
    const int MT = RP / 32;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % HW);
        const int r = (int)((e / HW) % RP);
        const int t = (int)(e / ((size_t)HW * RP));
        float v = 0.f;
        if (r < F) v = A[((size_t)t * F + r) * HW + n].x;
        else if (r < 2 * F) v = A[((size_t)t * F + (r - F)) * HW + n].y;
        else if (r == 2 * F) v = 1.f;
        a_rows[e] = v;
        const int c = n >> 3, h = (n >> 2) & 1, j = n & 3;
        const int mt = r >> 5, i = r & 31;
        a_frag[((((size_t)t * (HW >> 3) + c) * MT + mt) * 64 + (i + 32 * h)) * 4 + j] = v;
    }
}

extern "C" int bh_dense_pack(const void* A_vis_c64, int T, int F, int HW, float* a_frag,
                             float* a_rows, void* stream) {
    STEP_REQUIRE(A_vis_c64 && a_frag && a_rows, "bh_dense_pack: null pointer");
    STEP_REQUIRE(T > 0 && F > 0 && HW > 0 && HW % 8 == 0, "bh_dense_pack: need HW %% 8 == 0 (HW=%d)", HW);
    const int RP = rows_padded(F);
    hipLaunchKernelGGL(k_dense_pack, dim3(4096), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)A_vis_c64, T, F, HW, RP, a_frag, a_rows);
    STEP_CHECK_LAUNCH("bh_dense_pack");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- forward
// grid (KS, T, BG); 256 threads = 4 waves that interleave the 8-pixel chunks of the slice.
// a_frag holds, per (frame, chunk, row tile), one float4 per lane in MFMA operand order:
// lane (i = row, h) carries pixels 8c+4h+j, j = 0..3, so a wave's load is one contiguous KiB.
// The x operand lane (i = chain, h) reads the matching 16 bytes of its chain's frame.
template <int MT, int NT>
__global__ __launch_bounds__(256) void k_dft_fwd_dense(
    const float* __restrict__ a_frag, const float* __restrict__ x, float* __restrict__ part,
    int T, int HW, int B, int RP, int Bp, int chunks_per_slice, float scale, float shift,
    float clip_min) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [MT*NT*16][64]
    const int ks = blockIdx.x, t = blockIdx.y, bg = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int nchunk = HW >> 3;
    const int c0 = ks * chunks_per_slice;
    const int c1 = min(c0 + chunks_per_slice, nchunk);

    // acc = running total; sub = the current short MFMA chain (FLUSH chunks = 128 pixels),
    // folded into acc when it ends: keeps the rounding of the 65536-term sum near that of
    // torch's pairwise CPU sum instead of one long fma chain.
    constexpr int FLUSH = 16;
    f32x16 acc[MT][NT], sub[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = sub[m][n][r] = 0.f;

    const float* xrow[NT];
    bool xok[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int b = bg * 32 * NT + n * 32 + li;
        xok[n] = b < B;
        xrow[n] = x + ((size_t)(xok[n] ? b : 0) * T + t) * HW + 4 * h;
    }
    const float4* af = (const float4*)a_frag + ((size_t)t * nchunk * MT) * 64 + lane;

    int since = 0;
#pragma unroll 2
    for (int c = c0 + wave; c < c1; c += 4) {
        float4 a4[MT], x4[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a4[m] = af[((size_t)c * MT + m) * 64];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float4 v = *(const float4*)(xrow[n] + (size_t)c * 8);
            v.x = prep_pixel(v.x, scale, shift, clip_min);
            v.y = prep_pixel(v.y, scale, shift, clip_min);
            v.z = prep_pixel(v.z, scale, shift, clip_min);
            v.w = prep_pixel(v.w, scale, shift, clip_min);
            if (!xok[n]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            x4[n] = v;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                sub[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].x, x4[n].x, sub[m][n], 0, 0, 0);
                sub[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].y, x4[n].y, sub[m][n], 0, 0, 0);
                sub[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].z, x4[n].z, sub[m][n], 0, 0, 0);
                sub[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m].w, x4[n].w, sub[m][n], 0, 0, 0);
            }
        if (++since == FLUSH) {
            since = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc[m][n][r] += sub[m][n][r];
                        sub[m][n][r] = 0.f;
                    }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] += sub[m][n][r];

    // fixed-order sum of the four waves' partial tiles through LDS (deterministic)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* p = &red[((m * NT + n) * 16 + r) * 64 + lane];
                        *p = (w == 0) ? acc[m][n][r] : (*p + acc[m][n][r]);
                    }
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m * 32 + mfma32_row(r, h);
                    const int col = bg * 32 * NT + n * 32 + li;
                    part[(((size_t)t * gridDim.x + ks) * RP + row) * Bp + col] =
                        red[((m * NT + n) * 16 + r) * 64 + lane];
                }
    }
}

// vis/flux from the K-slice partials: one thread per (t, r <= 2F, b)
__global__ void k_dft_fwd_dense_finish(const float* __restrict__ part, int T, int F, int RP, int KS,
                                       int B, int Bp, float* __restrict__ vis, float* __restrict__ flux) {
    const int nr = 2 * F + 1;
    const size_t total = (size_t)T * nr * B;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e % B);
        const int r = (int)((e / B) % nr);
        const int t = (int)(e / ((size_t)B * nr));
        float s = 0.f;
        for (int k = 0; k < KS; ++k) s += part[(((size_t)t * KS + k) * RP + r) * Bp + b];
        if (r < F) vis[((size_t)b * T * F + (size_t)t * F + r) * 2] = s;
        else if (r < 2 * F) vis[((size_t)b * T * F + (size_t)t * F + (r - F)) * 2 + 1] = s;
        else flux[(size_t)b * T + t] = s;
    }
}

extern "C" int bh_dft_fwd_dense(const float* a_frag, int T, int F, int HW, const float* x, int B,
                                float scale, float shift, float clip_min, float* vis_c64,
                                float* flux, float* workspace, void* stream) {
    STEP_REQUIRE(a_frag && x && vis_c64 && flux && workspace, "bh_dft_fwd_dense: null pointer");
    STEP_REQUIRE(T > 0 && F > 0 && B > 0, "bh_dft_fwd_dense: bad sizes T=%d F=%d B=%d", T, F, B);
    STEP_REQUIRE(HW > 0 && HW % 8 == 0, "bh_dft_fwd_dense: need HW %% 8 == 0 (HW=%d)", HW);
    const int RP = rows_padded(F), MT = RP / 32;
    STEP_REQUIRE(MT <= 2, "bh_dft_fwd_dense: F=%d needs %d row tiles (max 2: F <= 31)", F, MT);
    const int NT = fwd_nt(B), Bp = fwd_bpad(B), KS = fwd_ksplit(HW);
    const int nchunk = HW / 8, cps = (nchunk + KS - 1) / KS;
    dim3 grid(KS, T, Bp / (32 * NT)), block(256);
    const size_t lds = (size_t)MT * NT * 16 * 64 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_FWD(M_, N_)                                                                      \
    hipLaunchKernelGGL((k_dft_fwd_dense<M_, N_>), grid, block, lds, s, a_frag, x, workspace, T, \
                       HW, B, RP, Bp, cps, scale, shift, clip_min)
    if (MT == 1 && NT == 1) LAUNCH_FWD(1, 1);
    else if (MT == 1) LAUNCH_FWD(1, 2);
    else if (NT == 1) LAUNCH_FWD(2, 1);
    else LAUNCH_FWD(2, 2);
#undef LAUNCH_FWD
    STEP_CHECK_LAUNCH("bh_dft_fwd_dense");
    const size_t total = (size_t)T * (2 * F + 1) * B;
    const int fb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_dft_fwd_dense_finish, dim3(fb), dim3(256), 0, s, workspace, T, F, RP, KS,
                       B, Bp, vis_c64, flux);
    STEP_CHECK_LAUNCH("bh_dft_fwd_dense_finish");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- adjoint
// grid (pixel groups, T, chain groups of 32*MB); each wave owns TPW pixel tiles of 32.
// A operand = g (LDS, [k][chain]), B operand = a_rows[t][k][n0 + lane] (coalesced dwords),
// accumulator rows = chains, cols = pixels -> 128-byte row stores.
#define ADJ_TPW 8
template <int MB>
__global__ __launch_bounds__(256) void k_dft_adj_dense(
    const float* __restrict__ a_rows, const float* __restrict__ gvis, const float* __restrict__ gflux,
    const float* __restrict__ chain_scale, float* __restrict__ grad, int T, int F, int HW, int RP,
    int B, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) float gT[];    // [KP][32*MB]
    const int t = blockIdx.y, b0 = blockIdx.z * 32 * MB;
    const int KP = (2 * F + 2) & ~1;                              // even number of k rows
    const int NB = 32 * MB;
    for (int e = threadIdx.x; e < KP * NB; e += blockDim.x) {
        const int k = e / NB, bl = e % NB, b = b0 + bl;
        float v = 0.f;
        if (b < B) {
            const float cs = out_scale * (chain_scale ? chain_scale[b] : 1.f);
            if (k < F) v = cs * gvis[((size_t)b * T * F + (size_t)t * F + k) * 2];
            else if (k < 2 * F) v = cs * gvis[((size_t)b * T * F + (size_t)t * F + (k - F)) * 2 + 1];
            else if (k == 2 * F) v = cs * gflux[(size_t)b * T + t];
        }
        gT[e] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int ntile = HW >> 5;
    const float* arow = a_rows + (size_t)t * RP * HW;
    for (int i = 0; i < ADJ_TPW; ++i) {
        const int tile = (blockIdx.x * 4 + wave) * ADJ_TPW + i;
        if (tile >= ntile) break;
        const int n = tile * 32 + li;
        f32x16 acc[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll 4
        for (int k = 0; k < KP; k += 2) {
            const float bv = arow[(size_t)(k + h) * HW + n];
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(gT[(k + h) * NB + m * 32 + li], bv,
                                                              acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = b0 + m * 32 + mfma32_row(r, h);
                if (b < B) grad[((size_t)b * T + t) * HW + n] = acc[m][r];
            }
    }
}

extern "C" int bh_dft_adj_dense(const float* a_rows, int T, int F, int HW, const float* gvis_c64,
                                const float* gflux, const float* chain_scale, int B,
                                float out_scale, float* grad, void* stream) {
    STEP_REQUIRE(a_rows && gvis_c64 && gflux && grad, "bh_dft_adj_dense: null pointer");
    STEP_REQUIRE(T > 0 && F > 0 && B > 0, "bh_dft_adj_dense: bad sizes T=%d F=%d B=%d", T, F, B);
    STEP_REQUIRE(HW > 0 && HW % 32 == 0, "bh_dft_adj_dense: need HW %% 32 == 0 (HW=%d)", HW);
    const int RP = rows_padded(F);
    STEP_REQUIRE(RP >= ((2 * F + 2) & ~1), "bh_dft_adj_dense: internal row padding");
    const int MB = B > 32 ? 2 : 1;
    const int ntile = HW / 32, per_wg = 4 * ADJ_TPW;
    dim3 grid((ntile + per_wg - 1) / per_wg, T, (B + 32 * MB - 1) / (32 * MB)), block(256);
    const size_t lds = (size_t)((2 * F + 2) & ~1) * 32 * MB * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (MB == 1)
        hipLaunchKernelGGL((k_dft_adj_dense<1>), grid, block, lds, s, a_rows, gvis_c64, gflux,
                           chain_scale, grad, T, F, HW, RP, B, out_scale);
    else
        hipLaunchKernelGGL((k_dft_adj_dense<2>), grid, block, lds, s, a_rows, gvis_c64, gflux,
                           chain_scale, grad, T, F, HW, RP, B, out_scale);
    STEP_CHECK_LAUNCH("bh_dft_adj_dense");
    return STEP_OK;
}
