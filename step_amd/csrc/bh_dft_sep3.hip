// Separable black-hole DFT with fp32 operands split into three bf16 planes ("3xbf16").
//
// Same contraction as bh_dft_sep.hip (ehtim ftmatrix is rank-1 per baseline, blackhole.py:226,
// forward blackhole.py:315-321), but the matrix cores run v_mfma_f32_32x32x16_bf16 (16x the
// fp32 MFMA rate) on an EXACT decomposition of every fp32 operand
//     a = a_hi + a_mid + a_lo        (8 + 8 + 8 mantissa bits, by truncation, all three bf16)
// and accumulate the six products whose weight is >= 2^-16 of the leading one
//     a x  ~=  a_hi x_hi + (a_hi x_mid + a_mid x_hi) + (a_hi x_lo + a_mid x_mid + a_lo x_hi)
// in fp32.  Every bf16 x bf16 product is exact in fp32; the dropped terms are <= 3 * 2^-24 of
// |a x|, i.e. the size of ONE fp32 rounding of the product, and ten times below the rounding
// noise of any fp32 sum over 65536 pixels (measured: oracle/README note in DESIGN.md section 4).
// Six bf16 MFMAs cost 6/16 of the fp32 MFMA they replace, which moves both directions off the
// fp32-matrix roofline onto the HBM roofline (one read of x / one write of grad).
//
// forward: persistent workgroups of 16 waves.  A workgroup owns frame t for `cpw` chains: it
//   splits the frame's ey table into bf16 planes in LDS ONCE (96 KB at H = 256) and then walks
//   its chains two at a time (waves 0-7: chain b, waves 8-15: chain b + 1; one 32-column strip
//   per wave).  The leading products go to `main`, the five corrections to `small`: the
//   corrections are 2^-8 of the leading term, so their rounding never matters, and `main`
//   sees only H/16 additions per output -- the short-chain discipline of the fp32 kernel
//   without its extra adds.
// adjoint: one workgroup (8 waves) per (chain, frame, half of the rows): G[k][y] (g folded into
//   conj(ey)) is built straight into bf16 planes in LDS, the ex rows of the strip are split
//   once into registers, and each 32x32 output tile takes <= 24 bf16 MFMAs before its 16 row
//   stores.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define S3_ADJ_THREADS 512

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// v[0..7] -> three packed bf16x8 planes with v[j] == hi[j] + mid[j] + lo[j] exactly.
__device__ __forceinline__ void split3(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[2 * j], b = v[2 * j + 1];
        const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        hi[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);          // top halves of (a, b)
        const float ra = a - __uint_as_float(ua & 0xffff0000u);      // exact
        const float rb = b - __uint_as_float(ub & 0xffff0000u);
        const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
        mid[j] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
        const float la = ra - __uint_as_float(va & 0xffff0000u);     // <= 8 significant bits
        const float lb = rb - __uint_as_float(vb & 0xffff0000u);
        lo[j] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
    }
}

#ifndef EXP3_NOMFMA
#define S3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0)
#else   // timing experiment: no matrix work, operands still consumed
__device__ __forceinline__ f32x16 s3_fake(u32x4 a, u32x4 b, f32x16 c) {
    c[0] += __uint_as_float((a[0] ^ b[0]) + (a[1] ^ b[1]) + (a[2] ^ b[2]) + (a[3] ^ b[3]));
    return c;
}
#define S3_MFMA(a, b, c) s3_fake(a, b, c)
#endif
#ifndef EXP3_NOSPLIT
#define S3_SPLIT(v, a, b, c) split3(v, a, b, c)
#else   // timing experiment: no operand split (wrong numbers)
#define S3_SPLIT(v, a, b, c)                                                              \
    {                                                                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                \
            a[j_] = __float_as_uint(v[2 * j_]); b[j_] = __float_as_uint(v[2 * j_ + 1]);   \
            c[j_] = a[j_];                                                                \
        }                                                                                 \
    }
#endif

// sum over the 16 lanes of a DPP row; lane 0 of the row adds in a fixed order
__device__ __forceinline__ float row16_sum(float v) {
#define S3_DPP(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false))
    v += S3_DPP(v, 0x128);   // row_ror:8
    v += S3_DPP(v, 0x124);   // row_ror:4
    v += S3_DPP(v, 0x122);   // row_ror:2
    v += S3_DPP(v, 0x121);   // row_ror:1
#undef S3_DPP
    return v;
}

// ------------------------------------------------------------------------------- forward
// LDS: ey planes, 16-byte units [(kb*3 + plane)*2 + h][64 rows] (kb = 16-row block of y, rows =
// Re f | Im f): the A fragment of lane (row, h) for k = 16 kb + 8 h + 0..7 is one ds_read_b128,
// consecutive lanes read consecutive units.  Then sred [parity][chain slot][2 * strips][64].
// NFR = chains in flight per workgroup (8 waves each); R = ring of 16-row blocks per wave
// (R - 1 blocks of 8 dwords per lane are in flight while one is consumed; the ring runs on
// across chains, so the stream never drains inside a workgroup).
#ifndef S3_NFR
#define S3_NFR 1
#endif
#ifndef S3_RING
#define S3_RING 4
#endif
template <int H, int W, int NFR, int R, bool CLIP>
__global__ __launch_bounds__(512 * NFR, 1) void k_dft_fwd_s3(
    const float* __restrict__ ey_t, const float* __restrict__ ex_ri, const float* __restrict__ x,
    float* __restrict__ vis, float* __restrict__ flux, int T, int F, int B, int cpw, float scale,
    float shift, float clip_min) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    constexpr int NKB = H / 16, NSTRIP = W / 32, NPART = 2 * NSTRIP, NTHR = 512 * NFR;
    static_assert(NKB % R == 0, "ring must divide the block count");
    u32x4* eyp = lds4;
    float* sred = (float*)(lds4 + NKB * 3 * 2 * 64);
    const int t = blockIdx.y, b0 = blockIdx.x * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = wave >> 3, strip = wave & 7;
    const int li = lane & 31, h = lane >> 5;

    for (int u = tid; u < (H / 8) * 64; u += NTHR) {
        const int row = u & 63, yg = u >> 6;                 // yg = 2 kb + h
        const float* src = ey_t + ((size_t)t * H + yg * 8) * 64 + row;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * 64];
        u32x4 p0, p1, p2;
        split3(v, p0, p1, p2);
        const int kb = yg >> 1, hh = yg & 1;
        eyp[((kb * 3 + 0) * 2 + hh) * 64 + row] = p0;
        eyp[((kb * 3 + 1) * 2 + hh) * 64 + row] = p1;
        eyp[((kb * 3 + 2) * 2 + hh) * 64 + row] = p2;
    }
    __syncthreads();

    const bool mine = strip < NSTRIP;                        // W < 256: the spare waves idle
    const int xcol = (mine ? strip : 0) * 32 + li;
    const u32x4* ap = eyp + h * 64 + li;
    const float* exr = ex_ri + ((size_t)t * 2 + 0) * 32 * W + xcol;
    const float* exi = ex_ri + ((size_t)t * 2 + 1) * 32 * W + xcol;
    const int n_it = cpw / NFR;
    const size_t frame = (size_t)H * W;
    auto frame_ptr = [&](int it) {
        const int b = min(b0 + NFR * it + fr, B - 1);        // past-the-end slots redo chain B-1
        return x + ((size_t)b * T + t) * frame + 8 * h * W + xcol;
    };

    f32x16 mn0, mn1, sm0, sm1;
    float xq[R][8];
#define S3_LOAD(dst, p, kb)                                                         \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) dst[j] = (p)[((kb) * 16 + j) * W];
#define S3_COMPUTE(src, kb)                                                         \
    {                                                                               \
        float pv[8];                                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                             \
            pv[j] = fmaf(src[j], scale, shift);                                     \
            if (CLIP) pv[j] = pv[j] < clip_min ? clip_min : pv[j];                  \
        }                                                                           \
        u32x4 xh, xm, xl;                                                           \
        S3_SPLIT(pv, xh, xm, xl);                                                   \
        const u32x4* a = ap + (kb) * (3 * 2 * 64);                                  \
        {                                                                           \
            const u32x4 ah = a[0], am = a[128], al = a[256];                        \
            sm0 = S3_MFMA(al, xh, sm0); sm0 = S3_MFMA(ah, xl, sm0);                 \
            sm0 = S3_MFMA(am, xm, sm0); sm0 = S3_MFMA(am, xh, sm0);                 \
            sm0 = S3_MFMA(ah, xm, sm0); mn0 = S3_MFMA(ah, xh, mn0);                 \
        }                                                                           \
        {                                                                           \
            const u32x4 ah = a[32], am = a[160], al = a[288];                       \
            sm1 = S3_MFMA(al, xh, sm1); sm1 = S3_MFMA(ah, xl, sm1);                 \
            sm1 = S3_MFMA(am, xm, sm1); sm1 = S3_MFMA(am, xh, sm1);                 \
            sm1 = S3_MFMA(ah, xm, sm1); mn1 = S3_MFMA(ah, xh, mn1);                 \
        }                                                                           \
    }

    const float* xcur = frame_ptr(0);
    if (mine) {
#pragma unroll
        for (int s = 0; s < R - 1; ++s) { S3_LOAD(xq[s], xcur, s); }
    }
#pragma unroll 1
    for (int it = 0; it < n_it; ++it) {
        const float* xnext = frame_ptr(min(it + 1, n_it - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) mn0[r] = mn1[r] = sm0[r] = sm1[r] = 0.f;
        if (mine) {
#pragma unroll 1
            for (int kb = 0; kb < NKB; kb += R) {
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    // fetch the block R - 1 ahead, unconditionally (branch-free, so that the
                    // wait before each use counts loads instead of draining them); past the
                    // last chain xnext == xcur and the extra blocks are simply not used
                    const int nb = kb + s + R - 1;
                    const float* pn = nb < NKB ? xcur + nb * 16 * W : xnext + (nb - NKB) * 16 * W;
                    S3_LOAD(xq[(s + R - 1) % R], pn, 0);
                    __builtin_amdgcn_sched_barrier(0);       // keep the loads ahead of the MFMAs
                    S3_COMPUTE(xq[s], kb + s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // p[f] = sum_x ex[f][x] Z[f][x] over the strip: complex multiply in the accumulator
            // layout, DPP sum over each 16-lane row, lanes 0/16/32/48 hand their halves to sred
            float* sp = sred + (((it & 1) * NFR + fr) * NPART + strip * 2 + ((lane >> 4) & 1)) * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = mfma32_row(r, h);
                const float er = exr[f * W], ei = exi[f * W];
                const float zr = mn0[r] + sm0[r], zi = mn1[r] + sm1[r];
                const float pr = row16_sum(er * zr - ei * zi);
                const float pi = row16_sum(er * zi + ei * zr);
                if ((lane & 15) == 0) { sp[f] = pr; sp[32 + f] = pi; }
            }
        }
        __syncthreads();
        if (strip == 0) {                                    // first wave of the chain: fixed-order sum
            const float* sp = sred + ((it & 1) * NFR + fr) * NPART * 64 + lane;
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < NPART; ++p) s += sp[p * 64];
            const int b = b0 + NFR * it + fr, c = lane >> 5, f = lane & 31;
            if (b < B) {
                if (f < F) vis[(((size_t)b * T + t) * F + f) * 2 + c] = s;
                else if (f == F && c == 0) flux[(size_t)b * T + t] = s;
            }
        }
        xcur = xnext;
    }
#undef S3_LOAD
#undef S3_COMPUTE
}

static int s3_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

extern "C" int bh_sep3_supported(int H, int W) {
    return (H == 256 && W == 256) || (H == 128 && W == 128) || (H == 64 && W == 64) || (H == 32 && W == 32);
}

template <int H, int W>
static void launch_fwd_s3(const float* ey_t, const float* ex_ri, int T, int F, const float* x, int B,
                          float scale, float shift, float clip_min, float* vis, float* flux,
                          hipStream_t st) {
    constexpr int NFR = S3_NFR, R = (H / 16 < S3_RING) ? H / 16 : S3_RING;
    // chains per workgroup: one resident workgroup per CU, every CU busy, no second round
    const int groups = (B + NFR - 1) / NFR;
    int per_t = s3_cu_count() / T;
    per_t = per_t < 1 ? 1 : (per_t > groups ? groups : per_t);
    const int cpw = NFR * ((groups + per_t - 1) / per_t);
    const int gx = (B + cpw - 1) / cpw;
    const size_t lds = (size_t)(H / 16) * 3 * 2 * 64 * 16 + (size_t)2 * NFR * 2 * (W / 32) * 64 * 4;
    if (clip_min == -INFINITY) {
        step_allow_lds((const void*)k_dft_fwd_s3<H, W, NFR, R, false>, lds);
        hipLaunchKernelGGL((k_dft_fwd_s3<H, W, NFR, R, false>), dim3(gx, T), dim3(512 * NFR), lds, st,
                           ey_t, ex_ri, x, vis, flux, T, F, B, cpw, scale, shift, clip_min);
    } else {
        step_allow_lds((const void*)k_dft_fwd_s3<H, W, NFR, R, true>, lds);
        hipLaunchKernelGGL((k_dft_fwd_s3<H, W, NFR, R, true>), dim3(gx, T), dim3(512 * NFR), lds, st,
                           ey_t, ex_ri, x, vis, flux, T, F, B, cpw, scale, shift, clip_min);
    }
}

extern "C" int bh_dft_fwd_sep3(const float* ey_t, const float* ex_ri, int T, int F, int H, int W,
                               const float* x, int B, float scale, float shift, float clip_min,
                               float* vis_c64, float* flux, void* stream) {
    STEP_REQUIRE(ey_t && ex_ri && x && vis_c64 && flux, "bh_dft_fwd_sep3: null pointer");
    STEP_REQUIRE(T > 0 && T <= 65535 && B > 0 && F > 0 && F <= 31,
                 "bh_dft_fwd_sep3: need 1 <= F <= 31, 1 <= T <= 65535 (F=%d T=%d)", F, T);
    STEP_REQUIRE(bh_sep3_supported(H, W), "bh_dft_fwd_sep3: image %dx%d not built (256, 128, 64, 32 square)", H, W);
    hipStream_t st = (hipStream_t)stream;
    if (H == 256) launch_fwd_s3<256, 256>(ey_t, ex_ri, T, F, x, B, scale, shift, clip_min, vis_c64, flux, st);
    else if (H == 128) launch_fwd_s3<128, 128>(ey_t, ex_ri, T, F, x, B, scale, shift, clip_min, vis_c64, flux, st);
    else if (H == 64) launch_fwd_s3<64, 64>(ey_t, ex_ri, T, F, x, B, scale, shift, clip_min, vis_c64, flux, st);
    else launch_fwd_s3<32, 32>(ey_t, ex_ri, T, F, x, B, scale, shift, clip_min, vis_c64, flux, st);
    STEP_CHECK_LAUNCH("bh_dft_fwd_sep3");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- adjoint
// LDS: G planes, 16-byte units [(kb*3 + plane)*2 + h][HS rows y] (kb = 16-row block of k),
// then g2 [2F + 2].  k rows: Re f | Im f | flux | zero pad, as in bh_dft_sep.hip.
template <int H, int W, int YS>
__global__ __launch_bounds__(S3_ADJ_THREADS, 2) void k_dft_adj_s3(
    const float* __restrict__ ey_ri, const float* __restrict__ ex_ri,
    const float* __restrict__ gvis, const float* __restrict__ gflux,
    const float* __restrict__ chain_scale, float* __restrict__ grad, int T, int F, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    constexpr int HS = H / YS, NSTRIP = W / 32, NTILE = HS / 32;
    const int nkb = (2 * F + 1 + 15) >> 4;                   // <= 4
    u32x4* gp = lds4;
    float* g2 = (float*)(lds4 + 4 * 3 * 2 * HS);
    const int b = blockIdx.x, t = blockIdx.y, y0 = blockIdx.z * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const float cs = out_scale * (chain_scale ? chain_scale[b] : 1.f);

    for (int e = tid; e < 2 * F + 2; e += S3_ADJ_THREADS) {
        float v = 0.f;
        if (e < F) v = gvis[(((size_t)b * T + t) * F + e) * 2];
        else if (e < 2 * F) v = gvis[(((size_t)b * T + t) * F + (e - F)) * 2 + 1];
        else if (e == 2 * F) v = gflux[(size_t)b * T + t];
        g2[e] = cs * v;
    }
    __syncthreads();
    // G[k][y]: k < F: Re(conj(ey) g);  F <= k < 2F: Im(conj(ey) g);  k = 2F: gflux;  else 0
    const float* eyr = ey_ri + ((size_t)t * 2 + 0) * 32 * H + y0;
    const float* eyi = ey_ri + ((size_t)t * 2 + 1) * 32 * H + y0;
    for (int u = tid; u < nkb * 2 * HS; u += S3_ADJ_THREADS) {
        const int y = u % HS, kg = u / HS;                   // kg = 2 kb + h
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kg * 8 + j;
            const bool im = k >= F;
            const int f = min(im ? k - F : k, F);            // f = F: the flux row (ey = 1 + 0i)
            const float a = eyr[f * H + y], c = eyi[f * H + y];
            const float gr = g2[min(f, F - 1)], gi = g2[F + min(f, F - 1)];
            float w = im ? a * gi - c * gr : a * gr + c * gi;
            if (k >= 2 * F) w = (k == 2 * F) ? g2[2 * F] : 0.f;
            v[j] = w;
        }
        u32x4 p0, p1, p2;
        split3(v, p0, p1, p2);
        const int kb = kg >> 1, hh = kg & 1;
        gp[((kb * 3 + 0) * 2 + hh) * HS + y] = p0;
        gp[((kb * 3 + 1) * 2 + hh) * HS + y] = p1;
        gp[((kb * 3 + 2) * 2 + hh) * HS + y] = p2;
    }
    __syncthreads();

    const float* exr = ex_ri + ((size_t)t * 2 + 0) * 32 * W;
    const float* exi = ex_ri + ((size_t)t * 2 + 1) * 32 * W;
    float* gf = grad + ((size_t)b * T + t) * H * W + (size_t)y0 * W;
    for (int strip = wave; strip < NSTRIP; strip += S3_ADJ_THREADS / 64) {
        const int xcol = strip * 32 + li;
        u32x4 bh[4], bm[4], bl[4];                           // EXK[k][x]: Re ex | Im ex | 1 | 0
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kb * 16 + 8 * h + j;
                const bool im = k >= F;
                const int f = min(im ? k - F : k, F);        // row F of ex_ri is 1 + 0i
                const float* tab = (im && k < 2 * F) ? exi : exr;
                const float w = tab[f * W + xcol];
                v[j] = (k <= 2 * F) ? w : 0.f;
            }
            split3(v, bh[kb], bm[kb], bl[kb]);
        }
        const u32x4* ap = gp + h * HS + li;
#pragma unroll 1
        for (int tile = 0; tile < NTILE; ++tile) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                if (kb < nkb) {
                    const u32x4* a = ap + kb * (3 * 2 * HS) + tile * 32;
                    const u32x4 ah = a[0], am = a[2 * HS], al = a[4 * HS];
                    acc = S3_MFMA(al, bh[kb], acc);
                    acc = S3_MFMA(ah, bl[kb], acc);
                    acc = S3_MFMA(am, bm[kb], acc);
                    acc = S3_MFMA(am, bh[kb], acc);
                    acc = S3_MFMA(ah, bm[kb], acc);
                    acc = S3_MFMA(ah, bh[kb], acc);
                }
            const int go = (tile * 32 + 4 * h) * W + xcol;
#pragma unroll
            for (int r = 0; r < 16; ++r) gf[go + ((r & 3) + 8 * (r >> 2)) * W] = acc[r];
        }
    }
}

template <int H, int W, int YS>
static void launch_adj_s3(const float* ey_ri, const float* ex_ri, int T, int F, const float* gvis,
                          const float* gflux, const float* chain_scale, int B, float out_scale,
                          float* grad, hipStream_t st) {
    const size_t lds = (size_t)4 * 3 * 2 * (H / YS) * 16 + (size_t)(2 * F + 2) * 4;
    step_allow_lds((const void*)k_dft_adj_s3<H, W, YS>, lds);
    hipLaunchKernelGGL((k_dft_adj_s3<H, W, YS>), dim3(B, T, YS), dim3(S3_ADJ_THREADS), lds, st, ey_ri,
                       ex_ri, gvis, gflux, chain_scale, grad, T, F, out_scale);
}

extern "C" int bh_dft_adj_sep3(const float* ey_ri, const float* ex_ri, int T, int F, int H, int W,
                               const float* gvis_c64, const float* gflux, const float* chain_scale,
                               int B, float out_scale, float* grad, void* stream) {
    STEP_REQUIRE(ey_ri && ex_ri && gvis_c64 && gflux && grad, "bh_dft_adj_sep3: null pointer");
    STEP_REQUIRE(T > 0 && T <= 65535 && B > 0 && F > 0 && F <= 31,
                 "bh_dft_adj_sep3: need 1 <= F <= 31, 1 <= T <= 65535 (F=%d T=%d)", F, T);
    STEP_REQUIRE(bh_sep3_supported(H, W), "bh_dft_adj_sep3: image %dx%d not built (256, 128, 64, 32 square)", H, W);
    hipStream_t st = (hipStream_t)stream;
    if (H == 256) launch_adj_s3<256, 256, 2>(ey_ri, ex_ri, T, F, gvis_c64, gflux, chain_scale, B, out_scale, grad, st);
    else if (H == 128) launch_adj_s3<128, 128, 1>(ey_ri, ex_ri, T, F, gvis_c64, gflux, chain_scale, B, out_scale, grad, st);
    else if (H == 64) launch_adj_s3<64, 64, 1>(ey_ri, ex_ri, T, F, gvis_c64, gflux, chain_scale, B, out_scale, grad, st);
    else launch_adj_s3<32, 32, 1>(ey_ri, ex_ri, T, F, gvis_c64, gflux, chain_scale, B, out_scale, grad, st);
    STEP_CHECK_LAUNCH("bh_dft_adj_sep3");
    return STEP_OK;
}
