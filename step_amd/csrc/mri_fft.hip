// Dynamic-MRI operator fused per frame: complex pack -> fftshift -> 2-D FFT -> k-space row
// mask (forward_operator/dynamic_mri.py:42-48), and for the Langevin step the residual,
// the per-chain loss of base.py:40-41 and the adjoint (SURVEY Appendix B "MRI") in the same
// launch, so a step moves 24*R*R bytes per frame: read x, read y, write grad.
//
// The Langevin step of the shipped masks runs k_mri_loss_grad2, whose body is mri_lg2.h; this file
// holds the first-generation kernels -- forward only, adjoint only, and the loss+gradient kernel
// that takes masks with more kept rows than one row buffer -- plus the C ABI.  Their scheme:
// one workgroup (512 threads) per frame, 192 = 16 x 12 in-register FFT passes over LDS:
//   1. per 32-column strip: load x, FFT along y (all 192 rows), keep only the masked k_y rows
//      (52 of 192 at acc 6 / 24 ACS lines) in an LDS row buffer  -> 80 KB instead of 295 KB
//   2. FFT along x on the kept rows                               = M * FFT2(x)
//   3. residual against y, loss, scale by 2                      (kept rows only carry gradient)
//   4. inverse FFT along x on the kept rows
//   5. per strip: scatter the kept rows into a zero strip, inverse FFT along y, store grad
// fftshift before the FFT is a roll by R/2 in both axes, i.e. a sign (-1)^(ky+kx) on the
// spectrum; the ifftshift of the adjoint is the same sign on its input.  No data is moved.
// Masks with more kept rows than the LDS row buffer holds are processed in groups (the
// gradient then accumulates over groups).
#include "common.h"
#include "fft_small.h"
#include "mri_lg2.h"
#include "mri_lg2p.h"
#include "mri_lg2s.h"
#include "mri_lg2mc.h"

#define MRI_R 192
#define MRI_THREADS 512
#define MRI_SW 32                 // strip width (columns)
#define MRI_RS 204                // row-buffer stride in complex elements (conflict-free pass A)
#define MRI_NKC_MAX 64

struct MriLds {
    cf32* tw;      // [192]
    cf32* S;       // [192][32]
    cf32* Rk;      // [nkc][204]
    int* rowpos;   // [192]  position of k_y row in the kept-row list, -1 if masked out
    int* rowl;     // [192]  the kept-row list (copy of the global table)
    int* unrow;    // [192]  the masked-out rows, ascending
    float* red;    // [16]
};

__device__ __forceinline__ MriLds carve(float* lds, int nkc) {
    MriLds L;
    L.tw = (cf32*)lds;
    L.S = L.tw + MRI_R;
    L.Rk = L.S + MRI_R * MRI_SW;
    L.rowpos = (int*)(L.Rk + (size_t)nkc * MRI_RS);
    L.rowl = L.rowpos + MRI_R;
    L.unrow = L.rowl + MRI_R;
    L.red = (float*)(L.unrow + MRI_R);
    return L;
}
static size_t mri_lds_bytes(int nkc) {
    return (size_t)(MRI_R + MRI_R * MRI_SW + (size_t)nkc * MRI_RS) * sizeof(cf32) +
           3 * MRI_R * sizeof(int) + 16 * sizeof(float);
}

__device__ __forceinline__ float sgn(int k) { return (k & 1) ? -1.f : 1.f; }

// Forward FFT along y of the strip S[192][32] (all 32 columns).  Pass A runs in place; pass B
// keeps its 12 outputs in registers and writes only the kept k_y rows of the current row group,
// sign-folded, straight into the row buffer: Rk[j][strip*32+col] = (-1)^ky X[ky][col].
__device__ __forceinline__ void strip_fft_fwd(const MriLds& L, int strip, int g0, int nk) {
    const int tid = threadIdx.x, col = tid & 31;
    if (tid < 12 * 32) {
        const int bb = tid >> 5;
        cf32 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = L.S[(12 * a + bb) * MRI_SW + col];
        fft192_pass_a<-1>(v, bb, L.tw);
#pragma unroll
        for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * MRI_SW + col] = v[c];
    }
    __syncthreads();
    {
        const int c = tid >> 5;                       // 16 x 32 tasks = 512 threads
        cf32 v[12];
#pragma unroll
        for (int b = 0; b < 12; ++b) v[b] = L.S[(12 * c + b) * MRI_SW + col];
        fft192_pass_b<-1>(v);
#pragma unroll
        for (int d = 0; d < 12; ++d) {
            const int ky = c + 16 * d;
            const int j = L.rowpos[ky] - g0;
            if (j >= 0 && j < nk) {
                const float sg = sgn(ky);
                L.Rk[(size_t)j * MRI_RS + strip * MRI_SW + col] = c_make(sg * v[d].x, sg * v[d].y);
            }
        }
    }
    __syncthreads();                                  // S may be refilled, Rk column block complete
}

// Inverse FFT along y for one strip: pass A gathers its inputs from the kept rows of the row
// buffer (all other k_y rows are zero), pass B stores image rows straight from registers.
template <bool MC = false>
__device__ __forceinline__ void strip_fft_inv(const MriLds& L, int strip, int g0, int nk,
                                              float* __restrict__ dr, float* __restrict__ di,
                                              bool accumulate, const float* __restrict__ sr = nullptr,
                                              const float* __restrict__ si = nullptr) {
    const int tid = threadIdx.x, col = tid & 31;
    if (tid < 12 * 32) {
        const int bb = tid >> 5;
        cf32 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const int j = L.rowpos[12 * a + bb] - g0;
            v[a] = (j >= 0 && j < nk) ? L.Rk[(size_t)j * MRI_RS + strip * MRI_SW + col] : c_make(0.f, 0.f);
        }
        fft192_pass_a<1>(v, bb, L.tw);
#pragma unroll
        for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * MRI_SW + col] = v[c];
    }
    __syncthreads();
    {
        const int c = tid >> 5;
        cf32 v[12];
#pragma unroll
        for (int b = 0; b < 12; ++b) v[b] = L.S[(12 * c + b) * MRI_SW + col];
        fft192_pass_b<1>(v);
#pragma unroll
        for (int d = 0; d < 12; ++d) {
            const int g = (c + 16 * d) * MRI_R + strip * MRI_SW + col;
            float ox = v[d].x, oy = v[d].y;
            if (MC) {                                 // conj(S_c) * v: the adjoint of the coil multiply
                const float a = sr[g], b = si[g];
                ox = v[d].x * a + v[d].y * b;
                oy = v[d].y * a - v[d].x * b;
            }
            if (accumulate) { dr[g] += ox; di[g] += oy; }
            else { dr[g] = ox; di[g] = oy; }
        }
    }
    __syncthreads();                                  // S is rewritten by the next strip's pass A
}

// FFT along x of the nk kept rows Rk[nk][204], in place, natural order.
template <int SIGN>
__device__ __forceinline__ void rows_fft_x(const MriLds& L, int nk) {
    const int tid = threadIdx.x;
    for (int e = tid; e < nk * 12; e += MRI_THREADS) {
        const int j = e / 12, bb = e - 12 * j;
        cf32* row = L.Rk + (size_t)j * MRI_RS;
        cf32 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = row[12 * a + bb];
        fft192_pass_a<SIGN>(v, bb, L.tw);
#pragma unroll
        for (int c = 0; c < 16; ++c) row[12 * c + bb] = v[c];
    }
    __syncthreads();
    // nk * 16 <= 1024 tasks: at most two per thread, all reads before any write
    cf32 v0[12], v1[12];
    const int e0 = tid, e1 = tid + MRI_THREADS;
    const bool has0 = e0 < nk * 16, has1 = e1 < nk * 16;
    if (has0) {
        const cf32* row = L.Rk + (size_t)(e0 >> 4) * MRI_RS + 12 * (e0 & 15);
#pragma unroll
        for (int b = 0; b < 12; ++b) v0[b] = row[b];
        fft192_pass_b<SIGN>(v0);
    }
    if (has1) {
        const cf32* row = L.Rk + (size_t)(e1 >> 4) * MRI_RS + 12 * (e1 & 15);
#pragma unroll
        for (int b = 0; b < 12; ++b) v1[b] = row[b];
        fft192_pass_b<SIGN>(v1);
    }
    __syncthreads();
    if (has0) {
        cf32* row = L.Rk + (size_t)(e0 >> 4) * MRI_RS + (e0 & 15);
#pragma unroll
        for (int d = 0; d < 12; ++d) row[16 * d] = v0[d];
    }
    if (has1) {
        cf32* row = L.Rk + (size_t)(e1 >> 4) * MRI_RS + (e1 & 15);
#pragma unroll
        for (int d = 0; d < 12; ++d) row[16 * d] = v1[d];
    }
    __syncthreads();
}

constexpr int MRI_NP = MRI_R * MRI_SW / MRI_THREADS;      // 12 elements of a strip per thread
constexpr int MRI_YQ = 6;   // float4 loads per thread and strip for the masked-out rows (<= 192 rows)

// pixels of strip `strip` owned by this thread: element i is row (tid >> 5) + 16 i, column tid & 31
template <bool MC = false>
__device__ __forceinline__ void load_strip(const float* __restrict__ xr, const float* __restrict__ xi,
                                           int strip, cf32 (&pre)[MRI_NP],
                                           const float* __restrict__ sr = nullptr,
                                           const float* __restrict__ si = nullptr) {
    const int col = threadIdx.x & 31, row0 = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < MRI_NP; ++i) {
        const int g = (row0 + 16 * i) * MRI_R + strip * MRI_SW + col;
        if (MC) {                                     // S_c * x (coil sensitivity, complex)
            const float a = sr[g], b = si[g], p = xr[g], q = xi[g];
            pre[i] = c_make(p * a - q * b, p * b + q * a);
        } else {
            pre[i] = c_make(xr[g], xi[g]);
        }
    }
}

// Steps 1-2 for kept rows [g0, g0+nk): leaves Rk[j][kx] = (-1)^ky * FFT2(z)[ky][kx], ky = rowlist[g0+j].
// `pre` holds strip 0 on entry (loaded by the caller ahead of the table set-up).
// y != nullptr (loss kernels, first row group): the |y|^2 of the masked-out rows -- where A(x) = 0,
// so the residual is -y -- is summed along the way: each strip every thread loads its share of a
// sixth of those rows as float4 before the strip's FFT and adds it up after, so the L2 latency
// hides behind the transform.
template <bool MC = false>
__device__ __forceinline__ void forward_rows(const MriLds& L, const float* __restrict__ xr,
                                             const float* __restrict__ xi, int g0, int nk,
                                             cf32 (&pre)[MRI_NP], const float* __restrict__ y,
                                             int n_un, float& acc, const float* __restrict__ sr = nullptr,
                                             const float* __restrict__ si = nullptr) {
    const int tid = threadIdx.x;
    constexpr int NP = MRI_NP, NS = MRI_R / MRI_SW, Q = MRI_R / 4;   // Q float4 per row
    const int col = tid & 31, row0 = tid >> 5;             // element i: row = row0 + 16 i
    for (int strip = 0; strip < NS; ++strip) {
#pragma unroll
        for (int i = 0; i < NP; ++i) L.S[(row0 + 16 * i) * MRI_SW + col] = pre[i];
        __syncthreads();
        // the next strip's pixels travel to registers while the current strip is transformed
        if (strip + 1 < NS) load_strip<MC>(xr, xi, strip + 1, pre, sr, si);
        float4 yq[MRI_YQ];
        const int q0 = strip * n_un / NS;
        const int total = y ? ((strip + 1) * n_un / NS - q0) * (2 * Q) : 0;   // both planes
        if (total > 0) {
#pragma unroll
            for (int i = 0; i < MRI_YQ; ++i) {
                const int e = min(tid + i * MRI_THREADS, total - 1);
                const int q = e / (2 * Q), r = e - q * (2 * Q);               // r: plane * Q + c
                const int pl = r >= Q;
                yq[i] = *(const float4*)(y + (size_t)pl * MRI_R * MRI_R + L.unrow[q0 + q] * MRI_R +
                                         4 * (r - pl * Q));
            }
        }
        strip_fft_fwd(L, strip, g0, nk);
        if (total > 0) {
#pragma unroll
            for (int i = 0; i < MRI_YQ; ++i)
                if (tid + i * MRI_THREADS < total)
                    acc += yq[i].x * yq[i].x + yq[i].y * yq[i].y + yq[i].z * yq[i].z + yq[i].w * yq[i].w;
        }
    }
    rows_fft_x<-1>(L, nk);
}

__device__ __forceinline__ void load_tables(const MriLds& L, const int* __restrict__ rowlist,
                                            int nk_total) {
    for (int e = threadIdx.x; e < MRI_R; e += MRI_THREADS) {
        float sn, cn;
        sincospif((float)e * (1.0f / 96.0f), &sn, &cn);      // (cos, sin)(2 pi e / 192)
        L.tw[e] = c_make(cn, sn);
        L.rowpos[e] = -1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nk_total; e += MRI_THREADS) {
        const int r = rowlist[e];
        L.rowpos[r] = e;
        L.rowl[e] = r;
    }
    __syncthreads();
    // rank of every masked-out row among the masked-out rows: ballot + popcount within each of
    // the three waves that cover the 192 rows, wave totals through LDS (red[0..2])
    const int e = threadIdx.x;
    const bool out_row = e < MRI_R && L.rowpos[e] < 0;
    const unsigned long long m = __ballot(out_row);
    const int below = __popcll(m & ((1ull << (e & 63)) - 1ull));
    if (e < MRI_R && (e & 63) == 0) L.red[e >> 6] = (float)__popcll(m);
    __syncthreads();
    if (out_row) {
        int base = 0;
        for (int w = 0; w < (e >> 6); ++w) base += (int)L.red[w];
        L.unrow[base + below] = e;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------- forward only
__global__ __launch_bounds__(MRI_THREADS) void k_mri_forward(
    const float* __restrict__ x, const int* __restrict__ rows, const int* __restrict__ nrows,
    int mask_frames, int T, int nkc, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MriLds L = carve(lds, nkc);
    const int frame = blockIdx.x, t = frame % T, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : t;
    const int* rowlist = rows + (size_t)mf * MRI_R;
    const int nk_total = nrows[mf];
    const float* xr = x + (size_t)frame * 2 * MRI_R * MRI_R;
    const float* xi = xr + MRI_R * MRI_R;
    float* outr = out + (size_t)frame * 2 * MRI_R * MRI_R;
    float* outi = outr + MRI_R * MRI_R;
    cf32 pre[MRI_NP];
    load_strip(xr, xi, 0, pre);                        // in flight during the table set-up
    load_tables(L, rowlist, nk_total);
    for (int e = tid; e < MRI_R * MRI_R; e += MRI_THREADS)
        if (L.rowpos[e / MRI_R] < 0) { outr[e] = 0.f; outi[e] = 0.f; }
    float unused = 0.f;
    for (int g0 = 0; g0 < nk_total; g0 += nkc) {
        const int nk = min(nkc, nk_total - g0);
        if (g0) load_strip(xr, xi, 0, pre);
        forward_rows(L, xr, xi, g0, nk, pre, nullptr, 0, unused);
        for (int e = tid; e < nk * MRI_R; e += MRI_THREADS) {
            const int j = e / MRI_R, kx = e - j * MRI_R;
            const int ky = L.rowl[g0 + j];
            const cf32 v = L.Rk[(size_t)j * MRI_RS + kx];
            const float s = sgn(kx);
            outr[(size_t)ky * MRI_R + kx] = s * v.x;
            outi[(size_t)ky * MRI_R + kx] = s * v.y;
        }
        __syncthreads();
    }
}

// Steps 4-5 for kept rows [g0, g0+nk): Rk holds the (sign-folded) k-space rows; writes (g0 == 0)
// or accumulates (later groups) the image-space result into the frame `dst` ([2][R][R]).
template <bool MC = false>
__device__ __forceinline__ void adjoint_rows(const MriLds& L, const int* __restrict__ rowlist,
                                             int g0, int nk, float* __restrict__ dst, bool accumulate,
                                             const float* __restrict__ sr = nullptr,
                                             const float* __restrict__ si = nullptr) {
    rows_fft_x<1>(L, nk);
    for (int strip = 0; strip < MRI_R / MRI_SW; ++strip)
        strip_fft_inv<MC>(L, strip, g0, nk, dst, dst + MRI_R * MRI_R, accumulate, sr, si);
}

// ------------------------------------------------------------------------------- loss + gradient
__global__ __launch_bounds__(MRI_THREADS) void k_mri_loss_grad(
    const float* __restrict__ x, const float* __restrict__ y, int y_batch,
    const int* __restrict__ rows, const int* __restrict__ nrows, int mask_frames, int T, int nkc,
    const float* __restrict__ chain_scale, float* __restrict__ loss_part, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MriLds L = carve(lds, nkc);
    const int frame = blockIdx.x, b = frame / T, t = frame % T, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : t;
    const int* rowlist = rows + (size_t)mf * MRI_R;
    const int nk_total = nrows[mf];
    const size_t fsz = (size_t)2 * MRI_R * MRI_R;
    const float* xr = x + (size_t)frame * fsz;
    const float* xi = xr + MRI_R * MRI_R;
    const float* yr = y + ((size_t)(y_batch > 1 ? b : 0) * T + t) * fsz;
    const float* yi = yr + MRI_R * MRI_R;
    const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
    cf32 pre[MRI_NP];
    load_strip(xr, xi, 0, pre);                        // in flight during the table set-up
    load_tables(L, rowlist, nk_total);

    // rows outside the mask are summed inside forward_rows; a frame with no kept row does it here
    float acc = 0.f;
    if (nk_total == 0) {
        const int total = MRI_R * MRI_R;
#pragma unroll 8
        for (int e = tid; e < total; e += MRI_THREADS) {
            const int q = e / MRI_R;
            const int g = L.unrow[q] * MRI_R + (e - q * MRI_R);
            const float a = yr[g], c = yi[g];
            acc += a * a + c * c;
        }
    }

    if (nk_total == 0 && grad) {
        float* gr = grad + (size_t)frame * fsz;
        for (int e = tid; e < 2 * MRI_R * MRI_R; e += MRI_THREADS) gr[e] = 0.f;
    }
    for (int g0 = 0; g0 < nk_total; g0 += nkc) {
        const int nk = min(nkc, nk_total - g0);
        if (g0) load_strip(xr, xi, 0, pre);
        forward_rows(L, xr, xi, g0, nk, pre, g0 == 0 ? yr : nullptr, MRI_R - nk_total, acc);
        // residual, loss, adjoint input 2 * (-1)^(ky+kx) * r  (the row sign is already in Rk)
#pragma unroll 4
        for (int e = tid; e < nk * MRI_R; e += MRI_THREADS) {
            const int j = e / MRI_R, kx = e - j * MRI_R;
            const int ky = L.rowl[g0 + j];
            const float sx = sgn(kx), sy = sgn(ky);
            const float ya = yr[ky * MRI_R + kx], yb = yi[ky * MRI_R + kx];
            const cf32 v = L.Rk[(size_t)j * MRI_RS + kx];
            const float rr = sx * v.x - ya;
            const float ri = sx * v.y - yb;
            acc += rr * rr + ri * ri;
            const float s = cs2 * sx * sy;
            L.Rk[(size_t)j * MRI_RS + kx] = c_make(s * rr, s * ri);
        }
        __syncthreads();
        if (grad) adjoint_rows(L, rowlist, g0, nk, grad + (size_t)frame * fsz, g0 != 0);
    }
    const float tot = block_sum(acc, L.red);
    if (tid == 0) loss_part[frame] = tot;
}

// ------------------------------------------------------- loss + gradient, pipelined (mri_lg2.h)
struct Lg2DevCtx {
    float* lds_;
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int block() const { return blockIdx.x; }
    __device__ __forceinline__ int nblocks() const { return gridDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ float* lds() const { return lds_; }
    // phase marker of the body: a no-op here (scripts/probes/mri_timing.hip builds the body over
    // a context that records timestamps)
    __device__ __forceinline__ void tick(int, int) const {}
    // the NaN flag of the in-kernel state update: one atomic per wave that saw a NaN, none once it is up
    __device__ __forceinline__ void raise_flag(int* f) const {
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(f, 1);
    }
    __device__ __forceinline__ cf32 cos_sin_2pi_over_192(int e) const {
        float sn, cn;
        sincospif((float)e * (1.0f / 96.0f), &sn, &cn);
        return c_make(cn, sn);
    }
};

template <bool YQ>
__global__ __launch_bounds__(LG2_THREADS) void k_mri_loss_grad2(
    const float* __restrict__ x, const float* __restrict__ y, int y_batch,
    const int* __restrict__ rows, const int* __restrict__ nrows, int mask_frames, int T,
    int nframes, int nkc, const float* __restrict__ chain_scale, float* __restrict__ loss_part,
    float* __restrict__ unk_part, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Lg2DevCtx ctx{lds};
    lg2_body<YQ>(ctx, x, y, y_batch, rows, nrows, mask_frames, T, nframes, nkc, chain_scale,
                 loss_part, unk_part, grad);
}

// the pipelined kernel with the state update in place of the gradient store (Lg2Upd, mri_lg2.h)
template <int UPD>
__global__ __launch_bounds__(LG2_THREADS) void k_mri_step_upd(
    const float* x, const float* __restrict__ y, const int* __restrict__ rows,
    const int* __restrict__ nrows, int mask_frames, int T, int nframes, int nkc,
    const float* __restrict__ chain_scale, float* __restrict__ loss_part, float* __restrict__ unk_part,
    Lg2Upd up) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Lg2DevCtx ctx{lds};
    lg2_body<false, UPD>(ctx, x, y, 1, rows, nrows, mask_frames, T, nframes, nkc, chain_scale, loss_part,
                         unk_part, nullptr, up);
}

// ------------------------------------------- loss + gradient, persistent pipelined kernel (mri_lg2p.h)
__global__ __launch_bounds__(LG2_THREADS) void k_mri_loss_grad2p(
    const float* __restrict__ x, const float* __restrict__ y, const int* __restrict__ rows,
    const int* __restrict__ nrows, int mask_frames, int T, int nframes, int nkc,
    const float* __restrict__ chain_scale, float* __restrict__ loss_part,
    float* __restrict__ unk_part, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Lg2DevCtx ctx{lds};
    lg2p_body(ctx, x, y, rows, nrows, mask_frames, T, nframes, nkc, chain_scale, loss_part, unk_part, grad);
}

// ------------------------------------------- loss + gradient, several workgroups per frame (mri_lg2s.h)
template <int SP>
__global__ __launch_bounds__(LG2_THREADS) void k_mri_loss_grad2s(
    const float* __restrict__ x, const float* __restrict__ y, const int* __restrict__ rows,
    const int* __restrict__ nrows, int mask_frames, int T, int nframes, int nkc,
    const float* __restrict__ chain_scale, float* part_out, float* unk_out, float* __restrict__ grad,
    cf32* zs, unsigned* cnt, float* __restrict__ loss, float* __restrict__ total) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Lg2DevCtx ctx{lds};
    lg2s_body<SP>(ctx, x, y, rows, nrows, mask_frames, T, nframes, nkc, chain_scale, part_out, unk_out,
                  grad, zs, cnt, loss, total);
}

// ------------------------------------------- multi-coil loss + gradient in one kernel (mri_lg2mc.h)
__global__ __launch_bounds__(LG2_THREADS) void k_mri_loss_grad_mcf(
    const float* __restrict__ x, const float* __restrict__ sens, int C, const float* __restrict__ y,
    int y_batch, const int* __restrict__ rows, const int* __restrict__ nrows, int mask_frames, int T,
    int nframes, int nkc, const float* __restrict__ chain_scale, float* __restrict__ loss_part,
    float* __restrict__ unk_part, float* __restrict__ grad, cf32* zs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Lg2DevCtx ctx{lds};
    lg2mc_body(ctx, x, sens, C, y, y_batch, rows, nrows, mask_frames, T, nframes, nkc, chain_scale,
               loss_part, unk_part, grad, zs);
}

// ------------------------------------------------------------------------------- adjoint only
// out = A^H g = ifftshift( R^2 * IFFT2( M * g ) ): the vector-Jacobian product of mri_forward.
__global__ __launch_bounds__(MRI_THREADS) void k_mri_adjoint(
    const float* __restrict__ g, const int* __restrict__ rows, const int* __restrict__ nrows,
    int mask_frames, int T, int nkc, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MriLds L = carve(lds, nkc);
    const int frame = blockIdx.x, t = frame % T, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : t;
    const int* rowlist = rows + (size_t)mf * MRI_R;
    const int nk_total = nrows[mf];
    const size_t fsz = (size_t)2 * MRI_R * MRI_R;
    const float* gr = g + (size_t)frame * fsz;
    const float* gi = gr + MRI_R * MRI_R;
    load_tables(L, rowlist, nk_total);
    if (nk_total == 0)
        for (int e = tid; e < 2 * MRI_R * MRI_R; e += MRI_THREADS) out[(size_t)frame * fsz + e] = 0.f;
    for (int g0 = 0; g0 < nk_total; g0 += nkc) {
        const int nk = min(nkc, nk_total - g0);
        for (int e = tid; e < nk * MRI_R; e += MRI_THREADS) {
            const int j = e / MRI_R, kx = e - j * MRI_R;
            const int ky = L.rowl[g0 + j];
            const float s = sgn(kx) * sgn(ky);
            L.Rk[(size_t)j * MRI_RS + kx] =
                c_make(s * gr[(size_t)ky * MRI_R + kx], s * gi[(size_t)ky * MRI_R + kx]);
        }
        __syncthreads();
        adjoint_rows(L, rowlist, g0, nk, out + (size_t)frame * fsz, g0 != 0);
    }
}

// ------------------------------------------------------------------------------- multi-coil
// y_c = M * FFT2(fftshift(S_c * x)), c < C: the coil loop runs INSIDE the per-frame workgroup, so
// x and the gradient belong to one workgroup from the first coil to the last (the gradient
// accumulates in place: sum_c conj(S_c) * A^H(2 r_c)).  NO REFERENCE COUNTERPART -- the reference
// operator is single-coil (dynamic_mri.py:42-48); C = 1 with S = 1 runs the very arithmetic of
// k_mri_loss_grad / k_mri_forward (1 * x and x - 0 are exact), bit for bit.
// sens: [C][2][R][R] (Re, Im planes, static over frames); y: [By][T][C][2][R][R].
__global__ __launch_bounds__(MRI_THREADS) void k_mri_loss_grad_mc(
    const float* __restrict__ x, const float* __restrict__ sens, int C, const float* __restrict__ y,
    int y_batch, const int* __restrict__ rows, const int* __restrict__ nrows, int mask_frames, int T,
    int nkc, const float* __restrict__ chain_scale, float* __restrict__ loss_part,
    float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MriLds L = carve(lds, nkc);
    const int frame = blockIdx.x, b = frame / T, t = frame % T, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : t;
    const int* rowlist = rows + (size_t)mf * MRI_R;
    const int nk_total = nrows[mf];
    const size_t fsz = (size_t)2 * MRI_R * MRI_R;
    const float* xr = x + (size_t)frame * fsz;
    const float* xi = xr + MRI_R * MRI_R;
    const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
    cf32 pre[MRI_NP];
    load_strip<true>(xr, xi, 0, pre, sens, sens + MRI_R * MRI_R);    // in flight during the table set-up
    load_tables(L, rowlist, nk_total);
    float acc = 0.f;
    if (nk_total == 0 && grad) {
        float* gr = grad + (size_t)frame * fsz;
        for (int e = tid; e < 2 * MRI_R * MRI_R; e += MRI_THREADS) gr[e] = 0.f;
    }
    for (int c = 0; c < C; ++c) {
        const float* sr = sens + (size_t)c * fsz;
        const float* si = sr + MRI_R * MRI_R;
        const float* yr = y + (((size_t)(y_batch > 1 ? b : 0) * T + t) * C + c) * fsz;
        const float* yi = yr + MRI_R * MRI_R;
        if (nk_total == 0) {                               // nothing kept: the loss is sum |y_c|^2
            for (int e = tid; e < MRI_R * MRI_R; e += MRI_THREADS) {
                const float a = yr[e], d = yi[e];
                acc += a * a + d * d;
            }
        }
        for (int g0 = 0; g0 < nk_total; g0 += nkc) {
            const int nk = min(nkc, nk_total - g0);
            if (c || g0) load_strip<true>(xr, xi, 0, pre, sr, si);
            forward_rows<true>(L, xr, xi, g0, nk, pre, g0 == 0 ? yr : nullptr, MRI_R - nk_total, acc, sr, si);
#pragma unroll 4
            for (int e = tid; e < nk * MRI_R; e += MRI_THREADS) {
                const int j = e / MRI_R, kx = e - j * MRI_R;
                const int ky = L.rowl[g0 + j];
                const float sx = sgn(kx), sy = sgn(ky);
                const float ya = yr[ky * MRI_R + kx], yb = yi[ky * MRI_R + kx];
                const cf32 v = L.Rk[(size_t)j * MRI_RS + kx];
                const float rr = sx * v.x - ya;
                const float ri = sx * v.y - yb;
                acc += rr * rr + ri * ri;
                const float s = cs2 * sx * sy;
                L.Rk[(size_t)j * MRI_RS + kx] = c_make(s * rr, s * ri);
            }
            __syncthreads();
            // the stores of one coil are read back by the same threads for the next (each thread
            // owns its pixels in strip_fft_inv), so no fence is needed between coils
            if (grad) adjoint_rows<true>(L, rowlist, g0, nk, grad + (size_t)frame * fsz, c != 0 || g0 != 0, sr, si);
        }
    }
    const float tot = block_sum(acc, L.red);
    if (tid == 0) loss_part[frame] = tot;
}

// out[b][t][c] = M * FFT2(fftshift(S_c * x[b][t]))
__global__ __launch_bounds__(MRI_THREADS) void k_mri_forward_mc(
    const float* __restrict__ x, const float* __restrict__ sens, int C, const int* __restrict__ rows,
    const int* __restrict__ nrows, int mask_frames, int T, int nkc, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MriLds L = carve(lds, nkc);
    const int frame = blockIdx.x, t = frame % T, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : t;
    const int* rowlist = rows + (size_t)mf * MRI_R;
    const int nk_total = nrows[mf];
    const size_t fsz = (size_t)2 * MRI_R * MRI_R;
    const float* xr = x + (size_t)frame * fsz;
    const float* xi = xr + MRI_R * MRI_R;
    cf32 pre[MRI_NP];
    load_strip<true>(xr, xi, 0, pre, sens, sens + MRI_R * MRI_R);
    load_tables(L, rowlist, nk_total);
    float unused = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* sr = sens + (size_t)c * fsz;
        const float* si = sr + MRI_R * MRI_R;
        float* outr = out + ((size_t)frame * C + c) * fsz;
        float* outi = outr + MRI_R * MRI_R;
        for (int e = tid; e < MRI_R * MRI_R; e += MRI_THREADS)
            if (L.rowpos[e / MRI_R] < 0) { outr[e] = 0.f; outi[e] = 0.f; }
        for (int g0 = 0; g0 < nk_total; g0 += nkc) {
            const int nk = min(nkc, nk_total - g0);
            if (c || g0) load_strip<true>(xr, xi, 0, pre, sr, si);
            forward_rows<true>(L, xr, xi, g0, nk, pre, nullptr, 0, unused, sr, si);
            for (int e = tid; e < nk * MRI_R; e += MRI_THREADS) {
                const int j = e / MRI_R, kx = e - j * MRI_R;
                const int ky = L.rowl[g0 + j];
                const cf32 v = L.Rk[(size_t)j * MRI_RS + kx];
                const float s = sgn(kx);
                outr[(size_t)ky * MRI_R + kx] = s * v.x;
                outi[(size_t)ky * MRI_R + kx] = s * v.y;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ multi-coil, split path
// z[f][c] = S_c * x[f] and grad[f] = sum_c conj(S_c) * g[f][c] as streaming kernels around the
// single-coil step kernel, which then sees B x (T C) frames: per coil the frame's pixels cross HBM
// four times (z written and read, g written and read) instead of the fused kernel's x, S_c and
// running gradient per coil through one workgroup -- but on the pipelined, persistent kernel.
// x, grad: [nframes][2][RR]; sens: [C][2][RR]; z, g: [nframes][C][2][RR]; RR % 4 == 0.
// A thread keeps its four pixels of eight sensitivity maps in registers and walks MC_FPB frames: the maps
// cross L2 once per MC_FPB frames instead of once per frame (they were as much L2 traffic as z itself).
#define MC_FPB 8
#define MC_CC 8
typedef float mc_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mri_coil_mul(const float4* __restrict__ x, const float4* __restrict__ sens,
                                                      int C, int rr4, int nframes, float4* __restrict__ z) {
    const int f0 = blockIdx.y * MC_FPB, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rr4) return;
    const int f1 = min(nframes, f0 + MC_FPB);
    for (int c0 = 0; c0 < C; c0 += MC_CC) {
        mc_f4 p[MC_CC], q[MC_CC];
#pragma unroll
        for (int k = 0; k < MC_CC; ++k)
            if (c0 + k < C) {
                p[k] = *(const mc_f4*)&sens[((size_t)(c0 + k) * 2 + 0) * rr4 + i];
                q[k] = *(const mc_f4*)&sens[((size_t)(c0 + k) * 2 + 1) * rr4 + i];
            }
        for (int f = f0; f < f1; ++f) {
            const mc_f4 a = *(const mc_f4*)&x[((size_t)f * 2 + 0) * rr4 + i], b = *(const mc_f4*)&x[((size_t)f * 2 + 1) * rr4 + i];
#pragma unroll
            for (int k = 0; k < MC_CC; ++k)
                if (c0 + k < C) {
                    float4* o = z + (((size_t)f * C + c0 + k) * 2) * rr4 + i;
                    // z is read once, by the next kernel, long after: non-temporal
                    __builtin_nontemporal_store(a * p[k] - b * q[k], (mc_f4*)o);
                    __builtin_nontemporal_store(a * q[k] + b * p[k], (mc_f4*)(o + rr4));
                }
        }
    }
}

// (the same frame walk in the combine -- eight frames per thread, maps in registers -- halves the loads in
// flight and takes 0.89 ms instead of 0.39: it stays one frame per thread)
__global__ __launch_bounds__(256) void k_mri_coil_combine(const float4* __restrict__ g, const float4* __restrict__ sens,
                                                          int C, int rr4, float4* __restrict__ out) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rr4) return;
    mc_f4 sr = {0.f, 0.f, 0.f, 0.f}, si = sr;
    for (int c = 0; c < C; ++c) {                              // fixed coil order
        const mc_f4 p = *(const mc_f4*)&sens[((size_t)c * 2 + 0) * rr4 + i], q = *(const mc_f4*)&sens[((size_t)c * 2 + 1) * rr4 + i];
        const float4* gi = g + (((size_t)f * C + c) * 2) * rr4 + i;
        const mc_f4 u = __builtin_nontemporal_load((const mc_f4*)gi), v = __builtin_nontemporal_load((const mc_f4*)(gi + rr4));
        sr += u * p + v * q;                                   // conj(S) * (u + i v)
        si += v * p - u * q;
    }
    *(mc_f4*)&out[((size_t)f * 2 + 0) * rr4 + i] = sr;
    *(mc_f4*)&out[((size_t)f * 2 + 1) * rr4 + i] = si;
}

extern "C" int mri_coil_mul(const float* x, const float* sens, int C, int nframes, int R, float* z,
                            void* stream) {
    STEP_REQUIRE(x && sens && z && C > 0 && nframes > 0, "mri_coil_mul: null pointer / sizes");
    STEP_REQUIRE(R > 0 && (R * R) % 4 == 0 && nframes <= 65535, "mri_coil_mul: R = %d, %d frames (<= 65535)", R, nframes);
    STEP_REQUIRE((((uintptr_t)x | (uintptr_t)sens | (uintptr_t)z) & 15) == 0, "mri_coil_mul: 16-byte aligned buffers");
    const int rr4 = R * R / 4;
    hipLaunchKernelGGL(k_mri_coil_mul, dim3((rr4 + 255) / 256, (nframes + MC_FPB - 1) / MC_FPB), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)x, (const float4*)sens, C, rr4, nframes, (float4*)z);
    STEP_CHECK_LAUNCH("mri_coil_mul");
    return STEP_OK;
}

extern "C" int mri_coil_combine(const float* g, const float* sens, int C, int nframes, int R, float* out,
                                void* stream) {
    STEP_REQUIRE(g && sens && out && C > 0 && nframes > 0, "mri_coil_combine: null pointer / sizes");
    STEP_REQUIRE(R > 0 && (R * R) % 4 == 0 && nframes <= 65535, "mri_coil_combine: R = %d, %d frames (<= 65535)", R, nframes);
    STEP_REQUIRE((((uintptr_t)g | (uintptr_t)sens | (uintptr_t)out) & 15) == 0, "mri_coil_combine: 16-byte aligned buffers");
    const int rr4 = R * R / 4;
    hipLaunchKernelGGL(k_mri_coil_combine, dim3((rr4 + 255) / 256, nframes), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g, (const float4*)sens, C, rr4, (float4*)out);
    STEP_CHECK_LAUNCH("mri_coil_combine");
    return STEP_OK;
}

// loss[b] = sum_t part[b][t] + sum of unk[0 .. n_unk): the shares of the masked-out rows' |y|^2
// of a batch-1 measurement, which the pipelined kernel sums once for all chains (n_unk = 0
// when the partial losses already hold them); total (optional) = sum_b loss[b], the scalar that
// gradient(return_loss=True) returns (sampler/daps.py:100) -- one launch less per step
__global__ __launch_bounds__(256) void k_mri_loss_finish(const float* __restrict__ part,
                                                         const float* __restrict__ unk, int n_unk,
                                                         int B, int T, float* __restrict__ loss,
                                                         float* __restrict__ total) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float v = 0.f;
    for (int i = tid; i < n_unk; i += 256) v += unk[i];
    red[tid] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {                      // fixed order: reproducible
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const float su = red[0];
    float mine = 0.f;                                        // one block: b = tid, tid + 256, ...
    for (int b = tid; b < B; b += 256) {
        const float* p = part + (size_t)b * T;
        float s = 0.f;
#pragma unroll 4
        for (int t = 0; t < T; ++t) s += p[t];
        loss[b] = s + su;
        mine += s + su;
    }
    if (total) {                                             // sum over the chains, fixed order
        __syncthreads();
        red[tid] = mine;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) red[tid] += red[tid + w];
            __syncthreads();
        }
        if (tid == 0) *total = red[0];
    }
}

static int mri_check(const char* who, int B, int T, int R, const int32_t* rows,
                     const int32_t* nrows, int mask_frames) {
    STEP_REQUIRE(B > 0 && T > 0, "%s: bad sizes B=%d T=%d", who, B, T);
    STEP_REQUIRE(R == MRI_R, "%s: only resolution %d is built (got %d)", who, MRI_R, R);
    STEP_REQUIRE(rows && nrows, "%s: null mask row tables", who);
    STEP_REQUIRE(mask_frames == 1 || mask_frames == T, "%s: mask_frames=%d must be 1 or T=%d", who, mask_frames, T);
    return STEP_OK;
}

// nkc = rows of the LDS row buffer; max_rows = the largest kept-row count of any mask frame
// (host knowledge: the row tables are built on the host).  Larger masks run in groups of 64.
static int pick_nkc(int max_rows) {
    int n = max_rows < 1 ? 1 : max_rows;
    return n > MRI_NKC_MAX ? MRI_NKC_MAX : n;
}

extern "C" int mri_forward(const float* x, int B, int T, int R, const int32_t* rows,
                           const int32_t* nrows, int mask_frames, int max_rows, float* out,
                           void* stream) {
    STEP_REQUIRE(x && out, "mri_forward: null pointer");
    if (int rc = mri_check("mri_forward", B, T, R, rows, nrows, mask_frames)) return rc;
    const int nkc = pick_nkc(max_rows);
    const size_t lds = mri_lds_bytes(nkc);
    step_allow_lds((const void*)k_mri_forward, lds);
    hipLaunchKernelGGL(k_mri_forward, dim3(B * T), dim3(MRI_THREADS), lds, (hipStream_t)stream, x,
                       rows, nrows, mask_frames, T, nkc, out);
    STEP_CHECK_LAUNCH("mri_forward");
    return STEP_OK;
}

extern "C" int mri_adjoint(const float* g, int B, int T, int R, const int32_t* rows,
                           const int32_t* nrows, int mask_frames, int max_rows, float* out,
                           void* stream) {
    STEP_REQUIRE(g && out, "mri_adjoint: null pointer");
    if (int rc = mri_check("mri_adjoint", B, T, R, rows, nrows, mask_frames)) return rc;
    const int nkc = pick_nkc(max_rows);
    const size_t lds = mri_lds_bytes(nkc);
    step_allow_lds((const void*)k_mri_adjoint, lds);
    hipLaunchKernelGGL(k_mri_adjoint, dim3(B * T), dim3(MRI_THREADS), lds, (hipStream_t)stream, g,
                       rows, nrows, mask_frames, T, nkc, out);
    STEP_CHECK_LAUNCH("mri_adjoint");
    return STEP_OK;
}

extern "C" int mri_forward_mc(const float* x, const float* sens, int C, int B, int T, int R,
                              const int32_t* rows, const int32_t* nrows, int mask_frames,
                              int max_rows, float* out, void* stream) {
    STEP_REQUIRE(x && sens && out && C > 0, "mri_forward_mc: null pointer / C");
    if (int rc = mri_check("mri_forward_mc", B, T, R, rows, nrows, mask_frames)) return rc;
    const int nkc = pick_nkc(max_rows);
    const size_t lds = mri_lds_bytes(nkc);
    step_allow_lds((const void*)k_mri_forward_mc, lds);
    hipLaunchKernelGGL(k_mri_forward_mc, dim3(B * T), dim3(MRI_THREADS), lds, (hipStream_t)stream, x,
                       sens, C, rows, nrows, mask_frames, T, nkc, out);
    STEP_CHECK_LAUNCH("mri_forward_mc");
    return STEP_OK;
}

extern "C" int mri_loss_grad_mc(const float* x, const float* sens, int C, const float* y, int y_batch,
                                int B, int T, int R, const int32_t* rows, const int32_t* nrows,
                                int mask_frames, int max_rows, const float* chain_scale,
                                float* loss_part, float* loss, float* loss_total, float* grad,
                                void* stream) {
    STEP_REQUIRE(x && sens && y && loss_part && loss && C > 0, "mri_loss_grad_mc: null pointer / C");
    STEP_REQUIRE(y_batch == 1 || y_batch == B, "mri_loss_grad_mc: measurement batch %d vs B=%d", y_batch, B);
    if (int rc = mri_check("mri_loss_grad_mc", B, T, R, rows, nrows, mask_frames)) return rc;
    const int nkc = pick_nkc(max_rows);
    const size_t lds = mri_lds_bytes(nkc);
    hipStream_t s = (hipStream_t)stream;
    step_allow_lds((const void*)k_mri_loss_grad_mc, lds);
    hipLaunchKernelGGL(k_mri_loss_grad_mc, dim3(B * T), dim3(MRI_THREADS), lds, s, x, sens, C, y, y_batch,
                       rows, nrows, mask_frames, T, nkc, chain_scale, loss_part, grad);
    STEP_CHECK_LAUNCH("mri_loss_grad_mc");
    hipLaunchKernelGGL(k_mri_loss_finish, dim3(1), dim3(256), 0, s, loss_part, loss_part, 0, B, T, loss, loss_total);
    STEP_CHECK_LAUNCH("mri_loss_grad_mc (finish)");
    return STEP_OK;
}

// Shared body of mri_loss_grad and mri_loss_grad_static.  variant: 0 = pick by mask (pipelined
// kernel when every mask frame fits one row buffer), 1 = first-generation kernel, 2 = pipelined
// kernel (an error if the mask does not fit).
int mri_loss_grad_impl(const char* who, const float* x, const float* y, int y_batch, int B,
                              int T, int R, const int32_t* rows, const int32_t* nrows,
                              int mask_frames, int max_rows, const float* chain_scale, int variant,
                              float* loss_part, float* loss, float* loss_total, float* grad,
                              hipStream_t s) {
    STEP_REQUIRE(x && y && loss_part && loss, "%s: null pointer", who);
    STEP_REQUIRE(y_batch == 1 || y_batch == B, "%s: measurement batch %d vs B=%d", who, y_batch, B);
    STEP_REQUIRE(variant == 0 || variant == 1 || variant == 2 || variant == 4,
                 "%s: variant %d (0 auto, 1 first generation, 2 pipelined, 4 persistent; 3, the two-per-CU kernel "
                 "that measured slower, left the library in round 3: scripts/probes/mri_lg4.h)", who, variant);
    STEP_REQUIRE(variant != 4 || (max_rows <= LG2_NKC_MAX && y_batch == 1),
                 "%s: the persistent kernel takes a batch-1 measurement and masks of <= %d kept rows", who, LG2_NKC_MAX);
    if (int rc = mri_check(who, B, T, R, rows, nrows, mask_frames)) return rc;
    STEP_REQUIRE(variant != 2 || max_rows <= LG2_NKC_MAX,
                 "%s: the pipelined kernel takes masks of <= %d kept rows (got %d)", who, LG2_NKC_MAX, max_rows);
    const int nkc = pick_nkc(max_rows);
    float* unk = loss_part + (size_t)B * T;              // second part of the workspace
    int n_unk = 0;
    // auto: the persistent form of the pipelined kernel for a shared measurement (bit-identical to
    // variant 2, 0-6 % faster), the pipelined kernel for a measurement per chain
    if (variant == 4 || (variant == 0 && y_batch == 1 && max_rows <= LG2_NKC_MAX)) {
        const size_t lds = lg2_lds_bytes(nkc);
        const int nframes = B * T;
        int cus = 256, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        n_unk = (B < LG2_UNK_PARTS ? B : LG2_UNK_PARTS) * T;
        step_allow_lds((const void*)k_mri_loss_grad2p, lds);
        hipLaunchKernelGGL(k_mri_loss_grad2p, dim3(nframes < cus ? nframes : cus), dim3(LG2_THREADS), lds, s, x, y,
                           rows, nrows, mask_frames, T, nframes, nkc, chain_scale, loss_part, unk, grad);
    } else if (max_rows <= LG2_NKC_MAX && variant != 1) {
        // every mask frame fits one row buffer: pipelined kernel
        const size_t lds = lg2_lds_bytes(nkc);
        const int nframes = B * T;
        if (y_batch == 1) {
            n_unk = (B < LG2_UNK_PARTS ? B : LG2_UNK_PARTS) * T;
            step_allow_lds((const void*)k_mri_loss_grad2<false>, lds);
            hipLaunchKernelGGL(k_mri_loss_grad2<false>, dim3(nframes), dim3(LG2_THREADS), lds, s, x, y,
                               y_batch, rows, nrows, mask_frames, T, nframes, nkc, chain_scale,
                               loss_part, unk, grad);
        } else {
            step_allow_lds((const void*)k_mri_loss_grad2<true>, lds);
            hipLaunchKernelGGL(k_mri_loss_grad2<true>, dim3(nframes), dim3(LG2_THREADS), lds, s, x, y,
                               y_batch, rows, nrows, mask_frames, T, nframes, nkc, chain_scale,
                               loss_part, unk, grad);
        }
    } else {
        const size_t lds = mri_lds_bytes(nkc);
        step_allow_lds((const void*)k_mri_loss_grad, lds);
        hipLaunchKernelGGL(k_mri_loss_grad, dim3(B * T), dim3(MRI_THREADS), lds, s, x, y, y_batch,
                           rows, nrows, mask_frames, T, nkc, chain_scale, loss_part, grad);
    }
    STEP_CHECK_LAUNCH(who);
    hipLaunchKernelGGL(k_mri_loss_finish, dim3(1), dim3(256), 0, s, loss_part, unk, n_unk, B, T, loss, loss_total);
    STEP_CHECK_LAUNCH(who);
    return STEP_OK;
}

extern "C" int mri_loss_grad(const float* x, const float* y, int y_batch, int B, int T, int R,
                             const int32_t* rows, const int32_t* nrows, int mask_frames,
                             int max_rows, const float* chain_scale, int variant, float* loss_part,
                             float* loss, float* loss_total, float* grad, void* stream) {
    return mri_loss_grad_impl("mri_loss_grad", x, y, y_batch, B, T, R, rows, nrows, mask_frames,
                              max_rows, chain_scale, variant, loss_part, loss, loss_total, grad,
                              (hipStream_t)stream);
}

// ------------------------------------------------------------ small batches: a frame on several workgroups
static int mri_cu_count() {
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    return cus;
}

// workgroups per frame: 6 or 3, the most with which the whole grid (frames rounded up to a multiple of
// 8, see lg2s_body) is resident at once; 0 = no split.  Two per frame are built and tested but not
// chosen: measured 39.6 us per step at 8 chains against 38.6 us for the unsplit kernel (the exchange
// costs what half a column phase saves).
static int lg2s_parts(long long nframes, int want) {
    const long long slots = mri_cu_count(), f8 = (nframes + 7) / 8 * 8;
    const int cand[3] = {6, 3, 2};
    for (int i = 0; i < 3; ++i)
        if ((want == 0 ? cand[i] > 2 : want == cand[i]) && f8 * cand[i] <= slots) return cand[i];
    return 0;
}

struct Lg2sWs { unsigned* cnt; float* part; float* unk; cf32* zs; size_t bytes; };
static Lg2sWs lg2s_carve_ws(void* ws, long long nframes, int nkc) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    Lg2sWs w;
    char* p = (char*)ws;
    size_t o = 0;
    w.cnt = (unsigned*)(p + o); o += up((size_t)(2 * nframes + 2) * 4);
    w.part = (float*)(p + o);   o += up((size_t)nframes * 6 * 4);
    w.unk = (float*)(p + o);    o += up((size_t)nframes * 6 * 4);
    w.zs = (cf32*)(p + o);      o += (size_t)nframes * nkc * MRI_R * sizeof(cf32);
    w.bytes = o;
    return w;
}

extern "C" size_t mri_loss_grad_split_ws_bytes(int B, int T, int max_rows, int parts) {
    if (B <= 0 || T <= 0 || max_rows > LG2_NKC_MAX) return 0;
    const long long nframes = (long long)B * T;
    if ((parts != 0 && parts != 2 && parts != 3 && parts != 6) || !lg2s_parts(nframes, parts)) return 0;
    return lg2s_carve_ws(nullptr, nframes, pick_nkc(max_rows)).bytes;
}

extern "C" int mri_loss_grad_split(const float* x, const float* y, int B, int T, int R,
                                   const int32_t* rows, const int32_t* nrows, int mask_frames,
                                   int max_rows, const float* chain_scale, int parts, void* ws,
                                   size_t ws_bytes, float* loss, float* loss_total, float* grad,
                                   void* stream) {
    const char* who = "mri_loss_grad_split";
    STEP_REQUIRE(x && y && ws && loss, "%s: null pointer", who);
    if (int rc = mri_check(who, B, T, R, rows, nrows, mask_frames)) return rc;
    STEP_REQUIRE(max_rows <= LG2_NKC_MAX, "%s: masks of <= %d kept rows (got %d)", who, LG2_NKC_MAX, max_rows);
    STEP_REQUIRE(parts == 0 || parts == 2 || parts == 3 || parts == 6, "%s: parts %d (0 auto, 2, 3, 6)", who, parts);
    const long long nframes = (long long)B * T;
    const int np = lg2s_parts(nframes, parts);
    STEP_REQUIRE(np > 0, "%s: %lld frames x %d workgroups do not fit the %d CUs at once; use mri_loss_grad", who,
                 nframes, parts ? parts : 2, mri_cu_count());
    const int nkc = pick_nkc(max_rows);
    const Lg2sWs w = lg2s_carve_ws(ws, nframes, nkc);
    STEP_REQUIRE(ws_bytes >= w.bytes, "%s: workspace of %zu bytes, %zu needed", who, ws_bytes, w.bytes);
    STEP_REQUIRE(((uintptr_t)ws & 15) == 0, "%s: workspace must be 16-byte aligned", who);
    const size_t lds = lg2_lds_bytes(nkc);
    const dim3 grid((unsigned)((nframes + 7) / 8 * 8 * np));
    hipStream_t s = (hipStream_t)stream;
#define LG2S_LAUNCH(SP)                                                                               \
    step_allow_lds((const void*)k_mri_loss_grad2s<SP>, lds);                                          \
    hipLaunchKernelGGL(k_mri_loss_grad2s<SP>, grid, dim3(LG2_THREADS), lds, s, x, y, rows, nrows,     \
                       mask_frames, T, (int)nframes, nkc, chain_scale, w.part, w.unk, grad, w.zs,     \
                       w.cnt, loss, loss_total)
    if (np == 6) { LG2S_LAUNCH(1); }
    else if (np == 3) { LG2S_LAUNCH(2); }
    else { LG2S_LAUNCH(3); }
#undef LG2S_LAUNCH
    STEP_CHECK_LAUNCH(who);
    return STEP_OK;
}

// ------------------------------------------------------------ multi-coil step in one kernel
static long long mcf_grid(long long nframes) {
    const long long cus = mri_cu_count();
    return nframes < cus ? nframes : cus;
}

extern "C" size_t mri_loss_grad_mc_fused_ws_bytes(int C, int B, int T, int max_rows) {
    if (C <= 0 || B <= 0 || T <= 0 || max_rows > LG2_NKC_MAX) return 0;
    return (size_t)mcf_grid((long long)B * T) * C * pick_nkc(max_rows) * MRI_R * sizeof(cf32);
}

extern "C" int mri_loss_grad_mc_fused(const float* x, const float* sens, int C, const float* y, int y_batch,
                                      int B, int T, int R, const int32_t* rows, const int32_t* nrows,
                                      int mask_frames, int max_rows, const float* chain_scale,
                                      float* loss_part, void* ws, size_t ws_bytes, float* loss,
                                      float* loss_total, float* grad, void* stream) {
    const char* who = "mri_loss_grad_mc_fused";
    STEP_REQUIRE(x && sens && y && loss_part && loss && C > 0, "%s: null pointer / C", who);
    STEP_REQUIRE(y_batch == 1 || y_batch == B, "%s: measurement batch %d vs B=%d", who, y_batch, B);
    if (int rc = mri_check(who, B, T, R, rows, nrows, mask_frames)) return rc;
    STEP_REQUIRE(max_rows <= LG2_NKC_MAX, "%s: masks of <= %d kept rows (got %d); use mri_loss_grad_mc", who,
                 LG2_NKC_MAX, max_rows);
    const size_t need = mri_loss_grad_mc_fused_ws_bytes(C, B, T, max_rows);
    STEP_REQUIRE(!grad || (ws && ws_bytes >= need), "%s: workspace of %zu bytes, %zu needed", who, ws_bytes, need);
    STEP_REQUIRE(((uintptr_t)ws & 7) == 0, "%s: workspace must be 8-byte aligned", who);
    const int nkc = pick_nkc(max_rows), nframes = B * T;
    const size_t lds = lg2_lds_bytes(nkc);
    float* unk = loss_part + (size_t)nframes;
    const int n_unk = y_batch == 1 ? (B < LG2_UNK_PARTS ? B : LG2_UNK_PARTS) * T : 0;
    hipStream_t s = (hipStream_t)stream;
    step_allow_lds((const void*)k_mri_loss_grad_mcf, lds);
    hipLaunchKernelGGL(k_mri_loss_grad_mcf, dim3((unsigned)mcf_grid(nframes)), dim3(LG2_THREADS), lds, s, x, sens, C,
                       y, y_batch, rows, nrows, mask_frames, T, nframes, nkc, chain_scale, loss_part, unk, grad,
                       (cf32*)ws);
    STEP_CHECK_LAUNCH(who);
    hipLaunchKernelGGL(k_mri_loss_finish, dim3(1), dim3(256), 0, s, loss_part, unk, n_unk, B, T, loss, loss_total);
    STEP_CHECK_LAUNCH(who);
    return STEP_OK;
}

// ------------------------------------------------------------ the step with the state update inside
// mri_loss_grad(variant 2) with MCMCSampler.optimize's update (mode 1), or the Langevin (2) / momentum-free HMC
// (3) step of MCMCSampler.sample with the noise read from `eps`, applied by the inverse column phase where
// it would have stored the gradient: the gradient never reaches HBM.
extern "C" int mri_step_update(float* x, const float* x0hat, const float* eps, const float* y, int B, int T, int R,
                               const int32_t* rows, const int32_t* nrows, int mask_frames, int max_rows,
                               int mode, const mcmc_params_t* p, float* loss_part, float* loss,
                               float* loss_total, int32_t* nan_flag, void* stream) {
    const char* who = "mri_step_update";
    STEP_REQUIRE(x && x0hat && y && p && loss_part && loss, "%s: null pointer", who);
    STEP_REQUIRE(mode >= 1 && mode <= 3, "%s: mode %d (1 optimize, 2 langevin, 3 hmc at momentum 0)", who, mode);
    STEP_REQUIRE(mode == 1 || eps, "%s: langevin / hmc read their noise from `eps`", who);
    STEP_REQUIRE(mode != 3 || p->momentum == 0.f, "%s: hmc with momentum %g carries a velocity; use mri_loss_grad + mcmc_update",
                 who, (double)p->momentum);
    STEP_REQUIRE(x != x0hat, "%s: x and x0hat must not share storage", who);
    if (int rc = mri_check(who, B, T, R, rows, nrows, mask_frames)) return rc;
    STEP_REQUIRE(max_rows <= LG2_NKC_MAX, "%s: masks of <= %d kept rows (got %d)", who, LG2_NKC_MAX, max_rows);
    const int nkc = pick_nkc(max_rows), nframes = B * T;
    const size_t lds = lg2_lds_bytes(nkc);
    float* unk = loss_part + (size_t)nframes;
    const int n_unk = (B < LG2_UNK_PARTS ? B : LG2_UNK_PARTS) * T;
    const Lg2Upd up{x, x0hat, eps, p->lr, p->inv_sigma2, p->inv_tau2, p->sqrt_lr, p->sqrt_2lr, p->noise_scale, nan_flag};
    hipStream_t s = (hipStream_t)stream;
#define MRI_UPD_LAUNCH(M)                                                                              \
    step_allow_lds((const void*)k_mri_step_upd<M>, lds);                                               \
    hipLaunchKernelGGL(k_mri_step_upd<M>, dim3(nframes), dim3(LG2_THREADS), lds, s, x, y, rows, nrows, \
                       mask_frames, T, nframes, nkc, nullptr, loss_part, unk, up)
    if (mode == 1) { MRI_UPD_LAUNCH(1); }
    else if (mode == 2) { MRI_UPD_LAUNCH(2); }
    else { MRI_UPD_LAUNCH(3); }
#undef MRI_UPD_LAUNCH
    STEP_CHECK_LAUNCH(who);
    hipLaunchKernelGGL(k_mri_loss_finish, dim3(1), dim3(256), 0, s, loss_part, unk, n_unk, B, T, loss, loss_total);
    STEP_CHECK_LAUNCH(who);
    return STEP_OK;
}
