// Dynamic-MRI operator at ANY frame size H x W (2 .. 1024 each): M * FFT2(fftshift(x_re + i x_im)) with the
// k_y row mask (forward_operator/dynamic_mri.py:42-48), its adjoint, and the loss + gradient of
// base.py:40-49.  The reference's operator takes any `resolution` (dynamic_mri.py:9); the tuned kernels of
// mri_fft.hip exist for 192 x 192 (the shipped configuration) only -- this file is what every other size runs
// on: mixed-radix Stockham passes over LDS (fft_any.h), three kernels per step with the kept k_y rows
// (nk of H) crossing a workspace in between:
//   1. k_any_cols_fwd   per (frame, strip of columns): FFT along y of all H rows, kept rows -> ws
//   2. k_any_rows<MODE> per (frame, group of kept rows): FFT along x; then MODE 0 writes the measurement,
//      MODE 1 takes the residual against y, sums the loss, scales by 2 and runs the inverse FFT along x back
//      into ws; MODE 2 (adjoint only) starts from a k-space input instead
//   3. k_any_cols_inv   per (frame, strip): kept rows of ws into a zero strip, inverse FFT along y -> image
// fftshift is a roll by n/2 (floor) in both axes = a phase exp(-2 pi i (n/2) k / n) on the spectrum (the sign
// (-1)^k for even n); the adjoint applies its conjugate.  No data is moved for it.
// Bound: the strips make 64..256-byte segments and the kept rows cross L2/HBM twice -- a general path, measured
// at 192 x 192 against the tuned kernels in DESIGN.md; results agree with them to fp32 rounding.
#include "common.h"
#include "fft_any_lds.h"

// index into an n-entry twiddle table of the fftshift phase of spectrum line k: (n/2) k mod n
__device__ __forceinline__ int any_shift_idx(int k, int n) {
    return (n & 1) ? (int)(((long long)(n / 2) * k) % n) : ((k & 1) ? n / 2 : 0);
}

// ws: [frames][max_rows][W] complex, row j of a frame = its j-th kept k_y row
// SW (a power of two) columns per workgroup, element (row i, column c) of the strip at a[i * SW + c]
__global__ __launch_bounds__(ANY_THREADS, 4) void k_any_cols_fwd(
    const float* __restrict__ x, const int* __restrict__ rows, const int* __restrict__ nrows,
    int mask_frames, int T, int H, int W, int SW, int lgSW, int nstrips, int max_rows, FftAnyPlan plan,
    cf32* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    cf32* tw = (cf32*)lds;
    cf32* a = tw + H;
    cf32* b = a + (size_t)H * SW;                     // only with a direct pass in the plan
    const int frame = blockIdx.x / nstrips, strip = blockIdx.x - frame * nstrips, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : frame % T;
    const int c0 = strip * SW, nl = min(SW, W - c0);
    const float* xr = x + (size_t)frame * 2 * H * W;
    const float* xi = xr + (size_t)H * W;
    any_twiddles(tw, H);
    // twelve pixels per thread in flight at a time (a plain loop waits for every load before the next)
    for (int e0 = tid; e0 < H * SW; e0 += 12 * ANY_THREADS) {
        cf32 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int e = e0 + u * ANY_THREADS, i = e >> lgSW, c = e & (SW - 1);
            const bool ok = e < H * SW && c < nl;
            const size_t o = ok ? (size_t)i * W + c0 + c : 0;
            v[u] = c_make(xr[o], xi[o]);
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int e = e0 + u * ANY_THREADS;
            if (e < H * SW) a[e] = v[u];
        }
    }
    __syncthreads();
    any_fft<-1>(a, b, tw, H, nl, SW, SW, plan);
    const int nk = nrows[mf];
    const int* rowl = rows + (size_t)mf * H;
    cf32* dst = ws + (size_t)frame * max_rows * W + c0;
    for (int e0 = tid; e0 < nk * SW; e0 += 4 * ANY_THREADS) {     // the row list four entries ahead of their use
        int ky[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ANY_THREADS;
            ky[u] = e < nk * SW ? rowl[e >> lgSW] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ANY_THREADS, j = e >> lgSW, c = e & (SW - 1);
            if (e < nk * SW && c < nl) {
                const cf32 w = tw[any_shift_idx(ky[u], H)];
                dst[(size_t)j * W + c] = c_rot<-1>(a[ky[u] * SW + c], w.x, w.y);
            }
        }
    }
}

// NR (a power of two) kept rows per workgroup, transposed in LDS: element (row jj, column x) at a[x * LD + jj]
// MODE 0: out = measurement rows; MODE 1: residual, loss, 2 A^H-side rows back into ws; MODE 2: rows of g -> ws
template <int MODE>
__global__ __launch_bounds__(ANY_THREADS, 4) void k_any_rows(
    cf32* __restrict__ ws, const float* __restrict__ y, int y_batch, const float* __restrict__ g,
    const int* __restrict__ rows, const int* __restrict__ nrows, int mask_frames, int T, int H, int W,
    int NR, int ngroups, int max_rows, FftAnyPlan plan, const float* __restrict__ chain_scale,
    float* __restrict__ out, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = NR + 1;                            // padded: the transposed fill is conflict-free
    cf32* tw = (cf32*)lds;
    cf32* phr = tw + W;                               // [NR] fftshift phase of each row of the group
    float* red = (float*)(phr + NR);
    int* rl = (int*)(red + 16);                       // [NR] the group's k_y rows (a copy of the global list)
    cf32* a = (cf32*)(rl + ((NR + 1) & ~1));
    cf32* b = a + (size_t)W * LD;                     // only with a direct pass in the plan
    const int frame = blockIdx.x / ngroups, grp = blockIdx.x - frame * ngroups, tid = threadIdx.x;
    const int bch = frame / T, t = frame - bch * T;
    const int mf = mask_frames == 1 ? 0 : t;
    const int nk = nrows[mf], j0 = grp * NR;
    const int nr = min(NR, nk - j0);
    if (nr <= 0) {                                    // uniform over the workgroup
        if (MODE == 1 && tid == 0) part[blockIdx.x] = 0.f;
        return;
    }
    const int* rowl = rows + (size_t)mf * H + j0;
    cf32* wrow = ws + ((size_t)frame * max_rows + j0) * W;
    any_twiddles(tw, W);
    if (tid < nr) {
        const int ky = rowl[tid];
        float sn, cn;
        sincospif((float)(2 * any_shift_idx(ky, H)) / (float)H, &sn, &cn);
        phr[tid] = c_make(cn, sn);
        rl[tid] = ky;                                 // from LDS below: a global load of the row id ahead of every
    }                                                 // measurement load was a second memory latency per batch
    __syncthreads();
    const size_t plane = (size_t)H * W;
    if (MODE == 2) {
        const float* gr = g + (size_t)frame * 2 * plane;
        const float* gi = gr + plane;
        for (int e = tid; e < nr * W; e += ANY_THREADS) {
            const int jj = e / W, kx = e - jj * W;
            const size_t o = (size_t)rl[jj] * W + kx;
            const cf32 wx = tw[any_shift_idx(kx, W)], wy = phr[jj];
            cf32 v = c_rot<1>(c_make(gr[o], gi[o]), wx.x, wx.y);
            a[kx * LD + jj] = c_rot<1>(v, wy.x, wy.y);
        }
        __syncthreads();
        any_fft<1>(a, b, tw, W, nr, NR, LD, plan);
        for (int e = tid; e < nr * W; e += ANY_THREADS) {
            const int jj = e / W, xx = e - jj * W;
            wrow[(size_t)jj * W + xx] = a[xx * LD + jj];
        }
        return;
    }
    for (int e0 = tid; e0 < nr * W; e0 += 4 * ANY_THREADS) {      // the rows are contiguous in ws
        cf32 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ANY_THREADS;
            v[u] = wrow[e < nr * W ? e : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ANY_THREADS, jj = e / W, xx = e - jj * W;
            if (e < nr * W) a[xx * LD + jj] = v[u];
        }
    }
    __syncthreads();
    any_fft<-1>(a, b, tw, W, nr, NR, LD, plan);
    if (MODE == 0) {
        float* outr = out + (size_t)frame * 2 * plane;
        float* outi = outr + plane;
        for (int e = tid; e < nr * W; e += ANY_THREADS) {
            const int jj = e / W, kx = e - jj * W;
            const cf32 wx = tw[any_shift_idx(kx, W)];
            const cf32 v = c_rot<-1>(a[kx * LD + jj], wx.x, wx.y);
            const size_t o = (size_t)rl[jj] * W + kx;
            outr[o] = v.x;
            outi[o] = v.y;
        }
        return;
    }
    // MODE 1
    const float* yr = y + ((size_t)(y_batch > 1 ? bch : 0) * T + t) * 2 * plane;
    const float* yi = yr + plane;
    const float cs2 = 2.f * (chain_scale ? chain_scale[bch] : 1.f);
    float acc = 0.f;
    for (int e0 = tid; e0 < nr * W; e0 += 8 * ANY_THREADS) {
        cf32 ym[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * ANY_THREADS, ec = e < nr * W ? e : 0, jj = ec / W, kx = ec - jj * W;
            const size_t o = (size_t)rl[jj] * W + kx;
            ym[u] = c_make(yr[o], yi[o]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * ANY_THREADS, jj = e / W, kx = e - jj * W;
            if (e < nr * W) {
                const cf32 wx = tw[any_shift_idx(kx, W)], wy = phr[jj];
                const cf32 v = c_rot<-1>(a[kx * LD + jj], wx.x, wx.y);
                const float rr = v.x - ym[u].x, ri = v.y - ym[u].y;
                acc += rr * rr + ri * ri;
                const cf32 w = c_rot<1>(c_make(cs2 * rr, cs2 * ri), wx.x, wx.y);
                a[kx * LD + jj] = c_rot<1>(w, wy.x, wy.y);
            }
        }
    }
    const float tot = block_sum(acc, red);            // its barriers also publish the rows
    if (tid == 0) part[blockIdx.x] = tot;
    if (out == nullptr) return;                       // loss only (uniform)
    any_fft<1>(a, b, tw, W, nr, NR, LD, plan);
    for (int e = tid; e < nr * W; e += ANY_THREADS) {
        const int jj = e / W, xx = e - jj * W;
        wrow[(size_t)jj * W + xx] = a[xx * LD + jj];
    }
}

__global__ __launch_bounds__(ANY_THREADS, 4) void k_any_cols_inv(
    const cf32* __restrict__ ws, const int* __restrict__ rows, const int* __restrict__ nrows,
    int mask_frames, int T, int H, int W, int SW, int lgSW, int nstrips, int max_rows, FftAnyPlan plan,
    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    cf32* tw = (cf32*)lds;
    cf32* a = tw + H;
    cf32* b = a + (size_t)H * SW;
    const int frame = blockIdx.x / nstrips, strip = blockIdx.x - frame * nstrips, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : frame % T;
    const int c0 = strip * SW, nl = min(SW, W - c0);
    any_twiddles(tw, H);
    for (int e = tid; e < H * SW; e += ANY_THREADS) a[e] = c_make(0.f, 0.f);
    __syncthreads();
    const int nk = nrows[mf];
    const int* rowl = rows + (size_t)mf * H;
    const cf32* src = ws + (size_t)frame * max_rows * W + c0;
    for (int e0 = tid; e0 < nk * SW; e0 += 4 * ANY_THREADS) {
        cf32 v[4];
        int ky[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ANY_THREADS, j = e >> lgSW, c = e & (SW - 1);
            const bool ok = e < nk * SW && c < nl;
            ky[u] = ok ? rowl[j] : -1;
            v[u] = src[ok ? (size_t)j * W + c : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ky[u] >= 0) a[ky[u] * SW + ((e0 + u * ANY_THREADS) & (SW - 1))] = v[u];
    }
    __syncthreads();
    any_fft<1>(a, b, tw, H, nl, SW, SW, plan);
    float* outr = out + (size_t)frame * 2 * H * W;
    float* outi = outr + (size_t)H * W;
    for (int e = tid; e < H * SW; e += ANY_THREADS) {
        const int i = e >> lgSW, c = e & (SW - 1);
        if (c < nl) {
            const cf32 v = a[e];
            outr[(size_t)i * W + c0 + c] = v.x;
            outi[(size_t)i * W + c0 + c] = v.y;
        }
    }
}

// ysq[f][ch] = sum of |y|^2 over the masked-out rows of chunk ch (of ANY_YCH) of measurement frame f (where
// A(x) = 0, the residual is -y)
#define ANY_YCH 16
__global__ __launch_bounds__(256) void k_any_ysq(const float* __restrict__ y, const int* __restrict__ rows,
                                                 const int* __restrict__ nrows, int mask_frames, int T, int H, int W,
                                                 float* __restrict__ ysq) {
    __shared__ int kept[FFT_ANY_MAX_N];
    __shared__ float red[16];
    const int f = blockIdx.x / ANY_YCH, ch = blockIdx.x - f * ANY_YCH, t = f % T, tid = threadIdx.x;
    const int mf = mask_frames == 1 ? 0 : t;
    for (int e = tid; e < H; e += 256) kept[e] = 0;
    __syncthreads();
    for (int e = tid; e < nrows[mf]; e += 256) kept[rows[(size_t)mf * H + e]] = 1;
    __syncthreads();
    const int r0 = ch * H / ANY_YCH, r1 = (ch + 1) * H / ANY_YCH;
    const float* yr = y + (size_t)f * 2 * H * W;
    const float* yi = yr + (size_t)H * W;
    float acc = 0.f;
    for (int e = r0 * W + tid; e < r1 * W; e += 256) {
        const float p = yr[e], q = yi[e];
        acc += kept[e / W] ? 0.f : p * p + q * q;
    }
    const float tot = block_sum(acc, red);
    if (tid == 0) ysq[blockIdx.x] = tot;
}

// loss[b] = sum_t (sum_groups part[b][t][.] + sum_chunks ysq[yb][t][.]), total = sum_b loss[b]: one block, fixed order
__global__ __launch_bounds__(256) void k_any_finish(const float* __restrict__ part, const float* __restrict__ ysq,
                                                    int y_batch, int B, int T, int ngroups,
                                                    float* __restrict__ loss, float* __restrict__ total) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float mine = 0.f;
    for (int b = tid; b < B; b += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) {
            const float* p = part + ((size_t)b * T + t) * ngroups;
            const float* u = ysq + ((size_t)(y_batch > 1 ? b : 0) * T + t) * ANY_YCH;
            float f = 0.f, m = 0.f;
            for (int g = 0; g < ngroups; ++g) f += p[g];
            for (int c = 0; c < ANY_YCH; ++c) m += u[c];
            s += f + m;
        }
        loss[b] = s;
        mine += s;
    }
    if (total) {
        red[tid] = mine;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) red[tid] += red[tid + w];
            __syncthreads();
        }
        if (tid == 0) *total = red[0];
    }
}

// ---------------------------------------------------------------------------------------------- C ABI
struct AnyGeom {
    FftAnyPlan ph, pw;
    int SW, lgSW, nstrips, NR, ngroups;
    size_t lds_c, lds_r;
    size_t rows_bytes, part_bytes, ysq_bytes;
};

static int any_geom(const char* who, int B, int T, int H, int W, int max_rows, AnyGeom* g) {
    STEP_REQUIRE(B > 0 && T > 0, "%s: bad sizes B=%d T=%d", who, B, T);
    STEP_REQUIRE(H >= 2 && W >= 2 && H <= FFT_ANY_MAX_N && W <= FFT_ANY_MAX_N, "%s: frame %d x %d (2 .. %d each)", who,
                 H, W, FFT_ANY_MAX_N);
    STEP_REQUIRE(max_rows >= 0 && max_rows <= H, "%s: max_rows %d of %d rows", who, max_rows, H);
    STEP_REQUIRE(fft_any_plan(H, &g->ph) && fft_any_plan(W, &g->pw), "%s: no pass plan for %d x %d", who, H, W);
    const int mr = max_rows < 1 ? 1 : max_rows;
    g->SW = fft_any_lines(H, &g->ph, ANY_THREADS, 0, ANY_LDS_BUF);
    g->NR = fft_any_lines(W, &g->pw, ANY_THREADS, 1, ANY_LDS_BUF);
    STEP_REQUIRE(g->SW > 0 && g->NR > 0, "%s: %d x %d does not fit the line buffers", who, H, W);
    while (g->SW / 2 >= W) g->SW /= 2;                // narrow frames: no wider than needed
    while (g->NR / 2 >= mr) g->NR /= 2;
    g->lgSW = 0;
    while ((1 << g->lgSW) < g->SW) ++g->lgSW;
    g->nstrips = (W + g->SW - 1) / g->SW;
    g->ngroups = (mr + g->NR - 1) / g->NR;
    STEP_REQUIRE((long long)B * T * (g->nstrips > g->ngroups ? g->nstrips : g->ngroups) < (1ll << 31),
                 "%s: %d x %d frames make too large a grid", who, B, T);
    g->lds_c = any_lds_bytes(H, g->SW, 0, g->ph.direct);
    g->lds_r = any_lds_bytes(W, g->NR + 1, g->NR, g->pw.direct);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    g->rows_bytes = up((size_t)B * T * mr * W * sizeof(cf32));
    g->part_bytes = up((size_t)B * T * g->ngroups * sizeof(float));
    g->ysq_bytes = up((size_t)B * T * ANY_YCH * sizeof(float));
    return STEP_OK;
}

static int any_check_mask(const char* who, int T, const int32_t* rows, const int32_t* nrows, int mask_frames) {
    STEP_REQUIRE(rows && nrows, "%s: null mask row tables", who);
    STEP_REQUIRE(mask_frames == 1 || mask_frames == T, "%s: mask_frames=%d must be 1 or T=%d", who, mask_frames, T);
    return STEP_OK;
}

extern "C" size_t mri_any_ws_bytes(int B, int T, int H, int W, int max_rows) {
    AnyGeom g;
    if (any_geom("mri_any_ws_bytes", B, T, H, W, max_rows, &g)) return 0;
    return g.rows_bytes + g.part_bytes + g.ysq_bytes;
}

#define ANY_WS_CHECK(who)                                                                                   \
    STEP_REQUIRE(ws && ws_bytes >= geo.rows_bytes + geo.part_bytes + geo.ysq_bytes,                         \
                 "%s: workspace of %zu bytes, %zu needed (mri_any_ws_bytes)", who, ws_bytes,                \
                 geo.rows_bytes + geo.part_bytes + geo.ysq_bytes);                                          \
    STEP_REQUIRE(((uintptr_t)ws & 15) == 0, "%s: workspace must be 16-byte aligned", who)

#define ANY_COLS_FWD(xin)                                                                                     \
    step_allow_lds((const void*)k_any_cols_fwd, geo.lds_c);                                                   \
    hipLaunchKernelGGL(k_any_cols_fwd, dim3((unsigned)nframes * geo.nstrips), dim3(ANY_THREADS), geo.lds_c, s, xin, \
                       rows, nrows, mask_frames, T, H, W, geo.SW, geo.lgSW, geo.nstrips, max_rows, geo.ph, (cf32*)ws); \
    STEP_CHECK_LAUNCH(who)
#define ANY_COLS_INV(dst)                                                                                     \
    step_allow_lds((const void*)k_any_cols_inv, geo.lds_c);                                                   \
    hipLaunchKernelGGL(k_any_cols_inv, dim3((unsigned)nframes * geo.nstrips), dim3(ANY_THREADS), geo.lds_c, s, \
                       (const cf32*)ws, rows, nrows, mask_frames, T, H, W, geo.SW, geo.lgSW, geo.nstrips, max_rows, \
                       geo.ph, dst);                                                                          \
    STEP_CHECK_LAUNCH(who)
#define ANY_ROWS(MODE, yy, yb, gg, cs, oo, pp)                                                                \
    step_allow_lds((const void*)k_any_rows<MODE>, geo.lds_r);                                                 \
    hipLaunchKernelGGL(k_any_rows<MODE>, dim3((unsigned)nframes * geo.ngroups), dim3(ANY_THREADS), geo.lds_r, s, \
                       (cf32*)ws, yy, yb, gg, rows, nrows, mask_frames, T, H, W, geo.NR, geo.ngroups, max_rows, \
                       geo.pw, cs, oo, pp);                                                                   \
    STEP_CHECK_LAUNCH(who)

extern "C" int mri_forward_any(const float* x, int B, int T, int H, int W, const int32_t* rows,
                               const int32_t* nrows, int mask_frames, int max_rows, void* ws, size_t ws_bytes,
                               float* out, void* stream) {
    const char* who = "mri_forward_any";
    STEP_REQUIRE(x && out, "%s: null pointer", who);
    AnyGeom geo;
    if (int rc = any_geom(who, B, T, H, W, max_rows, &geo)) return rc;
    if (int rc = any_check_mask(who, T, rows, nrows, mask_frames)) return rc;
    ANY_WS_CHECK(who);
    hipStream_t s = (hipStream_t)stream;
    const int nframes = B * T;
    if (hipMemsetAsync(out, 0, (size_t)nframes * 2 * H * W * sizeof(float), s) != hipSuccess) {
        step_set_error("%s: clearing the output failed", who);
        return STEP_E_LAUNCH;
    }
    if (max_rows == 0) return STEP_OK;
    ANY_COLS_FWD(x);
    ANY_ROWS(0, (const float*)nullptr, 1, (const float*)nullptr, (const float*)nullptr, out, (float*)nullptr);
    return STEP_OK;
}

extern "C" int mri_adjoint_any(const float* g, int B, int T, int H, int W, const int32_t* rows,
                               const int32_t* nrows, int mask_frames, int max_rows, void* ws, size_t ws_bytes,
                               float* out, void* stream) {
    const char* who = "mri_adjoint_any";
    STEP_REQUIRE(g && out, "%s: null pointer", who);
    AnyGeom geo;
    if (int rc = any_geom(who, B, T, H, W, max_rows, &geo)) return rc;
    if (int rc = any_check_mask(who, T, rows, nrows, mask_frames)) return rc;
    ANY_WS_CHECK(who);
    hipStream_t s = (hipStream_t)stream;
    const int nframes = B * T;
    if (max_rows > 0) {
        ANY_ROWS(2, (const float*)nullptr, 1, g, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    }
    ANY_COLS_INV(out);
    return STEP_OK;
}

extern "C" int mri_loss_grad_any(const float* x, const float* y, int y_batch, int B, int T, int H, int W,
                                 const int32_t* rows, const int32_t* nrows, int mask_frames, int max_rows,
                                 const float* chain_scale, void* ws, size_t ws_bytes, float* loss,
                                 float* loss_total, float* grad, void* stream) {
    const char* who = "mri_loss_grad_any";
    STEP_REQUIRE(x && y && loss, "%s: null pointer", who);
    STEP_REQUIRE(y_batch == 1 || y_batch == B, "%s: measurement batch %d vs B=%d", who, y_batch, B);
    AnyGeom geo;
    if (int rc = any_geom(who, B, T, H, W, max_rows, &geo)) return rc;
    if (int rc = any_check_mask(who, T, rows, nrows, mask_frames)) return rc;
    ANY_WS_CHECK(who);
    hipStream_t s = (hipStream_t)stream;
    const int nframes = B * T;
    float* part = (float*)((char*)ws + geo.rows_bytes);
    float* ysq = (float*)((char*)ws + geo.rows_bytes + geo.part_bytes);
    hipLaunchKernelGGL(k_any_ysq, dim3((unsigned)y_batch * T * ANY_YCH), dim3(256), 0, s, y, rows, nrows, mask_frames,
                       T, H, W, ysq);
    STEP_CHECK_LAUNCH(who);
    ANY_COLS_FWD(x);
    // `out` of MODE 1 only says whether the inverse row pass is wanted
    ANY_ROWS(1, y, y_batch, (const float*)nullptr, chain_scale, grad, part);
    if (grad) {
        ANY_COLS_INV(grad);
    }
    hipLaunchKernelGGL(k_any_finish, dim3(1), dim3(256), 0, s, part, ysq, y_batch, B, T, geo.ngroups, loss, loss_total);
    STEP_CHECK_LAUNCH(who);
    return STEP_OK;
}
