// The oversampled-grid transforms of the radial-MRI NUFFT (nufft_grid.hip) for R = 192, G = 384,
// fused with the padding / cropping and the de-apodisation and pruned to what is not zero.
// NO REFERENCE COUNTERPART (see nufft_grid.hip).
//
// A library FFT2 of the zero-padded grid moves every frame's 1.18 MB grid through HBM four times
// (pad, two transposed passes each way) and copies it once more; three quarters of what it
// transforms is zero (forward) or thrown away (adjoint).  Here each direction is two passes:
//
//   forward:  image rows    --pad + deapodise, FFT_384 along x-->  T [R][G]   (only the R rows that exist)
//             T columns     --FFT_384 along y over the R non-zero rows-->      grid [G][G]
//   adjoint:  grid columns  --IFFT_384 along y, only the R rows kept-->        T [R][G]
//             T rows        --IFFT_384 along x, crop + deapodise-->            image rows
//
// per frame 295 KB + 2 x 590 KB + 1.18 MB each way instead of ~8 x 1.18 MB.  384 = 16 x 24
// (fft_small.h: fft384_pass_a / fft384_pass_b, unit-tested on the host); pixel n of an axis sits at
// grid index (n - R/2) mod G, so of the 16 inputs 24 a + b of a pass-A task exactly a < 4 and
// a >= 12 are image samples, and of the 24 outputs c + 16 d of a pass-B task exactly d < 6 and
// d >= 18 are image rows.  Column passes: 32 columns per workgroup, tasks laid out column-fastest
// so that every global access is a 256-byte row segment and the LDS exchange is conflict-free;
// row passes: 8 rows per workgroup through one LDS tile (five workgroups per CU; 16 rows and two
// workgroups were 20 % slower), loads and stores as whole rows.
#include "common.h"
#include <algorithm>
#include "fft_small.h"

#define NF_R 192
#define NF_G 384
#define NF_H 96
#define NF_CW 32             // columns per workgroup (column passes; 16 measured the same)
#define NF_CT (24 * NF_CW)    // threads: 24 pass-A tasks per column
#define NF_RW 8              // rows per workgroup (row passes)
#define NF_RT (24 * NF_RW)    // threads: 24 pass-A tasks per row
#define NF_RS 408            // LDS row stride of the row passes (slots 25 c + b <= 398; 16 banks between rows)

__device__ __forceinline__ void nf_twiddles(cf32* tw, int tid, int nthreads) {
    for (int m = tid; m < NF_G; m += nthreads) {
        float sn, cs;
        sincospif((float)m * (1.0f / 192.0f), &sn, &cs);      // 2 pi m / 384
        tw[m] = c_make(cs, sn);
    }
}

// SIGN = -1: in = T [f][R][G] (image rows), out = grid [f][G][G]
// SIGN = +1: in = grid [f][G][G], out = T [f][R][G] (image rows), unnormalised
// rows of a 16-column tile column that can hold anything, as bounds on the centred tile-row coordinate
// (row >> 1, minus 192 from 96 on), packed (lo & 0xffff) | (hi << 16) per (trajectory frame, tile column):
// by value in the kernel arguments, so that the bounds cost no trip to memory ahead of the loads
#define NF_RANGE_FRAMES 12
struct NfRange {
    int frames;                                               // 0: no bounds (every row is read)
    int T;
    int v[NF_RANGE_FRAMES * (NF_G / 16)];
};

template <int SIGN>
__global__ __launch_bounds__(NF_CT) void k_nufft_fft_cols(const cf32* __restrict__ in, cf32* __restrict__ out,
                                                          const NfRange rg) {
    extern __shared__ __attribute__((aligned(16))) cf32 nf_lds[];
    cf32* tw = nf_lds;                                        // [384]
    cf32* E = nf_lds + NF_G;                                  // [384 slots][32 columns]
    const int tid = threadIdx.x, col = tid & (NF_CW - 1), q = tid / NF_CW;
    const int f = blockIdx.y, kx = blockIdx.x * NF_CW + col;
    cf32 v[16];
    {
        const int b = q;                                      // pass-A task: grid rows 24 a + b
        if (SIGN < 0) {
            const cf32* src = in + (size_t)f * NF_R * NF_G + kx;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                const int n = 24 * a + b;                     // a < 4: pixel row n + 96, a >= 12: n - 288
                v[a] = a < 4 ? src[(size_t)(n + NF_H) * NF_G]
                             : (a >= 12 ? src[(size_t)(n - (NF_G - NF_H)) * NF_G] : c_make(0.f, 0.f));
            }
        } else {
            const cf32* src = in + (size_t)f * NF_G * NF_G + kx;
            int lo = -32768, hi = 32767;
            if (rg.frames) {                                  // nufft_adjoint: the spread left the rows outside unwritten
                const int tf = rg.frames == 1 ? 0 : f % rg.T;
                const int pk = rg.v[tf * (NF_G / 16) + (kx >> 4)];
                lo = (short)(pk & 0xffff);
                hi = pk >> 16;
            }
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                const int tr = (24 * a + b) >> 1, cr = tr < NF_G / 4 ? tr : tr - NF_G / 2;
                v[a] = (cr >= lo && cr <= hi) ? src[(size_t)(24 * a + b) * NF_G] : c_make(0.f, 0.f);
            }
        }
        nf_twiddles(tw, tid, NF_CT);
        __syncthreads();
        fft384_pass_a<SIGN>(v, b, tw);
#pragma unroll
        for (int c = 0; c < 16; ++c) E[(24 * c + b) * NF_CW + col] = v[c];
    }
    __syncthreads();
    if (q < 16) {
        const int c = q;                                      // pass-B task: grid rows c + 16 d
        cf32 u[24];
#pragma unroll
        for (int b = 0; b < 24; ++b) u[b] = E[(24 * c + b) * NF_CW + col];
        fft384_pass_b<SIGN>(u);
        if (SIGN < 0) {
            cf32* dst = out + (size_t)f * NF_G * NF_G + kx;
#pragma unroll
            for (int d = 0; d < 24; ++d) dst[(size_t)(c + 16 * d) * NF_G] = u[d];
        } else {
            cf32* dst = out + (size_t)f * NF_R * NF_G + kx;
#pragma unroll
            for (int d = 0; d < 24; ++d) {
                const int k = c + 16 * d;                     // d < 6: pixel row k + 96, d >= 18: k - 288
                if (d < 6) dst[(size_t)(k + NF_H) * NF_G] = u[d];
                else if (d >= 18) dst[(size_t)(k - (NF_G - NF_H)) * NF_G] = u[d];
            }
        }
    }
}

// The forward column pass and the interpolation in one kernel: the grid never reaches HBM.  A
// workgroup owns a strip of 16 grid columns of one frame plus the W - 1 columns a footprint that starts
// in the strip can reach (a third of the column transforms are done twice, but two workgroups share a CU), runs the two passes of
// k_nufft_fft_cols<-1> on them, leaves the pass-B outputs in the exchange tile (thread (c, col) owns the
// 24 slots 24 c + b it has just read: row c + 16 d goes to slot 24 c + d, no barrier in between) and then
// takes the samples whose footprint starts in the strip -- binned by strip on the host, tables in bin
// order like the owner bins of k_nufft_interp_tiled, the first sample's tables requested ahead of the
// transforms.  Per frame 590 KB x (16 + W - 1) / 16 in instead of 590 KB in, 1.18 MB out and 1.47 MB in
// again.  The 36 products are summed in the order of k_nufft_interp.
#define NF_SC 16             // strip width: 74 KB of LDS, two workgroups per CU (32: one, 0.39 ms against 0.33)
#define NF_ES (NF_SC + 7)     // slot stride of the exchange tile: NF_SC + 7 columns at most, odd
struct NfStrips {
    int W, J, traj_frames;
    const int32_t* ptr;       // [traj_frames][G / NF_SC + 1]
    const int32_t* ent;       // [traj_frames][J] sample ids by strip
    const int32_t* x0;        // [traj_frames][J] first point (wrapped into [0, G)), in the order of ent
    const int32_t* y0;
    const float* wx;          // [traj_frames][J][W]
    const float* wy;
};
__global__ __launch_bounds__(1024) void k_nufft_cols_interp(const cf32* __restrict__ in, NfStrips p, int T,
                                                            const float2* __restrict__ y, int y_batch, float res_scale,
                                                            float2* __restrict__ out, float* __restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) cf32 nf_lds[];
    __shared__ float red[16];
    cf32* tw = nf_lds;                                        // [384]
    cf32* E = nf_lds + NF_G;                                  // [384 slots][NF_ES]
    const int W = p.W, ncol = NF_SC + W - 1, nthr = blockDim.x;
    const int tid = threadIdx.x, q = tid / ncol, col = tid - q * ncol;
    const int strip = blockIdx.x, f = blockIdx.y, t = f % T, b = f / T, tf = p.traj_frames == 1 ? 0 : t;
    const int nstrips = NF_G / NF_SC;
    // this thread's first sample: its tables travel under the transforms
    const int32_t* ptr = p.ptr + (size_t)tf * (nstrips + 1);
    const int e0 = ptr[strip], e1 = ptr[strip + 1];
    const size_t tb = (size_t)tf * p.J;
    int s = e0 + tid, j = 0, lx = 0, ly = 0;
    bool have = s < e1;
    float wxr[8], wyr[8];
    float2 m = make_float2(0.f, 0.f);
    auto fetch = [&]() {
        j = p.ent[tb + s];
        lx = p.x0[tb + s] - strip * NF_SC;
        ly = p.y0[tb + s];
        const float* wxp = p.wx + (tb + s) * W;
        const float* wyp = p.wy + (tb + s) * W;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wxr[k] = k < W ? wxp[k] : 0.f;
            wyr[k] = k < W ? wyp[k] : 0.f;
        }
        if (y) m = y[((size_t)(y_batch > 1 ? b : 0) * T + t) * p.J + j];
    };
    if (have) fetch();
    int kx = strip * NF_SC + col;
    kx = kx >= NF_G ? kx - NF_G : kx;
    {
        const int bq = q;                                     // pass-A task: grid rows 24 a + b (threads with q >= 24: none)
        cf32 v[16];
        if (q < 24) {
            const cf32* src = in + (size_t)f * NF_R * NF_G + kx;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                const int n = 24 * a + bq;
                v[a] = a < 4 ? src[(size_t)(n + NF_H) * NF_G]
                             : (a >= 12 ? src[(size_t)(n - (NF_G - NF_H)) * NF_G] : c_make(0.f, 0.f));
            }
        }
        nf_twiddles(tw, tid, nthr);
        __syncthreads();                                      // barriers outside the role tests: a wave may hold both roles
        if (q < 24) {
            fft384_pass_a<-1>(v, bq, tw);
#pragma unroll
            for (int c = 0; c < 16; ++c) E[(24 * c + bq) * NF_ES + col] = v[c];
        }
    }
    __syncthreads();
    if ((tid >> 5) < 16 && (tid & 31) < ncol) {
        // pass-B task: grid rows c + 16 d -> slots 24 c + d.  Half a wave per c (lanes = consecutive columns of one
        // slot: no bank conflict; with (c, col) = (tid / ncol, tid % ncol) neighbouring c landed 8 banks apart --
        // 18 % of the kernel's cycles by the counter, and yet the same 0.33 ms either way)
        const int c = tid >> 5, cb = tid & 31;
        cf32 u[24];
#pragma unroll
        for (int bb = 0; bb < 24; ++bb) u[bb] = E[(24 * c + bb) * NF_ES + cb];
        fft384_pass_b<-1>(u);
#pragma unroll
        for (int d = 0; d < 24; ++d) E[(24 * c + d) * NF_ES + cb] = u[d];
    }
    __syncthreads();
    float acc = 0.f;
    while (have) {
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a)
            if (a < W) {
                int gy = ly + a;
                gy = gy >= NF_G ? gy - NF_G : gy;
                const cf32* row = E + (24 * (gy & 15) + (gy >> 4)) * NF_ES + lx;
                float rr = 0.f, ri = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < W) {
                        const cf32 g = row[k];
                        rr += wxr[k] * g.x;
                        ri += wxr[k] * g.y;
                    }
                sr += wyr[a] * rr;
                si += wyr[a] * ri;
            }
        float2 o = make_float2(sr, si);
        if (y) {
            const float dr = sr - m.x, di = si - m.y;
            acc += dr * dr + di * di;
            o = make_float2(res_scale * dr, res_scale * di);
        }
        out[(size_t)f * p.J + j] = o;
        s += nthr;                                            // the strips through k = 0 hold more samples than threads
        have = s < e1;
        if (have) fetch();
    }
    if (loss_part) {
        const float tot = block_sum(acc, red);
        if (tid == 0) loss_part[(size_t)f * nstrips + strip] = tot;
    }
}

// The two passes of NF_RW rows held in one LDS tile (row stride NF_RS): in place, natural order.
// PRUNED_IN: only grid indices < 96 and >= 288 of a row hold data (the rest is taken as zero and
// need not be initialised).
template <int SIGN, bool PRUNED_IN>
__device__ __forceinline__ void nf_rows_in_lds(cf32* buf, const cf32* tw, int tid) {
    cf32 v[16];
    const int row = tid / 24, b = tid - row * 24;
    cf32* r = buf + row * NF_RS;
#pragma unroll
    for (int a = 0; a < 16; ++a)
        v[a] = (!PRUNED_IN || a < 4 || a >= 12) ? r[24 * a + b] : c_make(0.f, 0.f);
    fft384_pass_a<SIGN>(v, b, tw);
    __syncthreads();                                          // every input has been read
#pragma unroll
    for (int c = 0; c < 16; ++c) r[25 * c + b] = v[c];       // slot stride 25: pass B reads conflict-free
    __syncthreads();
    cf32 u[24];
    const int row2 = tid / 16, c2 = tid - row2 * 16;
    cf32* r2 = buf + row2 * NF_RS;
    if (tid < NF_RW * 16) {
#pragma unroll
        for (int bb = 0; bb < 24; ++bb) u[bb] = r2[25 * c2 + bb];
        fft384_pass_b<SIGN>(u);
    }
    __syncthreads();                                          // every slot has been read
    if (tid < NF_RW * 16) {
#pragma unroll
        for (int d = 0; d < 24; ++d) r2[c2 + 16 * d] = u[d];
    }
    __syncthreads();
}

// T[f][y][kx] = FFT_384 over gx of (apod[y] apod[x] (x_re + i x_im)[y][x] placed at gx = (x - R/2) mod G)
__global__ __launch_bounds__(NF_RT) void k_nufft_rows_fwd(const float* __restrict__ x, const float* __restrict__ apod,
                                                          cf32* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) cf32 nf_lds[];
    cf32* tw = nf_lds;
    cf32* buf = nf_lds + NF_G;
    const int tid = threadIdx.x, f = blockIdx.y, y0 = blockIdx.x * NF_RW;
    const float* xr = x + (size_t)f * 2 * NF_R * NF_R;
    const float* xi = xr + (size_t)NF_R * NF_R;
    nf_twiddles(tw, tid, NF_RT);
    static_assert(NF_RT == NF_R, "one thread per pixel column");
    {
        const int n = tid, g = n < NF_H ? n + (NF_G - NF_H) : n - NF_H;
        const float an = apod[n];
        float vr[NF_RW], vi[NF_RW];
#pragma unroll
        for (int row = 0; row < NF_RW; ++row) {
            vr[row] = xr[(y0 + row) * NF_R + n];
            vi[row] = xi[(y0 + row) * NF_R + n];
        }
#pragma unroll
        for (int row = 0; row < NF_RW; ++row) {
            const float a = apod[y0 + row] * an;
            buf[row * NF_RS + g] = c_make(a * vr[row], a * vi[row]);
        }
    }
    __syncthreads();
    nf_rows_in_lds<-1, true>(buf, tw, tid);
    cf32* dst = T + ((size_t)f * NF_R + y0) * NF_G;
#pragma unroll
    for (int row = 0; row < NF_RW; ++row) {
        dst[(size_t)row * NF_G + tid] = buf[row * NF_RS + tid];
        dst[(size_t)row * NF_G + NF_R + tid] = buf[row * NF_RS + NF_R + tid];
    }
}

// out[f][0/1][y][x] = scale * apod[y] apod[x] * Re/Im IFFT_384(T[f][y][:]) at gx = (x - R/2) mod G
__global__ __launch_bounds__(NF_RT) void k_nufft_rows_inv(const cf32* __restrict__ T, const float* __restrict__ apod,
                                                          int frames_per_chain, const float* __restrict__ chain_scale,
                                                          float out_scale, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) cf32 nf_lds[];
    cf32* tw = nf_lds;
    cf32* buf = nf_lds + NF_G;
    const int tid = threadIdx.x, f = blockIdx.y, y0 = blockIdx.x * NF_RW;
    const float cs = out_scale * (chain_scale ? chain_scale[f / frames_per_chain] : 1.f);
    nf_twiddles(tw, tid, NF_RT);
    const cf32* src = T + ((size_t)f * NF_R + y0) * NF_G;
    {
        cf32 v[2 * NF_RW];                                    // all of the tile in flight at once
#pragma unroll
        for (int row = 0; row < NF_RW; ++row) {
            v[2 * row] = src[(size_t)row * NF_G + tid];
            v[2 * row + 1] = src[(size_t)row * NF_G + NF_R + tid];
        }
#pragma unroll
        for (int row = 0; row < NF_RW; ++row) {
            buf[row * NF_RS + tid] = v[2 * row];
            buf[row * NF_RS + NF_R + tid] = v[2 * row + 1];
        }
    }
    __syncthreads();
    nf_rows_in_lds<1, false>(buf, tw, tid);
    float* outr = out + (size_t)f * 2 * NF_R * NF_R;
    float* outi = outr + (size_t)NF_R * NF_R;
    {
        const int n = tid, g = n < NF_H ? n + (NF_G - NF_H) : n - NF_H;
        const float an = cs * apod[n];
#pragma unroll
        for (int row = 0; row < NF_RW; ++row) {
            const float a = an * apod[y0 + row];
            const cf32 v = buf[row * NF_RS + g];
            outr[(y0 + row) * NF_R + n] = a * v.x;
            outi[(y0 + row) * NF_R + n] = a * v.y;
        }
    }
}

static const size_t NF_LDS_COLS = (size_t)(NF_G + NF_G * NF_CW) * sizeof(cf32);
static const size_t NF_LDS_ROWS = (size_t)(NF_G + NF_RW * NF_RS) * sizeof(cf32);

static int nf_check(const char* who, const nufft_plan_t* p, int nframes, int T) {
    STEP_REQUIRE(p && p->apod, "%s: null plan", who);
    STEP_REQUIRE(p->R == NF_R && p->G == NF_G, "%s: built for R = %d, G = %d (got R = %d, G = %d)", who, NF_R, NF_G,
                 p->R, p->G);
    STEP_REQUIRE(nframes > 0 && nframes <= 65535 && T > 0 && nframes % T == 0, "%s: %d frames (1..65535), T = %d", who,
                 nframes, T);
    return STEP_OK;
}

extern "C" int nufft_grid_fft_supported(int R, int G) { return R == NF_R && G == NF_G; }

extern "C" int nufft_fwd_grid(const float* x, int nframes, int T, const nufft_plan_t* plan, float* work_c64,
                              float* grid_c64, void* stream) {
    STEP_REQUIRE(x && work_c64 && grid_c64, "nufft_fwd_grid: null pointer");
    if (int rc = nf_check("nufft_fwd_grid", plan, nframes, T)) return rc;
    hipStream_t st = (hipStream_t)stream;
    step_allow_lds((const void*)k_nufft_rows_fwd, NF_LDS_ROWS);
    hipLaunchKernelGGL(k_nufft_rows_fwd, dim3(NF_R / NF_RW, nframes), dim3(NF_RT), NF_LDS_ROWS, st, x, plan->apod,
                       (cf32*)work_c64);
    STEP_CHECK_LAUNCH("nufft_fwd_grid (rows)");
    step_allow_lds((const void*)k_nufft_fft_cols<-1>, NF_LDS_COLS);
    hipLaunchKernelGGL((k_nufft_fft_cols<-1>), dim3(NF_G / NF_CW, nframes), dim3(NF_CT), NF_LDS_COLS, st,
                       (const cf32*)work_c64, (cf32*)grid_c64, NfRange{});
    STEP_CHECK_LAUNCH("nufft_fwd_grid (columns)");
    return STEP_OK;
}

static int nf_inv_grid(const char* who, const float* grid_c64, int nframes, int T, const nufft_plan_t* plan,
                       const float* chain_scale, float out_scale, float* work_c64, float* out, const NfRange& rg,
                       void* stream) {
    hipStream_t st = (hipStream_t)stream;
    step_allow_lds((const void*)k_nufft_fft_cols<1>, NF_LDS_COLS);
    hipLaunchKernelGGL((k_nufft_fft_cols<1>), dim3(NF_G / NF_CW, nframes), dim3(NF_CT), NF_LDS_COLS, st,
                       (const cf32*)grid_c64, (cf32*)work_c64, rg);
    STEP_CHECK_LAUNCH(who);
    step_allow_lds((const void*)k_nufft_rows_inv, NF_LDS_ROWS);
    hipLaunchKernelGGL(k_nufft_rows_inv, dim3(NF_R / NF_RW, nframes), dim3(NF_RT), NF_LDS_ROWS, st,
                       (const cf32*)work_c64, plan->apod, T, chain_scale, out_scale, out);
    STEP_CHECK_LAUNCH(who);
    return STEP_OK;
}

extern "C" int nufft_inv_grid(const float* grid_c64, int nframes, int T, const nufft_plan_t* plan,
                              const float* chain_scale, float out_scale, float* work_c64, float* out,
                              void* stream) {
    STEP_REQUIRE(grid_c64 && work_c64 && out, "nufft_inv_grid: null pointer");
    if (int rc = nf_check("nufft_inv_grid", plan, nframes, T)) return rc;
    return nf_inv_grid("nufft_inv_grid", grid_c64, nframes, T, plan, chain_scale, out_scale, work_c64, out, NfRange{}, stream);
}

extern "C" int nufft_adjoint_supported(const nufft_plan_t* plan, int B) {
    return plan && plan->R == NF_R && plan->G == NF_G && plan->wide_rec && plan->wide_ent && plan->wide_w && B >= 16 &&
           plan->traj_frames >= 1 && plan->traj_frames <= NF_RANGE_FRAMES;
}

extern "C" int nufft_adjoint(const float* vals_c64, int nframes, int T, const nufft_plan_t* plan,
                             const int32_t* range_host, const float* chain_scale, float out_scale, float* grid_c64,
                             float* work_c64, float* out, void* stream) {
    STEP_REQUIRE(vals_c64 && range_host && grid_c64 && work_c64 && out, "nufft_adjoint: null pointer");
    if (int rc = nf_check("nufft_adjoint", plan, nframes, T)) return rc;
    STEP_REQUIRE(nufft_adjoint_supported(plan, nframes / T),
                 "nufft_adjoint: needs R = %d, the lane-per-chain lists, at least 16 chains and at most %d trajectory frames",
                 NF_R, NF_RANGE_FRAMES);
    STEP_REQUIRE(plan->traj_frames == 1 || plan->traj_frames == T, "nufft_adjoint: traj_frames = %d must be 1 or T = %d",
                 plan->traj_frames, T);
    NfRange rg;
    rg.frames = plan->traj_frames;
    rg.T = T;
    for (int i = 0; i < plan->traj_frames * (NF_G / 16); ++i) {
        const int lo = range_host[2 * i], hi = range_host[2 * i + 1];
        STEP_REQUIRE(lo >= -NF_G / 4 && lo <= NF_G / 4 && hi >= -NF_G / 4 && hi <= NF_G / 4, "nufft_adjoint: range %d = [%d, %d]", i, lo, hi);
        rg.v[i] = (lo & 0xffff) | (hi << 16);
    }
    if (int rc = nufft_spread_impl(vals_c64, nframes, T, plan, grid_c64, 1, stream)) return rc;
    return nf_inv_grid("nufft_adjoint", grid_c64, nframes, T, plan, chain_scale, out_scale, work_c64, out, rg, stream);
}

// parts of loss_part per frame for nufft_fwd_interp, 0 if the plan has no strip bins
extern "C" int nufft_fwd_interp_parts(const nufft_plan_t* plan) {
    const bool ok = plan && plan->strip_ptr && plan->strip_ent && plan->strip_x0 && plan->strip_y0 && plan->strip_wx &&
                    plan->strip_wy && plan->R == NF_R && plan->G == NF_G && plan->W >= 2 && plan->W <= 8;
    return ok ? NF_G / NF_SC : 0;
}

extern "C" int nufft_fwd_interp(const float* x, int nframes, int T, const nufft_plan_t* plan, const float* y_c64,
                                int y_batch, float res_scale, float* work_c64, float* out_c64, float* loss_part,
                                float* loss, float* loss_total, void* stream) {
    STEP_REQUIRE(x && work_c64 && out_c64, "nufft_fwd_interp: null pointer");
    if (int rc = nf_check("nufft_fwd_interp", plan, nframes, T)) return rc;
    const int parts = nufft_fwd_interp_parts(plan), B = nframes / T;
    STEP_REQUIRE(parts > 0, "nufft_fwd_interp: the plan has no strip bins (or W = %d outside 2..8)", plan->W);
    STEP_REQUIRE(plan->traj_frames == 1 || plan->traj_frames == T, "nufft_fwd_interp: traj_frames = %d must be 1 or T = %d",
                 plan->traj_frames, T);
    STEP_REQUIRE(!y_c64 || y_batch == 1 || y_batch == B, "nufft_fwd_interp: measurement batch %d vs B=%d", y_batch, B);
    STEP_REQUIRE(!loss || (y_c64 && loss_part), "nufft_fwd_interp: the loss needs the measurement and loss_part");
    STEP_REQUIRE(!loss_total || loss, "nufft_fwd_interp: loss_total needs loss");
    hipStream_t st = (hipStream_t)stream;
    step_allow_lds((const void*)k_nufft_rows_fwd, NF_LDS_ROWS);
    hipLaunchKernelGGL(k_nufft_rows_fwd, dim3(NF_R / NF_RW, nframes), dim3(NF_RT), NF_LDS_ROWS, st, x, plan->apod,
                       (cf32*)work_c64);
    STEP_CHECK_LAUNCH("nufft_fwd_interp (rows)");
    const size_t lds = (size_t)(NF_G + NF_G * NF_ES) * sizeof(cf32);
    const int threads = std::max(512, (24 * (NF_SC + plan->W - 1) + 63) / 64 * 64);   // pass A: 24 x columns, pass B: 16 x 32
    step_allow_lds((const void*)k_nufft_cols_interp, lds);
    const NfStrips sp{plan->W, plan->J, plan->traj_frames, plan->strip_ptr, plan->strip_ent, plan->strip_x0,
                      plan->strip_y0, plan->strip_wx, plan->strip_wy};
    STEP_REQUIRE(nframes <= 65535, "nufft_fwd_interp: %d frames exceed the grid", nframes);
    hipLaunchKernelGGL(k_nufft_cols_interp, dim3(parts, nframes), dim3(threads), lds, st, (const cf32*)work_c64, sp, T,
                       (const float2*)y_c64, y_batch, res_scale, (float2*)out_c64, y_c64 ? loss_part : nullptr);
    STEP_CHECK_LAUNCH("nufft_fwd_interp (columns + interpolation)");
    if (loss) {
        if (int rc = nufft_loss_reduce(loss_part, B, T * parts, loss, loss_total, stream)) return rc;
    }
    return STEP_OK;
}
