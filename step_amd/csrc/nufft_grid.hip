// Non-uniform (radial) k-space sampling by gridding: the kernels around the oversampled FFT of a
// NUFFT.  NO REFERENCE COUNTERPART: the reference's MRI operator is Cartesian
// (forward_operator/dynamic_mri.py:42-48) and `pynfft` is only an install line (README.md:28);
// BASELINE.json's north_star names "the non-uniform-sampling path ... reimplemented as a gridding
// kernel".  Oracle: the direct non-uniform DFT in fp64 (oracle/nufft_oracle.py).
//
//   forward  (type 2):  x [R x R]  --pad + deapodise-->  [G x G]  --FFT2-->  --interpolate-->  y_j
//   adjoint  (type 1):  r_j  --spread (gridding)-->  [G x G]  --unnormalised IFFT2-->  --crop + deapodise-->  [R x R]
//
// G = 2 R oversampled grid, separable Kaiser-Bessel kernel of width W (<= 8) whose per-sample
// weights and first grid index are tabulated once per trajectory on the host (fp64 -> fp32), so
// the kernels do no transcendental work.  Pixel n of an axis has the centred coordinate n - R/2
// and sits at grid index (n - R/2) mod G; a sample at kappa cycles per field of view sits at grid
// coordinate 2 kappa.  The grid transforms between the two halves are nufft_fft.hip for R = 192
// (fused with pad / crop, pruned) and otherwise a library FFT2 (hipFFT through torch.fft on the
// same stream) between k_nufft_pad_apod and k_nufft_crop_apod.
//
// Spreading is a GATHER, not a scatter: the host bins the samples by 16 x 4 grid tile (every tile
// a sample's W x W footprint touches); one wave owns a tile of a frame, stages the tile's
// samples through LDS, and every lane sums the contributions to its own grid point in list
// order -- no atomics, bitwise reproducible, the grid is written exactly once, coalesced.
#include "common.h"

#define NU_TILE 16
#define NU_WMAX 8
#define NU_CHUNK 64          // samples staged in LDS at a time

struct NuPlan {
    int R, G, W, J, traj_frames, tiles;       // spread lists: (G / 16) x (G / 4) tiles of 16 x 4 points, row-major
    const int32_t* ix0;                       // [traj_frames][J] first grid column of the footprint (mod G applied later)
    const int32_t* iy0;
    const float* wx;                          // [traj_frames][J][W]
    const float* wy;
    const int32_t* tile_ptr;                  // [traj_frames][tiles*tiles + 1]
    const int32_t* tile_ent;                  // [traj_frames][...] sample ids, ascending per tile
    const int64_t* tile_base;                 // [traj_frames] offset of a trajectory frame's entries in tile_ent
    const float* apod;                        // [R] 1 / kernel transform at pixel n (per axis)
    const int32_t* own_ptr;                   // [traj_frames][(G/64)(G/32) + 1] samples by the 64 x 32 tile that holds
    const int32_t* own_ent;                   // [traj_frames][J]            their footprint's first point, or null;
    const int32_t* own_x0;                    // [traj_frames][J] first point (wrapped into [0, G)) and weights
    const int32_t* own_y0;                    //                  in the order of own_ent
    const float* own_wx;                      // [traj_frames][J][W]
    const float* own_wy;
    int wide_tiles;                           // (G / 16) x (G / 2) tiles of 16 x 2 points (k_nufft_spread_wide), or 0
    const int32_t* wide_rec;                  // [traj_frames][wide_tiles][16] (tile, first entry, entries, outside, first 8 ids, 0 x 4) in launch order
    const int32_t* wide_ent;                  // sample ids, ascending per tile, lists padded to multiples of 4
    const float* wide_w;                      // [entries][2][16] weight of the entry's sample at each point of its tile
};

__device__ __forceinline__ int nu_wrap(int i, int G) { return i < 0 ? i + G : (i >= G ? i - G : i); }

// grid[f][gy][gx] = apod[y] apod[x] (x_re + i x_im)[y][x] at (y - R/2, x - R/2) mod G, zero elsewhere
__global__ __launch_bounds__(256) void k_nufft_pad_apod(const float* __restrict__ x, NuPlan p,
                                                        float2* __restrict__ grid) {
    const int f = blockIdx.y, G = p.G, R = p.R, h = R / 2;
    const size_t fsz = (size_t)R * R;
    const float* xr = x + (size_t)f * 2 * fsz;
    const float* xi = xr + fsz;
    float2* g = grid + (size_t)f * G * G;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < G * G; e += gridDim.x * 256) {
        const int gy = e / G, gx = e - gy * G;
        // grid index -> pixel: indices [0, h) hold pixels h .. R-1, indices [G-h, G) hold pixels 0 .. h-1
        const int py = gy < h ? gy + h : (gy >= G - h ? gy - (G - h) : -1);
        const int px = gx < h ? gx + h : (gx >= G - h ? gx - (G - h) : -1);
        float2 v = make_float2(0.f, 0.f);
        if (py >= 0 && px >= 0) {
            const float a = p.apod[py] * p.apod[px];
            v = make_float2(a * xr[py * R + px], a * xi[py * R + px]);
        }
        g[e] = v;
    }
}

// out[f][j] = sum_{a,b} wy[j][a] wx[j][b] grid[f][iy0+a][ix0+b]; with y != null the residual
// against the measurement is formed in the same pass: out = res_scale * (yhat - y) and the block's
// sum |yhat - y|^2 goes to loss_part[f][blockIdx.x]
__global__ __launch_bounds__(256) void k_nufft_interp(const float2* __restrict__ grid, NuPlan p, int T,
                                                      const float2* __restrict__ y, int y_batch,
                                                      float res_scale, float2* __restrict__ out,
                                                      float* __restrict__ loss_part) {
    __shared__ float red[8];
    const int f = blockIdx.y, b = f / T, t = f - b * T, tf = p.traj_frames == 1 ? 0 : t;
    const int j = blockIdx.x * 256 + threadIdx.x, G = p.G, W = p.W;
    const float2* g = grid + (size_t)f * G * G;
    float acc = 0.f;
    if (j < p.J) {
        const size_t sj = (size_t)tf * p.J + j;
        const int x0 = p.ix0[sj], y0 = p.iy0[sj];
        const float* wx = p.wx + sj * W;
        const float* wy = p.wy + sj * W;
        float wxr[NU_WMAX];
        int cx[NU_WMAX];
#pragma unroll
        for (int k = 0; k < NU_WMAX; ++k) {
            wxr[k] = k < W ? wx[k] : 0.f;
            cx[k] = nu_wrap(x0 + (k < W ? k : 0), G);
        }
        float sr = 0.f, si = 0.f;
        for (int a = 0; a < W; ++a) {
            const float2* row = g + (size_t)nu_wrap(y0 + a, G) * G;
            float rr = 0.f, ri = 0.f;
#pragma unroll
            for (int k = 0; k < NU_WMAX; ++k)
                if (k < W) {
                    const float2 v = row[cx[k]];
                    rr += wxr[k] * v.x;
                    ri += wxr[k] * v.y;
                }
            const float w = wy[a];
            sr += w * rr;
            si += w * ri;
        }
        float2 o = make_float2(sr, si);
        if (y) {
            const float2 m = y[((size_t)(y_batch > 1 ? b : 0) * T + t) * p.J + j];
            const float dr = sr - m.x, di = si - m.y;
            acc = dr * dr + di * di;
            o = make_float2(res_scale * dr, res_scale * di);
        }
        out[(size_t)f * p.J + j] = o;
    }
    if (loss_part) {
        const float tot = block_sum(acc, red);
        if (threadIdx.x == 0) loss_part[(size_t)f * gridDim.x + blockIdx.x] = tot;
    }
}

// The same interpolation with the grid staged through LDS: a workgroup owns a 64 x 32 tile of
// frame index t for NI_CH chains (one), loads the tile plus the W - 1 rows / columns a footprint that
// starts in it can reach (each grid point crosses HBM about 1.25 times instead of being gathered
// 36 times through L2), then takes the samples whose footprint starts in the tile: the host has
// binned them (every sample in exactly one tile) and laid their tables out in bin order, so a
// thread reads its sample's first point and 2 W weights from consecutive memory -- no dependent
// gathers, no staging, no barrier per batch of samples.  One (sample, chain) pair per thread at a
// time, the 36 products from LDS in the order of k_nufft_interp.
// loss_part[f][tile] = the tile's share of sum |yhat - y|^2.
#define NI_TW 64             // tile width: 69-point row segments touch 5 lines of 128 bytes for 552 useful bytes
#define NI_TH 32
#define NI_CH 1              // chains per workgroup (2: 291 us against 279; 23 KB of LDS and 66 registers leave room for six workgroups per CU)
#define NI_SW (NI_TW + NU_WMAX - 1)
#define NI_SH (NI_TH + NU_WMAX - 1)
#define NI_RB ((NI_SH + 3) / 4)   // staged rows per wave and chain
__global__ __launch_bounds__(256) void k_nufft_interp_tiled(const float2* __restrict__ grid, NuPlan p, int T, int B,
                                                            const float2* __restrict__ y, int y_batch,
                                                            float res_scale, float2* __restrict__ out,
                                                            float* __restrict__ loss_part) {
    __shared__ float2 s_g[NI_CH][NI_SH * NI_SW];
    __shared__ float red[256];
    const int G = p.G, W = p.W, ntx = G / NI_TW, ntiles = ntx * (G / NI_TH);
    const int sw = NI_TW + W - 1, sh = NI_TH + W - 1;
    const int tile = blockIdx.x, ty = tile / ntx, tx = tile - ty * ntx;
    const int t = blockIdx.y % T, b0 = (blockIdx.y / T) * NI_CH, tf = p.traj_frames == 1 ? 0 : t;
    const int nc = min(NI_CH, B - b0), tid = threadIdx.x;
    const int32_t* ptr = p.own_ptr + (size_t)tf * (ntiles + 1);
    const int e0 = ptr[tile], e1 = ptr[tile + 1];
    if (e0 == e1) {                                           // nothing starts here (uniform)
        if (loss_part && tid < nc) loss_part[((size_t)(b0 + tid) * T + t) * ntiles + tile] = 0.f;
        return;
    }
    const int q = tid / NI_CH, c = tid - q * NI_CH;           // this thread's sample slot and chain
    const int nslot = 256 / NI_CH;
    const size_t tb = (size_t)tf * p.J;
    const int b = b0 + c;
    // the tables (and the measurement) of this thread's first sample are requested ahead of the
    // grid staging and arrive under it; most tiles hold fewer samples than there are slots
    int s = e0 + q, j = 0, lx = 0, ly = 0;
    bool have = c < nc && s < e1;
    float wxr[NU_WMAX], wyr[NU_WMAX];
    float2 m = make_float2(0.f, 0.f);
    auto fetch = [&]() {
        j = p.own_ent[tb + s];
        lx = p.own_x0[tb + s] - tx * NI_TW;
        ly = p.own_y0[tb + s] - ty * NI_TH;
        const float* wxp = p.own_wx + (tb + s) * W;
        const float* wyp = p.own_wy + (tb + s) * W;
#pragma unroll
        for (int k = 0; k < NU_WMAX; ++k) {
            wxr[k] = k < W ? wxp[k] : 0.f;
            wyr[k] = k < W ? wyp[k] : 0.f;
        }
        if (y) m = y[((size_t)(y_batch > 1 ? b : 0) * T + t) * p.J + j];
    };
    if (have) fetch();
    {
        // A wave takes whole rows of the staged region (rows w, w + 4, ... of every chain): its lanes load the
        // first 64 points of the row, the sw - 64 points beyond are taken eight to a row; everything is in
        // flight before the first LDS store.  No division anywhere: the flat index this loop used to
        // split into (chain, row, column) by run-time divisors cost 1800 vector instructions per wave and
        // made the kernel ALU-bound (77 % of the cycles) while it moved 2.7 TB/s.
        const int wv = tid >> 6, lane = tid & 63;
        const int gx0 = nu_wrap(tx * NI_TW + lane, G);
        const float2* gb = grid + ((size_t)b0 * T + t) * G * G;
        const size_t cstride = (size_t)T * G * G;
        static_assert(4 * NI_RB >= NI_SH && 64 >= NI_SH && NI_TW == 64 && NU_WMAX - 1 <= 8, "row split of the staging loop");
        float2 v[NI_CH][NI_RB];
#pragma unroll
        for (int cc = 0; cc < NI_CH; ++cc)
#pragma unroll
            for (int i = 0; i < NI_RB; ++i) {
                const int ry = wv + 4 * i;
                if (cc < nc && ry < sh) v[cc][i] = gb[cc * cstride + (size_t)nu_wrap(ty * NI_TH + ry, G) * G + gx0];
            }
        const int rx = 64 + (tid & 7);
        const int gx1 = nu_wrap(tx * NI_TW + rx, G);
        float2 v1[NI_CH][2];
#pragma unroll
        for (int cc = 0; cc < NI_CH; ++cc)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ry = (tid >> 3) + 32 * i;
                if (cc < nc && ry < sh && rx < sw) v1[cc][i] = gb[cc * cstride + (size_t)nu_wrap(ty * NI_TH + ry, G) * G + gx1];
            }
#pragma unroll
        for (int cc = 0; cc < NI_CH; ++cc)
#pragma unroll
            for (int i = 0; i < NI_RB; ++i) {
                const int ry = wv + 4 * i;
                if (cc < nc && ry < sh) s_g[cc][ry * sw + lane] = v[cc][i];
            }
#pragma unroll
        for (int cc = 0; cc < NI_CH; ++cc)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ry = (tid >> 3) + 32 * i;
                if (cc < nc && ry < sh && rx < sw) s_g[cc][ry * sw + rx] = v1[cc][i];
            }
    }
    __syncthreads();
    float acc = 0.f;
    while (have) {
        const float2* g0 = s_g[c] + ly * sw + lx;
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int a = 0; a < NU_WMAX; ++a)
            if (a < W) {
                const float2* row = g0 + a * sw;
                float rr = 0.f, ri = 0.f;
#pragma unroll
                for (int k = 0; k < NU_WMAX; ++k)
                    if (k < W) {
                        const float2 v = row[k];
                        rr += wxr[k] * v.x;
                        ri += wxr[k] * v.y;
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
        out[((size_t)b * T + t) * p.J + j] = o;
        s += nslot;
        have = s < e1;
        if (have) fetch();
    }
    if (loss_part) {
#if NI_CH == 1
        const float tot = block_sum(acc, red);                // a fixed tree: deterministic
        if (tid == 0) loss_part[((size_t)b0 * T + t) * ntiles + tile] = tot;
#else
        red[tid] = acc;
        __syncthreads();
        if (tid < nc) {                                       // thread c: its chain's partials, in slot order
            float sum = 0.f;
            for (int i = 0; i < nslot; ++i) sum += red[NI_CH * i + tid];
            loss_part[((size_t)(b0 + tid) * T + t) * ntiles + tile] = sum;
        }
#endif
    }
}

// grid[f][gy][gx] = sum over the samples j (ascending) whose footprint reaches the point of
// wy[j][gy - iy0] wx[j][gx - ix0] v[f][j].
// A workgroup owns a 16 x 16 block of frame index t for NU_CH chains, each of its four waves a
// 16 x 4 tile of it (a wave's 64 lanes = 4 rows of 16 points: 128-byte row segments) with the
// tile's OWN sample list from the host: a lane tests every sample of its wave's list, and with a
// 6 x 6 footprint on a 16 x 4 tile a sample is on 3 lists instead of 1.7 but a list is 2.4 times
// shorter than a 16 x 16 tile's -- the tests, not the sums or the writes, are the work (measured:
// without any hit the kernel took 0.29 of 0.39 ms, without the table loads 0.27).  The waves are
// independent: a wave stages 32 samples of its list at a time in its own slice of LDS, no
// workgroup barrier anywhere.  Whether (and with which weight) a sample reaches a grid point is
// decided once for all chains; per chain and grid point the terms are added in list order
// (ascending j): bitwise reproducible, independent of NU_CH and of the tiling.
#define NU_CH 8
#define NU_WCH 32            // samples a wave stages at a time
__global__ __launch_bounds__(256) void k_nufft_spread(const float2* __restrict__ vals, NuPlan p,
                                                      int T, int B, float2* __restrict__ grid) {
    __shared__ int s_x0[4][NU_WCH], s_y0[4][NU_WCH];
    __shared__ float s_wx[4][NU_WCH][NU_WMAX], s_wy[4][NU_WCH][NU_WMAX];
    __shared__ float2 s_v[4][NU_WCH][NU_CH];
    const int t = blockIdx.y % T, b0 = (blockIdx.y / T) * NU_CH, tf = p.traj_frames == 1 ? 0 : t;
    const int nc = min(NU_CH, B - b0);
    const int G = p.G, W = p.W, nbx = G / NU_TILE;
    const int blk = blockIdx.x, by = blk / nbx, bx = blk - by * nbx;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, ly = lane >> 4, lx = lane & 15;
    const int gy = by * NU_TILE + wv * 4 + ly, gx = bx * NU_TILE + lx;
    const int wt = (by * 4 + wv) * nbx + bx;                  // this wave's 16 x 4 tile
    const int32_t* ptr = p.tile_ptr + (size_t)tf * (p.tiles + 1);
    const int32_t* ent = p.tile_ent + p.tile_base[tf];
    const int e0 = ptr[wt], e1 = ptr[wt + 1];
    float ar[NU_CH], ai[NU_CH];
#pragma unroll
    for (int c = 0; c < NU_CH; ++c) ar[c] = ai[c] = 0.f;
    for (int c0 = e0; c0 < e1; c0 += NU_WCH) {
        const int n = min(NU_WCH, e1 - c0);
        // the wave's previous batch has been consumed: LDS accesses of a wave complete in order
        if (lane < n) {
            const size_t sj = (size_t)tf * p.J + ent[c0 + lane];
            s_x0[wv][lane] = p.ix0[sj];
            s_y0[wv][lane] = p.iy0[sj];
        }
        for (int u = lane; u < n * NU_CH; u += 64) {
            const int q = u / NU_CH, c = u - q * NU_CH;
            s_v[wv][q][c] = c < nc ? vals[((size_t)(b0 + c) * T + t) * p.J + ent[c0 + q]] : make_float2(0.f, 0.f);
        }
        for (int u = lane; u < n * W; u += 64) {
            const int q = u / W, k = u - q * W;
            const size_t sj = (size_t)tf * p.J + ent[c0 + q];
            s_wx[wv][q][k] = p.wx[sj * W + k];
            s_wy[wv][q][k] = p.wy[sj * W + k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int q = 0; q < n; ++q) {
            // offset of this grid point inside the sample's footprint, on the torus
            int dy = gy - s_y0[wv][q], dx = gx - s_x0[wv][q];
            dy = dy < 0 ? dy + G : (dy >= G ? dy - G : dy);
            dx = dx < 0 ? dx + G : (dx >= G ? dx - G : dx);
            if (dy < W && dx < W) {
                const float w = s_wy[wv][q][dy] * s_wx[wv][q][dx];
#pragma unroll
                for (int c = 0; c < NU_CH; ++c) {
                    const float2 v = s_v[wv][q][c];
                    ar[c] += w * v.x;
                    ai[c] += w * v.y;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < NU_CH; ++c)
        if (c < nc) grid[(((size_t)(b0 + c) * T + t) * G + gy) * G + gx] = make_float2(ar[c], ai[c]);
}

// The same sums with A LANE PER CHAIN, for batches that fill a wave.  k_nufft_spread spends its time
// finding out which of a tile's 64 points a sample reaches and broadcasting the sample's values to
// them through LDS (8 ds_read_b64 per list entry and 8 chains: the LDS pipe, shared by the CU's four
// SIMDs, is what pins it at 0.38 ms whatever the instruction count).  Here a single-wave workgroup
// owns a 16 x 2 tile of frame index t for 64 chains: lane c holds chain c's 32 sums, an entry's 32
// weights (host table: w_y w_x at each point of the tile, zero outside the footprint) are the same for
// every lane and arrive in SCALAR registers, an entry costs one 8-byte load per lane and 64 v_fmac
// with a scalar operand -- no test, no LDS, no divergence; each entry serves 64 chains, so the list
// walk is paid once per 64 instead of once per 8.  The sums are transposed through LDS (one row of
// 16 points at a time, XOR-swizzled: conflict-free both ways) so that the grid is still written as
// 128-byte row segments.  Terms are added in list order (ascending j) with the same rounding
// (fp32 product of the two weights, then fma): bit-identical to k_nufft_spread for finite input; a
// zero weight times a non-finite value would reach points the footprint does not (the other kernel
// skips them).
// A tile's record (id, first entry, entries) is one 16-byte scalar load; the tiles are launched longest
// list first (the four tiles around k = 0 hold ~200 entries against a median of 6).
#define NW_TW 16
#define NW_TH 2
#define NW_PTS (NW_TW * NW_TH)
#define NW_PAD 4             // the host pads every list to a multiple of this
#define NW_CHUNK 4           // entries whose values are requested together (= NW_PAD)
#define NW_RP 16             // points per transposition round: 128-byte row segments (64-byte ones: 285 -> 389 us)
#define NW_REC 16            // ints per tile record
#define NW_M 1               // tiles per wave (2: both tiles' first values in flight need 139 registers, 12 waves per CU: 0.33 ms against 0.28)
#define NW_MIN_CHAINS 16
__global__ __launch_bounds__(64) void k_nufft_spread_wide(const float2* __restrict__ vals, NuPlan p, int T, int B,
                                                          int skip_outside, float2* __restrict__ grid) {
    __shared__ float2 s_tr[64 * NW_RP];                       // 8 KB, XOR-swizzled instead of padded: 20 waves per CU, not 18
    const int lane = threadIdx.x;
    const int t = blockIdx.y % T, b0 = (blockIdx.y / T) * 64, tf = p.traj_frames == 1 ? 0 : t;
    const int nc = min(64, B - b0);
    const int G = p.G, ntx = G / NW_TW;
    // A tile's record: id, first entry, entries, outside, then the sample ids of its first eight entries -- their
    // values are requested after ONE trip to memory.  NW_M > 1 lets a wave take NW_M consecutive records with
    // all their first values requested at once; measured slower at 2 (139 registers), so a wave takes one tile.
    const int32_t* __restrict__ rp = p.wide_rec + ((size_t)tf * p.wide_tiles + blockIdx.x * NW_M) * NW_REC;
    const int32_t* __restrict__ ent = p.wide_ent;
    const float* __restrict__ wts = p.wide_w;
    const float2* __restrict__ v = vals + ((size_t)(b0 + min(lane, nc - 1)) * T + t) * p.J;
    int4 rec[NW_M];
    float2 xa[NW_M][NW_CHUNK], xb[NW_M][NW_CHUNK];
#pragma unroll
    for (int m = 0; m < NW_M; ++m) {
        rec[m] = *(const int4*)(rp + m * NW_REC);
        const int4 ja = *(const int4*)(rp + m * NW_REC + 4), jb = *(const int4*)(rp + m * NW_REC + 8);
        xa[m][0] = v[ja.x]; xa[m][1] = v[ja.y]; xa[m][2] = v[ja.z]; xa[m][3] = v[ja.w];   // valid ids even where the list is shorter
        xb[m][0] = v[jb.x]; xb[m][1] = v[jb.y]; xb[m][2] = v[jb.z]; xb[m][3] = v[jb.w];
    }
    const int cl = lane / NW_RP, px = lane % NW_RP;           // store role: 64 / NW_RP chains x one row segment
#pragma unroll
    for (int m = 0; m < NW_M; ++m) {
        if (skip_outside && rec[m].w) continue;               // nufft_adjoint: nobody reads the tiles outside the column ranges
        float ar[NW_PTS], ai[NW_PTS];
#pragma unroll
        for (int k = 0; k < NW_PTS; ++k) ar[k] = ai[k] = 0.f;
        auto add = [&](int e, const float2 (&x)[NW_CHUNK]) {
#pragma unroll
            for (int u = 0; u < NW_CHUNK; ++u) {
                const float* __restrict__ w = wts + (size_t)(e + u) * NW_PTS;
#pragma unroll
                for (int k = 0; k < NW_PTS; ++k) {
                    ar[k] = fmaf(w[k], x[u].x, ar[k]);
                    ai[k] = fmaf(w[k], x[u].y, ai[k]);
                }
            }
        };
        if (rec[m].z > 0) add(rec[m].y, xa[m]);
        if (rec[m].z > NW_CHUNK) add(rec[m].y + NW_CHUNK, xb[m]);
        for (int e = rec[m].y + 2 * NW_CHUNK; e < rec[m].y + rec[m].z; e += NW_CHUNK) {
            float2 x[NW_CHUNK];
#pragma unroll
            for (int u = 0; u < NW_CHUNK; ++u) x[u] = v[ent[e + u]];
            add(e, x);
        }
        const int ty = rec[m].x / ntx, tx = rec[m].x - ty * ntx;
#pragma unroll
        for (int r = 0; r < NW_PTS / NW_RP; ++r) {
#pragma unroll
            for (int k = 0; k < NW_RP; ++k) s_tr[lane * NW_RP + (k ^ ((lane >> 1) & 15))] = make_float2(ar[r * NW_RP + k], ai[r * NW_RP + k]);
            __syncthreads();
            float2* dst = grid + (((size_t)b0 * T + t) * G + ty * NW_TH + (r * NW_RP) / NW_TW) * G + tx * NW_TW + (r * NW_RP) % NW_TW + px;
#pragma unroll
            for (int i = 0; i < NW_RP; ++i) {
                const int c = (64 / NW_RP) * i + cl;
                if (c < nc) {
                    typedef float f2v __attribute__((ext_vector_type(2)));
                    const float2 o = s_tr[c * NW_RP + (px ^ ((c >> 1) & 15))];
                    __builtin_nontemporal_store(f2v{o.x, o.y}, (f2v*)(dst + (size_t)c * T * G * G));
                }
            }
            __syncthreads();
        }
    }
}

// out[f][0/1][y][x] = scale * apod[y] apod[x] * Re/Im grid[f][(y - R/2) mod G][(x - R/2) mod G]
__global__ __launch_bounds__(256) void k_nufft_crop_apod(const float2* __restrict__ grid, NuPlan p, int T,
                                                         const float* __restrict__ chain_scale,
                                                         float out_scale, float* __restrict__ out) {
    const int f = blockIdx.y, G = p.G, R = p.R, h = R / 2;
    const float cs = out_scale * (chain_scale ? chain_scale[f / T] : 1.f);
    const float2* g = grid + (size_t)f * G * G;
    float* outr = out + (size_t)f * 2 * R * R;
    float* outi = outr + (size_t)R * R;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < R * R; e += gridDim.x * 256) {
        const int py = e / R, px = e - py * R;
        const int gy = py >= h ? py - h : py + G - h, gx = px >= h ? px - h : px + G - h;
        const float a = cs * p.apod[py] * p.apod[px];
        const float2 v = g[(size_t)gy * G + gx];
        outr[e] = a * v.x;
        outi[e] = a * v.y;
    }
}

// loss[b] = sum of the chain's partials: a workgroup per chain, thread i adds entries i, i + 256, ...
// in order and the 256 sums are folded by a fixed tree; k_nufft_loss_total adds the chains in order.
__global__ __launch_bounds__(256) void k_nufft_loss_finish(const float* __restrict__ part, int per_chain,
                                                           float* __restrict__ loss) {
    __shared__ float red[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* q = part + (size_t)b * per_chain;
    float s = 0.f;
    for (int i = tid; i < per_chain; i += 256) s += q[i];
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) loss[b] = red[0];
}

__global__ void k_nufft_loss_total(const float* __restrict__ loss, int B, float* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += loss[b];
        *total = s;
    }
}

static int nu_check(const char* who, const nufft_plan_t* p, int nframes, int T) {
    STEP_REQUIRE(p, "%s: null plan", who);
    STEP_REQUIRE(nframes > 0 && T > 0 && nframes % T == 0, "%s: %d frames, T = %d", who, nframes, T);
    STEP_REQUIRE(p->R > 0 && p->R % 2 == 0 && p->G == 2 * p->R && p->G % NU_TILE == 0,
                 "%s: need an even R and G = 2 R divisible by %d (R=%d G=%d)", who, NU_TILE, p->R, p->G);
    STEP_REQUIRE(p->W >= 2 && p->W <= NU_WMAX && p->J > 0, "%s: kernel width %d (2..%d), J = %d", who, p->W, NU_WMAX, p->J);
    STEP_REQUIRE(p->traj_frames == 1 || p->traj_frames == T, "%s: traj_frames = %d must be 1 or T = %d", who, p->traj_frames, T);
    STEP_REQUIRE(p->ix0 && p->iy0 && p->wx && p->wy && p->tile_ptr && p->tile_ent && p->tile_base && p->apod,
                 "%s: null plan table", who);
    return STEP_OK;
}

static NuPlan nu_plan(const nufft_plan_t* p) {
    return NuPlan{p->R, p->G, p->W, p->J, p->traj_frames, (p->G / NU_TILE) * (p->G / 4), p->ix0, p->iy0, p->wx, p->wy,
                  p->tile_ptr, p->tile_ent, p->tile_base, p->apod, p->own_ptr, p->own_ent, p->own_x0, p->own_y0,
                  p->own_wx, p->own_wy,
                  (p->wide_rec && p->wide_ent && p->wide_w) ? (p->G / NW_TW) * (p->G / NW_TH) : 0,
                  p->wide_rec, p->wide_ent, p->wide_w};
}

extern "C" int nufft_interp_blocks(int J) { return (J + 255) / 256; }

extern "C" int nufft_pad_apod(const float* x, int nframes, int T, const nufft_plan_t* plan, float* grid_c64,
                              void* stream) {
    STEP_REQUIRE(x && grid_c64, "nufft_pad_apod: null pointer");
    if (int rc = nu_check("nufft_pad_apod", plan, nframes, T)) return rc;
    hipLaunchKernelGGL(k_nufft_pad_apod, dim3(64, nframes), dim3(256), 0, (hipStream_t)stream, x, nu_plan(plan),
                       (float2*)grid_c64);
    STEP_CHECK_LAUNCH("nufft_pad_apod");
    return STEP_OK;
}

// the tiled kernel needs the owner bins and whole 32 x 32 tiles
static bool nu_tiled(const nufft_plan_t* p) {
    return p->own_ptr && p->own_ent && p->own_x0 && p->own_y0 && p->own_wx && p->own_wy && p->G % NI_TW == 0 && p->G % NI_TH == 0;
}

extern "C" int nufft_interp_parts(const nufft_plan_t* plan) {
    if (!plan) return 0;
    return nu_tiled(plan) ? (plan->G / NI_TW) * (plan->G / NI_TH) : nufft_interp_blocks(plan->J);
}

// loss[b] = sum of the chain's per_chain partial sums (fixed order), loss_total = their sum (may be NULL)
int nufft_loss_reduce(const float* loss_part, int B, int per_chain, float* loss, float* loss_total, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_nufft_loss_finish, dim3(B), dim3(256), 0, st, loss_part, per_chain, loss);
    if (loss_total) hipLaunchKernelGGL(k_nufft_loss_total, dim3(1), dim3(64), 0, st, loss, B, loss_total);
    STEP_CHECK_LAUNCH("nufft loss reduction");
    return STEP_OK;
}

extern "C" int nufft_interp(const float* grid_c64, int nframes, int T, const nufft_plan_t* plan,
                            const float* y_c64, int y_batch, float res_scale, float* out_c64,
                            float* loss_part, float* loss, float* loss_total, void* stream) {
    STEP_REQUIRE(grid_c64 && out_c64, "nufft_interp: null pointer");
    if (int rc = nu_check("nufft_interp", plan, nframes, T)) return rc;
    const int B = nframes / T;
    STEP_REQUIRE(!y_c64 || y_batch == 1 || y_batch == B, "nufft_interp: measurement batch %d vs B=%d", y_batch, B);
    STEP_REQUIRE(!loss || (y_c64 && loss_part), "nufft_interp: the loss needs the measurement and loss_part");
    STEP_REQUIRE(!loss_total || loss, "nufft_interp: loss_total needs loss");
    hipStream_t st = (hipStream_t)stream;
    const int parts = nufft_interp_parts(plan);
    if (nu_tiled(plan)) {
        STEP_REQUIRE(T * ((B + NI_CH - 1) / NI_CH) <= 65535, "nufft_interp: %d chains x %d frames exceed the grid", B, T);
        hipLaunchKernelGGL(k_nufft_interp_tiled, dim3(parts, T * ((B + NI_CH - 1) / NI_CH)), dim3(256), 0, st,
                           (const float2*)grid_c64, nu_plan(plan), T, B, (const float2*)y_c64, y_batch, res_scale,
                           (float2*)out_c64, y_c64 ? loss_part : nullptr);
    } else {
        hipLaunchKernelGGL(k_nufft_interp, dim3(parts, nframes), dim3(256), 0, st, (const float2*)grid_c64,
                           nu_plan(plan), T, (const float2*)y_c64, y_batch, res_scale, (float2*)out_c64,
                           y_c64 ? loss_part : nullptr);
    }
    STEP_CHECK_LAUNCH("nufft_interp");
    if (loss) return nufft_loss_reduce(loss_part, B, T * parts, loss, loss_total, stream);
    return STEP_OK;
}

// skip_outside: leave the tiles whose record says "outside the column ranges" unwritten (nufft_adjoint,
// whose column pass does not read them); needs the lane-per-chain lists and B >= 16, else STEP_E_INVALID
int nufft_spread_impl(const float* vals_c64, int nframes, int T, const nufft_plan_t* plan, float* grid_c64,
                      int skip_outside, void* stream) {
    STEP_REQUIRE(vals_c64 && grid_c64, "nufft_spread: null pointer");
    if (int rc = nu_check("nufft_spread", plan, nframes, T)) return rc;
    const NuPlan p = nu_plan(plan);
    const int B = nframes / T;
    if (p.wide_tiles && B >= NW_MIN_CHAINS) {                   // a lane per chain
        const int ny = T * ((B + 63) / 64);
        STEP_REQUIRE(p.G % NW_TW == 0 && p.wide_tiles % NW_M == 0 && ny <= 65535, "nufft_spread: G = %d, %d chains x %d frames", p.G, B, T);
        hipLaunchKernelGGL(k_nufft_spread_wide, dim3(p.wide_tiles / NW_M, ny), dim3(64), 0, (hipStream_t)stream,
                           (const float2*)vals_c64, p, T, B, skip_outside, (float2*)grid_c64);
        STEP_CHECK_LAUNCH("nufft_spread (wide)");
        return STEP_OK;
    }
    STEP_REQUIRE(!skip_outside, "nufft_adjoint: needs the plan's lane-per-chain lists and at least %d chains", NW_MIN_CHAINS);
    hipLaunchKernelGGL(k_nufft_spread, dim3((p.G / NU_TILE) * (p.G / NU_TILE), T * ((B + NU_CH - 1) / NU_CH)),
                       dim3(256), 0, (hipStream_t)stream, (const float2*)vals_c64, p, T, B,
                       (float2*)grid_c64);
    STEP_CHECK_LAUNCH("nufft_spread");
    return STEP_OK;
}

extern "C" int nufft_spread(const float* vals_c64, int nframes, int T, const nufft_plan_t* plan,
                            float* grid_c64, void* stream) {
    return nufft_spread_impl(vals_c64, nframes, T, plan, grid_c64, 0, stream);
}

extern "C" int nufft_crop_apod(const float* grid_c64, int nframes, int T, const nufft_plan_t* plan,
                               const float* chain_scale, float out_scale, float* out, void* stream) {
    STEP_REQUIRE(grid_c64 && out, "nufft_crop_apod: null pointer");
    if (int rc = nu_check("nufft_crop_apod", plan, nframes, T)) return rc;
    hipLaunchKernelGGL(k_nufft_crop_apod, dim3(32, nframes), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)grid_c64, nu_plan(plan), T, chain_scale, out_scale, out);
    STEP_CHECK_LAUNCH("nufft_crop_apod");
    return STEP_OK;
}
