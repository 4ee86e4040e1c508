// Separable black-hole DFT on the bf16 matrix cores: fp32 operands split exactly into three
// bf16 planes ("3 x bf16"), and the image folded about its horizontal mid-line.
//
// Same contraction as bh_dft_sep.hip (ehtim ftmatrix is rank-1 per baseline, blackhole.py:226,
// forward blackhole.py:315-321).  Two ideas on top of it:
//
// 1. Operand split.  v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate.  Every fp32
//    operand is decomposed EXACTLY (by truncation) into three bf16 values
//        a = a_hi + a_mid + a_lo        (8 + 8 + 8 mantissa bits)
//    and the six products whose weight is >= 2^-16 of the leading one
//        a x ~= a_hi x_hi + (a_hi x_mid + a_mid x_hi) + (a_hi x_lo + a_mid x_mid + a_lo x_hi)
//    are accumulated in fp32.  A bf16 x bf16 product is exact in fp32; the dropped terms are
//    <= 3 * 2^-24 |a x|, the size of ONE fp32 rounding of the product and ten times below the
//    rounding noise of any fp32 sum over 65536 pixels.
//
// 2. Fold.  ehtim's pixel grid is centred, so ey[f][H-1-y] = conj(ey[f][y]) (bit-exact in the
//    fp32 tables; the caller checks it, bh_sep3_tables_ok).  Re ey is even and Im ey is odd in y:
//        Re Z[f][x] = sum_{y < H/2} Re ey[f][y] (X[y][x] + X[H-1-y][x])
//        Im Z[f][x] = sum_{y < H/2} Im ey[f][y] (X[y][x] - X[H-1-y][x])
//    and for the adjoint, with w[f][x] = conj(ex[f][x]) g[f],
//        E[y][x] = sum_f Re ey[f][y] Re w[f][x],   O[y][x] = sum_f Im ey[f][y] Im w[f][x]
//        grad[y][x] = E + O,   grad[H-1-y][x] = E - O          (y < H/2)
//    which halves the matrix work and the LDS image of ey in both directions, and makes the
//    adjoint's LDS operand independent of the chain (g moves into the register operand).
//
// Both kernels are persistent: a workgroup (8 waves, one 32-column strip each) owns frame t for
// `cpw` chains, stages the frame's ey planes in LDS once (48 KB at H = 256) and walks its chains.
// The grid is 1-D with t = id % T, so that XCD c streams the frames with t % 8 == c (s3_grid).
// forward: a ring of R blocks (16 rows from the top + their 16 mirror rows, 16 dwords per lane
//   per block) keeps R - 1 blocks in flight across chain boundaries; the leading products go to
//   `mn`, the five corrections to `sm` (2^-8 of the leading term, their rounding never matters),
//   so `mn` sees H/32 additions per output.  Epilogue: complex multiply by ex (held in
//   registers: nothing is loaded in the chain loop but the ring) in the accumulator layout,
//   DPP row sums, fixed-order sum over strips.
// adjoint: g of up to 64 chains is staged in LDS; per chain each wave folds g into its ex columns
//   (w), splits w into planes in registers, and produces two mirrored 32x32 output tiles from
//   24 MFMAs; no barrier and no vector load in the chain loop.
#include "common.h"
#include "closure_math.h"
#include "mcmc_elem.h"
#include <atomic>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define S3_THREADS 512
#define S3_RING 4
// chain handled in iteration `it` by chain group grp: blocked (dealing the chains round-robin,
// all workgroups sweeping the batch as one front, measured the same: DESIGN.md section 5)
#define S3_CHAIN(it) (grp * cpw + (it))
#define S3_NIT() min(cpw, B - grp * cpw)
#define S3_GCH 64          // adjoint: chains whose g is staged in LDS at a time
#define S3_GROW 68         // floats per staged chain: 32 x (re, im), flux, pad to 16 bytes

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

#define S3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0)

// the six plane products of one operand pair: corrections into sm, leading term into mn
#define S3_SIX(ah, am, al, bh, bm, bl, sm, mn)                  \
    sm = S3_MFMA(al, bh, sm); sm = S3_MFMA(ah, bl, sm);         \
    sm = S3_MFMA(am, bm, sm); sm = S3_MFMA(am, bh, sm);         \
    sm = S3_MFMA(ah, bm, sm); mn = S3_MFMA(ah, bh, mn);

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

static int s3_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

// chains per workgroup: one resident workgroup per CU, every CU busy, no second round.
// The grid is 1-D with t = id % T, chain group = id / T: workgroup ids go round the 8 XCDs, so XCD c
// streams the frames with t % 8 == c.  Which XCD touches which frames matters to the memory
// system: the same stores from a (gx, T) grid -- XCD c gets one chain group and the t of one
// parity -- take 8 % longer (scripts/probes/store_probe.*: 0.209 -> 0.194 ms per GB).
static void s3_grid(int B, int T, int* cpw, int* gx) {
    int per_t = s3_cu_count() / T;
    per_t = per_t < 1 ? 1 : (per_t > B ? B : per_t);
    *cpw = (B + per_t - 1) / per_t;
    *gx = (B + *cpw - 1) / *cpw;
}

extern "C" int bh_sep3_supported(int H, int W) {
    return (H == 256 && W == 256) || (H == 128 && W == 128) || (H == 64 && W == 64);
}

// Host check of the fold's precondition on the planar table ey_ri [T][2][32][H] (host memory):
// Re ey even, Im ey odd about the horizontal mid-line, bit for bit.
extern "C" int bh_sep3_tables_ok(const float* ey_ri_host, int T, int H) {
    if (!ey_ri_host || T <= 0 || H <= 0) return 0;
    for (int t = 0; t < T; ++t)
        for (int f = 0; f < 32; ++f) {
            const float* re = ey_ri_host + (((size_t)t * 2 + 0) * 32 + f) * H;
            const float* im = ey_ri_host + (((size_t)t * 2 + 1) * 32 + f) * H;
            for (int y = 0; y < H / 2; ++y)
                if (re[y] != re[H - 1 - y] || im[y] != -im[H - 1 - y]) return 0;
        }
    return 1;
}

// ------------------------------------------------------------------------------- fused step
// bh_loss_grad_sep3 runs the whole Langevin data step -- DFT, closure quantities, chi^2, loss,
// their backward, DFT adjoint -- in TWO launches: the closure stage has no kernel of its own.
//   forward epilogue: the frame's closure phases / log closure amplitudes only involve the frame's
//     own visibilities (process_obs keys every baseline by its time, blackhole.py:237-241), so the
//     wave that sums the strips goes on to evaluate them (wave 0 the triangles, wave 1 the
//     quadrangles; one closure per lane, table entries and the batch-1 measurement held in
//     registers for all chains: nothing is loaded in the chain loop), writes each closure's
//     derivative term d/dphi, d/dl and the frame's partial chi^2 sums S[b][t];
//   adjoint prologue: per chain, sum S over the frames -> chi^2 -> e'(chi^2) (normalize_chisq) ->
//     the factors of the backward; the frame's g_vis follows from the derivative terms by the
//     same incidence gather as in k_closure_loss and goes straight into the LDS staging of g.
//     Workgroup 0 also evaluates every chain's loss and their total.
struct S3FusedFwd {
    const int32_t* ftab;      // [T][2][64][2]: (closure id k or -1, packed legs) -- bh_build_frame_closures
    const float* y_cp;        // measurement (batch 1): cphase, sigma_cphase, logcamp, sigma_logcamp, flux
    const float* s_cp;
    const float* y_ca;
    const float* s_ca;
    const float* y_fl;
    float* dcp;               // [B][Kcp]  d/dphi of (1 - cos(y - phi)) / s^2
    float* dca;               // [B][Kca]  d/dl of ((l - y) / s)^2
    float* S;                 // [B][T][4] partial sums: cphase, logcamp, flux^2 residual, unused
    int Kcp, Kca, cphase_form;
};

struct S3FusedAdj {
    const float2* vis;        // [B][T*F]
    const float* flux;        // [B][T]
    const float* S;
    const float* dcp;
    const float* dca;
    const int32_t* cp_ptr;    // incidence CSR over the T*F baselines (bh_build_incidence)
    const int32_t* cp_ent;
    const int32_t* ca_ptr;
    const int32_t* ca_ent;
    const float* y_fl;
    float w_cp, w_camp, w_flux;
    int normalize, Kcp, Kca;
    float* loss;              // [B]
    float* chi2;              // [B][5] or null
    float* total;             // sum of loss[] or null
};

// UPD instantiations of the adjoint: the gradient is not stored; the gradient-descent update of
// MCMCSampler.optimize (daps.py:210-223), x -= lr (g + (x - x0hat) / sigma^2), is applied where
// the tile would have been written: x and x0hat are read, x is written, and the gradient never
// reaches HBM (one 1 GB write and one 1 GB read less per step at 64 chains).
// UPD == 4 / 5: the same two steps with the noise READ from a tensor laid out like x (the reference's
// own draw, torch.randn_like at daps.py:192, replayed): one extra stream instead of the update
// kernel's six.
// UPD == 2 / 3: the Langevin / momentum-free HMC step of MCMCSampler.sample (daps.py:135-144,
// :93-104 with the score terms that depend on x collapsed into (x0 - x) / sigma^2: what the
// 'gaussian' prior solver's (xt - x) / sigma^2 + (x0hat - xt) / sigma^2 adds up to) in the same
// place, noise from Philox in the column-quad layout (mcmc_elem.h): a lane's four accumulator
// rows of one column are one generator call.  Same arithmetic as k_mcmc_update on the stored
// gradient with xt = x0hat, no prior tensor and quad_w = W.
struct S3Update {
    float* x;                 // [B][T][H*W] state, updated in place
    const float* x0;          // x0hat
    float lr, inv_sigma2;
    mcmc_params_t p;          // UPD >= 2
    const float* eps;         // UPD >= 4: the noise, laid out like x (torch's draw, read once) -- no generator
    const int64_t* seeds;     // UPD 2 / 3: [B] device
    int64_t step;
    const int64_t* step_dev;  // device step counter (graph replay) or null
    int32_t* nan_flag;        // or null
};

// chi^2 sums of one chain -> loss and the backward factors (same order of operations as
// k_closure_loss).  s_* are the sums over all frames.
__device__ __forceinline__ void s3_chain_factors(const S3FusedAdj& z, int T, float s_cp, float s_ca,
                                                 float s_fl, float& f_cp, float& f_ca, float& L,
                                                 float& chi_cp, float& chi_ca, float& chi_fl) {
    chi_cp = z.w_cp > 0.f ? 2.0f * s_cp / z.Kcp : 0.f;
    chi_ca = z.w_camp > 0.f ? s_ca / z.Kca : 0.f;
    chi_fl = z.w_flux > 0.f ? s_fl / T : 0.f;
    float e_cp = 0.f, d_cp = 0.f, e_ca = 0.f, d_ca = 0.f;
    if (z.w_cp > 0.f) fold(chi_cp, z.normalize, e_cp, d_cp);
    if (z.w_camp > 0.f) fold(chi_ca, z.normalize, e_ca, d_ca);
    L = 0.f;
    if (z.w_cp > 0.f) L += z.w_cp * e_cp;
    if (z.w_camp > 0.f) L += z.w_camp * e_ca;
    if (z.w_flux > 0.f) L += z.w_flux * chi_fl;
    f_cp = z.w_cp * d_cp * 2.0f / z.Kcp;
    f_ca = z.w_camp * d_ca / z.Kca;
}

// ------------------------------------------------------------------------------- forward
// LDS: ey planes, 16-byte units [((kb*2 + part)*3 + plane)*2 + h][32 rows f] (kb = block of 16
// rows of the upper half, part 0 = Re ey, 1 = Im ey): the A fragment of lane (f, h) for
// k = 16 kb + 8 h + 0..7 is one ds_read_b128, consecutive lanes read consecutive units.
// Then sred [parity][2 * strips][64].
template <int H, int W, int R, bool CLIP, bool FUSED>
__global__ __launch_bounds__(S3_THREADS, 1) void k_dft_fwd_s3(
    const float* __restrict__ ey_t, const float* __restrict__ ex_ri, const float* __restrict__ x,
    float* __restrict__ vis, float* __restrict__ flux, int T, int F, int B, int cpw, float scale,
    float shift, float clip_min, S3FusedFwd fz) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    constexpr int NKB = H / 32, NSTRIP = W / 32, NPART = 2 * NSTRIP;
    static_assert(NKB % R == 0, "ring must divide the block count");
    u32x4* eyp = lds4;
    float* sred = (float*)(lds4 + NKB * 384);
    const int t = blockIdx.x % T, grp = blockIdx.x / T;      // s3_grid: why a 1-D grid
    const int tid = threadIdx.x, lane = tid & 63, strip = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_it = S3_NIT();
    const bool mine = strip < NSTRIP;                        // W < 256: the spare waves idle
    const int xcol = (mine ? strip : 0) * 32 + li;
    // ex of this lane's accumulator rows, held in registers for all chains: loads in the
    // per-chain epilogue would wait behind the x blocks already in flight for the next chain
    // (vmcnt counts in order) and drain the ring at every chain boundary
    float exv_r[16], exv_i[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = mfma32_row(r, h);
        exv_r[r] = ex_ri[(((size_t)t * 2 + 0) * 32 + f) * W + xcol];
        exv_i[r] = ex_ri[(((size_t)t * 2 + 1) * 32 + f) * W + xcol];
    }
    // FUSED: this lane's closure of frame t (wave 0: triangle `lane`, wave 1: quadrangle `lane`)
    // and its measurement, in registers for all chains (a load in the chain loop would drain the ring)
    int ck = -1, cpk = 0;
    float cy = 0.f, cis = 0.f, yfl = 0.f;
    if (FUSED && strip < 2) {
        const int32_t* e = fz.ftab + (((size_t)t * 2 + strip) * 64 + lane) * 2;
        ck = (strip == 0 ? fz.y_cp : fz.y_ca) ? e[0] : -1;   // a term with zero weight: no closures
        cpk = e[1];
        if (ck >= 0) {
            cy = (strip == 0 ? fz.y_cp : fz.y_ca)[ck];
            const float sg = (strip == 0 ? fz.s_cp : fz.s_ca)[ck];
            cis = strip == 0 ? 1.0f / (sg * sg) : 1.0f / sg;
        }
        yfl = fz.y_fl[t];
    }

    const u32x4* ap = eyp + h * 32 + li;
    const size_t frame = (size_t)H * W;
    // pointer to row 8h of the chain's frame; the mirror rows are addressed from it too
    auto frame_ptr = [&](int it) { return x + ((size_t)S3_CHAIN(it) * T + t) * frame + 8 * h * W + xcol; };

    f32x16 mn0, mn1, sm0, sm1;
    float xq[R][16];
    // block kb of the frame at p: rows 16 kb + 8h + j (top) and H-1 - (16 kb + 8h + j) (mirror)
#define S3_LOAD(dst, p, kb)                                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                  \
        dst[j] = S3_LDX((p) + ((kb) * 16 + j) * W);                                  \
        dst[8 + j] = S3_LDX((p) + (H - 1 - 16 * h - (kb) * 16 - j) * W);             \
    }
#define S3_COMPUTE(src, kb)                                                          \
    {                                                                                \
        float ev[8], od[8];                                                          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                              \
            float pt = fmaf(src[j], scale, shift), pb = fmaf(src[8 + j], scale, shift); \
            if (CLIP) {                                                              \
                pt = pt < clip_min ? clip_min : pt;                                  \
                pb = pb < clip_min ? clip_min : pb;                                  \
            }                                                                        \
            ev[j] = pt + pb;                                                         \
            od[j] = pt - pb;                                                         \
        }                                                                            \
        u32x4 eh, em, el, oh, om, ol;                                                \
        split3(ev, eh, em, el);                                                      \
        split3(od, oh, om, ol);                                                      \
        const u32x4* a = ap + (kb) * 384;                                            \
        {                                                                            \
            const u32x4 ah = a[0], am = a[64], al = a[128];                          \
            S3_SIX(ah, am, al, eh, em, el, sm0, mn0)                                 \
        }                                                                            \
        {                                                                            \
            const u32x4 ah = a[192], am = a[256], al = a[320];                       \
            S3_SIX(ah, am, al, oh, om, ol, sm1, mn1)                                 \
        }                                                                            \
    }

    // x is streamed exactly once: non-temporal loads keep it out of L2 / Infinity Cache, which
    // still hold the previous adjoint's dirty gradient lines (0.29 -> 0.19 ms inside the step)
#define S3_LDX(ptr) __builtin_nontemporal_load(ptr)
    const float* xcur = frame_ptr(0);
    if (mine) {
#pragma unroll
        for (int s = 0; s < R - 1; ++s) {
            // block after block, as in the loop: the wait counts of the loop's stages are the
            // merge of both paths, and a shuffled issue order here makes every stage over-wait
            S3_LOAD(xq[s], xcur, s);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the frame's ey planes are staged BEHIND the first x blocks: both are cold fetches, and at
    // small batches (one or two chains per workgroup) the prologue is a large part of the kernel
#pragma unroll
    for (int u = tid; u < (H / 16) * 64; u += S3_THREADS) {
        const int col = u & 63, yg = u >> 6;                 // col = part*32 + f, yg = 2 kb + h
        const float* src = ey_t + ((size_t)t * H + yg * 8) * 64 + col;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * 64];
        u32x4 p0, p1, p2;
        split3(v, p0, p1, p2);
        const int kb = yg >> 1, hh = yg & 1, part = col >> 5, f = col & 31;
        u32x4* dst = eyp + (((kb * 2 + part) * 3) * 2 + hh) * 32 + f;
        dst[0] = p0; dst[64] = p1; dst[128] = p2;
    }
    __syncthreads();
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
                    const bool wrap = nb >= NKB;
                    const float* pn = wrap ? xnext : xcur;
                    const int kn = wrap ? nb - NKB : nb;
                    S3_LOAD(xq[(s + R - 1) % R], pn, kn);
                    __builtin_amdgcn_sched_barrier(0);       // keep the loads ahead of the MFMAs
                    S3_COMPUTE(xq[s], kb + s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // p[f] = sum_x ex[f][x] Z[f][x] over the strip: complex multiply in the accumulator
            // layout, DPP sum over each 16-lane row, lanes 0/16/32/48 hand their halves to sred
            float* sp = sred + ((it & 1) * NPART + strip * 2 + ((lane >> 4) & 1)) * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = mfma32_row(r, h);
                const float er = exv_r[r], ei = exv_i[r];
                const float zr = mn0[r] + sm0[r], zi = mn1[r] + sm1[r];
                const float pr = row16_sum(er * zr - ei * zi);
                const float pi = row16_sum(er * zi + ei * zr);
                if ((lane & 15) == 0) { sp[f] = pr; sp[32 + f] = pi; }
            }
        }
        __syncthreads();
        if (strip == 0) {                                    // wave 0: fixed-order sum over strips
            const float* sp = sred + (it & 1) * NPART * 64 + lane;
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < NPART; ++p) s += sp[p * 64];
            const int b = S3_CHAIN(it), c = lane >> 5, f = lane & 31;
            if (f < F) vis[(((size_t)b * T + t) * F + f) * 2 + c] = s;
            else if (f == F && c == 0) flux[(size_t)b * T + t] = s;
            if (FUSED) {
                // closure phases of the frame, one triangle per lane (blackhole.py:341-363, 472-474)
                const int f0 = cpk & 31, f1 = (cpk >> 5) & 31, f2 = (cpk >> 10) & 31;
                const float2 v0 = make_float2(__shfl(s, f0), __shfl(s, 32 + f0));
                const float2 v1 = make_float2(__shfl(s, f1), __shfl(s, 32 + f1));
                const float2 v2 = make_float2(__shfl(s, f2), __shfl(s, 32 + f2));
                const float fl = __shfl(s, F);
                float acc = 0.f;
                if (ck >= 0) {
                    const float2 p = cmul(cmul(cdir(v0, (cpk >> 15) & 1), cdir(v1, (cpk >> 16) & 1)),
                                          cdir(v2, (cpk >> 17) & 1));
                    const float d = cy - atan2f(p.y, p.x);
                    acc = one_minus_cos(d, fz.cphase_form) * cis;
                    fz.dcp[(size_t)b * fz.Kcp + ck] = -sinf(d) * cis;
                }
                acc = wave_sum(acc);
                if (lane == 0) {
                    const float r = (fl - yfl) * 0.5f;                     // blackhole.py:490-491
                    float* Sp = fz.S + ((size_t)b * T + t) * 4;
                    Sp[0] = acc;
                    Sp[2] = r * r;
                }
            }
        }
        if (FUSED && strip == 1) {
            // wave 1 sums the strips again for itself and takes the log closure amplitudes, one
            // quadrangle per lane (blackhole.py:365-391, 482-483)
            const float* sp = sred + (it & 1) * NPART * 64 + lane;
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < NPART; ++p) s += sp[p * 64];
            const int b = S3_CHAIN(it);
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int fj = (cpk >> (5 * j)) & 31;
                a[j] = cabs(make_float2(__shfl(s, fj), __shfl(s, 32 + fj)));
            }
            float acc = 0.f;
            if (ck >= 0) {
                const float l = logf(a[0]) + logf(a[1]) - logf(a[2]) - logf(a[3]);
                const float r = (l - cy) * cis;
                acc = r * r;
                fz.dca[(size_t)b * fz.Kca + ck] = 2.0f * r * cis;
            }
            acc = wave_sum(acc);
            if (lane == 0) fz.S[((size_t)b * T + t) * 4 + 1] = acc;
        }
        xcur = xnext;
    }
#undef S3_LOAD
#undef S3_COMPUTE
#undef S3_LDX
}

template <int H, int W>
static void launch_fwd_s3(const float* ey_t, const float* ex_ri, int T, int F, const float* x, int B,
                          float scale, float shift, float clip_min, float* vis, float* flux,
                          const S3FusedFwd* fz, hipStream_t st) {
    constexpr int R = (H / 32 < S3_RING) ? H / 32 : S3_RING;
    int cpw, gx;
    s3_grid(B, T, &cpw, &gx);
    const size_t lds = (size_t)(H / 32) * 384 * 16 + (size_t)2 * 2 * (W / 32) * 64 * 4;
    if (fz) {                                                // the fused step never clips
        step_allow_lds((const void*)k_dft_fwd_s3<H, W, R, false, true>, lds);
        hipLaunchKernelGGL((k_dft_fwd_s3<H, W, R, false, true>), dim3(gx * T), dim3(S3_THREADS), lds, st,
                           ey_t, ex_ri, x, vis, flux, T, F, B, cpw, scale, shift, clip_min, *fz);
    } else if (clip_min == -INFINITY) {
        step_allow_lds((const void*)k_dft_fwd_s3<H, W, R, false, false>, lds);
        hipLaunchKernelGGL((k_dft_fwd_s3<H, W, R, false, false>), dim3(gx * T), dim3(S3_THREADS), lds, st,
                           ey_t, ex_ri, x, vis, flux, T, F, B, cpw, scale, shift, clip_min, S3FusedFwd{});
    } else {
        step_allow_lds((const void*)k_dft_fwd_s3<H, W, R, true, false>, lds);
        hipLaunchKernelGGL((k_dft_fwd_s3<H, W, R, true, false>), dim3(gx * T), dim3(S3_THREADS), lds, st,
                           ey_t, ex_ri, x, vis, flux, T, F, B, cpw, scale, shift, clip_min, S3FusedFwd{});
    }
}

static void dispatch_fwd_s3(const float* ey_t, const float* ex_ri, int T, int F, int H, const float* x,
                            int B, float scale, float shift, float clip_min, float* vis, float* flux,
                            const S3FusedFwd* fz, hipStream_t st) {
    if (H == 256) launch_fwd_s3<256, 256>(ey_t, ex_ri, T, F, x, B, scale, shift, clip_min, vis, flux, fz, st);
    else if (H == 128) launch_fwd_s3<128, 128>(ey_t, ex_ri, T, F, x, B, scale, shift, clip_min, vis, flux, fz, st);
    else launch_fwd_s3<64, 64>(ey_t, ex_ri, T, F, x, B, scale, shift, clip_min, vis, flux, fz, st);
}

extern "C" int bh_dft_fwd_sep3(const float* ey_t, const float* ex_ri, int T, int F, int H, int W,
                               const float* x, int B, float scale, float shift, float clip_min,
                               float* vis_c64, float* flux, void* stream) {
    STEP_REQUIRE(ey_t && ex_ri && x && vis_c64 && flux, "bh_dft_fwd_sep3: null pointer");
    STEP_REQUIRE(T > 0 && T <= 65535 && B > 0 && F > 0 && F <= 31,
                 "bh_dft_fwd_sep3: need 1 <= F <= 31, 1 <= T <= 65535 (F=%d T=%d)", F, T);
    STEP_REQUIRE(bh_sep3_supported(H, W), "bh_dft_fwd_sep3: image %dx%d not built (256, 128, 64 square)", H, W);
    dispatch_fwd_s3(ey_t, ex_ri, T, F, H, x, B, scale, shift, clip_min, vis_c64, flux, nullptr, (hipStream_t)stream);
    STEP_CHECK_LAUNCH("bh_dft_fwd_sep3");
    return STEP_OK;
}

// ------------------------------------------------------------------------------- adjoint
// LDS: ey planes as the A operand (M = y of the upper half, K = f), 16-byte units
// [(((part*2 + kb)*3 + plane)*2 + h)][H/2 rows y]; part 0 = Re ey (with the flux row f = F,
// ey = 1), part 1 = Im ey; kb = block of 16 baselines.  B operand per chain in registers:
// wR[f][x] = Re(conj(ex) g) = exR gR + exI gI,  wI[f][x] = Im(conj(ex) g) = exR gI - exI gR.
template <int H, int W, bool FUSED, int UPD>
__global__ __launch_bounds__(S3_THREADS, 1) void k_dft_adj_s3(
    const float* __restrict__ ey_ri, const float* __restrict__ ex_ri,
    const float* __restrict__ gvis, const float* __restrict__ gflux,
    const float* __restrict__ chain_scale, float* __restrict__ grad, int T, int F, int B, int cpw,
    float out_scale, const float* __restrict__ sum_in, int sum_n, float* __restrict__ sum_out,
    S3FusedAdj fz, S3Update up) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    constexpr int HH = H / 2, NSTRIP = W / 32, NTILE = HH / 32;
    // optional rider: sum_out = sum(sum_in[0 .. sum_n)) in a fixed order (the scalar loss that
    // gradient(return_loss=True) returns, daps.py:100) -- saves a reduction launch per step
    if (!FUSED && sum_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        float sacc = 0.f;
        for (int i = threadIdx.x; i < sum_n; i += 64) sacc += sum_in[i];
        sacc = wave_sum(sacc);
        if (threadIdx.x == 0) *sum_out = sacc;
    }
    const int nkb = (F + 1 + 15) >> 4;                       // 1 or 2 blocks of 16 baselines
    constexpr bool EPSMEM = UPD >= 4;                        // noise from memory instead of Philox
    constexpr int UALGO = UPD >= 2 ? ((UPD - 2) & 1) : 0;    // 0 langevin, 1 hmc at momentum 0
    const uint64_t nstep = (UPD >= 2 && !EPSMEM) ? (uint64_t)(up.step_dev ? *up.step_dev : up.step) : 0;
    bool bad = false;
    u32x4* eyp = lds4;
    float* gl = (float*)(lds4 + 24 * HH);                    // [S3_GCH chains][S3_GROW]
    float* fac = gl + S3_GCH * S3_GROW;                      // FUSED: [S3_GCH][2] backward factors, then [8]
    const int t = blockIdx.x % T, grp = blockIdx.x / T;      // s3_grid: why a 1-D grid
    const int tid = threadIdx.x, lane = tid & 63, strip = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_it = S3_NIT();
    const bool mine = strip < NSTRIP;                        // W < 256: the spare waves only stage
    const int xcol = (mine ? strip : 0) * 32 + li;
    const uint64_t qbase = (uint64_t)t * (H / 4) * W + xcol; // UPD >= 2: this column's first quad of the frame

    // this lane's baselines f = 16 kb + 8 h + j and their ex values (chain independent); issued
    // ahead of the table staging so that the two cold fetches overlap
    float exr[16], exi[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int f = (q >> 3) * 16 + 8 * h + (q & 7);
        exr[q] = ex_ri[(((size_t)t * 2 + 0) * 32 + f) * W + xcol];
        exi[q] = ex_ri[(((size_t)t * 2 + 1) * 32 + f) * W + xcol];
    }
    // FUSED: the incidence lists of this frame's baselines, relative to the frame, into LDS:
    // lcsr = cp_ptr[F + 1] (at 0) | ca_ptr[F + 1] (at 34) | cp_ent[<= 192] (at 68) | ca_ent[<= 256]
    int* lcsr = (int*)(fac + 2 * S3_GCH + 8);
    if (FUSED) {
        const int m0 = t * F;
        const int bcp = fz.w_cp > 0.f ? fz.cp_ptr[m0] : 0, bca = fz.w_camp > 0.f ? fz.ca_ptr[m0] : 0;
        const int ncp = fz.w_cp > 0.f ? fz.cp_ptr[m0 + F] - bcp : 0;
        const int nca = fz.w_camp > 0.f ? fz.ca_ptr[m0 + F] - bca : 0;
        if (tid <= F) {
            lcsr[tid] = fz.w_cp > 0.f ? fz.cp_ptr[m0 + tid] - bcp : 0;
            lcsr[34 + tid] = fz.w_camp > 0.f ? fz.ca_ptr[m0 + tid] - bca : 0;
        }
        if (tid < ncp) lcsr[68 + tid] = fz.cp_ent[bcp + tid];           // <= 192 and <= 256 entries:
        if (tid < nca) lcsr[68 + 192 + tid] = fz.ca_ent[bca + tid];     // bh_build_frame_closures
    }
    // gridDim.y workgroups share a frame's tiles when the chains alone do not fill the chip (one or
    // two chains per GPU): workgroup blockIdx.y takes the tiles with tile % gridDim.y == blockIdx.y
    // and stages only their rows of the planes; everything per chain ahead of the tiles is repeated
    const int tpart = blockIdx.y, tsplit = gridDim.y;
#pragma unroll
    for (int u = tid; u < 8 * HH; u += S3_THREADS) {
        const int y = u % HH, q = u / HH;                    // q = (part*2 + kb)*2 + h
        if (tsplit > 1 && (y >> 5) % tsplit != tpart) continue;
        const int hh = q & 1, kb = (q >> 1) & 1, part = q >> 2;
        const float* src = ey_ri + (((size_t)t * 2 + part) * 32 + kb * 16 + hh * 8) * H + y;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * H];
        u32x4 p0, p1, p2;
        split3(v, p0, p1, p2);
        u32x4* dst = eyp + (((part * 2 + kb) * 3) * 2 + hh) * HH + y;
        dst[0] = p0; dst[2 * HH] = p1; dst[4 * HH] = p2;
    }
    // plain stores: non-temporal ones were 5 % slower here and in the step
#define S3_STG(ptr, v) (*(ptr) = (v))
    const u32x4* ap = eyp + h * HH + li;
    // g of up to S3_GCH chains goes through LDS: a global load of the next chain's g inside the
    // chain loop would be waited for with vmcnt(0) -- loads and stores share that counter, in
    // order -- and every chain would start by draining the previous chain's 128 stores per lane
#pragma unroll 1
    for (int c0 = 0; c0 < n_it; c0 += S3_GCH) {
        const int nch = min(S3_GCH, n_it - c0);
        __syncthreads();                                     // planes staged / gl free again
        if (FUSED) {
            // per chain: chi^2 sums over the frames -> the two backward factors; a wave takes chains
            // strip, strip + 8, ... two at a time (both loads in flight together)
            for (int c = strip; c < nch; c += 2 * (S3_THREADS / 64)) {
                const int c2 = c + S3_THREADS / 64 < nch ? c + S3_THREADS / 64 : c;
                const int b1 = S3_CHAIN(c0 + c), b2 = S3_CHAIN(c0 + c2);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f;
                for (int tt = lane; tt < T; tt += 64) {
                    const float4 s4 = ((const float4*)fz.S)[(size_t)b1 * T + tt];
                    const float4 r4 = ((const float4*)fz.S)[(size_t)b2 * T + tt];
                    a0 += s4.x; a1 += s4.y; a2 += s4.z;
                    e0 += r4.x; e1 += r4.y; e2 += r4.z;
                }
                a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
                e0 = wave_sum(e0); e1 = wave_sum(e1); e2 = wave_sum(e2);
                float f_cp, f_ca, L, c_cp, c_ca, c_fl;
                s3_chain_factors(fz, T, a0, a1, a2, f_cp, f_ca, L, c_cp, c_ca, c_fl);
                if (lane == 0) { fac[2 * c] = f_cp; fac[2 * c + 1] = f_ca; }
                s3_chain_factors(fz, T, e0, e1, e2, f_cp, f_ca, L, c_cp, c_ca, c_fl);
                if (lane == 0) { fac[2 * c2] = f_cp; fac[2 * c2 + 1] = f_ca; }
            }
            __syncthreads();
            // g_vis of this frame's baselines: gather of the derivative terms through the incidence
            // lists, in the order of k_closure_loss.  The lists come
            // from LDS and the terms are fetched eight at a time, triangles and quadrangles
            // together: one round trip to memory instead of two per list entry.
            const int M = T * F;
            for (int u = tid; u < nch * 33; u += S3_THREADS) {
                const int c = u / 33, e = u - c * 33;
                const int b = S3_CHAIN(c0 + c);
                if (e < 32) {
                    const int fe = min(e, F - 1);
                    const float2 v = fz.vis[(size_t)b * M + t * F + fe];
                    const float* dP = fz.dcp + (size_t)b * fz.Kcp;
                    const float* dA = fz.dca + (size_t)b * fz.Kca;
                    const int p0 = lcsr[fe], p1 = lcsr[fe + 1], q0 = lcsr[34 + fe], q1 = lcsr[34 + fe + 1];
                    float gp = 0.f, ga = 0.f;
                    for (int o = 0; p0 + o < p1 || q0 + o < q1; o += 8) {
                        int en[8], eq[8];
                        float dv[8], dq[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const bool okp = p0 + o + j < p1, okq = q0 + o + j < q1;
                            en[j] = okp ? lcsr[68 + p0 + o + j] : -1;
                            eq[j] = okq ? lcsr[68 + 192 + q0 + o + j] : -1;
                            dv[j] = okp ? dP[en[j] >> 3] : 0.f;
                            dq[j] = okq ? dA[eq[j] >> 3] : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (en[j] >= 0) gp += (en[j] & 1) ? -dv[j] : dv[j];       // conj leg: -arg
                            if (eq[j] >= 0) ga += (((eq[j] >> 1) & 3) < 2) ? dq[j] : -dq[j];
                        }
                    }
                    gp *= fac[2 * c];
                    ga *= fac[2 * c + 1];
                    const float n2 = v.x * v.x + v.y * v.y;
                    float2 g_ = make_float2(0.f, 0.f);
                    if (fz.w_cp > 0.f || fz.w_camp > 0.f)
                        g_ = make_float2((ga * v.x - gp * v.y) / n2, (ga * v.y + gp * v.x) / n2);
                    *(float2*)(gl + c * S3_GROW + 2 * e) = g_;
                } else {
                    float g_ = 0.f;
                    if (fz.w_flux > 0.f)
                        g_ = fz.w_flux * (fz.flux[(size_t)b * T + t] - fz.y_fl[t]) * 0.5f / T;
                    gl[c * S3_GROW + 64] = g_;
                }
            }
        } else {
            for (int u = tid; u < nch * 33; u += S3_THREADS) {
                const int c = u / 33, e = u - c * 33;
                const size_t bt = (size_t)S3_CHAIN(c0 + c) * T + t;
                if (e < 32) {
                    const float2 g_ = ((const float2*)gvis)[bt * F + min(e, F - 1)];
                    *(float2*)(gl + c * S3_GROW + 2 * e) = g_;
                } else {
                    gl[c * S3_GROW + 64] = gflux[bt];
                }
            }
        }
        __syncthreads();
        if (!mine) continue;
#pragma unroll 1
        for (int ic = 0; ic < nch; ++ic) {
            const int b = S3_CHAIN(c0 + ic);
            const float cs = out_scale * (chain_scale ? chain_scale[b] : 1.f);
            const float* gc = gl + ic * S3_GROW;
            const float gfl = gc[64];
            u32x4 rh[2], rm[2], rl[2], ih[2], im[2], il[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float4* g4 = (const float4*)(gc + 2 * (kb * 16 + 8 * h));
                float gr[8], gi[8];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 g_ = g4[jj];
                    gr[2 * jj] = g_.x; gi[2 * jj] = g_.y; gr[2 * jj + 1] = g_.z; gi[2 * jj + 1] = g_.w;
                }
                float wr[8], wi[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = kb * 8 + j, f = kb * 16 + 8 * h + j;
                    const float a = cs * gr[j], c = cs * gi[j];
                    float vr = exr[q] * a + exi[q] * c, vi = exr[q] * c - exi[q] * a;
                    if (f >= F) { vr = (f == F) ? cs * gfl : 0.f; vi = 0.f; }
                    wr[j] = vr; wi[j] = vi;
                }
                split3(wr, rh[kb], rm[kb], rl[kb]);
                split3(wi, ih[kb], im[kb], il[kb]);
            }
            const size_t fo = ((size_t)b * T + t) * H * W + xcol;
            const uint64_t seed = (UPD >= 2 && !EPSMEM) ? (uint64_t)up.seeds[b] : 0;
            float* gf = (UPD ? up.x : grad) + fo;
            const float* x0f = UPD ? up.x0 + fo : nullptr;
            const float* epf = EPSMEM ? up.eps + fo : nullptr;
#pragma unroll 1
            for (int tile = tpart; tile < NTILE; tile += tsplit) {

                f32x16 ev, od;
#pragma unroll
                for (int r = 0; r < 16; ++r) ev[r] = od[r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    if (kb < nkb) {
                        const u32x4* a = ap + kb * (6 * HH) + tile * 32;
                        {
                            const u32x4 ah = a[0], am = a[2 * HH], al = a[4 * HH];
                            S3_SIX(ah, am, al, rh[kb], rm[kb], rl[kb], ev, ev)
                        }
                        {
                            const u32x4 ah = a[12 * HH], am = a[14 * HH], al = a[16 * HH];
                            S3_SIX(ah, am, al, ih[kb], im[kb], il[kb], od, od)
                        }
                    }
                const int ytop = tile * 32 + 4 * h;
                if (UPD >= 2) {
                    // eight rows (two quads) at a time: their state and x0hat are fetched, the two
                    // generator calls run while the loads are in flight, then the update is stored
#pragma unroll
                    for (int part = 0; part < 4; ++part) {
                        const bool mir = part >= 2;
                        const int r0 = 8 * (part & 1);
                        float xo[8], xh[8], nz[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int yy = ytop + (i & 3) + 8 * ((r0 + i) >> 2);
                            xo[i] = gf[(mir ? H - 1 - yy : yy) * W];
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int yy = ytop + (i & 3) + 8 * ((r0 + i) >> 2);
                            xh[i] = x0f[(mir ? H - 1 - yy : yy) * W];
                        }
                        if (EPSMEM) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int yy = ytop + (i & 3) + 8 * ((r0 + i) >> 2);
                                nz[i] = __builtin_nontemporal_load(epf + (mir ? H - 1 - yy : yy) * W);
                            }
                        } else {
#pragma unroll
                        for (int jq = 0; jq < 2; ++jq) {
                            const int yq = ytop + 8 * ((r0 >> 2) + jq);          // first row of the quad, upper half
                            const int m = (mir ? H - 4 - yq : yq) >> 2;
                            float n4[4];
                            philox_normals4(qbase + (uint64_t)m * W, nstep, seed, n4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) nz[4 * jq + k] = n4[mir ? 3 - k : k];
                        }
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int r = r0 + i;
                            const int yy = ytop + (r & 3) + 8 * (r >> 2);
                            const float g_ = mir ? ev[r] - od[r] : ev[r] + od[r];
                            float vdummy;
                            const float xn = mcmc_elem_a<UALGO>(xo[i], g_, xh[i], 0.f, nz[i], 0.f, up.p, vdummy);
                            bad |= (xn != xn);
                            S3_STG(gf + (mir ? H - 1 - yy : yy) * W, xn);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (UPD == 1) {
                    // state and x0hat behind the matrix work, upper rows then mirror rows: 32 values
                    // in flight per lane next to the accumulators is what the register file takes;
                    // the other seven waves of the workgroup cover a wave's wait
                    float xo[16], xh[16], xb[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) xo[r] = gf[(ytop + (r & 3) + 8 * (r >> 2)) * W];
#pragma unroll
                    for (int r = 0; r < 16; ++r) xh[r] = x0f[(ytop + (r & 3) + 8 * (r >> 2)) * W];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int y = ytop + (r & 3) + 8 * (r >> 2);
                        S3_STG(gf + y * W, xo[r] - up.lr * ((ev[r] + od[r]) + (xo[r] - xh[r]) * up.inv_sigma2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) xb[r] = gf[(H - 1 - (ytop + (r & 3) + 8 * (r >> 2))) * W];
#pragma unroll
                    for (int r = 0; r < 16; ++r) xh[r] = x0f[(H - 1 - (ytop + (r & 3) + 8 * (r >> 2))) * W];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int y = H - 1 - (ytop + (r & 3) + 8 * (r >> 2));
                        S3_STG(gf + y * W, xb[r] - up.lr * ((ev[r] - od[r]) + (xb[r] - xh[r]) * up.inv_sigma2));
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int y = ytop + (r & 3) + 8 * (r >> 2);
                    S3_STG(gf + y * W, ev[r] + od[r]);
                    S3_STG(gf + (H - 1 - y) * W, ev[r] - od[r]);
                }
                }
            }
        }
    }
    if (UPD >= 2 && up.nan_flag && __any(bad) && lane == 0 &&
        __hip_atomic_load(up.nan_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicOr(up.nan_flag, 1);
    if (FUSED && blockIdx.x == 0 && blockIdx.y == 0) {
        // workgroup 0, once its tiles are stored (off the path of everybody's first store): every
        // chain's loss (and raw chi^2) and their total, in a fixed order:
        // wave w takes chains w, w + 8, ...; the eight partial totals are added in wave order
        float wtot = 0.f;
        for (int b = strip; b < B; b += S3_THREADS / 64) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int tt = lane; tt < T; tt += 64) {
                const float4 s4 = ((const float4*)fz.S)[(size_t)b * T + tt];
                a0 += s4.x; a1 += s4.y; a2 += s4.z;
            }
            a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
            float f_cp, f_ca, L, c_cp, c_ca, c_fl;
            s3_chain_factors(fz, T, a0, a1, a2, f_cp, f_ca, L, c_cp, c_ca, c_fl);
            wtot += L;
            if (lane == 0) {
                fz.loss[b] = L;
                if (fz.chi2) {
                    float* c = fz.chi2 + (size_t)b * 5;
                    c[0] = 0.f; c[1] = 0.f; c[2] = c_cp; c[3] = c_ca; c[4] = c_fl;
                }
            }
        }
        if (fz.total) {
            if (lane == 0) fac[2 * S3_GCH + strip] = wtot;
            __syncthreads();
            if (tid == 0) {
                float tsum = 0.f;
#pragma unroll
                for (int w = 0; w < S3_THREADS / 64; ++w) tsum += fac[2 * S3_GCH + w];
                *fz.total = tsum;
            }
        }
    }
#undef S3_STG
}

template <int H, int W>
static void launch_adj_s3(const float* ey_ri, const float* ex_ri, int T, int F, const float* gvis,
                          const float* gflux, const float* chain_scale, int B, float out_scale,
                          float* grad, const float* sum_in, int sum_n, float* sum_out,
                          const S3FusedAdj* fz, const S3Update* up, hipStream_t st) {
    int cpw, gx;
    s3_grid(B, T, &cpw, &gx);
    const size_t lds = (size_t)24 * (H / 2) * 16 + (size_t)S3_GCH * S3_GROW * 4 + (2 * S3_GCH + 8) * 4 +
                       (68 + 192 + 256) * 4;
    // fewer workgroups than half the CUs (one or two chains): split a frame's H / 64 tiles over 2 or 4
    int tsplit = 1;
    while (2 * tsplit <= (H / 64) && gx * T * tsplit * 2 <= s3_cu_count()) tsplit *= 2;
#define S3_ADJ_LAUNCH(FU, UP)                                                                              \
    step_allow_lds((const void*)k_dft_adj_s3<H, W, FU, UP>, lds);                                          \
    hipLaunchKernelGGL((k_dft_adj_s3<H, W, FU, UP>), dim3(gx * T, tsplit), dim3(S3_THREADS), lds, st, ey_ri, ex_ri, \
                       gvis, gflux, chain_scale, grad, T, F, B, cpw, out_scale, sum_in, sum_n, sum_out,    \
                       fz ? *fz : S3FusedAdj{}, up ? *up : S3Update{})
    // 0 gradient, 1 optimize, 2 / 3 langevin / hmc (momentum 0) with Philox noise, 4 / 5 the same with the noise read
    const int mode = up ? (up->eps ? 4 + up->p.algo : up->seeds ? 2 + up->p.algo : 1) : 0;
    if (fz) {
        switch (mode) {
            case 0: S3_ADJ_LAUNCH(true, 0); break;
            case 1: S3_ADJ_LAUNCH(true, 1); break;
            case 2: S3_ADJ_LAUNCH(true, 2); break;
            case 3: S3_ADJ_LAUNCH(true, 3); break;
            case 4: S3_ADJ_LAUNCH(true, 4); break;
            default: S3_ADJ_LAUNCH(true, 5); break;
        }
    } else {
        switch (mode) {
            case 0: S3_ADJ_LAUNCH(false, 0); break;
            case 1: S3_ADJ_LAUNCH(false, 1); break;
            case 2: S3_ADJ_LAUNCH(false, 2); break;
            case 3: S3_ADJ_LAUNCH(false, 3); break;
            case 4: S3_ADJ_LAUNCH(false, 4); break;
            default: S3_ADJ_LAUNCH(false, 5); break;
        }
    }
#undef S3_ADJ_LAUNCH
}

static void dispatch_adj_s3(const float* ey_ri, const float* ex_ri, int T, int F, int H, const float* gvis,
                            const float* gflux, const float* chain_scale, int B, float out_scale,
                            float* grad, const float* sum_in, int sum_n, float* sum_out,
                            const S3FusedAdj* fz, const S3Update* up, hipStream_t st) {
    if (H == 256) launch_adj_s3<256, 256>(ey_ri, ex_ri, T, F, gvis, gflux, chain_scale, B, out_scale, grad, sum_in, sum_n, sum_out, fz, up, st);
    else if (H == 128) launch_adj_s3<128, 128>(ey_ri, ex_ri, T, F, gvis, gflux, chain_scale, B, out_scale, grad, sum_in, sum_n, sum_out, fz, up, st);
    else launch_adj_s3<64, 64>(ey_ri, ex_ri, T, F, gvis, gflux, chain_scale, B, out_scale, grad, sum_in, sum_n, sum_out, fz, up, st);
}

extern "C" int bh_dft_adj_sep3(const float* ey_ri, const float* ex_ri, int T, int F, int H, int W,
                               const float* gvis_c64, const float* gflux, const float* chain_scale,
                               int B, float out_scale, float* grad, const float* sum_in, int sum_n,
                               float* sum_out, void* stream) {
    STEP_REQUIRE(ey_ri && ex_ri && gvis_c64 && gflux && grad, "bh_dft_adj_sep3: null pointer");
    STEP_REQUIRE(!sum_out || (sum_in && sum_n > 0), "bh_dft_adj_sep3: sum_out needs sum_in and sum_n > 0");
    STEP_REQUIRE(T > 0 && T <= 65535 && B > 0 && F > 0 && F <= 31,
                 "bh_dft_adj_sep3: need 1 <= F <= 31, 1 <= T <= 65535 (F=%d T=%d)", F, T);
    STEP_REQUIRE(bh_sep3_supported(H, W), "bh_dft_adj_sep3: image %dx%d not built (256, 128, 64 square)", H, W);
    dispatch_adj_s3(ey_ri, ex_ri, T, F, H, gvis_c64, gflux, chain_scale, B, out_scale, grad, sum_in, sum_n,
                    sum_out, nullptr, nullptr, (hipStream_t)stream);
    STEP_CHECK_LAUNCH("bh_dft_adj_sep3");
    return STEP_OK;
}

// bh_update_t (checked) -> the kernel's argument block
static int s3_make_update(const bh_update_t* u, S3Update* out, const char* who) {
    STEP_REQUIRE(u->x_inout && u->x0hat, "%s: update needs x_inout and x0hat", who);
    *out = S3Update{};
    out->x = u->x_inout; out->x0 = u->x0hat; out->lr = u->lr; out->inv_sigma2 = u->inv_sigma2;
    if (!u->mcmc) return STEP_OK;
    const mcmc_params_t& p = *u->mcmc;
    STEP_REQUIRE(p.algo == 0 || (p.algo == 1 && p.momentum == 0.f),
                 "%s: the in-adjoint step is langevin or hmc with momentum 0 (algo %d, momentum %g)", who,
                 p.algo, (double)p.momentum);
    STEP_REQUIRE(u->seeds || u->eps, "%s: the in-adjoint Langevin step takes its noise from per-chain seeds or from `eps`", who);
    STEP_REQUIRE(u->step >= 0, "%s: negative step", who);
    out->p = p; out->eps = u->eps; out->seeds = u->eps ? nullptr : u->seeds; out->step = u->step; out->step_dev = u->step_dev;
    out->nan_flag = u->nan_flag;
    return STEP_OK;
}

// bh_dft_adj_sep3 with the state update in place of the gradient store (bh_update_t):
// MCMCSampler.optimize's x[b][t][n] -= lr * (g + (x - x0hat) * inv_sigma2) (daps.py:210-223) or,
// with update->mcmc, the Langevin / momentum-free HMC step of MCMCSampler.sample with in-kernel
// noise; g = the gradient bh_dft_adj_sep3 would have written.
extern "C" int bh_dft_adj_sep3_update(const float* ey_ri, const float* ex_ri, int T, int F, int H, int W,
                                      const float* gvis_c64, const float* gflux, const float* chain_scale,
                                      int B, float out_scale, const bh_update_t* update,
                                      const float* sum_in, int sum_n, float* sum_out, void* stream) {
    STEP_REQUIRE(ey_ri && ex_ri && gvis_c64 && gflux && update, "bh_dft_adj_sep3_update: null pointer");
    STEP_REQUIRE(!sum_out || (sum_in && sum_n > 0), "bh_dft_adj_sep3_update: sum_out needs sum_in and sum_n > 0");
    STEP_REQUIRE(T > 0 && T <= 65535 && B > 0 && F > 0 && F <= 31,
                 "bh_dft_adj_sep3_update: need 1 <= F <= 31, 1 <= T <= 65535 (F=%d T=%d)", F, T);
    STEP_REQUIRE(bh_sep3_supported(H, W), "bh_dft_adj_sep3_update: image %dx%d not built (256, 128, 64 square)", H, W);
    S3Update up;
    if (int rc = s3_make_update(update, &up, "bh_dft_adj_sep3_update")) return rc;
    dispatch_adj_s3(ey_ri, ex_ri, T, F, H, gvis_c64, gflux, chain_scale, B, out_scale, nullptr, sum_in, sum_n,
                    sum_out, nullptr, &up, (hipStream_t)stream);
    STEP_CHECK_LAUNCH("bh_dft_adj_sep3_update");
    return STEP_OK;
}

// frame_tab is dereferenced by the forward kernel: the table that bh_build_frame_closures fills on the
// HOST has to be handed over as a device copy (or pinned host memory).  A pageable host pointer
// would be a GPU memory fault, so it is refused here; the last accepted pointer is remembered, the
// lookup (no allocation, no synchronisation) runs once per table, not once per step.
static bool s3_gpu_readable(const void* p) {
    static std::atomic<const void*> accepted{nullptr};
    if (accepted.load(std::memory_order_relaxed) == p) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                             // pageable host memory: not an error of ours to keep
        return false;
    }
    if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeHost && a.type != hipMemoryTypeManaged &&
        a.type != hipMemoryTypeArray)
        return false;                                        // hipMemoryTypeUnregistered
    accepted.store(p, std::memory_order_relaxed);
    return true;
}

// ------------------------------------------------------------------------------- fused step
extern "C" int bh_loss_grad_sep3(const float* ey_t, const float* ey_ri, const float* ex_ri, int T, int F,
                                 int H, int W, const float* x, int B, const bh_tables_t* tab,
                                 const int32_t* frame_tab, const bh_meas_t* y, const bh_weights_t* w,
                                 const bh_step_ws_t* ws, const float* chain_scale, float out_scale,
                                 float* loss, float* chi2, float* loss_total, float* grad,
                                 const bh_update_t* update, void* stream) {
    STEP_REQUIRE(ey_t && ey_ri && ex_ri && x && tab && frame_tab && y && w && ws && loss,
                 "bh_loss_grad_sep3: null pointer");
    STEP_REQUIRE(s3_gpu_readable(frame_tab),
                 "bh_loss_grad_sep3: frame_tab must be a DEVICE copy of the table bh_build_frame_closures fills on the host");
    STEP_REQUIRE((grad != nullptr) != (update != nullptr), "bh_loss_grad_sep3: pass either grad or update");
    STEP_REQUIRE(!update || update->x_inout == x, "bh_loss_grad_sep3: update->x_inout must be x");
    S3Update up{};
    if (update)
        if (int rc = s3_make_update(update, &up, "bh_loss_grad_sep3")) return rc;
    STEP_REQUIRE(T > 0 && T <= 65535 && B > 0 && F > 0 && F <= 31,
                 "bh_loss_grad_sep3: need 1 <= F <= 31, 1 <= T <= 65535 (F=%d T=%d)", F, T);
    STEP_REQUIRE(bh_sep3_supported(H, W), "bh_loss_grad_sep3: image %dx%d not built (256, 128, 64 square)", H, W);
    STEP_REQUIRE(tab->n_vis == T * F && tab->T == T, "bh_loss_grad_sep3: tables are for n_vis=%d T=%d", tab->n_vis, tab->T);
    STEP_REQUIRE(y->batch == 1, "bh_loss_grad_sep3: one measurement for all chains (batch %d); use the three-kernel path", y->batch);
    STEP_REQUIRE(!(w->w_vis > 0) && !(w->w_amp > 0), "bh_loss_grad_sep3: visibility / amplitude terms need the three-kernel path");
    STEP_REQUIRE(w->cphase_form == 0 || w->cphase_form == 1, "bh_loss_grad_sep3: cphase_form %d", w->cphase_form);
    STEP_REQUIRE(!(w->w_cp > 0) || (tab->n_tri > 0 && y->cphase && y->sigma_cphase && tab->cp_ptr && tab->cp_ent),
                 "bh_loss_grad_sep3: cphase data / tables missing");
    STEP_REQUIRE(!(w->w_camp > 0) || (tab->n_quad > 0 && y->logcamp && y->sigma_logcamp && tab->ca_ptr && tab->ca_ent),
                 "bh_loss_grad_sep3: logcamp data / tables missing");
    STEP_REQUIRE(y->flux, "bh_loss_grad_sep3: flux data missing");
    STEP_REQUIRE(ws->vis_c64 && ws->flux && ws->dcp && ws->dca && ws->S, "bh_loss_grad_sep3: null workspace");
    hipStream_t st = (hipStream_t)stream;
    const bool use_cp = w->w_cp > 0, use_ca = w->w_camp > 0;
    S3FusedFwd ff{frame_tab, use_cp ? y->cphase : nullptr, use_cp ? y->sigma_cphase : nullptr,
                  use_ca ? y->logcamp : nullptr, use_ca ? y->sigma_logcamp : nullptr, y->flux,
                  ws->dcp, ws->dca, ws->S, tab->n_tri, tab->n_quad, w->cphase_form};
    dispatch_fwd_s3(ey_t, ex_ri, T, F, H, x, B, 0.5f, 0.5f, -INFINITY, ws->vis_c64, ws->flux, &ff, st);
    STEP_CHECK_LAUNCH("bh_loss_grad_sep3 (forward)");
    S3FusedAdj fa{(const float2*)ws->vis_c64, ws->flux, ws->S, ws->dcp, ws->dca, tab->cp_ptr, tab->cp_ent,
                  tab->ca_ptr, tab->ca_ent, y->flux, w->w_cp, w->w_camp, w->w_flux, w->normalize,
                  tab->n_tri, tab->n_quad, loss, chi2, loss_total};
    dispatch_adj_s3(ey_ri, ex_ri, T, F, H, nullptr, nullptr, chain_scale, B, out_scale, grad, nullptr, 0,
                    nullptr, &fa, update ? &up : nullptr, st);
    STEP_CHECK_LAUNCH("bh_loss_grad_sep3 (adjoint)");
    return STEP_OK;
}
