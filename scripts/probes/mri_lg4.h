// Dynamic-MRI Langevin step, two workgroups per CU: forward FFT2 + mask + residual + loss +
// adjoint of forward_operator/dynamic_mri.py:42-48 with base.py:40-41, fused per frame (SURVEY
// Appendix B "MRI"), for masks of at most LG4_NKC_MAX kept rows per frame (every shipped mode).
//
// mri_lg2.h keeps one 140 KB workgroup per CU, so a frame's load, transform and store phases
// overlap with nothing but each other's pipeline.  Here a workgroup needs 80 KB of LDS and has
// one wave per SIMD, so two share a CU and one frame's transforms run under the other's streams
// (six waves per workgroup do not work: the second workgroup's waves want the same SIMDs as the
// first one's, a SIMD would need four waves of 168 registers, and the CU takes one workgroup):
//   * column FFT of 192 points = 4 lanes x 48 points, entirely in registers: a wave owns 16
//     columns, the four lanes of a quad (lane & 3 = r) hold the rows n = 4 n' + r of one column
//     (96 registers); each lane runs a 48-point FFT (12 x 4, compile-time twiddles), multiplies by
//     W192^(r k') and the quad finishes with a 4-point DFT on DPP quad permutes -- no strip
//     buffer, no barrier;
//   * 256 threads (4 waves x 16 columns, one wave per SIMD) take the frame in three passes of 64
//     columns; the next pass's pixels are requested group by group into the registers that the
//     current pass has just finished with;
//   * the kept k_y rows go to an LDS row buffer of exactly nkc x 192 complex values -- no padding:
//     52 rows + the twiddle table + byte-sized row tables are 81 696 of the 81 920 bytes that two
//     workgroups can each have;
//   * row phase as in mri_lg2.h (three in-place passes, the inverse in the transposed
//     decomposition), four rounds of 16 rows, measurement rows fetched a round ahead.
// Device code only (DPP); parity against the pipelined kernel and the oracle in tests/test_gpu_mri.py.
//
// STATUS (round 2): variant 3 of mri_loss_grad, not the default.  Two
// workgroups do share a CU (wave-cycle counters), nothing spills, but a wave's load or store
// instruction covers 4 rows x 16 columns = 64-byte row segments, and a copy kernel with exactly this
// pattern runs 1.81 GB in 0.39-0.41 ms where 128-byte segments (mri_lg2.h) or a linear copy take
// 0.34 ms (scripts/probes/mri_pattern_probe.hip): measured 0.523 ms at 256 chains against 0.503 ms
// for mri_lg2.h, 0.084 against 0.082 ms at 32.
#pragma once
#include "mri_lg2.h"

#define LG4_THREADS 256
#define LG4_NP 48                 // points per lane
#define LG4_NKC_MAX 52            // (52 * 192 + 192) * 8 + 192 + 64 + 32 = 81696 <= 81920
#define LG4_NPASS 3               // column passes of LG4_THREADS / 4 = 64 columns
#define LG4_GPRE 7                // register groups of a pass requested under the previous pass
#define LG4_ROUNDS 4              // row-phase rounds of LG4_THREADS / 16 = 16 rows
#define LG4_FENCE() LG2_FENCE()

struct Lg4Lds {
    cf32* Rk;               // [nkc][192] kept k_y rows
    cf32* tw;               // [192] (cos, sin)(2 pi m / 192)
    signed char* rowpos;    // [192] position of a k_y row in the kept-row list, -1 if masked out
    unsigned char* rowl;    // [64]  the kept-row list
    float* red;             // [8]
};

static inline size_t lg4_lds_bytes(int nkc) {
    return (size_t)(nkc + 1) * LG2_R * sizeof(cf32) + LG2_R + 64 + 8 * sizeof(float);
}

__device__ __forceinline__ Lg4Lds lg4_carve(float* lds, int nkc) {
    Lg4Lds L;
    L.Rk = (cf32*)lds;
    L.tw = L.Rk + (size_t)nkc * LG2_R;
    L.rowpos = (signed char*)(L.tw + LG2_R);
    L.rowl = (unsigned char*)(L.rowpos + LG2_R);
    L.red = (float*)(L.rowl + 64);
    return L;
}

// (cos, sin)(2 pi k / 48), k = b c <= 33
__device__ __forceinline__ constexpr float lg4_c48(int k) {
    constexpr float t[34] = {1.f, 0.991444861f, 0.965925826f, 0.923879533f, 0.866025404f, 0.79335334f,
        0.707106781f, 0.608761429f, 0.5f, 0.382683432f, 0.258819045f, 0.130526192f, 0.f, -0.130526192f,
        -0.258819045f, -0.382683432f, -0.5f, -0.608761429f, -0.707106781f, -0.79335334f, -0.866025404f,
        -0.923879533f, -0.965925826f, -0.991444861f, -1.f, -0.991444861f, -0.965925826f, -0.923879533f,
        -0.866025404f, -0.79335334f, -0.707106781f, -0.608761429f, -0.5f, -0.382683432f};
    return t[k];
}
__device__ __forceinline__ constexpr float lg4_s48(int k) {
    constexpr float t[34] = {0.f, 0.130526192f, 0.258819045f, 0.382683432f, 0.5f, 0.608761429f,
        0.707106781f, 0.79335334f, 0.866025404f, 0.923879533f, 0.965925826f, 0.991444861f, 1.f, 0.991444861f,
        0.965925826f, 0.923879533f, 0.866025404f, 0.79335334f, 0.707106781f, 0.608761429f, 0.5f, 0.382683432f,
        0.258819045f, 0.130526192f, 0.f, -0.130526192f, -0.258819045f, -0.382683432f, -0.5f, -0.608761429f,
        -0.707106781f, -0.79335334f, -0.866025404f, -0.923879533f};
    return t[k];
}

// Values declared modified at this point of the program: everything they depend on is computed
// before it, every use comes after.  A scheduling fence alone does not do that -- it binds the
// machine scheduler, but the order in which a basic block of thousands of instructions is first
// laid out already has all the sums of a transform ahead of all its differences.
__device__ __forceinline__ void lg4_pin(cf32& a, cf32& b, cf32& c) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y));
}
__device__ __forceinline__ void lg4_pin(cf32& a, cf32& b, cf32& c, cf32& d) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y), "+v"(d.x), "+v"(d.y));
}

// fft12 of fft_small.h with every butterfly pinned: with 96 registers of data per lane the budget
// of 168 leaves no room for the order the compiler likes (all the sums of a 12-point transform
// first, then all the differences: 24 inputs + 24 outputs + the temporaries live at once).
template <int SIGN>
__device__ __forceinline__ void lg4_fft12(cf32 (&v)[12]) {
    const float h = 0.8660254037844386f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        fft3<SIGN>(v[b], v[4 + b], v[8 + b]);
        lg4_pin(v[b], v[4 + b], v[8 + b]);
    }
    v[4 * 1 + 1] = c_rot<SIGN>(v[4 * 1 + 1], h, 0.5f);
    v[4 * 1 + 2] = c_rot<SIGN>(v[4 * 1 + 2], 0.5f, h);
    v[4 * 1 + 3] = c_muli<SIGN>(v[4 * 1 + 3]);
    v[4 * 2 + 1] = c_rot<SIGN>(v[4 * 2 + 1], 0.5f, h);
    v[4 * 2 + 2] = c_rot<SIGN>(v[4 * 2 + 2], -0.5f, h);
    v[4 * 2 + 3] = c_make(-v[4 * 2 + 3].x, -v[4 * 2 + 3].y);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
        lg4_pin(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    }
    cf32 o[12];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) o[c + 3 * d] = v[4 * c + d];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = o[i];
}

// 48-point FFT in registers, input in natural order v[n'], n' = 4 a + b; output F[c + 12 d] at
// v[4 c + d]:  12-point over a for each b, twiddle W48^(b c), 4-point over b for each c.
template <int SIGN>
__device__ __forceinline__ void lg4_fft48_dit(cf32 (&v)[LG4_NP]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cf32 t[12];
#pragma unroll
        for (int a = 0; a < 12; ++a) t[a] = v[4 * a + b];
        lg4_fft12<SIGN>(t);
#pragma unroll
        for (int c = 0; c < 12; ++c) v[4 * c + b] = t[c];
        LG4_FENCE();
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
#pragma unroll
        for (int b = 1; b < 4; ++b)
            if (c) v[4 * c + b] = c_rot<SIGN>(v[4 * c + b], lg4_c48(b * c), lg4_s48(b * c));
        fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
        lg4_pin(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    }
}

// The mirror image: input H[c + 12 d] at v[4 c + d]; output in natural order v[n'], n' = 4 a + b.
template <int SIGN>
__device__ __forceinline__ void lg4_fft48_dif(cf32 (&v)[LG4_NP]) {
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
#pragma unroll
        for (int b = 1; b < 4; ++b)
            if (c) v[4 * c + b] = c_rot<SIGN>(v[4 * c + b], lg4_c48(b * c), lg4_s48(b * c));
        lg4_pin(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cf32 t[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) t[c] = v[4 * c + b];
        lg4_fft12<SIGN>(t);
#pragma unroll
        for (int a = 0; a < 12; ++a) v[4 * a + b] = t[a];
        LG4_FENCE();
    }
}


// value of the quad lane (lane ^ M), M = 1 or 2
template <int M>
__device__ __forceinline__ float lg4_qx(float v) {
    constexpr int ctrl = M == 1 ? 0xB1 : 0x4E;       // quad_perm [1,0,3,2] / [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}

__device__ __forceinline__ float lg4_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the workgroup, valid in thread 0; red = float[8]; ends unsynced
__device__ __forceinline__ float lg4_block_sum(float v, float* red) {
    v = lg4_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int k = 0; k < LG4_THREADS / 64; ++k) t += red[k];
    return t;
}

// One workgroup (256 threads) per frame.  Arguments as lg2_body: YQ = a measurement per chain
// (every workgroup sums the masked-out rows of its own measurement frame), !YQ = one measurement
// for all chains (the first 16 chains' workgroups each sum a 16th, into unk_part).
template <bool YQ>
__device__ __forceinline__ void lg4_body(float* lds, const float* __restrict__ x, const float* __restrict__ y,
                                         int y_batch, const int* __restrict__ rows,
                                         const int* __restrict__ nrows, int mask_frames, int T, int nframes,
                                         int nkc, const float* __restrict__ chain_scale,
                                         float* __restrict__ loss_part, float* __restrict__ unk_part,
                                         float* __restrict__ grad) {
    constexpr int R = LG2_R, Q = LG2_R / 4, NT = LG4_THREADS, NP = LG4_NP, PW = LG4_THREADS / 4;
    const Lg4Lds L = lg4_carve(lds, nkc);
    const int tid = threadIdx.x;
    const size_t fsz = (size_t)2 * R * R;
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int b = frame / T, t = frame - b * T;
    const int mf = mask_frames == 1 ? 0 : t;
    const float* xr = x + (size_t)frame * fsz;
    const float* xi = xr + R * R;
    const float* yr = y + ((size_t)(y_batch > 1 ? b : 0) * T + t) * fsz;
    const float* yi = yr + R * R;
    const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);

    // column roles: wave = 16 columns of a pass, quad = one column, lane & 3 = r
    const int r = tid & 3, colq = (tid & 63) >> 2, wave = tid >> 6;
    const int column0 = 16 * wave + colq;                        // + 64 per pass
    const unsigned off0 = (unsigned)(r * R + column0) * 4u;      // (row r, column), bytes
    const int mq = (r >> 1) + 2 * (r & 1);           // this lane ends up with k = k' + 48 mq
    const float s_r0 = (r & 1) ? -1.f : 1.f, s_r1 = (r & 2) ? -1.f : 1.f;
    const bool lane3 = r == 3;

    // A lane's 48 values of a pass: register v[4 c + d] holds image row 4 (c + 12 d) + r.  The
    // forward transform runs in the form whose first stage works on four CONSECUTIVE registers
    // (4-point over d, twiddle, then 12-point over c; result in natural order v[k']), and the
    // quad stage walks the registers in order, so the registers of group g = {4 g .. 4 g + 3} are
    // free as soon as its kept rows are in the row buffer: the next pass's pixels are requested
    // into them right there, and are first needed when the next transform reaches group g -- a
    // whole pass later for the first groups, 11 four-point transforms later for the last one.
    // Each phase derives its addresses from an opaque copy of the offset: left to itself the
    // compiler shares the 96 load and store addresses of a lane across the whole kernel and keeps
    // them in registers from the first load to the last store.
#define LG4_LOAD_GROUP(ob, g)                                                          \
    _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                    \
        const unsigned o = (ob) + (unsigned)(4 * ((g) + 12 * d)) * (R * 4u);           \
        v[4 * (g) + d] = c_make(lg2_ld(xr, o), lg2_ld(xi, o));                         \
    }
    // (a wave can have 63 loads outstanding and the compiler gives up exact wait counts beyond
    // that: groups 0 .. LG4_GPRE-1, 56 loads, travel under the previous pass, the other five are
    // requested at the top of their own pass, behind them in the queue)
    cf32 v[NP];
    {
        unsigned ob = off0;
        LG2_OPAQUE(ob);
#pragma unroll
        for (int g = 0; g < LG4_GPRE; ++g) LG4_LOAD_GROUP(ob, g)
    }

    // tables of the mask frame
    const int nk = nrows[mf];
    for (int e = tid; e < R; e += NT) {
        float sn, cn;
        sincospif((float)e * (1.0f / 96.0f), &sn, &cn);
        L.tw[e] = c_make(cn, sn);
        L.rowpos[e] = -1;
    }
    __syncthreads();
    if (tid < nk) {
        const int rr = rows[(size_t)mf * R + tid];
        L.rowpos[rr] = (signed char)tid;
        L.rowl[tid] = (unsigned char)rr;
    }
    __syncthreads();

    // bit k' of keep48: one of the rows k' + 48 q, q < 4 -- the four outputs of a quad for the
    // register that holds k' -- is kept.  The same for every lane of the workgroup (the mask
    // belongs to the frame), so the quad stage of a value nobody keeps is skipped by a scalar branch.
    unsigned long long keep48;
    {
        const int e = tid & 63, k = e < 48 ? e : 0;
        const int any = (int)L.rowpos[k] & (int)L.rowpos[k + 48] & (int)L.rowpos[k + 96] & (int)L.rowpos[k + 144];
        keep48 = __ballot(e < 48 && any >= 0);
    }

    // the x-independent part of the loss: |y|^2 of the masked-out rows
    float acc = 0.f, acc_u = 0.f;
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;
    const bool has_share = !YQ && b < nparts;        // uniform over the workgroup
    if (YQ || has_share) {
        const int r0 = YQ ? 0 : b * R / nparts, r1 = YQ ? R : (b + 1) * R / nparts;
        float s = 0.f;
#pragma unroll 4
        for (int e = tid; e < (r1 - r0) * 2 * Q; e += NT) {
            const int row = r0 + e / (2 * Q), rq = e % (2 * Q), pl = rq >= Q;
            if (L.rowpos[row] < 0) {
                const float4 w = *(const float4*)(yr + (size_t)pl * R * R + row * R + 4 * (rq - pl * Q));
                s += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
            }
        }
        if (YQ) acc = s; else acc_u = s;
    }

    // ---------------------------------------------------------------- forward columns
    const signed char* rp = L.rowpos + 48 * mq;
    // one pass: transform, twiddle W192^(r k'), 4-point DFT across the quad (partner r ^ 2, lane 3
    // times -i, partner r ^ 1; lane r then holds X[k' + 48 mq]), kept rows into the row buffer;
    // PREFETCH: group by group the next pass's pixels into the registers just freed
#define LG4_FWD_PASS(pass, PREFETCH)                                                   \
    {                                                                                  \
        {                                                                              \
            unsigned oc = off0 + (unsigned)(pass) * (PW * 4u);                         \
            LG2_OPAQUE(oc);                                                            \
            _Pragma("unroll") for (int g = LG4_GPRE; g < 12; ++g) LG4_LOAD_GROUP(oc, g) \
        }                                                                              \
        lg4_fft48_dif<-1>(v);                                                          \
        LG4_FENCE();                                                                   \
        cf32* col = L.Rk + column0 + PW * (pass);                                      \
        unsigned ob = off0 + (unsigned)((pass) + 1) * (PW * 4u);                       \
        LG2_OPAQUE(ob);                                                                \
        _Pragma("unroll") for (int g = 0; g < 12; ++g) {                               \
            cf32 w[4];                                                                 \
            int jj[4];                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) w[i] = L.tw[r * (4 * g + i)]; \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) jj[i] = rp[4 * g + i];       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) if ((keep48 >> (4 * g + i)) & 1) { \
                const cf32 u = c_rot<-1>(v[4 * g + i], w[i].x, w[i].y);                \
                cf32 w1 = c_make(s_r1 * u.x + lg4_qx<2>(u.x), s_r1 * u.y + lg4_qx<2>(u.y)); \
                if (lane3) w1 = c_make(w1.y, -w1.x);                                   \
                const cf32 w2 = c_make(s_r0 * w1.x + lg4_qx<1>(w1.x), s_r0 * w1.y + lg4_qx<1>(w1.y)); \
                const float sg = (i & 1) ? -1.f : 1.f;   /* (-1)^k, k = 4 g + i + 48 mq */ \
                /* value and address are complete before the branch, so that it guards one */ \
                /* LDS store and needs no jump around it */                            \
                cf32 o = c_make(sg * w2.x, sg * w2.y);                                 \
                cf32* dst = col + jj[i] * R;                                           \
                asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(dst));                    \
                if (jj[i] >= 0) *dst = o;                                              \
            }                                                                          \
            if (PREFETCH && g < LG4_GPRE) LG4_LOAD_GROUP(ob, g)                        \
            LG4_FENCE();                                                               \
        }                                                                              \
    }
    static_assert(LG4_NPASS == 3, "two prefetching passes in a loop, the last one written out");
    // (the last pass is its own code: inside the loop, a pass that requests nothing would leave
    // the wait counts of every pass at "wait for everything")
#pragma unroll 1
    for (int pass = 0; pass < LG4_NPASS - 1; ++pass) LG4_FWD_PASS(pass, true)
    LG4_FWD_PASS(LG4_NPASS - 1, false)
#undef LG4_FWD_PASS
#undef LG4_LOAD_GROUP

    // ---------------------------------------------------------------- row phase
    // Row j of round q is handled by lanes 16 j' .. 16 j' + 15 (j = 16 q + j'): pass A (16-point,
    // twiddle) on lanes c16 < 12, then pass B (12-point) + residual + loss + the first pass of the
    // inverse (transposed decomposition k = c + 16 d -> n = q + 12 p) on all 16, then the
    // inverse's 16-point pass on lanes c16 < 12.
    const bool with_grad = grad != nullptr;
    {
        const int c16 = tid & 15, jr = tid >> 4;
        // measurement rows of this thread's four pass-B tasks (row j, group c): kx = c + 16 d.  All
        // four rounds are requested here -- the column registers are free in the row phase -- and
        // none under a branch (a round past the last kept row fetches row 0 and drops it), so
        // that each round's wait count is exact and pass A and a barrier lie between request and use
        float y0r[12], y0i[12], y1r[12], y1i[12], y2r[12], y2i[12], y3r[12], y3i[12];
#define LG4_YLOAD(j_, re_, im_)                                                        \
        {                                                                              \
            const unsigned o_ = (unsigned)((int)L.rowl[(j_) < nk ? (j_) : 0] * R + c16) * 4u; \
            _Pragma("unroll") for (int d = 0; d < 12; ++d) {                           \
                re_[d] = lg2_ld(yr, o_ + 64u * d);                                     \
                im_[d] = lg2_ld(yi, o_ + 64u * d);                                     \
            }                                                                          \
        }
        LG4_YLOAD(jr, y0r, y0i)
        LG4_YLOAD(jr + 16, y1r, y1i)
        LG4_YLOAD(jr + 32, y2r, y2i)
        LG4_YLOAD(jr + 48, y3r, y3i)
        __syncthreads();                             // Rk complete
#pragma unroll 1
        for (int q = 0; q < LG4_ROUNDS; ++q) {
            const int j = jr + 16 * q;
            if (j < nk && c16 < 12) {
                cf32* row = L.Rk + j * R + c16;
                cf32 a[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = row[12 * i];
                fft192_pass_a<-1>(a, c16, L.tw);
#pragma unroll
                for (int i = 0; i < 16; ++i) row[12 * i] = a[i];
            }
        }
        __syncthreads();
#define LG4_ROW_B(j_, re_, im_)                                                        \
        if ((j_) < nk) {                                                               \
            cf32* row = L.Rk + (j_) * R + 12 * c16;                                    \
            cf32 a[12];                                                                \
            _Pragma("unroll") for (int i = 0; i < 12; ++i) a[i] = row[i];              \
            fft192_pass_b<-1>(a);            /* a[d] = (-1)^ky FFT2(z)[ky][c + 16 d] */ \
            const float sx = lg2_sgn(c16), sc = cs2 * sx * lg2_sgn((int)L.rowl[j_]);   \
            _Pragma("unroll") for (int d = 0; d < 12; ++d) {                           \
                const float rr = sx * a[d].x - re_[d], ri = sx * a[d].y - im_[d];      \
                acc += rr * rr + ri * ri;                                              \
                a[d] = c_make(sc * rr, sc * ri);                                       \
            }                                                                          \
            if (with_grad) {                                                           \
                fft12<1>(a);                                                           \
                _Pragma("unroll") for (int q_ = 1; q_ < 12; ++q_) {                    \
                    const cf32 w = L.tw[q_ * c16];                                     \
                    a[q_] = c_rot<1>(a[q_], w.x, w.y);                                 \
                }                                                                      \
                _Pragma("unroll") for (int i = 0; i < 12; ++i) row[i] = a[i];          \
            }                                                                          \
        }
        LG4_ROW_B(jr, y0r, y0i)
        LG4_FENCE();
        LG4_ROW_B(jr + 16, y1r, y1i)
        LG4_FENCE();
        LG4_ROW_B(jr + 32, y2r, y2i)
        LG4_FENCE();
        LG4_ROW_B(jr + 48, y3r, y3i)
#undef LG4_ROW_B
#undef LG4_YLOAD
        if (with_grad) {
            __syncthreads();
#pragma unroll 1
            for (int q = 0; q < LG4_ROUNDS; ++q) {
                const int j = jr + 16 * q;
                if (j < nk && c16 < 12) {
                    cf32* row = L.Rk + j * R + c16;
                    cf32 a[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = row[12 * i];
                    fft16<1>(a);
#pragma unroll
                    for (int i = 0; i < 16; ++i) row[12 * i] = a[i];
                }
            }
        }
    }

    if (with_grad) {
        __syncthreads();
        // ------------------------------------------------------------ inverse columns
        float* dr = grad + (size_t)frame * fsz;
        float* di = dr + R * R;
#pragma unroll 1
        for (int pass = 0; pass < LG4_NPASS; ++pass) {
            const cf32* col = L.Rk + column0 + PW * pass;
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                int jj[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) jj[i] = rp[4 * g + i];
                cf32 w[4], tq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {        // no branch: a masked-out row reads row 0 and drops it
                    const cf32 q = col[(jj[i] >= 0 ? jj[i] : 0) * R];
                    w[i] = c_make(jj[i] >= 0 ? q.x : 0.f, jj[i] >= 0 ? q.y : 0.f);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) tq[i] = L.tw[r * (4 * g + i)];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if ((keep48 >> (4 * g + i)) & 1) {
                        cf32 q1 = c_make(s_r0 * w[i].x + lg4_qx<1>(w[i].x), s_r0 * w[i].y + lg4_qx<1>(w[i].y));
                        if (lane3) q1 = c_make(-q1.y, q1.x);
                        const cf32 q2 = c_make(s_r1 * q1.x + lg4_qx<2>(q1.x), s_r1 * q1.y + lg4_qx<2>(q1.y));
                        v[4 * g + i] = c_rot<1>(q2, tq[i].x, tq[i].y);
                    } else {
                        v[4 * g + i] = c_make(0.f, 0.f);
                    }
                }
                LG4_FENCE();
            }
            lg4_fft48_dit<1>(v);                     // natural k' in, image row 4 (c + 12 d) + r at v[4 c + d]
            unsigned oh = off0 + (unsigned)pass * (PW * 4u);
            LG2_OPAQUE(oh);
#pragma unroll
            for (int c = 0; c < 12; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned o = oh + (unsigned)(4 * (c + 12 * d)) * (R * 4u);
                    lg2_st(dr, o, v[4 * c + d].x);
                    lg2_st(di, o, v[4 * c + d].y);
                }
        }
    }
    {
        __syncthreads();
        const float tot = lg4_block_sum(acc, L.red);
        if (tid == 0) loss_part[frame] = tot;
        if (has_share) {
            __syncthreads();
            const float tot_u = lg4_block_sum(acc_u, L.red);
            if (tid == 0) unk_part[frame] = tot_u;
        }
    }
}
