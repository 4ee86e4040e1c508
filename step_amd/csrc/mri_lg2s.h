// Dynamic-MRI Langevin step for batches that leave most of the chip idle: NP = 6 / SP workgroups
// SHARE a frame (mri_lg2.h gives a frame to one workgroup, so B chains keep B * T CUs busy: 48 of
// 256 at the 4 chains per GPU that BASELINE config 5 produces, 12 at the reference's B = 1).
//
// Workgroup (frame, part) owns SP of the frame's six 32-column strips in the two column phases --
// the long ones: 17 and 14 of a frame's 41 us in mri_lg2.h -- and the kept k_y rows cross from the
// forward column phase to the row phase through a global scratch of nkc x 192 complex values per
// frame (80 KB, L2-resident):
//   1. forward column passes of the own strips, kept rows -> LDS row buffer and scratch
//      (arrive / wait on the frame's counter)
//   2. the other parts' columns scratch -> LDS; lg2_row_phase on ALL kept rows in every part (row
//      FFT, residual against the measurement, loss, inverse row FFT: 7 us, cheaper than a second
//      exchange, which costs a release, a wait for the slowest partner and an acquire: 3-6 us,
//      plus the trip through the scratch)
//   3. inverse column passes of the own strips out of the LDS row buffer, gradient stores.
// Arrive / wait: thread 0 releases at agent scope, adds, spins (bounded: a wait that does not end
// sets the error word and goes on rather than hang), acquires.  The launcher keeps the grid
// within the number of CUs, so every workgroup is resident while its partners wait.  Workgroup
// ids are dealt so that the parts of a frame share an XCD (ids go round the XCDs): the exchange
// then stays in one L2; correctness does not depend on it.
// The last workgroup to finish adds the partial losses up (no finish launch) and zeroes the
// counters: a zeroed workspace is left zeroed.
// Every transform runs the code of mri_lg2.h on the same values in the same order: the gradient
// is bit-identical to k_mri_loss_grad2 / 2p; the loss differs in its summation order only.
#pragma once
#include "mri_lg2.h"

#define LG2S_SPIN_LIMIT 4000000u
// -DLG2S_TIMING: workgroup 0 leaves 100 MHz timestamps of its phases behind the counters (probe builds only)
#ifdef LG2S_TIMING
#define LG2S_TICK(k) do { if (ctx.block() == 0 && threadIdx.x == 0) ((unsigned long long*)(cnt + 2 * nframes + 2))[k] = wall_clock64(); } while (0)
#define LG2S_TK(k) (ctx.block() == 0 ? (unsigned long long*)(cnt + 2 * nframes + 2) + (k) : nullptr)
#else
#define LG2S_TK(k) nullptr
#define LG2S_TICK(k) do { } while (0)
#endif

// counters: [0, nframes) arrivals per frame, [nframes] finished workgroups,
// [nframes + 1] error word (a wait gave up)
__device__ __forceinline__ void lg2s_arrive_wait(unsigned* c, unsigned target, unsigned* err,
                                                 unsigned long long* tk = nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (tk) tk[0] = wall_clock64();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tk) tk[1] = wall_clock64();
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > LG2S_SPIN_LIMIT) {
                __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (tk) tk[2] = wall_clock64();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tk) tk[3] = wall_clock64();
    }
    __syncthreads();
}

// sum of two values over the workgroup in a fixed order (waves by shuffle, then the eight wave
// sums in wave order), results in every thread; ends synced
__device__ __forceinline__ void lg2s_sum2(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int k = 0; k < LG2_THREADS / 64; ++k) { ta += red[2 * k]; tb += red[2 * k + 1]; }
    __syncthreads();
    a = ta; b = tb;
}

// x, grad: [nframes][2][R][R]; y: [T][2][R][R] (one measurement for all chains);
// part_out: [nframes]; unk_out: [nframes * NP]; zs_all: [nframes][nkc][R] complex; cnt: see above.
template <int SP, class Ctx>
__device__ __forceinline__ void lg2s_body(const Ctx& ctx, const float* __restrict__ x,
                                          const float* __restrict__ y, const int* __restrict__ rows,
                                          const int* __restrict__ nrows, int mask_frames, int T,
                                          int nframes, int nkc, const float* __restrict__ chain_scale,
                                          float* part_out, float* unk_out, float* __restrict__ grad,
                                          cf32* zs_all, unsigned* cnt, float* __restrict__ loss,
                                          float* __restrict__ total) {
    constexpr int R = LG2_R, SW = LG2_SW, RS = LG2_RS, Q = LG2_R / 4, NP = LG2_NS / SP;
    static_assert(SP == 1 || SP == 2 || SP == 3, "strips per workgroup");
    const Lg2Lds L = lg2_carve(ctx.lds(), nkc);
    const int tid = ctx.tid();
    const size_t fsz = (size_t)2 * R * R;
#define LG2_IDS(t_)                                                                    \
    int t_ = tid;                                                                      \
    LG2_OPAQUE(t_);                                                                    \
    const int col = t_ & 31, g = t_ >> 5;                                              \
    const bool a_task = g < 12;                                                        \
    const int bb = g < 12 ? g : 0;                                                     \
    const unsigned offA = (unsigned)(bb * R + col) * 4u;                               \
    const unsigned offB = (unsigned)(g * R + col) * 4u;                                \
    (void)offA; (void)offB; (void)a_task;
#define LG2_RP(tab, i) ((int)(signed char)((tab)[(i) >> 2] >> (8 * ((i) & 3))))

    // block id -> (frame, part): ids id, id + 8, id + 16, ... sit on one XCD
    const int bid = ctx.block(), q = bid >> 3;
    const int part = q % NP, frame = (q / NP) * 8 + (bid & 7);
    if (frame >= nframes) return;
    const int b = frame / T, t = frame - b * T;
    const int mf = mask_frames == 1 ? 0 : t;
    const int s0 = part * SP;
    unsigned* err = cnt + nframes + 1;

    const float* xr = x + (size_t)frame * fsz;
    const float* xi = xr + R * R;
    const float* yr = y + (size_t)t * fsz;
    const float* yi = yr + R * R;
    cf32* zs = zs_all + (size_t)frame * nkc * R;

    LG2S_TICK(0);
    cf32 a[SP][16];
    const int my_row = rows[(size_t)mf * R + (tid < R ? tid : 0)];
    {
        LG2_IDS(t0)
#pragma unroll
        for (int k = 0; k < SP; ++k) lg2_load_strip(xr, xi, offA + (unsigned)(s0 + k) * (SW * 4u), a[k]);
    }
    const int nk = nrows[mf];
    lg2_tables(ctx, L, my_row, nk);
    unsigned rpB[3], rpA[4];
    {
        LG2_IDS(t1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rpA[i] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rpA[i] |= (unsigned)((a_task ? L.rowpos[12 * (4 * i + k) + bb] : -1) & 0xff) << (8 * k);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            rpB[i] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rpB[i] |= (unsigned)(L.rowpos[g + 16 * (4 * i + k)] & 0xff) << (8 * k);
        }
    }
    const float cs2 = 2.f * (chain_scale ? chain_scale[b] : 1.f);
    float acc = 0.f, acc_u = 0.f;
    LG2S_TICK(1);

    // ---------------------------------------------------------------- forward column phase
    {
        LG2_IDS(t2)
        const float sgB = lg2_sgn(g);                    // (-1)^ky for ky = g + 16 d
        cf32 u[12];
#define LG2S_PASS_B(strip)                                                             \
        {                                                                              \
            fft192_pass_b<-1>(u);                                                      \
            const int n = (strip) * SW + col, np_ = lg2_phys(n);                       \
            _Pragma("unroll") for (int d = 0; d < 12; ++d) {                           \
                const int j = LG2_RP(rpB, d);                                          \
                if (j >= 0) {                                                          \
                    const cf32 v_ = c_make(sgB * u[d].x, sgB * u[d].y);                \
                    zs[j * R + n] = v_;                                                \
                    L.Rk[j * RS + np_] = v_;                                           \
                }                                                                      \
            }                                                                          \
        }
#pragma unroll
        for (int k = 0; k < SP; ++k) {
            if (k > 0) LG2S_PASS_B(s0 + k - 1)
            LG2_FENCE();
            if (a_task) {
                fft16<-1>(a[k]);
#pragma unroll
                for (int c = 1; c < 16; ++c) {
                    const cf32 w = L.tw[bb * c];
                    a[k][c] = c_rot<-1>(a[k][c], w.x, w.y);
                }
            }
            ctx.sync();
            if (a_task) {
#pragma unroll
                for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = a[k][c];
            }
            ctx.sync();
#pragma unroll
            for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
        }
        LG2S_PASS_B(s0 + SP - 1)
#undef LG2S_PASS_B
    }
    LG2S_TICK(2);
    // this workgroup's share of the masked-out measurement rows (a constant of the measurement
    // that every chain's loss contains): the first 16 chains' workgroups sum a share each.
    // Here, ahead of the wait: the loads travel while the partners arrive
    const int nparts = nframes / T < LG2_UNK_PARTS ? nframes / T : LG2_UNK_PARTS;
    if (b < nparts) {
        const int nshare = nparts * NP, sid = b * NP + part;
        const int r0 = sid * R / nshare, r1 = (sid + 1) * R / nshare, ne = (r1 - r0) * 2 * Q;
#pragma unroll 1
        for (int e0 = tid; e0 < ne; e0 += 4 * LG2_THREADS) {
            float4 v[4];
            bool m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + i * LG2_THREADS, ec = e < ne ? e : 0;
                const int row = r0 + ec / (2 * Q), r = ec % (2 * Q), pl = r >= Q;
                m[i] = e < ne && L.rowpos[row] < 0;
                v[i] = *(const float4*)(yr + (size_t)pl * R * R + row * R + 4 * (r - pl * Q));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float qv = v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
                acc_u += m[i] ? qv : 0.f;
            }
        }
    }
    lg2s_arrive_wait(cnt + frame, NP, err, LG2S_TK(10));
    LG2S_TICK(3);
    // the partners' columns of the kept rows: scratch -> LDS row buffer, all loads of a thread
    // in flight before its first LDS store
    {
        constexpr int OW = R - SP * SW;                  // columns the partners own
        const int ne = nk * OW;
#pragma unroll 1
        for (int e0 = tid; e0 < ne; e0 += 8 * LG2_THREADS) {
            cf32 v[8];
            int dst[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * LG2_THREADS, ec = e < ne ? e : 0;
                const int j = ec / OW, c = ec - j * OW;
                const int n = c < s0 * SW ? c : c + SP * SW;
                dst[i] = e < ne ? j * RS + lg2_phys(n) : -1;
                v[i] = zs[(size_t)j * R + n];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (dst[i] >= 0) L.Rk[dst[i]] = v[i];
        }
    }
    LG2S_TICK(4);
    // ---------------------------------------------------------------- row phase, every kept row
    lg2_row_phase<LG2_THREADS>(ctx, L.Rk, L.tw, L.rowl, rows + (size_t)mf * R, nk, yr, yi, cs2,
                               grad != nullptr, frame, acc);
    LG2S_TICK(5);
    if (grad) {
        ctx.sync();
        // ------------------------------------------------------------ inverse column phase
        float* dr = grad + (size_t)frame * fsz;
        float* di = dr + R * R;
        LG2_IDS(t3)
        cf32 av[16], u[12];
#define LG2_INV_A(strip)                                                              \
        if (a_task) {                                                                 \
            const int n_ = lg2_phys((strip) * SW + col);                              \
            unsigned rq_[4] = {rpA[0], rpA[1], rpA[2], rpA[3]};                       \
            LG2_OPAQUE(rq_[0]); LG2_OPAQUE(rq_[1]); LG2_OPAQUE(rq_[2]); LG2_OPAQUE(rq_[3]); \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) {                          \
                const int j_ = LG2_RP(rq_, i);                                        \
                av[i] = j_ >= 0 ? L.Rk[j_ * RS + n_] : c_make(0.f, 0.f);              \
            }                                                                         \
            fft16<1>(av);                                                             \
            _Pragma("unroll") for (int c = 1; c < 16; ++c) {                          \
                const cf32 w_ = L.tw[bb * c];                                         \
                av[c] = c_rot<1>(av[c], w_.x, w_.y);                                  \
            }                                                                         \
        }
        LG2_INV_A(s0)
        if (a_task) {
#pragma unroll
            for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = av[c];
        }
        ctx.sync();
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
#pragma unroll 1
        for (int s = s0; s < s0 + SP; ++s) {
            if (s + 1 < s0 + SP) { LG2_INV_A(s + 1) }
            LG2_FENCE();
            fft192_pass_b<1>(u);
            unsigned ob = offB + (unsigned)s * (SW * 4u);
            LG2_OPAQUE(ob);
#pragma unroll
            for (int d = 0; d < 12; ++d) {
                const unsigned o = ob + (unsigned)d * (16u * R * 4u);
                lg2_st(dr, o, u[d].x);
                lg2_st(di, o, u[d].y);
            }
            if (s + 1 < s0 + SP) {
                ctx.sync();
                if (a_task) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) L.S[(12 * c + bb) * SW + col] = av[c];
                }
                ctx.sync();
#pragma unroll
                for (int i = 0; i < 12; ++i) u[i] = L.S[(12 * g + i) * SW + col];
            }
        }
#undef LG2_INV_A
    }
    // ---------------------------------------------------------------- losses
    ctx.sync();
    LG2S_TICK(6);
    lg2s_sum2(acc, acc_u, L.red);
    int* flag = (int*)L.rowpos;                          // the tables are no longer needed
    if (tid == 0) {
        if (part == 0) part_out[frame] = acc;            // every part has summed every kept row
        unk_out[frame * NP + part] = acc_u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned nlive = (unsigned)nframes * NP;
        const unsigned old = __hip_atomic_fetch_add(cnt + nframes, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = old + 1 == nlive;
        if (old + 1 == nlive) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ctx.sync();
    LG2S_TICK(7);
    if (flag[0]) {
        // the last workgroup: loss[b] = sum of the chain's frame sums + every share of the
        // masked-out rows; their total; and the counters back to zero for the next call
        const int Bn = nframes / T;
        float v = 0.f, zero = 0.f;
        for (int i = tid; i < nframes * NP; i += LG2_THREADS) v += unk_out[i];
        lg2s_sum2(v, zero, L.red);
        const float su = v;
        float mine = 0.f;                                // a chain per thread, frames in order
        for (int bi = tid; bi < Bn; bi += LG2_THREADS) {
            float sb = 0.f;
            for (int i = 0; i < T; ++i) sb += part_out[(size_t)bi * T + i];
            loss[bi] = sb + su;
            mine += sb + su;
        }
        lg2s_sum2(mine, zero, L.red);
        if (total && tid == 0) *total = mine;
        for (int i = tid; i < nframes + 1; i += LG2_THREADS) cnt[i] = 0u;
    }
    LG2S_TICK(8);
#undef LG2_RP
#undef LG2_IDS
}
