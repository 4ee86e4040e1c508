// Shared helpers for the gfx950 kernels of libstep_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/step_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// thread-local error text behind step_last_error()
void step_set_error(const char* fmt, ...);

// body of mri_loss_grad (mri_fft.hip), shared with mri_loss_grad_static (mri_noalloc.hip)
int mri_loss_grad_impl(const char* who, const float* x, const float* y, int y_batch, int B, int T,
                       int R, const int32_t* rows, const int32_t* nrows, int mask_frames,
                       int max_rows, const float* chain_scale, int variant, float* loss_part,
                       float* loss, float* loss_total, float* grad, hipStream_t s);

#define STEP_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            step_set_error(__VA_ARGS__);        \
            return STEP_E_INVALID;              \
        }                                       \
    } while (0)

#define STEP_CHECK_LAUNCH(name)                                                    \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            step_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return STEP_E_LAUNCH;                                                  \
        }                                                                          \
    } while (0)

// A kernel that asks for more than 64 KiB of dynamic LDS needs its limit raised first
// (no allocation, no synchronisation; idempotent).
static inline void step_allow_lds(const void* fn, size_t bytes) {
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Row of the 32x32 f32 MFMA accumulator held in register `reg` of a lane in half `h`
// (C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31).
__device__ __forceinline__ int mfma32_row(int reg, int h) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * h;
}

// x' = x*scale + shift, then clip(min=clip_min) the way torch.clip does (NaN stays NaN;
// clip_min = -inf disables the clip).  unnormalize blackhole.py:720-722, clip :672.
__device__ __forceinline__ float prep_pixel(float v, float scale, float shift, float clip_min) {
    v = fmaf(v, scale, shift);
    return v < clip_min ? clip_min : v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over the block; every thread gets the result.  `red` = LDS float[blockDim/64 + 1].
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// nufft_grid.hip: loss[b] = sum of a chain's per_chain partial sums in fixed order (used by nufft_fft.hip too)
int nufft_loss_reduce(const float* loss_part, int B, int per_chain, float* loss, float* loss_total, void* stream);
// nufft_grid.hip: nufft_spread with the option of leaving the tiles outside the column ranges unwritten
int nufft_spread_impl(const float* vals_c64, int nframes, int T, const nufft_plan_t* plan, float* grid_c64,
                      int skip_outside, void* stream);
