// Probe build (NOT part of libstep_amd.so): the PERSISTENT pipelined MRI step body (mri_lg2p.h) over a context
// whose tick() records wall-clock timestamps of workgroup 0 on its k-th frame (steady state, every other
// workgroup running the same code), for the phase breakdown in DESIGN.md.
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -shared \
//         -I step_amd/csrc scripts/probes/mri_timing_p.hip -o variants/mri_timing_p.so
#include <hip/hip_runtime.h>
#include "mri_lg2p.h"

__device__ unsigned long long g_mri_t[16];
__device__ int g_mri_frame;

struct TimingCtx {
    float* lds_;
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int block() const { return blockIdx.x; }
    __device__ __forceinline__ int nblocks() const { return gridDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ float* lds() const { return lds_; }
    __device__ __forceinline__ void tick(int i, int frame) const {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (frame == g_mri_frame) g_mri_t[i] = wall_clock64();
            if (i == 6 && frame == g_mri_frame - (int)gridDim.x) g_mri_t[0] = wall_clock64();   // end of the frame before
        }
    }
    __device__ __forceinline__ void raise_flag(int*) const {}
    __device__ __forceinline__ cf32 cos_sin_2pi_over_192(int e) const {
        float sn, cn;
        sincospif((float)e * (1.0f / 96.0f), &sn, &cn);
        return c_make(cn, sn);
    }
};

__global__ __launch_bounds__(LG2_THREADS) void k_probe(const float* x, const float* y, const int* rows,
                                                       const int* nrows, int T, int nframes, int nkc,
                                                       float* loss_part, float* unk_part, float* grad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    TimingCtx ctx{lds};
    lg2p_body(ctx, x, y, rows, nrows, 1, T, nframes, nkc, nullptr, loss_part, unk_part, grad);
}

extern "C" int mri_probe_step(const float* x, const float* y, int B, int T, const int* rows,
                              const int* nrows, int nkc, float* loss_part, float* grad, int k, void* stream) {
    const size_t lds = lg2_lds_bytes(nkc);
    (void)hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int cus = 256;
    const int grid = B * T < cus ? B * T : cus, fr = k * grid;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mri_frame), &fr, sizeof(int));
    hipLaunchKernelGGL(k_probe, dim3(grid), dim3(LG2_THREADS), lds, (hipStream_t)stream, x, y, rows,
                       nrows, T, B * T, nkc, loss_part, loss_part + (size_t)B * T, grad);
    return (int)hipGetLastError();
}

extern "C" int mri_probe_timing(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mri_t), sizeof(unsigned long long) * 16);
}
