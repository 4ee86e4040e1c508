// Store-pattern probe: how fast can 1 GB be written with (a) the adjoint's pattern (each wave owns a
// 32-column strip of a 256-wide fp32 image; one instruction = two 128-byte row segments),
// (b) whole rows as float4 (one instruction = one 1-KB row), from persistent workgroups.
#include <hip/hip_runtime.h>
// grid (gx, 64): workgroup (g, t) writes frames (b*64 + t) for b in [g*cpw, (g+1)*cpw)
template <int MODE>
__global__ __launch_bounds__(512) void k_store(float* out, int B, int cpw) {
    const int t = blockIdx.y, b0 = blockIdx.x * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        if (MODE == 0) {           // adjoint pattern: tiles of 32 rows, top + mirrored
            for (int tile = 0; tile < 4; ++tile)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                    f[y * 256 + wave * 32 + li] = (float)r;
                    f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
                }
        } else if (MODE == 1) {    // whole rows, float4: wave w writes rows w, w+8, ...
            const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll 8
            for (int y = wave; y < 256; y += 8) ((float4*)(f + y * 256))[lane] = v;
        } else {                   // whole rows, dword: one instruction = 256 contiguous bytes
#pragma unroll 8
            for (int q = wave; q < 1024; q += 8) f[q * 64 + lane] = 1.f;
        }
    }
}
// (d) 4 waves per workgroup, each owning 64 columns: the adjoint's row order (32-row tiles, top +
// mirrored) with one instruction = one row segment of 256 contiguous bytes -- what the adjoint
// would store after a v_permlane32_swap between the accumulators of two adjacent strips
__global__ __launch_bounds__(256) void k_store4(float* out, int B, int cpw) {
    const int t = blockIdx.y, b0 = blockIdx.x * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        for (int tile = 0; tile < 4; ++tile)
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int y = tile * 32 + r;
                f[y * 256 + wave * 64 + lane] = (float)r;
                f[(255 - y) * 256 + wave * 64 + lane] = (float)tile;
            }
    }
}
// (e) the adjoint's pattern with the work dealt the other way round: workgroup (q, t) writes tile q
// (32 rows from the top + their mirror rows) of frame t of EVERY chain, so that all workgroups
// sweep the same chain at the same time (one write front, like an elementwise kernel)
__global__ __launch_bounds__(512) void k_store5(float* out, int B) {
    const int t = blockIdx.y, tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    for (int b = 0; b < B; ++b) {
        float* f = out + ((size_t)b * 64 + t) * 65536;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
            f[y * 256 + wave * 32 + li] = (float)r;
            f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
        }
    }
}
extern "C" void store_probe(int mode, float* out, int B, void* stream) {
    if (mode == 4) {
        hipLaunchKernelGGL(k_store5, dim3(4, 64), dim3(512), 0, (hipStream_t)stream, out, B);
        return;
    }
    if (mode == 3) {
        const int cpw4 = (B + 3) / 4;
        hipLaunchKernelGGL(k_store4, dim3((B + cpw4 - 1) / cpw4, 64), dim3(256), 0, (hipStream_t)stream, out, B, cpw4);
        return;
    }
    const int cpw = (B + 3) / 4;
    dim3 g((B + cpw - 1) / cpw, 64);
    if (mode == 0) hipLaunchKernelGGL(k_store<0>, g, dim3(512), 0, (hipStream_t)stream, out, B, cpw);
    else if (mode == 1) hipLaunchKernelGGL(k_store<1>, g, dim3(512), 0, (hipStream_t)stream, out, B, cpw);
    else hipLaunchKernelGGL(k_store<2>, g, dim3(512), 0, (hipStream_t)stream, out, B, cpw);
}
