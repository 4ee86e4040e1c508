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
// (f) the adjoint's pattern from NW waves per workgroup (one workgroup per CU): 16 waves = two per
// strip, alternate tiles; 4 waves = two strips per wave -- does the number of stores in flight
// per CU (63 per wave) bound the pattern?
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_store6(float* out, int B, int cpw) {
    const int t = blockIdx.y, b0 = blockIdx.x * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        for (int u = wave; u < 32; u += NW) {              // unit = (tile, strip)
            const int strip = u & 7, tile = u >> 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                f[y * 256 + strip * 32 + li] = (float)r;
                f[(255 - y) * 256 + strip * 32 + li] = (float)tile;
            }
        }
    }
}
// (g) the adjoint's pattern with the workgroups staggered: all workgroups walk their frames in
// lockstep, so at any moment they write the same offset of 256 frames (addresses equal modulo
// 256 KB).  STAG 1: tile order rotated by workgroup; STAG 2: also the row order inside a tile.
template <int STAG>
__global__ __launch_bounds__(512) void k_store7(float* out, int B, int cpw) {
    const int t = blockIdx.y, b0 = blockIdx.x * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int rot = (blockIdx.x + t) & 3, rrot = STAG == 2 ? ((t >> 2) & 15) : 0;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        for (int tl = 0; tl < 4; ++tl) {
            const int tile = (tl + rot) & 3;
#pragma unroll
            for (int r0 = 0; r0 < 16; ++r0) {
                const int r = (r0 + rrot) & 15;
                const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                f[y * 256 + wave * 32 + li] = (float)r;
                f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
            }
        }
    }
}
// (h) not persistent: one workgroup per (tile, frame, chain), dispatched in memory order, the
// adjoint's pattern inside the tile -- a moving write front like an elementwise kernel's
__global__ __launch_bounds__(512) void k_store9(float* out) {
    const int tile = blockIdx.x, t = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    float* f = out + ((size_t)b * 64 + t) * 65536;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
        f[y * 256 + wave * 32 + li] = (float)r;
        f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
    }
}
// (i) the same with one workgroup per frame (4 tiles)
__global__ __launch_bounds__(512) void k_store10(float* out) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    float* f = out + ((size_t)b * 64 + t) * 65536;
    for (int tile = 0; tile < 4; ++tile)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
            f[y * 256 + wave * 32 + li] = (float)r;
            f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
        }
}
// (j) persistent like (a), but workgroup (g, t) takes chains g, g + gx, g + 2 gx, ...: at any
// moment the workgroups write gx * 64 consecutive frames (one write front) instead of gx fronts
__global__ __launch_bounds__(512) void k_store12(float* out, int B) {
    const int t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        float* f = out + ((size_t)b * 64 + t) * 65536;
        for (int tile = 0; tile < 4; ++tile)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                f[y * 256 + wave * 32 + li] = (float)r;
                f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
            }
    }
}
// (k) persistent like (a) with the waves of a workgroup kept together: a barrier per frame
// (SYNC 1) or per tile (SYNC 2)
template <int SYNC>
__global__ __launch_bounds__(512) void k_store13(float* out, int B, int cpw) {
    const int t = blockIdx.y, b0 = blockIdx.x * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        if (SYNC == 1) __syncthreads();
        for (int tile = 0; tile < 4; ++tile) {
            if (SYNC == 2) __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                f[y * 256 + wave * 32 + li] = (float)r;
                f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
            }
        }
    }
}
// (l) persistent like (a) on a 1-D grid with t = id % 64, g = id / 64: workgroup ids go round
// the 8 XCDs, so XCD c now writes the frames with t % 8 == c, as in (i)
template <int MAP>
__global__ __launch_bounds__(512) void k_store15(float* out, int B, int cpw) {
    const int id = blockIdx.x;
    // MAP 0: XCD c writes t % 8 == c;  1: t / 8 == c;  2: t % 4 == c % 4 and chain group parity c / 4
    const int t = MAP == 0 ? (id & 63) : MAP == 1 ? (((id & 7) << 3) | ((id >> 3) & 7))
                                                  : ((id & 3) | ((id >> 3) & 7) << 2 | ((id >> 6) & 1) << 5);
    const int g = MAP == 2 ? (((id >> 2) & 1) | ((id >> 7) << 1)) : (id >> 6);
    const int b0 = g * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        for (int tile = 0; tile < 4; ++tile)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                f[y * 256 + wave * 32 + li] = (float)r;
                f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
            }
    }
}
// (m) XCD mapping of (l); ORDER 0: the frame written linearly, 2 KB (two rows) per round of the
// workgroup; ORDER 1: tile by tile -- 32 rows from the top linearly, then their 32 mirror rows
// linearly -- what the adjoint could store if its tiles went through LDS
template <int ORDER>
__global__ __launch_bounds__(512) void k_store18(float* out, int B, int cpw) {
    const int t = blockIdx.x & 63, b0 = (blockIdx.x >> 6) * cpw;
    const int tid = threadIdx.x;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        if (ORDER == 0) {
#pragma unroll 8
            for (int q = 0; q < 128; ++q) f[q * 512 + tid] = 1.f;
        } else {
            for (int tile = 0; tile < 4; ++tile) {
#pragma unroll
                for (int q = 0; q < 16; ++q) f[(tile * 32 + 2 * q) * 256 + tid] = 1.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) f[(224 - tile * 32 + 2 * q) * 256 + tid] = 2.f;
            }
        }
    }
}
// (n) XCD mapping of (l), the adjoint's pattern, with DELAY x 64 cycles of s_sleep between tiles:
// stands in for the matrix phase of a tile, during which the wave issues no store
template <int DELAY>
__global__ __launch_bounds__(512) void k_store20(float* out, int B, int cpw) {
    const int t = blockIdx.x & 63, b0 = (blockIdx.x >> 6) * cpw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    for (int it = 0; it < cpw && b0 + it < B; ++it) {
        float* f = out + ((size_t)(b0 + it) * 64 + t) * 65536;
        for (int tile = 0; tile < 4; ++tile) {
            if (DELAY) __builtin_amdgcn_s_sleep(DELAY);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int y = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                f[y * 256 + wave * 32 + li] = (float)r;
                f[(255 - y) * 256 + wave * 32 + li] = (float)tile;
            }
        }
    }
}
extern "C" void store_probe(int mode, float* out, int B, void* stream) {
    if (mode >= 20 && mode <= 23) {
        const int cpw20 = (B + 3) / 4;
        dim3 g20(((B + cpw20 - 1) / cpw20) * 64);
        if (mode == 20) hipLaunchKernelGGL(k_store20<0>, g20, dim3(512), 0, (hipStream_t)stream, out, B, cpw20);
        else if (mode == 21) hipLaunchKernelGGL(k_store20<12>, g20, dim3(512), 0, (hipStream_t)stream, out, B, cpw20);
        else if (mode == 22) hipLaunchKernelGGL(k_store20<24>, g20, dim3(512), 0, (hipStream_t)stream, out, B, cpw20);
        else hipLaunchKernelGGL(k_store20<48>, g20, dim3(512), 0, (hipStream_t)stream, out, B, cpw20);
        return;
    }
    if (mode == 18 || mode == 19) {
        const int cpw18 = (B + 3) / 4;
        if (mode == 18) hipLaunchKernelGGL(k_store18<0>, dim3(((B + cpw18 - 1) / cpw18) * 64), dim3(512), 0, (hipStream_t)stream, out, B, cpw18);
        else hipLaunchKernelGGL(k_store18<1>, dim3(((B + cpw18 - 1) / cpw18) * 64), dim3(512), 0, (hipStream_t)stream, out, B, cpw18);
        return;
    }
    if (mode == 15) {
        const int cpw15 = (B + 3) / 4;
        hipLaunchKernelGGL(k_store15<0>, dim3(((B + cpw15 - 1) / cpw15) * 64), dim3(512), 0, (hipStream_t)stream, out, B, cpw15);
        return;
    }
    if (mode == 16 || mode == 17) {
        const int cpw15 = (B + 3) / 4;
        if (mode == 16) hipLaunchKernelGGL(k_store15<1>, dim3(((B + cpw15 - 1) / cpw15) * 64), dim3(512), 0, (hipStream_t)stream, out, B, cpw15);
        else hipLaunchKernelGGL(k_store15<2>, dim3(((B + cpw15 - 1) / cpw15) * 64), dim3(512), 0, (hipStream_t)stream, out, B, cpw15);
        return;
    }
    if (mode == 13 || mode == 14) {
        const int cpw13 = (B + 3) / 4;
        dim3 g13((B + cpw13 - 1) / cpw13, 64);
        if (mode == 13) hipLaunchKernelGGL(k_store13<1>, g13, dim3(512), 0, (hipStream_t)stream, out, B, cpw13);
        else hipLaunchKernelGGL(k_store13<2>, g13, dim3(512), 0, (hipStream_t)stream, out, B, cpw13);
        return;
    }
    if (mode == 12) { hipLaunchKernelGGL(k_store12, dim3(4, 64), dim3(512), 0, (hipStream_t)stream, out, B); return; }
    if (mode == 9) { hipLaunchKernelGGL(k_store9, dim3(4, 64, B), dim3(512), 0, (hipStream_t)stream, out); return; }
    if (mode == 10) { hipLaunchKernelGGL(k_store10, dim3(64, B), dim3(512), 0, (hipStream_t)stream, out); return; }
    if (mode == 11) {          // (i) at one workgroup per CU (100 KB of dynamic LDS)
        (void)hipFuncSetAttribute((const void*)k_store10, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipLaunchKernelGGL(k_store10, dim3(64, B), dim3(512), 100 * 1024, (hipStream_t)stream, out);
        return;
    }
    if (mode == 7 || mode == 8) {
        const int cpw7 = (B + 3) / 4;
        dim3 g7((B + cpw7 - 1) / cpw7, 64);
        if (mode == 7) hipLaunchKernelGGL(k_store7<1>, g7, dim3(512), 0, (hipStream_t)stream, out, B, cpw7);
        else hipLaunchKernelGGL(k_store7<2>, g7, dim3(512), 0, (hipStream_t)stream, out, B, cpw7);
        return;
    }
    if (mode == 5 || mode == 6) {
        const int cpw6 = (B + 3) / 4;
        dim3 g6((B + cpw6 - 1) / cpw6, 64);
        if (mode == 5) hipLaunchKernelGGL(k_store6<16>, g6, dim3(1024), 0, (hipStream_t)stream, out, B, cpw6);
        else hipLaunchKernelGGL(k_store6<4>, g6, dim3(256), 0, (hipStream_t)stream, out, B, cpw6);
        return;
    }
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
