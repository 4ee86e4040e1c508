// How much dynamic LDS may a workgroup ask for and still share a CU with a second one?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor for a trivial kernel, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ __launch_bounds__(384) void k_probe(float* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = lds[(threadIdx.x + 1) % 384];
}
int main() {
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {65536, 79872, 80896, 81408, 81664, 81920, 82176, 83968, 98304, 163840};
    for (int threads : {256, 384, 512})
        for (int s : sizes) {
            int nb = -1;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_probe, threads, s);
            printf("threads %d lds %d -> blocks/CU %d (%s)\n", threads, s, nb, hipGetErrorString(e));
        }
    return 0;
}
