// Which XCD should touch which addresses?  256 persistent workgroups (id % 8 = XCD); the 1 GB
// buffer is cut into chunks of 2^S bytes and XCD c takes the chunks q with q % 8 == (c + k) % 8
// (k = 0 .. 7), 32 workgroups sharing them.  Writes (dword stores, 256 B per wave instruction)
// or reads (dword loads, summed).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512) void k_xcd(float* buf, size_t bytes, int S, int k, int rd, float* sink) {
    const int id = blockIdx.x, c = id & 7, j = id >> 3, nj = gridDim.x >> 3;
    const size_t nchunk = bytes >> S, per = (size_t)1 << (S - 2);      // floats per chunk
    float acc = 0.f;
    for (size_t m = j; m * 8 + 7 < nchunk; m += nj) {
        const size_t q = m * 8 + ((c + k) & 7);
        float* p = buf + q * per;
        if (rd) {
#pragma unroll 8
            for (size_t e = threadIdx.x; e < per; e += 512) acc += __builtin_nontemporal_load(p + e);
        } else {
#pragma unroll 8
            for (size_t e = threadIdx.x; e < per; e += 512) p[e] = 1.f;
        }
    }
    if (rd && acc == 123.456f) *sink = acc;
}
extern "C" void xcd_probe(float* buf, size_t bytes, int S, int k, int rd, float* sink, void* stream) {
    hipLaunchKernelGGL(k_xcd, dim3(256), dim3(512), 0, (hipStream_t)stream, buf, bytes, S, k, rd, sink);
}
