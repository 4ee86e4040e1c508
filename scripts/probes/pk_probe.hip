// Issue rate of packed f32 VALU instructions on gfx950 against their scalar forms: N independent
// chains per wave, W waves per SIMD, wall clock over a long loop.  Reports lane-flop/cycle/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2{a + i + threadIdx.x, b - i};
    const f2 m = {a, b}, c = {b, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) {            // 2 x v_fma_f32
                    asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v[i].x), "+v"(v[i].y) : "v"(a), "v"(b));
                } else if (MODE == 1) {     // 1 x v_pk_fma_f32
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
                } else if (MODE == 2) {     // 2 x v_add_f32
                    asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(v[i].x), "+v"(v[i].y) : "v"(a));
                } else if (MODE == 3) {     // 1 x v_pk_add_f32
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
                } else if (MODE == 4) {     // 1 x v_pk_mul_f32
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
                } else if (MODE == 5) {     // v_pk_fma_f32 with op_sel swap + neg (complex rotate second half)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(v[i]) : "v"(m), "v"(c));
                } else if (MODE == 6) {     // v_add_f32 with a quad_perm DPP source
                    asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v[i].x), "+v"(v[i].y));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, int waves_per_simd) {
    const int iters = 4000, threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.9999f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) {
            // per wave: iters * 4 * 8 asm statements, each = 2 lane-results per lane
            const double stmts = (double)iters * 32;
            const double ns_per_stmt_per_simd = ms * 1e6 / (stmts * waves_per_simd);
            printf("%-34s waves/SIMD %d: %.3f ms, %.2f ns per statement (2 f32 results/lane) per SIMD -> %.2f cycles at 2.4 GHz\n",
                   name, waves_per_simd, ms, ns_per_stmt_per_simd, ns_per_stmt_per_simd * 2.4);
        }
    }
}
int main() {
    float* out; CHK(hipMalloc(&out, 256 * 1024 * 4));
    for (int w : {1, 2, 4}) {
        run<0>("2 x v_fma_f32", out, w);
        run<1>("1 x v_pk_fma_f32", out, w);
        run<2>("2 x v_add_f32", out, w);
        run<3>("1 x v_pk_add_f32", out, w);
        run<4>("1 x v_pk_mul_f32", out, w);
        run<5>("1 x v_pk_fma_f32 op_sel/neg", out, w);
        run<6>("2 x v_add_f32_dpp quad_perm", out, w);
    }
    return 0;
}
