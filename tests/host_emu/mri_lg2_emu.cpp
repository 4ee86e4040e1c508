// Host emulation of the MRI kernel body (step_amd/csrc/mri_lg2.h): the same C++ that the GPU runs,
// one pthread per GPU thread (512 per workgroup), __syncthreads = pthread_barrier, LDS = a
// NaN-filled heap block.  Built and driven by tests/test_mri_lg2_host.py.
#include <pthread.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include "mri_lg2.h"

struct HostCtx {
    int tid_, blk_, nblk_;
    pthread_barrier_t* bar_;
    float* lds_;
    int tid() const { return tid_; }
    int block() const { return blk_; }
    int nblocks() const { return nblk_; }
    void sync() const { pthread_barrier_wait(bar_); }
    float* lds() const { return lds_; }
    void tick(int, int) const {}
    void raise_flag(int* f) const { *f = 1; }
    cf32 cos_sin_2pi_over_192(int e) const {
        return c_make((float)cos(2 * M_PI * e / 192), (float)sin(2 * M_PI * e / 192));
    }
};

extern "C" int lg2_emu(const float* x, const float* y, int y_batch, const int* rows, const int* nrows,
                       int mask_frames, int T, int nframes, int nkc, const float* chain_scale,
                       float* loss_part, float* unk_part, float* grad, int nblocks) {
    for (int blk = 0; blk < nblocks; ++blk) {
        std::vector<float> lds(lg2_lds_bytes(nkc) / 4 + 16, NAN);
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, LG2_THREADS);
        std::vector<std::thread> th;
        for (int t = 0; t < LG2_THREADS; ++t)
            th.emplace_back([&, t]() {
                HostCtx ctx{t, blk, nblocks, &bar, lds.data()};
                if (y_batch > 1)
                    lg2_body<true>(ctx, x, y, y_batch, rows, nrows, mask_frames, T, nframes, nkc,
                                   chain_scale, loss_part, unk_part, grad);
                else
                    lg2_body<false>(ctx, x, y, y_batch, rows, nrows, mask_frames, T, nframes, nkc,
                                    chain_scale, loss_part, unk_part, grad);
            });
        for (auto& t : th) t.join();
        pthread_barrier_destroy(&bar);
    }
    return 0;
}


