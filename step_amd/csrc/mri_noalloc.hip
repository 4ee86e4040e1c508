// mri_loss_grad with its workspace in a file-scope device array: the entry point that allocates
// nothing (include/step_amd.h: mri_loss_grad_static).  The kernels are those of mri_fft.hip.
#include "common.h"
#include <mutex>

// ------------------------------------------------------------ the entry point that allocates nothing
// mri_loss_grad with its workspace (the per-frame partial losses and the shares of the masked-out
// rows, 2*B*T floats) in a file-scope device array instead of a caller-provided buffer, for
// callers that cannot allocate (HIP graph capture, serving loops).  Each GPU holds its own copy
// of the array at its own address: the address is looked up once per device, with that device
// current, and only that pointer -- never the array's name, which in host code is a stand-in and
// not a device address -- is handed to the kernels.
static __device__ __align__(256) float scratch[2 * STEP_MRI_STATIC_MAX_FRAMES];

#define STEP_MAX_DEVICES 64
static float* g_scratch_ptr[STEP_MAX_DEVICES];            // per device, filled on first use
static std::mutex g_scratch_mu;

static int mri_scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        step_set_error("mri_loss_grad_static: hipGetDevice: %s", hipGetErrorString(e));
        return STEP_E_LAUNCH;
    }
    STEP_REQUIRE(dev >= 0 && dev < STEP_MAX_DEVICES, "mri_loss_grad_static: device %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            step_set_error("mri_loss_grad_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return STEP_E_LAUNCH;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return STEP_OK;
}

extern "C" int mri_loss_grad_static(const float* x, const float* y, int y_batch, int B, int T, int R,
                                    const int32_t* rows, const int32_t* nrows, int mask_frames,
                                    int max_rows, const float* chain_scale, float* loss,
                                    float* loss_total, float* grad, void* stream) {
    STEP_REQUIRE(B > 0 && T > 0 && (long long)B * T <= STEP_MRI_STATIC_MAX_FRAMES,
                 "mri_loss_grad_static: B*T = %lld frames exceeds the static workspace (%d frames); "
                 "use mri_loss_grad with a caller workspace", (long long)B * T, STEP_MRI_STATIC_MAX_FRAMES);
    float* ws = nullptr;
    if (int rc = mri_scratch_for_current_device(&ws)) return rc;
    // every workspace entry the finish kernel reads is written by this call's first kernel, so
    // nothing has to be cleared between calls
    return mri_loss_grad_impl("mri_loss_grad_static", x, y, y_batch, B, T, R, rows, nrows,
                              mask_frames, max_rows, chain_scale, 0, ws, loss, loss_total, grad,
                              (hipStream_t)stream);
}
