/* step_amd.h -- C ABI of libstep_amd.so: the MI355X (gfx950) kernels behind STeP's
 * DAPS measurement-consistency path.
 *
 * The reference (zhangbingliang2019/STeP) is pure Python/PyTorch and has no FFI of
 * its own; each entry point below names the reference code (file:line under the
 * reference root) whose torch-op sequence it replaces.  The Python host layer
 * (step_amd/forward_operator, step_amd/sampler) mirrors the reference's
 * operator/sampler interface on top of exactly these symbols via ctypes.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer (tensor.data_ptr()) unless the
 *     parameter comment says "host"; the caller owns every buffer;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every launch is asynchronous on it; no entry point allocates, frees or synchronises;
 *   - return value: 0 = ok, negative = error (STEP_E_*); step_last_error() returns a
 *     thread-local message; no C++ exception crosses this boundary;
 *   - float data is IEEE fp32; complex data is interleaved (re, im) fp32 = torch.complex64;
 *   - video tensors are contiguous [B][T][C][H][W].
 */
#ifndef STEP_AMD_H
#define STEP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEP_OK            0
#define STEP_E_INVALID    -1   /* bad argument / unsupported shape (nothing launched) */
#define STEP_E_LAUNCH     -2   /* hip launch error (hipGetLastError text in step_last_error) */

const char* step_last_error(void);
int         step_abi_version(void);   /* 12 since mri_*_any; 11 since mri_step_update; 10 since mri_loss_grad_mc_fused; 9 since bh_update_t carries `eps`; 8 since mri_loss_grad_split; 7 since nufft_plan_t carries the lane-per-chain spread lists; 6 since it carries the owner bins; 5 since bh_update_t carries the Langevin step and mcmc_update_philox takes `quad_w`; 4 since bh_loss_grad_sep3 takes `update` (3: mri_loss_grad `variant`, bh_weights_t `cphase_form`) */

/* ------------------------------------------------------------------ host-side tables */

/* Closure index construction, reference forward_operator/blackhole.py:235-264.
 * Visibility rows i = 0..n_rows-1 carry (row_time[i], row_t1[i], row_t2[i]) (station ids);
 * a later row with the same key overwrites an earlier one (dict semantics).  Triangle
 * (a,b,c) -> rows of (a,b),(b,c),(c,a); quadrangle (a,b,c,d) -> rows of (a,b),(c,d),(a,d),(b,c);
 * conj = 1 where the row is stored with the stations swapped.  All pointers HOST.
 * Outputs int64 [n_tri][3] x2 and [n_quad][4] x2 (the reference's torch.long tensors).
 * A key with no row -> STEP_E_INVALID (the reference raises KeyError). */
int bh_build_closure_tables(const double* row_time, const int32_t* row_t1, const int32_t* row_t2,
                            int64_t n_rows,
                            const double* tri_time, const int32_t* tri_sites, int64_t n_tri,
                            const double* quad_time, const int32_t* quad_sites, int64_t n_quad,
                            int64_t* cp_index, int64_t* cp_conj,
                            int64_t* camp_index, int64_t* camp_conj);

/* Transposed (baseline -> incident closures) tables used by the adjoint gather so that
 * the scatter of Appendix-B items 2-3 is a deterministic gather.  All pointers HOST.
 * ptr arrays have n_vis+1 entries; ent arrays 3*n_tri / 4*n_quad entries, each
 * (k << 3) | (leg << 1) | conj.  Index out of [0,n_vis) -> STEP_E_INVALID. */
int bh_build_incidence(const int64_t* cp_index, const int64_t* cp_conj, int64_t n_tri,
                       const int64_t* camp_index, const int64_t* camp_conj, int64_t n_quad,
                       int64_t n_vis,
                       int32_t* cp_ptr, int32_t* cp_ent, int32_t* ca_ptr, int32_t* ca_ent);

/* k-space row mask, reference forward_operator/dynamic_mri.py:9-39.
 * mode 0 = same_mask (mask_frames = 1), 1 = split_mask, 2 = random_split_mask
 * (np.random.seed(42) + np.random.shuffle restated: MT19937 + masked rejection sampling).
 * mask: HOST float[mask_frames][resolution] of 0/1. */
int mri_build_mask(int acc_factor, int mode, int resolution, int frames, int num_acs_lines,
                   float* mask);

/* ------------------------------------------------------------------ black-hole DFT, dense A */

/* Rows per frame of the packed real matrix: 2F+1 (Re rows, Im rows, flux row of ones)
 * rounded up to a multiple of 32. */
int    bh_dense_rows_padded(int F);
size_t bh_dense_packed_floats(int T, int F, int HW);           /* per packed copy */
size_t bh_dense_fwd_workspace_floats(int T, int F, int HW, int B);

/* Repack A_vis (complex64 [T][F][HW], the reference attribute blackhole.py:155) once at
 * operator construction into the two fp32 images the kernels stream:
 * a_frag (MFMA-fragment order, forward) and a_rows (row-planar [T][RP][HW], adjoint). */
int bh_dense_pack(const void* A_vis_c64, int T, int F, int HW,
                  float* a_frag, float* a_rows, void* stream);

/* vis[b][t*F+f] = sum_n A[t][f][n] * x'[b][t][n], flux[b][t] = sum_n x'[b][t][n] with
 * x' = max(x*scale+shift, clip_min) (clip_min = -inf -> no clip).
 * Replaces forward_vis blackhole.py:315-321 (+ unnormalize :720-722, forward_flux :328-329,
 * the clip of evaluate_chisq :672).  HW % 8 == 0. */
int bh_dft_fwd_dense(const float* a_frag, int T, int F, int HW,
                     const float* x, int B, float scale, float shift, float clip_min,
                     float* vis_c64, float* flux, float* workspace, void* stream);

/* grad[b][t][n] = out_scale * chain_scale[b] * ( sum_f Re(conj(A[t][f][n]) gvis[b][t*F+f])
 *                                               + gflux[b][t] ).
 * The autograd backward of forward_vis/forward_flux.
 * chain_scale may be NULL (=1).  HW % 32 == 0. */
int bh_dft_adj_dense(const float* a_rows, int T, int F, int HW,
                     const float* gvis_c64, const float* gflux, const float* chain_scale,
                     int B, float out_scale, float* grad, void* stream);

/* ------------------------------------------------------------------ black-hole DFT, separable
 * A[t][f][y*W+x] = ey[t][f][y] * ex[t][f][x] (what ehtim's ftmatrix produces: pulse * e^{2 pi i y v}
 * * e^{2 pi i x u}); never materialises A.  Tables (fp32, built by the host layer):
 *   ey_t  [T][H][64]     LDS image for the forward: col f = Re ey, col 32+f = Im ey, f = F is the
 *                        flux row (1 + 0i), other cols 0
 *   ey_ri [T][2][32][H]  planar Re / Im of ey (same row convention)
 *   ex_ri [T][2][32][W]  planar Re / Im of ex (flux row = 1 + 0i)
 * Requires F <= 31, H % 32 == 0, W % 32 == 0, H <= 512. */
int bh_dft_fwd_sep(const float* ey_t, const float* ex_ri, int T, int F, int H, int W,
                   const float* x, int B, float scale, float shift, float clip_min,
                   float* vis_c64, float* flux, void* stream);
int bh_dft_adj_sep(const float* ey_ri, const float* ex_ri, int T, int F, int H, int W,
                   const float* gvis_c64, const float* gflux, const float* chain_scale,
                   int B, float out_scale, float* grad, void* stream);

/* The same two contractions (same tables, same arguments, same reference lines) on the bf16
 * matrix cores: every fp32 operand is split exactly into three bf16 planes in-kernel and the six
 * leading plane products are accumulated in fp32 (error <= one fp32 rounding per product), and
 * the image is folded about its horizontal mid-line, which needs ey[f][H-1-y] == conj(ey[f][y])
 * bit for bit (true for ehtim's centred pixel grid; bh_sep3_tables_ok checks a host copy of
 * ey_ri -- the kernels cannot, and give wrong results on other tables).  HBM-bound instead of
 * fp32-MFMA-bound.  Built for the square sizes bh_sep3_supported() accepts (256, 128, 64);
 * other shapes return STEP_E_INVALID -- use the fp32 entry points above for them.  Inputs with
 * Inf pixels give NaN (Inf - Inf in the split). */
int bh_sep3_supported(int H, int W);
int bh_sep3_tables_ok(const float* ey_ri_host, int T, int H);
int bh_dft_fwd_sep3(const float* ey_t, const float* ex_ri, int T, int F, int H, int W,
                    const float* x, int B, float scale, float shift, float clip_min,
                    float* vis_c64, float* flux, void* stream);
/* sum_out (may be NULL) = sum(sum_in[0 .. sum_n)) in a fixed order, computed by one wave of the
 * same launch: the scalar loss of gradient(x, y, return_loss=True) (base.py:43-49, daps.py:100)
 * without a reduction launch of its own.  sum_in must be complete before this launch starts
 * (same stream as its producer). */
int bh_dft_adj_sep3(const float* ey_ri, const float* ex_ri, int T, int F, int H, int W,
                    const float* gvis_c64, const float* gflux, const float* chain_scale,
                    int B, float out_scale, float* grad, const float* sum_in, int sum_n,
                    float* sum_out, void* stream);

/* ------------------------------------------------------------------ closure quantities */

typedef struct {
    int32_t n_vis;            /* T*F */
    int32_t n_tri, n_quad;    /* K_cp, K_ca */
    int32_t T;                /* frames (flux length) */
    const int32_t* cp_index;  /* [n_tri][3]  */
    const int32_t* cp_conj;   /* [n_tri][3]  0/1 */
    const int32_t* camp_index;/* [n_quad][4] */
    const int32_t* cp_ptr;    /* incidence, see bh_build_incidence */
    const int32_t* cp_ent;
    const int32_t* ca_ptr;
    const int32_t* ca_ent;
} bh_tables_t;

typedef struct {               /* a measurement (BlackHoleObservation, blackhole.py:23-132) */
    const float* vis_c64;      /* [By][n_vis] complex   (may be NULL when w_vis == 0)  */
    const float* sigma_vis;    /* [By][n_vis]                                         */
    const float* amp;          /* [By][n_vis]           (may be NULL when w_amp == 0)  */
    const float* sigma_amp;
    const float* cphase;       /* [By][n_tri]  radians                                */
    const float* sigma_cphase;
    const float* logcamp;      /* [By][n_quad]                                        */
    const float* sigma_logcamp;
    const float* flux;         /* [By][T]                                             */
    int32_t batch;             /* By: 1 (broadcast over chains) or B                  */
} bh_meas_t;

typedef struct {               /* blackhole.py:175-180 (weights already multiplied by their dims) */
    float w_vis, w_amp, w_cp, w_camp, w_flux;
    int32_t normalize;         /* loss_normalize */
    int32_t cphase_form;       /* how 1 - cos(d) of chi2_cphase (blackhole.py:472-474) is evaluated:
                                * 0 = 2 sin^2(d/2): accurate to fp32 rounding for every d, what the
                                *     fp64 evaluation of the reference formula converges to;
                                * 1 = 1 - cosf(d): the reference's literal fp32 form, which cancels
                                *     once a chain fits the data (d ~ 1e-3 keeps ~3 bits) */
} bh_weights_t;

/* Per-chain loss and its gradient w.r.t. vis / flux, from vis and flux of x01.
 * Replaces chi2_* blackhole.py:448-491, normalize_chisq :665-669, the weighted sum of
 * loss :736-771 and the autograd backward of all of it (Appendix B items 2-5).
 *   loss [B]; chi2 [B][5] raw (vis, amp, cphase, logcamp, flux; 0 where weight == 0);
 *   gvis [B][n_vis] complex = dL/dRe + i dL/dIm; gflux [B][T].  gvis/gflux may be NULL
 *   (forward only, e.g. Metropolis-Hastings sample_MH daps.py:157-158). */
int bh_closure_loss(const float* vis_c64, const float* flux, int B,
                    const bh_tables_t* tab, const bh_meas_t* y, const bh_weights_t* w,
                    float* loss, float* chi2, float* gvis_c64, float* gflux, void* stream);

/* chi^2 of every data product for the evaluator (evaluate_chisq blackhole.py:671-699):
 * vis_scale[b][t] (may be NULL) multiplies frame t's visibilities first (the per-frame flux
 * alignment y_flux/x_flux is linear in the DFT).  chi2 [B][4] = vis, amp, cphase, logcamp, raw.
 * cphase_form as in bh_weights_t. */
int bh_closure_chisq(const float* vis_c64, const float* vis_scale, int B, int F,
                     const bh_tables_t* tab, const bh_meas_t* y, int cphase_form, float* chi2,
                     void* stream);

/* Full observation from vis (forward_from_vis blackhole.py:393-401): amp, sigma_amp
 * (= sigma broadcast), cphase, sigma_cphase, logcamp, sigma_logcamp.  sigma [n_vis]. */
int bh_closure_observe(const float* vis_c64, const float* sigma, int B, const bh_tables_t* tab,
                       float* amp, float* sigma_amp, float* cphase, float* sigma_cphase,
                       float* logcamp, float* sigma_logcamp, void* stream);

/* ------------------------------------------------------------------ fused black-hole step
 * One Langevin data step of the black-hole operator -- operator.gradient(x, y, return_loss=True),
 * daps.py:100 -> blackhole.py:736-771 + base.py:43-49 -- in TWO launches instead of three: the
 * closure stage rides in the DFT kernels (forward epilogue: per-frame closure terms and chi^2
 * partial sums; adjoint prologue: chi^2 -> normalize_chisq -> g_vis by the incidence gather;
 * bh_dft_sep3.hip).  Same arithmetic as bh_dft_fwd_sep3 -> bh_closure_loss -> bh_dft_adj_sep3
 * except for the order in which the chi^2 terms are summed (per frame, then over frames).
 * Preconditions (else STEP_E_INVALID; callers fall back to the three calls):
 *   - the bh_dft_*_sep3 preconditions (bh_sep3_supported, bh_sep3_tables_ok);
 *   - every triangle / quadrangle involves baselines of ONE frame and a frame has at most 64 of
 *     each (bh_build_frame_closures succeeds) -- true for tables built by process_obs, whose
 *     baseline keys include the time (blackhole.py:237-241);
 *   - one measurement for all chains (y->batch == 1) and w_vis == w_amp == 0 (the shipped config). */

/* frame_tab: HOST int32 [T][2][64][2]: for frame t, kind (0 triangles, 1 quadrangles), slot j:
 * (closure id k or -1, packed local baseline indices 5 bits each + conjugate flags from bit 15).
 * This builder runs on the host and writes host memory; bh_loss_grad_sep3 below reads the table
 * from inside a kernel and needs a DEVICE copy of it (hipMemcpy the T*2*64*2 int32 once). */
int bh_build_frame_closures(const int64_t* cp_index, const int64_t* cp_conj, int64_t n_tri,
                            const int64_t* camp_index, int64_t n_quad, int T, int F,
                            int32_t* frame_tab);

typedef struct {               /* device workspace of the fused step (caller-owned, any contents) */
    float* vis_c64;            /* [B][T*F] complex: the visibilities of x (an output as well) */
    float* flux;               /* [B][T] */
    float* dcp;                /* [B][n_tri] */
    float* dca;                /* [B][n_quad] */
    float* S;                  /* [B][T][4], 16-byte aligned */
} bh_step_ws_t;

/* Constants of the state update (mcmc_update below; here because bh_update_t carries them). */
typedef struct {
    float inv_tau2;      /* 1/tau^2            daps.py:101 */
    float inv_sigma2;    /* 1/sigma^2          daps.py:102 */
    float lr;            /* get_lr(ratio)      daps.py:233-239 */
    float sqrt_lr;       /* np.sqrt(lr)        daps.py:139 */
    float sqrt_2lr;      /* np.sqrt(2 lr)      daps.py:137 */
    float momentum;      /* daps.py:140 */
    float noise_scale;   /* np.sqrt(2 (1 - momentum)) */
    int32_t algo;        /* 0 = langevin, 1 = hmc, 2 = gradient descent of optimize() */
} mcmc_params_t;

typedef struct {               /* the state update fused into the adjoint: no gradient in HBM */
    float* x_inout;            /* the state: must be the `x` the step was evaluated at; updated in place */
    const float* x0hat;
    float lr, inv_sigma2;      /* mcmc == NULL: MCMCSampler.optimize (daps.py:210-223),
                                * x -= lr * (g + (x - x0hat) * inv_sigma2), g = d loss / d x */
    /* mcmc != NULL (host pointer, read during the call; lr / inv_sigma2 above are then unused):
     * the step of MCMCSampler.sample (daps.py:93-104, :135-144) for algo 0 (langevin) or algo 1
     * with momentum 0 (no velocity buffer: its old value has weight zero), with the score
     *     -g * inv_tau2 + (x0hat - x) * inv_sigma2
     * -- what the 'gaussian' prior solver's (xt - x)/sigma^2 + (x0hat - xt)/sigma^2 adds up to --
     * and the noise drawn in the kernel: Philox4x32-10 keyed by seeds[b] (DEVICE int64 [B]),
     * counter (column quad, step) = the numbers mcmc_update_philox draws with quad_w = W.  The
     * result equals mcmc_update_philox(x, NULL, g, xt = x0hat, prior = NULL, ..., quad_w = W).
     * step_dev (may be NULL): device int64 read instead of `step`; nan_flag (may be NULL): set
     * to 1 when an updated element is NaN. */
    const mcmc_params_t* mcmc;
    const int64_t* seeds;
    int64_t step;
    const int64_t* step_dev;
    int32_t* nan_flag;
    /* eps (may be NULL; mcmc != NULL only): DEVICE noise tensor laid out like x.  When given, the
     * step uses it instead of drawing (seeds / step / step_dev unused): the reference's own draw,
     * torch.randn_like at daps.py:192, replayed -- the result equals
     * mcmc_update(x, NULL, g, xt = x0hat, prior = NULL, eps, ...).  Read once, never written. */
    const float* eps;
} bh_update_t;

/* frame_tab: DEVICE pointer to a copy of the host table bh_build_frame_closures fills (like every
 * pointer of this call except the structs themselves, `w` and update->mcmc); a pageable host
 * pointer is refused with STEP_E_INVALID instead of faulting in the kernel.
 * loss [B], chi2 [B][5] (may be NULL), loss_total (may be NULL) = sum of loss[],
 * grad [B][T][H*W] = out_scale * chain_scale[b] * d loss[b] / d x' with x' = 0.5 x + 0.5
 * (out_scale = 0.5 gives d loss / d x; chain_scale may be NULL).
 * Exactly one of `grad` and `update` is non-NULL: with `update` the gradient is not written
 * anywhere -- the adjoint applies the state update to x where it would have stored it. */
int bh_loss_grad_sep3(const float* ey_t, const float* ey_ri, const float* ex_ri, int T, int F,
                      int H, int W, const float* x, int B, const bh_tables_t* tab,
                      const int32_t* frame_tab, const bh_meas_t* y, const bh_weights_t* w,
                      const bh_step_ws_t* ws, const float* chain_scale, float out_scale,
                      float* loss, float* chi2, float* loss_total, float* grad,
                      const bh_update_t* update, void* stream);

/* The three-kernel path's adjoint with the same fused update (see bh_update_t): replaces
 * bh_dft_adj_sep3 + mcmc_update(algo 2) or + mcmc_update_philox. */
int bh_dft_adj_sep3_update(const float* ey_ri, const float* ex_ri, int T, int F, int H, int W,
                           const float* gvis_c64, const float* gflux, const float* chain_scale,
                           int B, float out_scale, const bh_update_t* update, const float* sum_in,
                           int sum_n, float* sum_out, void* stream);

/* ------------------------------------------------------------------ dynamic MRI
 * x, y, out: [B][T][2][R][R] fp32 (channel 0 = Re, 1 = Im); R == 192.
 * rows: int32 [mask_frames][R] = kept k_y rows of each mask frame, ascending, padded with -1;
 * nrows: int32 [mask_frames]; mask_frames = 1 (same_mask) or T; max_rows = max(nrows) (host
 * knowledge; sizes the LDS row buffer, masks with more than 64 kept rows run in groups). */

/* out = M * FFT2(fftshift(x_re + i x_im)), unnormalised (dynamic_mri.py:42-48). */
int mri_forward(const float* x, int B, int T, int R, const int32_t* rows, const int32_t* nrows,
                int mask_frames, int max_rows, float* out, void* stream);

/* out = A^H g = ifftshift(R^2 IFFT2(M * g)): the vector-Jacobian product of mri_forward, i.e.
 * what torch autograd computes for dynamic_mri.py:42-48 when a caller differentiates A(x). */
int mri_adjoint(const float* g, int B, int T, int R, const int32_t* rows, const int32_t* nrows,
                int mask_frames, int max_rows, float* out, void* stream);

/* loss[b] = sum |A(x) - y|^2 over every k-space entry (base.py:40-41);
 * grad = chain_scale[b] * d loss[b] / d x  (autograd of the above; Appendix B "MRI").
 * grad may be NULL (loss only).  chain_scale may be NULL (=1).
 * loss_part: workspace float[2*B*T] (per-frame partial losses, then shares of the masked-out
 * rows' sum |y|^2 of a batch-1 measurement); loss [B]; loss_total (may be NULL): sum of loss[].
 * variant: 0 = choose by mask and measurement (when every mask frame keeps <= 64 rows, i.e. every
 * shipped mode: the persistent pipelined kernel for y_batch = 1, the pipelined kernel otherwise),
 * 1 = first-generation kernel (any mask; cross-checks),
 * 2 = pipelined kernel, one workgroup per frame (STEP_E_INVALID if a mask frame keeps more than 64 rows),
 * 4 = its persistent form (a workgroup walks several frames and requests the next frame's first
 * strips under the current frame's row and inverse phases; y_batch = 1 only; bit-identical to 2).
 * (3 was a kernel with in-register column transforms and two workgroups per CU; it measured slower
 * at every batch and left the library in round 3 -- scripts/probes/mri_lg4.h keeps it.) */
int mri_loss_grad(const float* x, const float* y, int y_batch, int B, int T, int R,
                  const int32_t* rows, const int32_t* nrows, int mask_frames, int max_rows,
                  const float* chain_scale, int variant, float* loss_part, float* loss,
                  float* loss_total, float* grad, void* stream);

/* mri_loss_grad for batches that leave most of the chip idle (the reference samples one chain at a
 * time, sample.py:64-65: 12 frames; BASELINE config 5 on 8 GPUs is 4 chains = 48 frames per GPU):
 * 6 or 3 workgroups share a frame -- each takes column strips in the two column phases; the kept
 * rows go from the forward column phase to the row phase through `ws` with a per-frame arrive /
 * wait counter -- and the last workgroup adds the losses up (one launch, no finish kernel).  Same reference lines as mri_loss_grad
 * (dynamic_mri.py:42-48, base.py:40-49); one measurement for all chains (y: [T][2][R][R]); masks of
 * <= 64 kept rows per frame.  The gradient is bit-identical to mri_loss_grad's, the loss equal up to
 * its summation order.
 * mri_loss_grad_split_ws_bytes: size of `ws` for (B, T, max_rows, parts), or 0 when the split does not
 * apply (the frames x 3 workgroups would not all be resident at once, i.e. more than 6 chains of 12
 * frames on 256 CUs: use mri_loss_grad).
 * ws: device memory, 16-byte aligned, ZEROED ONCE by the caller before its first use (its head is
 * the counters); every call leaves it zeroed for the next one, whatever its shapes.  Calls that
 * share a workspace must not overlap.  parts: 0 = 6 or 3, whichever fits; or 6, 3, 2 (2: measured no faster
 * than the unsplit kernel). */
size_t mri_loss_grad_split_ws_bytes(int B, int T, int max_rows, int parts);
int mri_loss_grad_split(const float* x, const float* y, int B, int T, int R, const int32_t* rows,
                        const int32_t* nrows, int mask_frames, int max_rows,
                        const float* chain_scale, int parts, void* ws, size_t ws_bytes, float* loss,
                        float* loss_total, float* grad, void* stream);

/* One step of the MRI data-consistency loop with the STATE UPDATE inside the kernel: mri_loss_grad's
 * pipelined kernel applies the update where its inverse column phase would have stored the gradient
 * (x and x0hat are read at a thread's output pixels while its last pass runs; the gradient never
 * reaches HBM).  x: the state [B][T][2][R][R], evaluated and then updated in place; one measurement for
 * all chains (y: [T][2][R][R]); masks of <= 64 kept rows per frame.
 *   mode 1: MCMCSampler.optimize (daps.py:210-223):  x -= p->lr (g + (x - x0hat) p->inv_sigma2)
 *   mode 2 / 3: the Langevin / momentum-free HMC step of MCMCSampler.sample (daps.py:93-104, :135-144)
 *     for the 'gaussian' prior solver, score = -g p->inv_tau2 + (x0hat - x) p->inv_sigma2, with the
 *     noise READ from `eps` (laid out like x, e.g. the reference's torch.randn_like, daps.py:192)
 * -- the results of mri_loss_grad followed by mcmc_update(algo 2 / 0 / 1, xt = x0hat, prior = NULL).
 * loss [B] / loss_total: the loss at the OLD x; loss_part: float[2*B*T]; nan_flag (may be NULL, modes
 * 2 / 3): set to 1 when an updated element is NaN.  p: HOST pointer, read during the call. */
int mri_step_update(float* x, const float* x0hat, const float* eps, const float* y, int B, int T, int R,
                    const int32_t* rows, const int32_t* nrows, int mask_frames, int max_rows, int mode,
                    const mcmc_params_t* p, float* loss_part, float* loss, float* loss_total,
                    int32_t* nan_flag, void* stream);

/* The dynamic-MRI operator at ANY frame size H x W, 2 .. 1024 each (round 3; csrc/mri_any.hip): the reference's
 * operator takes any `resolution` (forward_operator/dynamic_mri.py:9) and any frame width (its mask broadcasts
 * over the last axis, :15,24,35,48); the tuned entry points above are built for 192 x 192 only and refuse
 * other sizes.  Same reference lines, argument meaning and results (to fp32 rounding) as mri_forward
 * (dynamic_mri.py:42-48), mri_adjoint (its vector-Jacobian product) and mri_loss_grad (base.py:40-49;
 * chain_scale, loss, loss_total, grad as there, grad may be NULL), with rows = int32[mask_frames][H] and any
 * number of kept rows.  Mixed-radix passes over LDS, three kernels per call; the kept k_y rows cross `ws`
 * (device, 16-byte aligned, mri_any_ws_bytes(B, T, H, W, max_rows) bytes, any contents; calls that share it
 * must not overlap).  mri_any_ws_bytes returns 0 for sizes outside the limits. */
size_t mri_any_ws_bytes(int B, int T, int H, int W, int max_rows);
int mri_forward_any(const float* x, int B, int T, int H, int W, const int32_t* rows, const int32_t* nrows,
                    int mask_frames, int max_rows, void* ws, size_t ws_bytes, float* out, void* stream);
int mri_adjoint_any(const float* g, int B, int T, int H, int W, const int32_t* rows, const int32_t* nrows,
                    int mask_frames, int max_rows, void* ws, size_t ws_bytes, float* out, void* stream);
int mri_loss_grad_any(const float* x, const float* y, int y_batch, int B, int T, int H, int W,
                      const int32_t* rows, const int32_t* nrows, int mask_frames, int max_rows,
                      const float* chain_scale, void* ws, size_t ws_bytes, float* loss, float* loss_total,
                      float* grad, void* stream);

/* Multi-coil variants -- NO REFERENCE COUNTERPART (the reference operator is single-coil,
 * dynamic_mri.py:42-48; BASELINE.json names the coil-sensitivity multiply): y_c = M * FFT2(fftshift(
 * S_c * x)), c < C, with complex sensitivity maps sens [C][2][R][R] (Re, Im planes; static over
 * frames and chains).  The coil loop runs inside the per-frame workgroup: x is owned by one
 * workgroup for all coils and the gradient sum_c conj(S_c) A^H(2 r_c) accumulates in place.
 * out / y: [B or By][T][C][2][R][R].  loss[b] = sum_c sum |A_c(x) - y_c|^2; loss_part: float[B*T].
 * C = 1 with S = 1 + 0i gives mri_forward / mri_loss_grad(variant = 1) bit for bit. */
int mri_forward_mc(const float* x, const float* sens, int C, int B, int T, int R,
                   const int32_t* rows, const int32_t* nrows, int mask_frames, int max_rows,
                   float* out, void* stream);
int mri_loss_grad_mc(const float* x, const float* sens, int C, const float* y, int y_batch, int B,
                     int T, int R, const int32_t* rows, const int32_t* nrows, int mask_frames,
                     int max_rows, const float* chain_scale, float* loss_part, float* loss,
                     float* loss_total, float* grad, void* stream);

/* The multi-coil step in ONE kernel on the pipelined transforms (round 3; NO REFERENCE COUNTERPART):
 * a persistent workgroup owns a frame for all coils -- S_c * x is formed in registers from the strip
 * loads (z never exists in memory), each coil's processed kept rows go to the workgroup's scratch
 * slot, and the inverse column phase runs strip-outer / coil-inner so that sum_c conj(S_c) * A^H(2 r_c)
 * accumulates in registers and every gradient pixel is stored once.  Same arguments and results as
 * mri_loss_grad_mc (to fp32 rounding; C = 1 with S = 1 + 0i gives mri_loss_grad(variant 2)'s gradient
 * bit for bit); masks of <= 64 kept rows per frame.  loss_part: float[2*B*T].
 * ws: device scratch of mri_loss_grad_mc_fused_ws_bytes(C, B, T, max_rows) bytes (min(B*T, CUs) slots
 * of C x 64 x 192 complex; any contents; not needed when grad is NULL); calls that share it must not
 * overlap. */
size_t mri_loss_grad_mc_fused_ws_bytes(int C, int B, int T, int max_rows);
int mri_loss_grad_mc_fused(const float* x, const float* sens, int C, const float* y, int y_batch, int B,
                           int T, int R, const int32_t* rows, const int32_t* nrows, int mask_frames,
                           int max_rows, const float* chain_scale, float* loss_part, void* ws,
                           size_t ws_bytes, float* loss, float* loss_total, float* grad, void* stream);

/* The same operator as two streaming kernels around the SINGLE-coil step: z[f][c] = S_c * x[f]
 * ([nframes][C][2][R][R]), then mri_loss_grad on B x (T*C) frames (z as x, the multi-coil
 * measurement as y, per-frame mask tables repeated C times), then grad[f] = sum_c conj(S_c) * g[f][c].
 * Needs the two [nframes][C][2][R][R] buffers from the caller but runs on the pipelined kernels
 * (2.41 -> 1.74 ms at C = 8 and 64 chains; slower than mri_loss_grad_mc below 3 coils).
 * nframes <= 65535; 16-byte aligned buffers. */
int mri_coil_mul(const float* x, const float* sens, int C, int nframes, int R, float* z, void* stream);
int mri_coil_combine(const float* g, const float* sens, int C, int nframes, int R, float* out,
                     void* stream);

/* ------------------------------------------------------------------ non-uniform sampling (gridding)
 * NO REFERENCE COUNTERPART: the reference samples k-space on Cartesian rows only
 * (dynamic_mri.py:42-48); pynfft is an install line (README.md:28).  BASELINE.json's north_star
 * names a gridding kernel for the non-uniform path.  These are the kernels around the oversampled
 * FFT of a NUFFT (G = 2 R, separable Kaiser-Bessel kernel of width W <= 8, per-sample weights
 * tabulated on the host); the FFT itself is a library FFT called by the host layer between them.
 *   type 2 (image -> samples):  nufft_pad_apod -> FFT2 (G x G) -> nufft_interp
 *   type 1 (samples -> image):  nufft_spread   -> unnormalised IFFT2 -> nufft_crop_apod
 * Pixel n has the centred coordinate n - R/2; a sample at kappa cycles per field of view is the
 * sum over pixels of x[n] exp(-2 pi i kappa . (n - R/2) / R).  Grids are complex64 [frames][G][G];
 * sample arrays complex64 [frames][J]; frames = B * T; the trajectory may differ per frame t
 * (traj_frames = T) or be shared (1).  All plan tables are DEVICE pointers. */
typedef struct {
    int32_t R, G, W, J, traj_frames;
    const int32_t* ix0;        /* [traj_frames][J] first grid column of a sample's W x W footprint (may be negative: wraps) */
    const int32_t* iy0;
    const float* wx;           /* [traj_frames][J][W] kernel weights along x */
    const float* wy;
    const int32_t* tile_ptr;   /* [traj_frames][(G/16)(G/4) + 1] samples touching each 16 (x) by 4 (y) grid tile (CSR, tiles row-major) */
    const int32_t* tile_ent;   /* sample ids, ascending within a tile */
    const int64_t* tile_base;  /* [traj_frames] offset of a trajectory frame's list in tile_ent */
    const float* apod;         /* [R] de-apodisation per axis: 1 / (kernel transform at pixel n) */
    /* the six below may all be NULL (then nufft_interp gathers from global memory): every sample once, by the
     * 64 (x) by 32 (y) grid tile that holds the first point (ix0 mod G, iy0 mod G) of its footprint */
    const int32_t* own_ptr;    /* [traj_frames][(G/64)(G/32) + 1] (CSR, per trajectory frame from 0; tiles row-major) */
    const int32_t* own_ent;    /* [traj_frames][J] sample ids, ascending within a tile */
    const int32_t* own_x0;     /* [traj_frames][J] ix0 mod G, iy0 mod G and the weights once more, */
    const int32_t* own_y0;     /*                  in the order of own_ent (read consecutively) */
    const float* own_wx;       /* [traj_frames][J][W] */
    const float* own_wy;
    /* the three below may all be NULL (then nufft_spread always runs the kernel that takes 8 chains per tile):
     * the samples touching each 16 (x) by 2 (y) grid tile with their weights AT THE TILE'S POINTS, for the
     * spread kernel that gives every chain a lane (used from 16 chains on; bit-identical for finite input) */
    const int32_t* wide_rec;   /* [traj_frames][(G/16)(G/2)][16]: tile (row-major id), its first entry in wide_ent /
                                  wide_w, its entries (a multiple of 4: padded with zero-weight entries), `outside` (see nufft_adjoint; 0 if unused), the
                                  sample ids of its first 8 entries (any valid id where the list is shorter), 0 x 4
                                  -- in the order the kernel should take the tiles (longest lists first) */
    const int32_t* wide_ent;   /* sample ids, ascending within a tile (padding: any valid id) */
    const float* wide_w;       /* [entries][2][16] fp32 product wy[gy - iy0] * wx[gx - ix0] on the torus, 0 outside the footprint */
    /* the six below may all be NULL (then nufft_fwd_interp is not available): every sample once, by the strip of 16
     * grid columns that holds ix0 mod G, for the kernel that interpolates straight from the forward column pass */
    const int32_t* strip_ptr;  /* [traj_frames][G/16 + 1] (CSR, per trajectory frame from 0) */
    const int32_t* strip_ent;  /* [traj_frames][J] sample ids, ascending within a strip */
    const int32_t* strip_x0;   /* [traj_frames][J] ix0 mod G, iy0 mod G and the weights in the order of strip_ent */
    const int32_t* strip_y0;
    const float* strip_wx;     /* [traj_frames][J][W] */
    const float* strip_wy;
} nufft_plan_t;

int nufft_interp_blocks(int J);     /* partial sums per frame of the gathering kernel */
/* loss_part holds frames * nufft_interp_parts(plan) floats: (G/64)(G/32) per frame with the owner
 * bins (the interpolation then runs tile by tile out of LDS), nufft_interp_blocks(J) without. */
int nufft_interp_parts(const nufft_plan_t* plan);
int nufft_pad_apod(const float* x, int nframes, int T, const nufft_plan_t* plan, float* grid_c64,
                   void* stream);
/* out = interpolated samples, or with y_c64 != NULL: res_scale * (samples - y) and, with loss !=
 * NULL, loss[b] = sum |samples - y|^2 (loss_total = their sum, may be NULL). */
int nufft_interp(const float* grid_c64, int nframes, int T, const nufft_plan_t* plan,
                 const float* y_c64, int y_batch, float res_scale, float* out_c64, float* loss_part,
                 float* loss, float* loss_total, void* stream);
int nufft_spread(const float* vals_c64, int nframes, int T, const nufft_plan_t* plan, float* grid_c64,
                 void* stream);
int nufft_crop_apod(const float* grid_c64, int nframes, int T, const nufft_plan_t* plan,
                    const float* chain_scale, float out_scale, float* out, void* stream);

/* The grid transforms themselves for R = 192, G = 384 (nufft_grid_fft_supported), fused with the
 * padding / cropping and pruned to the rows and columns that are not zero: 384 = 16 x 24 point
 * passes, rows through LDS, columns 32 at a time.  NO REFERENCE COUNTERPART.
 *   nufft_fwd_grid == nufft_pad_apod + a forward FFT2 of the grid (torch.fft.fft2 convention);
 *   nufft_inv_grid == an unnormalised inverse FFT2 of the grid + nufft_crop_apod.
 * work_c64: caller-owned scratch of nframes * R * G complex64 (any contents). */
int nufft_grid_fft_supported(int R, int G);
/* FFT2 of the oversampled grids at any size G <= 1024, in place (round 3; csrc/nufft_any.hip): the transform
 * between nufft_pad_apod and nufft_interp, and between nufft_spread and nufft_crop_apod, for resolutions other
 * than the fused transforms' (nufft_grid_fft_supported); mixed-radix passes over LDS, rows then column strips.
 * grid_c64: [nframes][G][G] complex64; sign -1 forward (torch.fft.fft2), +1 unnormalised inverse.
 * NO REFERENCE COUNTERPART (as the whole gridding path). */
int nufft_fft2_any_supported(int G);
int nufft_fft2_any(float* grid_c64, int nframes, int G, int sign, void* stream);
int nufft_fwd_grid(const float* x, int nframes, int T, const nufft_plan_t* plan, float* work_c64,
                   float* grid_c64, void* stream);
int nufft_inv_grid(const float* grid_c64, int nframes, int T, const nufft_plan_t* plan,
                   const float* chain_scale, float out_scale, float* work_c64, float* out,
                   void* stream);

/* nufft_fwd_grid + nufft_interp without the grid: the forward column pass keeps a strip of 16 + W - 1 grid
 * columns in LDS and the samples whose footprint starts in the strip are interpolated from there (the
 * oversampled grid is neither written nor read: per frame 0.8 MB instead of 3.2 MB through HBM).  Needs the
 * plan's strip bins and R = 192; same arguments and results as nufft_interp (to fp32 rounding of the same sums),
 * loss_part holds frames * nufft_fwd_interp_parts(plan) floats (0 parts = not available for this plan).
 * NO REFERENCE COUNTERPART. */
int nufft_fwd_interp_parts(const nufft_plan_t* plan);
int nufft_fwd_interp(const float* x, int nframes, int T, const nufft_plan_t* plan, const float* y_c64,
                     int y_batch, float res_scale, float* work_c64, float* out_c64, float* loss_part,
                     float* loss, float* loss_total, void* stream);

/* nufft_spread + nufft_inv_grid (type 1: samples -> image) for R = 192 with the lane-per-chain lists, from 16
 * chains on (nufft_adjoint_supported), skipping what holds nothing: `range_host` is a HOST array
 * [traj_frames][G/16][2] of bounds (lo, hi) on the centred tile-row coordinate (grid row >> 1, minus 192 from
 * 96 on) outside which no sample reaches the 16-column tile column; tiles whose wide_rec record carries
 * `outside` = 1 (4th int; must mean: outside these bounds) are not written by the spread and not read by the
 * inverse column pass (a radial trajectory leaves 21 % of the grid empty).  grid_c64: scratch of nframes * G * G
 * complex64, work_c64 as for nufft_inv_grid.  Same result as the two calls.  NO REFERENCE COUNTERPART. */
int nufft_adjoint_supported(const nufft_plan_t* plan, int B);
int nufft_adjoint(const float* vals_c64, int nframes, int T, const nufft_plan_t* plan,
                  const int32_t* range_host, const float* chain_scale, float out_scale, float* grid_c64,
                  float* work_c64, float* out, void* stream);

/* mri_loss_grad without a workspace argument, for callers that cannot allocate: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in
 * the reference.  Same arguments, kernels and results as mri_loss_grad(variant = 0); the
 * 2*B*T-float workspace lives in a file-scope device array of the library instead
 * (2 * STEP_MRI_STATIC_MAX_FRAMES floats = 256 KiB, held by every GPU the library runs on until
 * the process ends).  B*T > STEP_MRI_STATIC_MAX_FRAMES -> STEP_E_INVALID.  The call allocates and
 * frees nothing.  All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (different streams or threads) race on it -- serialise them or use mri_loss_grad. */
#define STEP_MRI_STATIC_MAX_FRAMES 32768
int mri_loss_grad_static(const float* x, const float* y, int y_batch, int B, int T, int R,
                         const int32_t* rows, const int32_t* nrows, int mask_frames, int max_rows,
                         const float* chain_scale, float* loss, float* loss_total, float* grad,
                         void* stream);

/* ------------------------------------------------------------------ MCMC state update */

/* mcmc_params_t: defined above, next to bh_update_t */

/* score = -grad*inv_tau2 + (xt - x)*inv_sigma2 + prior  (score_fn daps.py:93-104), then
 * mc_update daps.py:135-144 in place on x (and velocity for hmc).  Sets *nan_flag (int32,
 * device) to 1 if any updated x is NaN (the guard of daps.py:202-203 without a per-step sync
 * unless the host chooses to read it).  hmc with momentum 0: the velocity buffer is neither read
 * nor written (its old value has weight zero in every step).
 * algo 2: x -= lr * (grad + (x - xt) * inv_sigma2) with xt = x0hat (MCMCSampler.optimize,
 * daps.py:210-223); velocity, prior and eps are not read and may be NULL. */
int mcmc_update(float* x, float* velocity, const float* grad, const float* xt, const float* prior,
                const float* eps, int64_t n, const mcmc_params_t* p, int32_t* nan_flag,
                void* stream);

/* mcmc_update with the noise drawn inside the kernel (algo 0 / 1): epsilon of daps.py:192 never
 * exists in HBM.  Philox4x32-10, key = seeds[b] (the chain's own 64-bit seed, DEVICE int64
 * [n / chain_len]), counter = (float4 index within the chain, step); normals by Box-Muller.  The
 * noise of a chain depends only on (seed, step, element) -- not on the batch it runs in, the rank
 * or the launch geometry -- which is what makes sharded runs reproducible per chain.  It is NOT
 * torch's random stream: parity with the reference's draws is statistical (the reference's
 * stream is replayed through mcmc_update with an explicit eps).
 * step_dev (may be NULL): device int64 holding the step, read instead of `step` (graph replay).
 * quad_w: 0 = one generator call per four consecutive elements (call j -> elements 4j .. 4j+3);
 * W > 0 = the chain is rows of W elements and a call covers four consecutive rows of one column
 * (call (row / 4) * W + column, its k-th normal for row 4 (row / 4) + k) -- the layout the
 * black-hole adjoint draws in when it applies the update itself (bh_update_t), so that a chain
 * gets the same noise on both routes; costs four calls per four elements here.  chain_len % W == 0.
 * eps_out (may be NULL): receives the noise that was used.
 * With 16-byte aligned buffers, chain_len % 4 == 0 (and W % 4 == 0, chain_len % (4 W) == 0 in the
 * quad layout) the kernel moves float4s; otherwise it draws the same numbers one element at a time
 * (element j of a chain: normal j & 3 of call j >> 2), so any chain length and any offset view works. */
int mcmc_update_philox(float* x, float* velocity, const float* grad, const float* xt,
                       const float* prior, int64_t n, const mcmc_params_t* p, const int64_t* seeds,
                       int64_t chain_len, int64_t step, const int64_t* step_dev, int64_t quad_w,
                       float* eps_out, int32_t* nan_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STEP_AMD_H */
