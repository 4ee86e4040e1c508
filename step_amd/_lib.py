"""ctypes binding of libstep_amd.so (C ABI: include/step_amd.h).

There is no fallback: if the shared library is missing or a symbol is absent the
import of any operator raises, and every non-zero return code raises RuntimeError
with the library's own message."""
import ctypes as C
import os

# torch must be imported BEFORE the library is opened: libstep_amd.so needs libamdhip64.so.7 and
# has to bind to the HIP runtime torch bundles (same soname), so that both share one runtime,
# one device context and torch's streams.  Loaded the other way round the process ends up with
# ROCm's system runtime underneath torch and no device is visible.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEP_AMD_LIB") or os.path.join(_HERE, "libstep_amd.so")   # env: A/B builds

c_f = C.c_float
c_i = C.c_int
c_i64 = C.c_int64
c_p = C.c_void_p


class bh_tables_t(C.Structure):
    _fields_ = [("n_vis", C.c_int32), ("n_tri", C.c_int32), ("n_quad", C.c_int32), ("T", C.c_int32),
                ("cp_index", c_p), ("cp_conj", c_p), ("camp_index", c_p),
                ("cp_ptr", c_p), ("cp_ent", c_p), ("ca_ptr", c_p), ("ca_ent", c_p)]


class bh_meas_t(C.Structure):
    _fields_ = [("vis_c64", c_p), ("sigma_vis", c_p), ("amp", c_p), ("sigma_amp", c_p),
                ("cphase", c_p), ("sigma_cphase", c_p), ("logcamp", c_p), ("sigma_logcamp", c_p),
                ("flux", c_p), ("batch", C.c_int32)]


class bh_weights_t(C.Structure):
    _fields_ = [("w_vis", c_f), ("w_amp", c_f), ("w_cp", c_f), ("w_camp", c_f), ("w_flux", c_f),
                ("normalize", C.c_int32), ("cphase_form", C.c_int32)]


class bh_step_ws_t(C.Structure):
    _fields_ = [("vis_c64", c_p), ("flux", c_p), ("dcp", c_p), ("dca", c_p), ("S", c_p)]


class nufft_plan_t(C.Structure):
    _fields_ = [("R", C.c_int32), ("G", C.c_int32), ("W", C.c_int32), ("J", C.c_int32), ("traj_frames", C.c_int32),
                ("ix0", c_p), ("iy0", c_p), ("wx", c_p), ("wy", c_p), ("tile_ptr", c_p), ("tile_ent", c_p),
                ("tile_base", c_p), ("apod", c_p), ("own_ptr", c_p), ("own_ent", c_p),
                ("own_x0", c_p), ("own_y0", c_p), ("own_wx", c_p), ("own_wy", c_p),
                ("wide_rec", c_p), ("wide_ent", c_p), ("wide_w", c_p),
                ("strip_ptr", c_p), ("strip_ent", c_p), ("strip_x0", c_p), ("strip_y0", c_p), ("strip_wx", c_p), ("strip_wy", c_p)]


class mcmc_params_t(C.Structure):
    _fields_ = [("inv_tau2", c_f), ("inv_sigma2", c_f), ("lr", c_f), ("sqrt_lr", c_f),
                ("sqrt_2lr", c_f), ("momentum", c_f), ("noise_scale", c_f), ("algo", C.c_int32)]


class bh_update_t(C.Structure):
    _fields_ = [("x_inout", c_p), ("x0hat", c_p), ("lr", c_f), ("inv_sigma2", c_f),
                ("mcmc", C.POINTER(mcmc_params_t)), ("seeds", c_p), ("step", C.c_int64),
                ("step_dev", c_p), ("nan_flag", c_p), ("eps", c_p)]


# name -> (restype, argtypes); must list every symbol include/step_amd.h declares
# the version of include/step_amd.h these signatures were written against (step_abi_version(); checked at load)
ABI_VERSION = 12

SIGNATURES = {
    "step_last_error": (C.c_char_p, []),
    "step_abi_version": (c_i, []),
    "bh_build_closure_tables": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_p, c_i64,
                                      c_p, c_p, c_p, c_p]),
    "bh_build_incidence": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p]),
    "mri_build_mask": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p]),
    "bh_dense_rows_padded": (c_i, [c_i]),
    "bh_dense_packed_floats": (C.c_size_t, [c_i, c_i, c_i]),
    "bh_dense_fwd_workspace_floats": (C.c_size_t, [c_i, c_i, c_i, c_i]),
    "bh_dense_pack": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "bh_dft_fwd_dense": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p]),
    "bh_dft_adj_dense": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p]),
    "bh_dft_fwd_sep": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_f, c_f, c_f, c_p, c_p, c_p]),
    "bh_dft_adj_sep": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p]),
    "bh_sep3_supported": (c_i, [c_i, c_i]),
    "bh_sep3_tables_ok": (c_i, [c_p, c_i, c_i]),
    "bh_dft_fwd_sep3": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_f, c_f, c_f, c_p, c_p, c_p]),
    "bh_dft_adj_sep3": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_i,
                              c_p, c_p]),
    "bh_build_frame_closures": (c_i, [c_p, c_p, c_i64, c_p, c_i64, c_i, c_i, c_p]),
    "bh_loss_grad_sep3": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, C.POINTER(bh_tables_t), c_p,
                                C.POINTER(bh_meas_t), C.POINTER(bh_weights_t), C.POINTER(bh_step_ws_t),
                                c_p, c_f, c_p, c_p, c_p, c_p, C.POINTER(bh_update_t), c_p]),
    "bh_dft_adj_sep3_update": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_f,
                                     C.POINTER(bh_update_t), c_p, c_i, c_p, c_p]),
    "bh_closure_loss": (c_i, [c_p, c_p, c_i, C.POINTER(bh_tables_t), C.POINTER(bh_meas_t),
                              C.POINTER(bh_weights_t), c_p, c_p, c_p, c_p, c_p]),
    "bh_closure_chisq": (c_i, [c_p, c_p, c_i, c_i, C.POINTER(bh_tables_t), C.POINTER(bh_meas_t),
                               c_i, c_p, c_p]),
    "bh_closure_observe": (c_i, [c_p, c_p, c_i, C.POINTER(bh_tables_t), c_p, c_p, c_p, c_p, c_p,
                                 c_p, c_p]),
    "mri_forward": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "mri_adjoint": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "mri_loss_grad": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_p, c_p,
                            c_p, c_p, c_p]),
    "mri_loss_grad_split_ws_bytes": (C.c_size_t, [c_i, c_i, c_i, c_i]),
    "mri_loss_grad_split": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_p, C.c_size_t,
                                  c_p, c_p, c_p, c_p]),
    "mri_step_update": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, C.POINTER(mcmc_params_t),
                              c_p, c_p, c_p, c_p, c_p]),
    "mri_any_ws_bytes": (C.c_size_t, [c_i, c_i, c_i, c_i, c_i]),
    "mri_forward_any": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, C.c_size_t, c_p, c_p]),
    "mri_adjoint_any": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, C.c_size_t, c_p, c_p]),
    "mri_loss_grad_any": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, C.c_size_t,
                                c_p, c_p, c_p, c_p]),
    "mri_forward_mc": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "mri_loss_grad_mc": (c_i, [c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p,
                               c_p, c_p, c_p]),
    "mri_loss_grad_mc_fused_ws_bytes": (C.c_size_t, [c_i, c_i, c_i, c_i]),
    "mri_loss_grad_mc_fused": (c_i, [c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p,
                                     C.c_size_t, c_p, c_p, c_p, c_p]),
    "mri_coil_mul": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "mri_coil_combine": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "nufft_interp_blocks": (c_i, [c_i]),
    "nufft_pad_apod": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_p]),
    "nufft_interp": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p]),
    "nufft_spread": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_p]),
    "nufft_crop_apod": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_f, c_p, c_p]),
    "nufft_fft2_any_supported": (c_i, [c_i]),
    "nufft_fft2_any": (c_i, [c_p, c_i, c_i, c_i, c_p]),
    "nufft_grid_fft_supported": (c_i, [c_i, c_i]),
    "nufft_interp_parts": (c_i, [C.POINTER(nufft_plan_t)]),
    "nufft_adjoint_supported": (c_i, [C.POINTER(nufft_plan_t), c_i]),
    "nufft_adjoint": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_p, c_f, c_p, c_p, c_p, c_p]),
    "nufft_fwd_interp_parts": (c_i, [C.POINTER(nufft_plan_t)]),
    "nufft_fwd_interp": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "nufft_fwd_grid": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_p, c_p]),
    "nufft_inv_grid": (c_i, [c_p, c_i, c_i, C.POINTER(nufft_plan_t), c_p, c_f, c_p, c_p, c_p]),
    "mri_loss_grad_static": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p,
                                   c_p, c_p]),
    "mcmc_update_philox": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, C.POINTER(mcmc_params_t), c_p, c_i64,
                                 c_i64, c_p, c_i64, c_p, c_p, c_p]),
    "mcmc_update": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, C.POINTER(mcmc_params_t), c_p, c_p]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "step_amd: %s is not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C step_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        got = L.step_abi_version()
        if got != ABI_VERSION:             # a stale .so under newer bindings (or the reverse) would misread arguments
            raise RuntimeError("step_amd: %s speaks ABI %d, these bindings %d -- rebuild with `make -C step_amd/csrc`"
                               % (LIB_PATH, got, ABI_VERSION))
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().step_last_error()
        raise RuntimeError("step_amd %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def stream_of(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream
