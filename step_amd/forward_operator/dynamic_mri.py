"""Dynamic-MRI operator on MI355X kernels behind the reference's interface
(reference forward_operator/dynamic_mri.py): registry name ``'mri'``, constructor kwargs
``acc_factor, mode, resolution, frames, num_acs_lines, sigma``, ``self.mask`` of the
reference's shape, ``__call__`` = M * FFT2(fftshift(x_re + i x_im)).

``loss`` / ``gradient`` (inherited semantics of base.py:40-49) run the fused
forward + residual + adjoint kernel: one launch per Langevin step."""
import numpy as np
import torch

from .. import _lib
from .base import Operator, register_operator

__all__ = ["DynamicMRI", "MultiCoilMRI", "birdcage_maps"]

_MODES = {"same_mask": 0, "split_mask": 1, "random_split_mask": 2}


def _f32c(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class _ForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        ctx.op, ctx.dtype = op, x.dtype
        return op._launch_forward(_f32c(x, x.device))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return ctx.op._launch_adjoint(_f32c(g, g.device)).to(ctx.dtype), None


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, y, unit_upstream):
        need = ctx.needs_input_grad[0]
        loss, grad = op._launch_loss_grad(_f32c(x, x.device), y, need, None)
        ctx.dtype, ctx.unit = x.dtype, bool(unit_upstream)
        if need:
            ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        if ctx.unit:                       # LatentWrapper.gradient: the loss is only summed
            return grad.to(ctx.dtype), None, None, None
        g = grad * grad_loss.to(grad.dtype).view(-1, 1, 1, 1, 1)
        return g.to(ctx.dtype), None, None, None


@register_operator('mri')
class DynamicMRI(Operator):
    def __init__(self, acc_factor, mode='same_mask', resolution=192, frames=12, num_acs_lines=12,
                 sigma=0):
        super().__init__(sigma)
        self.acc_factor = acc_factor
        if mode not in _MODES:
            raise ValueError('Unknown mode: {}'.format(mode))
        self.mode, self.resolution, self.frames = mode, resolution, frames
        mf = 1 if mode == 'same_mask' else frames
        m = np.zeros((mf, resolution), np.float32)
        rc = _lib.lib().mri_build_mask(int(acc_factor), _MODES[mode], int(resolution), int(frames),
                                       int(num_acs_lines), _lib.ptr(m))
        _lib.check(rc, "mri_build_mask")
        # reference layout: [1, 1 or T, 1, H, 1] float32 (dynamic_mri.py:15,24,35)
        self.mask = torch.from_numpy(m).view(1, mf, 1, resolution, 1)
        rows = np.full((mf, resolution), -1, np.int32)
        nrows = np.zeros(mf, np.int32)
        for f in range(mf):
            kept = np.nonzero(m[f])[0].astype(np.int32)
            rows[f, :len(kept)] = kept
            nrows[f] = len(kept)
        self._rows_host, self._nrows_host = rows, nrows
        self._max_rows = int(nrows.max())
        self._dev_tables = {}

    _prof = None      # bench.py sets this to {} to collect (start, end) event pairs
    kernel_variant = 0       # mri_loss_grad's `variant`: 0 auto, 1 first generation, 2 pipelined, 4 persistent
    # Batches whose frames leave most CUs idle (B * T * 3 <= CUs: up to 6 chains at T = 12) share a frame
    # between 6 or 3 workgroups (mri_loss_grad_split; bit-identical gradient).  None = whenever it
    # applies and kernel_variant is 0, False = never, 2 / 3 / 6 = that many workgroups per frame.
    split_frames = None

    def _split_ws(self, device, nbytes):
        """The split kernel's workspace: zeroed once, left zeroed by every call (grown when needed)."""
        key = ("split_ws", str(device))
        ws = self._dev_tables.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.zeros(nbytes, device=device, dtype=torch.uint8)
            self._dev_tables[key] = ws
        return ws

    def _tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            self._dev_tables[key] = (torch.from_numpy(self._rows_host).to(device),
                                     torch.from_numpy(self._nrows_host).to(device))
        return self._dev_tables[key]

    # Frames other than 192 x 192 (the shipped configuration, for which the tuned kernels are built) run on the
    # mixed-radix kernels of csrc/mri_any.hip: any resolution 2 .. 1024, any frame width (the reference's mask
    # broadcasts over the last axis).  True sends 192 x 192 there too (tests: an independent implementation).
    generic_kernels = False
    any_width = True         # MultiCoilMRI (192 x 192 kernels only) turns this off

    def _generic(self, x):
        return self.generic_kernels or x.shape[3] != 192 or x.shape[4] != 192

    def _any_ws(self, x):
        B, T, H, W = x.shape[0], x.shape[1], x.shape[3], x.shape[4]
        nb = _lib.lib().mri_any_ws_bytes(B, T, H, W, self._max_rows)
        if nb == 0:
            raise ValueError("mri operator: frames of %d x %d are outside the kernels' range (2 .. 1024)" % (H, W))
        key = ("any_ws", str(x.device))
        ws = self._dev_tables.get(key)
        if ws is None or ws.numel() < nb:
            ws = torch.empty(nb, device=x.device, dtype=torch.uint8)
            self._dev_tables[key] = ws
        return ws

    def _dims(self, x):
        if x.dim() != 5 or x.shape[2] != 2 or x.shape[3] != self.resolution \
                or (x.shape[4] != self.resolution and not self.any_width):
            raise ValueError("mri operator expects [B, T, 2, %d, %s], got %s"
                             % (self.resolution, "W" if self.any_width else self.resolution, tuple(x.shape)))
        mf = self.mask.shape[1]
        if mf != 1 and x.shape[1] != mf:
            raise ValueError("mask has %d frames, input has %d" % (mf, x.shape[1]))
        return x.shape[0], x.shape[1]

    def _launch_forward(self, x):
        B, T = self._dims(x)
        rows, nrows = self._tables(x.device)
        out = torch.empty_like(x)
        if self._generic(x):
            ws = self._any_ws(x)
            rc = _lib.lib().mri_forward_any(x.data_ptr(), B, T, x.shape[3], x.shape[4], rows.data_ptr(),
                                            nrows.data_ptr(), self.mask.shape[1], self._max_rows, ws.data_ptr(),
                                            ws.numel(), out.data_ptr(), _lib.stream_of(x))
            _lib.check(rc, "mri_forward_any")
            return out
        rc = _lib.lib().mri_forward(x.data_ptr(), B, T, self.resolution, rows.data_ptr(),
                                    nrows.data_ptr(), self.mask.shape[1], self._max_rows,
                                    out.data_ptr(), _lib.stream_of(x))
        _lib.check(rc, "mri_forward")
        return out

    def _launch_adjoint(self, g):
        B, T = self._dims(g)
        rows, nrows = self._tables(g.device)
        out = torch.empty_like(g)
        if self._generic(g):
            ws = self._any_ws(g)
            rc = _lib.lib().mri_adjoint_any(g.data_ptr(), B, T, g.shape[3], g.shape[4], rows.data_ptr(),
                                            nrows.data_ptr(), self.mask.shape[1], self._max_rows, ws.data_ptr(),
                                            ws.numel(), out.data_ptr(), _lib.stream_of(g))
            _lib.check(rc, "mri_adjoint_any")
            return out
        rc = _lib.lib().mri_adjoint(g.data_ptr(), B, T, self.resolution, rows.data_ptr(),
                                    nrows.data_ptr(), self.mask.shape[1], self._max_rows,
                                    out.data_ptr(), _lib.stream_of(g))
        _lib.check(rc, "mri_adjoint")
        return out

    def _launch_loss_grad(self, x, y, need_grad, chain_scale):
        B, T = self._dims(x)
        y = _f32c(y, x.device)
        if y.dim() != 5 or y.shape[1:] != x.shape[1:] or y.shape[0] not in (1, B):
            raise ValueError("measurement shape %s does not match input %s"
                             % (tuple(y.shape), tuple(x.shape)))
        rows, nrows = self._tables(x.device)
        loss = torch.empty(B + 1, device=x.device, dtype=torch.float32)   # [B] per chain + their sum
        grad = torch.empty_like(x) if need_grad else None
        if self._generic(x):
            ws = self._any_ws(x)
            t0 = None
            if self._prof is not None:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(torch.cuda.current_stream(x.device))
            rc = _lib.lib().mri_loss_grad_any(x.data_ptr(), y.data_ptr(), y.shape[0], B, T, x.shape[3], x.shape[4],
                                              rows.data_ptr(), nrows.data_ptr(), self.mask.shape[1], self._max_rows,
                                              _lib.ptr(chain_scale), ws.data_ptr(), ws.numel(), loss.data_ptr(),
                                              loss.data_ptr() + 4 * B, _lib.ptr(grad), _lib.stream_of(x))
            if t0 is not None:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record(torch.cuda.current_stream(x.device))
                self._prof.setdefault("mri_loss_grad", []).append((t0, t1))
            _lib.check(rc, "mri_loss_grad_any")
            self._loss_total = loss[B]
            return loss[:B], grad
        ws_bytes = 0
        if y.shape[0] == 1 and self.kernel_variant == 0 and self.split_frames is not False:
            ws_bytes = _lib.lib().mri_loss_grad_split_ws_bytes(B, T, self._max_rows, int(self.split_frames or 0))
        part = None if ws_bytes else torch.empty(2 * B * T, device=x.device, dtype=torch.float32)
        t0 = None
        if self._prof is not None:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(torch.cuda.current_stream(x.device))
        if ws_bytes:
            ws = self._split_ws(x.device, ws_bytes)
            rc = _lib.lib().mri_loss_grad_split(x.data_ptr(), y.data_ptr(), B, T, self.resolution,
                                                rows.data_ptr(), nrows.data_ptr(), self.mask.shape[1],
                                                self._max_rows, _lib.ptr(chain_scale),
                                                int(self.split_frames or 0), ws.data_ptr(), ws.numel(),
                                                loss.data_ptr(), loss.data_ptr() + 4 * B, _lib.ptr(grad),
                                                _lib.stream_of(x))
        else:
            rc = _lib.lib().mri_loss_grad(x.data_ptr(), y.data_ptr(), y.shape[0], B, T,
                                          self.resolution, rows.data_ptr(), nrows.data_ptr(),
                                          self.mask.shape[1], self._max_rows, _lib.ptr(chain_scale),
                                          self.kernel_variant, part.data_ptr(), loss.data_ptr(),
                                          loss.data_ptr() + 4 * B, _lib.ptr(grad), _lib.stream_of(x))
        if t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream(x.device))
            self._prof.setdefault("mri_loss_grad", []).append((t0, t1))
        _lib.check(rc, "mri_loss_grad")
        self._loss_total = loss[B]           # scalar sum over the chains, summed by the finish kernel
        return loss[:B], grad

    # ------------------------------------------------------------------ the step with the update inside
    @property
    def has_descent_step(self):
        """descent_step / langevin_step(eps=...) live in the pipelined kernel: masks of <= 64 kept rows."""
        return (type(self) is DynamicMRI and self._max_rows <= 64 and int(self.resolution) == 192
                and not self.generic_kernels)

    has_langevin_step = has_descent_step
    langevin_draws_noise = False     # langevin_step needs `eps`: no in-kernel generator here

    def prefers_langevin_step(self, batch):
        return self.has_langevin_step

    def _step_update(self, x, y, x0hat, eps, mode, params, nan_flag, return_loss, who):
        import ctypes as C
        B, T = self._dims(x)
        if x.shape[3] != 192 or x.shape[4] != 192:
            raise ValueError("%s lives in the 192 x 192 kernels; got frames of %d x %d" % (who, x.shape[3], x.shape[4]))
        if (x.dtype != torch.float32 or not x.is_contiguous() or x.device.type != 'cuda' or x.requires_grad):
            raise ValueError("%s needs a contiguous fp32 CUDA state without autograd history" % who)
        x0 = x0hat.detach()
        if x0.dtype != torch.float32 or not x0.is_contiguous() or x0.shape != x.shape or x0.device != x.device:
            x0 = x0.to(device=x.device, dtype=torch.float32).expand_as(x).contiguous()
        if x0.data_ptr() == x.data_ptr():
            raise ValueError("%s: x and x0hat must not share storage" % who)
        y = _f32c(y, x.device)
        if y.dim() != 5 or y.shape[0] != 1 or y.shape[1:] != x.shape[1:]:
            raise ValueError("%s takes one measurement for all chains, got %s" % (who, tuple(y.shape)))
        if eps is not None and (eps.dtype != torch.float32 or eps.shape != x.shape or eps.device != x.device
                                or not eps.is_contiguous()):
            raise ValueError("%s: eps must be contiguous float32 %s on %s" % (who, tuple(x.shape), x.device))
        rows, nrows = self._tables(x.device)
        part = torch.empty(2 * B * T, device=x.device, dtype=torch.float32)
        loss = torch.empty(B + 1, device=x.device, dtype=torch.float32)
        rc = _lib.lib().mri_step_update(x.data_ptr(), x0.data_ptr(), _lib.ptr(eps), y.data_ptr(), B, T,
                                        self.resolution, rows.data_ptr(), nrows.data_ptr(), self.mask.shape[1],
                                        self._max_rows, mode, C.byref(params), part.data_ptr(), loss.data_ptr(),
                                        loss.data_ptr() + 4 * B, _lib.ptr(nan_flag), _lib.stream_of(x))
        _lib.check(rc, "mri_step_update")
        return loss[B] if return_loss else None

    def descent_step(self, x, y, x0hat, lr, inv_sigma2, return_loss=False):
        """One step of MCMCSampler.optimize (daps.py:210-223) in place on x:
        ``x -= lr * (gradient(x, y) + (x - x0hat) * inv_sigma2)``, applied by the kernel's inverse column
        phase where it would have stored the gradient (mri_step_update); -> the summed loss at the old x."""
        p = _lib.mcmc_params_t(1.0, float(inv_sigma2), float(lr), 0.0, 0.0, 0.0, 0.0, 2)
        return self._step_update(x, y, x0hat, None, 1, p, None, return_loss, "descent_step")

    def langevin_step(self, x, y, x0hat, params, seeds=None, step=0, step_dev=None, nan_flag=None,
                      return_loss=False, eps=None):
        """One step of MCMCSampler.sample (daps.py:184-200) in place on x for the 'gaussian' prior solver
        (score ``-g / tau^2 + (x0hat - x) / sigma^2``), langevin (params.algo 0) or hmc at momentum 0
        (algo 1), with the step's noise handed over as the tensor ``eps`` (the reference's
        ``torch.randn_like``): the kernel applies the update, the gradient never reaches HBM.  This
        operator draws no noise itself: ``eps`` is required (``seeds`` / ``step`` are ignored)."""
        if eps is None:
            raise NotImplementedError("the MRI kernel reads its noise from `eps`; in-kernel Philox noise is the "
                                      "black-hole adjoint's (use gradient + mcmc_update_philox here)")
        return self._step_update(x, y, x0hat, eps, 2 + int(params.algo), params, nan_flag, return_loss,
                                 "langevin_step")

    # ------------------------------------------------------------------ reference API
    def __call__(self, x):
        # [B, T, C, H, W] -> masked k-space, same layout (dynamic_mri.py:42-48)
        return _ForwardFn.apply(x, self)

    def loss(self, x, y, unit_upstream=False):
        """``unit_upstream=True``: the caller only sums the loss (LatentWrapper.gradient), so the
        backward hands on the gradient computed in the forward without a scaling pass."""
        return _LossFn.apply(x, self, y, unit_upstream)

    def gradient(self, x, y, return_loss=False):
        loss, grad = self._launch_loss_grad(_f32c(x, x.device), y, True, None)
        grad = grad.to(x.dtype)
        if return_loss:
            return grad, self._loss_total
        return grad


def birdcage_maps(coils, resolution=192, radius=1.2, dtype=torch.complex64):
    """Smooth synthetic coil sensitivities [C, R, R] (complex): coil c sits on a circle around the
    field of view, magnitude ~ 1 / distance, phase = the angle towards the coil; normalised so that
    sum_c |S_c|^2 = 1 in every pixel (the usual SENSE convention)."""
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, resolution, dtype=torch.float64),
                            torch.linspace(-1, 1, resolution, dtype=torch.float64), indexing="ij")
    maps = []
    for c in range(coils):
        a = 2 * np.pi * c / coils
        dx, dy = xx - radius * np.cos(a), yy - radius * np.sin(a)
        maps.append(torch.polar(1.0 / torch.sqrt(dx * dx + dy * dy + 0.05), torch.atan2(dy, dx)))
    s = torch.stack(maps)
    return (s / s.abs().pow(2).sum(0, keepdim=True).sqrt()).to(dtype)


class _ForwardMCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        ctx.op, ctx.dtype = op, x.dtype
        return op._launch_forward(_f32c(x, x.device))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        # A^H g through the fused kernel: the gradient of sum |A x - y|^2 at x = 0 with y = -g / 2
        # is 2 A^H(0 - y) = A^H g
        op = ctx.op
        z = torch.zeros(g.shape[0], g.shape[1], 2, op.resolution, op.resolution, device=g.device)
        _, grad = op._launch_loss_grad(z, -0.5 * _f32c(g, g.device), True, None)
        return grad.to(ctx.dtype), None


@register_operator('mri_multicoil')
class MultiCoilMRI(DynamicMRI):
    """Multi-coil dynamic MRI, y_c = M * FFT2(fftshift(S_c * x)): NO REFERENCE COUNTERPART (the
    reference's operator is single-coil, dynamic_mri.py:42-48); the kernel-backed operator that
    BASELINE.json's north_star names.  ``sens``: complex [C, R, R] (or real [C, 2, R, R]); ``coils``:
    build ``birdcage_maps(coils)`` instead.  ``__call__`` returns [B, T, C, 2, R, R]; ``loss`` /
    ``gradient`` keep the single-coil semantics summed over coils.  With C = 1 and S = 1 every
    result equals the single-coil operator's first-generation kernels bit for bit.  Resolutions other
    than 192 (the tuned kernels') run the coil multiply / combine kernels around the any-size
    single-coil kernels (csrc/mri_any.hip) on B x (T C) frames."""

    def __init__(self, acc_factor, mode='same_mask', resolution=192, frames=12, num_acs_lines=12,
                 sigma=0, sens=None, coils=None):
        super().__init__(acc_factor, mode, resolution, frames, num_acs_lines, sigma)
        if (int(resolution) * int(resolution)) % 4:
            raise ValueError("mri_multicoil: resolution %r (the coil multiply moves four pixels at a time: R * R must "
                             "be a multiple of 4)" % (resolution,))
        if sens is None:
            if not coils:
                raise ValueError("mri_multicoil needs `sens` ([C, R, R] complex) or `coils`")
            sens = birdcage_maps(int(coils), resolution)
        sens = torch.as_tensor(sens)
        if torch.is_complex(sens):
            sens = torch.stack([sens.real, sens.imag], dim=1)
        if sens.dim() != 4 or sens.shape[1] != 2 or sens.shape[2] != resolution or sens.shape[3] != resolution:
            raise ValueError("sens must be [C, R, R] complex or [C, 2, R, R], got %s" % (tuple(sens.shape),))
        self.sens = sens.to(torch.float32).contiguous()
        self.coils = self.sens.shape[0]
        self._sens_dev = {}

    def _sens_on(self, device):
        key = str(device)
        if key not in self._sens_dev:
            self._sens_dev[key] = self.sens.to(device)
        return self._sens_dev[key]

    def _any_ws_mc(self, x):
        """Workspace of the any-size kernels for the B x (T C) single-coil frames of the split path."""
        B, T, R = x.shape[0], self.frames * self.coils, self.resolution
        nb = _lib.lib().mri_any_ws_bytes(B, T, R, R, self._max_rows)
        if nb == 0:
            raise ValueError("mri_multicoil: frames of %d x %d are outside the kernels' range (2 .. 1024)" % (R, R))
        key = ("any_ws_mc", str(x.device))
        ws = self._dev_tables.get(key)
        if ws is None or ws.numel() < nb:
            ws = torch.empty(nb, device=x.device, dtype=torch.uint8)
            self._dev_tables[key] = ws
        return ws

    def _launch_forward(self, x):
        B, T = self._dims(x)
        rows, nrows = self._tables(x.device)
        out = torch.empty(B, T, self.coils, 2, self.resolution, self.resolution, device=x.device)
        if int(self.resolution) != 192:
            # other resolutions: coil multiply, then the any-size single-coil kernels on B x (T C) frames
            L, st, Cn, R = _lib.lib(), _lib.stream_of(x), self.coils, self.resolution
            rows, nrows = self._tables_mc(x.device)
            z = torch.empty_like(out)
            _lib.check(L.mri_coil_mul(x.data_ptr(), self._sens_on(x.device).data_ptr(), Cn, B * T, R, z.data_ptr(), st),
                       "mri_coil_mul")
            ws = self._any_ws_mc(x)
            mf = self.mask.shape[1] * (Cn if self.mask.shape[1] > 1 else 1)
            _lib.check(L.mri_forward_any(z.data_ptr(), B, T * Cn, R, R, rows.data_ptr(), nrows.data_ptr(), mf,
                                         self._max_rows, ws.data_ptr(), ws.numel(), out.data_ptr(), st),
                       "mri_forward_any (multi-coil)")
            return out
        rc = _lib.lib().mri_forward_mc(x.data_ptr(), self._sens_on(x.device).data_ptr(), self.coils, B, T,
                                       self.resolution, rows.data_ptr(), nrows.data_ptr(),
                                       self.mask.shape[1], self._max_rows, out.data_ptr(), _lib.stream_of(x))
        _lib.check(rc, "mri_forward_mc")
        return out

    # loss / gradient as coil multiply -> single-coil step on B x (T C) frames -> coil combine
    # (three launches on the pipelined kernels, two [B, T, C, 2, R, R] buffers) instead of the one
    # fused first-generation kernel: 2.41 -> 1.85 ms at 8 coils and 64 chains, 0.256 -> 0.280 ms at one
    # coil.  None = by coil count (split from 3 coils on), True / False force one or the other.
    split_coils = None
    any_width = False

    def _tables_mc(self, device):
        """Row tables of the single-coil step seen as T * C frames: a per-frame mask repeats C times."""
        key = ("mc", str(device))
        if key not in self._dev_tables:
            rows, nrows = self._tables(device)
            if self.mask.shape[1] > 1:
                rows = rows.repeat_interleave(self.coils, dim=0).contiguous()
                nrows = nrows.repeat_interleave(self.coils, dim=0).contiguous()
            self._dev_tables[key] = (rows, nrows)
        return self._dev_tables[key]

    def _launch_split(self, x, y, need_grad, chain_scale):
        B, T, Cn, R = x.shape[0], self.frames, self.coils, self.resolution
        L, st, dev = _lib.lib(), _lib.stream_of(x), x.device
        sens = self._sens_on(dev)
        rows, nrows = self._tables_mc(dev)
        t0 = None
        if self._prof is not None:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(torch.cuda.current_stream(dev))
        z = torch.empty(B, T, Cn, 2, R, R, device=dev, dtype=torch.float32)
        _lib.check(L.mri_coil_mul(x.data_ptr(), sens.data_ptr(), Cn, B * T, R, z.data_ptr(), st), "mri_coil_mul")
        part = torch.empty(2 * B * T * Cn, device=dev, dtype=torch.float32)
        loss = torch.empty(B + 1, device=dev, dtype=torch.float32)
        g = torch.empty_like(z) if need_grad else None
        mf = self.mask.shape[1] * (Cn if self.mask.shape[1] > 1 else 1)
        if int(R) != 192:
            ws = self._any_ws_mc(x)
            rc = L.mri_loss_grad_any(z.data_ptr(), y.data_ptr(), y.shape[0], B, T * Cn, R, R, rows.data_ptr(),
                                     nrows.data_ptr(), mf, self._max_rows, _lib.ptr(chain_scale), ws.data_ptr(),
                                     ws.numel(), loss.data_ptr(), loss.data_ptr() + 4 * B, _lib.ptr(g), st)
        else:
            rc = L.mri_loss_grad(z.data_ptr(), y.data_ptr(), y.shape[0], B, T * Cn, R, rows.data_ptr(),
                                 nrows.data_ptr(), mf, self._max_rows, _lib.ptr(chain_scale), self.kernel_variant,
                                 part.data_ptr(), loss.data_ptr(), loss.data_ptr() + 4 * B, _lib.ptr(g), st)
        _lib.check(rc, "mri_loss_grad (multi-coil, split)")
        grad = None
        if need_grad:
            grad = torch.empty_like(x)
            _lib.check(L.mri_coil_combine(g.data_ptr(), sens.data_ptr(), Cn, B * T, R, grad.data_ptr(), st),
                       "mri_coil_combine")
        if t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream(dev))
            self._prof.setdefault("mri_loss_grad_mc", []).append((t0, t1))
        self._loss_total = loss[B]
        return loss[:B], grad

    # loss / gradient in ONE kernel on the pipelined transforms (mri_loss_grad_mc_fused, round 3): the coil
    # multiply in the strip loads, the coil sum in the inverse column phase's registers.  None = whenever
    # every mask frame keeps <= 64 rows (every shipped mode) and split_coils is not forced; False = never.
    fused_coils = None

    def _launch_fused(self, x, y, need_grad, chain_scale):
        B, T, Cn, R = x.shape[0], self.frames, self.coils, self.resolution
        L, dev = _lib.lib(), x.device
        rows, nrows = self._tables(dev)
        part = torch.empty(2 * B * T, device=dev, dtype=torch.float32)
        loss = torch.empty(B + 1, device=dev, dtype=torch.float32)
        grad = torch.empty_like(x) if need_grad else None
        ws, nbytes = None, 0
        if need_grad:
            nbytes = L.mri_loss_grad_mc_fused_ws_bytes(Cn, B, T, self._max_rows)
            key = ("mcf_ws", str(dev))
            ws = self._dev_tables.get(key)
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
                self._dev_tables[key] = ws
        t0 = None
        if self._prof is not None:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(torch.cuda.current_stream(dev))
        rc = L.mri_loss_grad_mc_fused(x.data_ptr(), self._sens_on(dev).data_ptr(), Cn, y.data_ptr(), y.shape[0], B, T,
                                      R, rows.data_ptr(), nrows.data_ptr(), self.mask.shape[1], self._max_rows,
                                      _lib.ptr(chain_scale), part.data_ptr(), _lib.ptr(ws), nbytes, loss.data_ptr(),
                                      loss.data_ptr() + 4 * B, _lib.ptr(grad), _lib.stream_of(x))
        if t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream(dev))
            self._prof.setdefault("mri_loss_grad_mc", []).append((t0, t1))
        _lib.check(rc, "mri_loss_grad_mc_fused")
        self._loss_total = loss[B]
        return loss[:B], grad

    def _launch_loss_grad(self, x, y, need_grad, chain_scale):
        B, T = self._dims(x)
        y = _f32c(y, x.device)
        want = (T, self.coils, 2, self.resolution, self.resolution)
        if y.dim() != 6 or tuple(y.shape[1:]) != want or y.shape[0] not in (1, B):
            raise ValueError("measurement shape %s, expected [1 or %d, %s]" % (tuple(y.shape), B, want))
        # one kernel per frame for all coils pays once the frames alone fill most of the chip; below
        # that (scripts/kbench_mri_coils.py: 4 chains x 8 coils 0.34 ms fused, 0.14 ms split) the split
        # path's B x T x C single-coil frames spread over the idle CUs
        if int(self.resolution) != 192:                  # coil multiply / any-size kernels / coil combine
            if B * T > 65535:
                raise ValueError("mri_multicoil at resolution %d: at most 65535 frames per call" % self.resolution)
            return self._launch_split(x, y, need_grad, chain_scale)
        fused = (self.split_coils is None and self._max_rows <= 64 and (self.coils == 1 or B * T >= 144)) \
            if self.fused_coils is None else self.fused_coils
        if fused:
            return self._launch_fused(x, y, need_grad, chain_scale)
        split = self.coils >= 3 if self.split_coils is None else self.split_coils
        if split and B * T <= 65535:
            return self._launch_split(x, y, need_grad, chain_scale)
        rows, nrows = self._tables(x.device)
        part = torch.empty(B * T, device=x.device, dtype=torch.float32)
        loss = torch.empty(B + 1, device=x.device, dtype=torch.float32)
        grad = torch.empty_like(x) if need_grad else None
        t0 = None
        if self._prof is not None:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(torch.cuda.current_stream(x.device))
        rc = _lib.lib().mri_loss_grad_mc(x.data_ptr(), self._sens_on(x.device).data_ptr(), self.coils,
                                         y.data_ptr(), y.shape[0], B, T, self.resolution, rows.data_ptr(),
                                         nrows.data_ptr(), self.mask.shape[1], self._max_rows,
                                         _lib.ptr(chain_scale), part.data_ptr(), loss.data_ptr(),
                                         loss.data_ptr() + 4 * B, _lib.ptr(grad), _lib.stream_of(x))
        if t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream(x.device))
            self._prof.setdefault("mri_loss_grad_mc", []).append((t0, t1))
        _lib.check(rc, "mri_loss_grad_mc")
        self._loss_total = loss[B]
        return loss[:B], grad

    def __call__(self, x):
        return _ForwardMCFn.apply(x, self)
