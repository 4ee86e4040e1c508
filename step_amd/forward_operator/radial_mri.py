"""Dynamic MRI with NON-UNIFORM (radial) k-space sampling on the gridding kernels
(step_amd/csrc/nufft_grid.hip).  NO REFERENCE COUNTERPART: the reference's MRI operator samples
Cartesian rows (forward_operator/dynamic_mri.py:42-48) and `pynfft` appears only as an install
line (README.md:28); BASELINE.json's north_star names the non-uniform path as a gridding kernel.
Parity is therefore UNPINNED; the oracle is the direct non-uniform DFT in fp64
(oracle/nufft_oracle.py), which the operator matches to the accuracy of the Kaiser-Bessel design
(width 6, oversampling 2: ~1e-5 relative, asserted at 1e-4 in tests/test_gpu_nufft.py).

    y[b, t, j] = sum_n (x_re + i x_im)[b, t, n] exp(-2 pi i kappa[t, j] . (n - R/2) / R)

with kappa in cycles per field of view.  Same operator interface as the other tasks
(``__call__``, ``loss``, ``gradient``, autograd through ``loss``); registry name ``'mri_radial'``.
The default trajectory is golden-angle radial: ``spokes`` spokes of ``R`` read-out points per
frame, the angle advancing by pi (sqrt(5) - 1) / 2 from spoke to spoke across the frames.
Hand-written: the interpolation (+ residual + loss; grid tiles through LDS), the spread (a
deterministic gather over host-binned tiles; from 16 chains on with a lane per chain) and, for
R = 192, the oversampled-grid transforms themselves (nufft_fft.hip: 384-point passes fused with
pad / crop / de-apodisation and pruned to the rows that are not zero), the forward column pass
interpolating straight from its LDS strip (``fwd_interp = 'grid'`` forces the grid route).  Other sizes (grids up to 1024): pad and crop kernels
around the library's own any-size FFT2 (nufft_any.hip, mixed-radix passes over LDS); ``grid_fft = 'library'`` puts
hipFFT (``torch.fft`` on the same stream) there instead."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from .base import Operator, register_operator

__all__ = ["RadialMRI", "golden_angle_radial", "kb_plan"]

_TW, _TH = 16, 4                    # tile of a wave of the spread kernel (nufft_grid.hip)
_OWN_TW, _OWN_TH = 64, 32           # tile of the interpolation kernel (NI_TW, NI_TH in nufft_grid.hip)
_WIDE_TW, _WIDE_TH, _WIDE_PAD = 16, 2, 4     # tile and list padding of the lane-per-chain spread kernel (NW_*)
_STRIP = 16                         # columns per strip of the fused column pass + interpolation (NF_SC in nufft_fft.hip)


def golden_angle_radial(frames, spokes, resolution):
    """kappa [frames, spokes * resolution, 2] = (kx, ky) in cycles per field of view."""
    rho = np.arange(resolution, dtype=np.float64) - resolution / 2
    s = np.arange(frames * spokes, dtype=np.float64).reshape(frames, spokes)
    th = s * (math.pi * (math.sqrt(5.0) - 1.0) / 2.0)
    kx = rho[None, None, :] * np.cos(th)[..., None]
    ky = rho[None, None, :] * np.sin(th)[..., None]
    return np.stack([kx, ky], -1).reshape(frames, spokes * resolution, 2)


def _kb(d, W, beta):
    a = 1.0 - (2.0 * d / W) ** 2
    return np.where(a >= 0, np.i0(beta * np.sqrt(np.maximum(a, 0.0))), 0.0) / np.i0(beta)


def _wide_lists(i0, w32, G, W):
    """Lists of the spread kernel that gives every chain a lane (k_nufft_spread_wide): per
    trajectory frame the samples touching each 16 x 2 grid tile, ascending, every list padded to a
    multiple of 4 with zero-weight entries; per entry the fp32 product wy * wx at each of the tile's
    32 points (zero outside the sample's footprint); per trajectory frame the records (tile, first
    entry, entries, outside, ids of the first 8 entries, 0 x 4) in the order in which the kernel's waves take them."""
    Tt, J = i0.shape[:2]
    ntx, ntiles = G // _WIDE_TW, (G // _WIDE_TW) * (G // _WIDE_TH)
    k = np.arange(W)
    recs, ents, wts, ranges, base = [], [], [], [], 0
    for t in range(Tt):
        x0, y0 = i0[t, :, 0] % G, i0[t, :, 1] % G
        tx = ((x0[:, None] + k) % G) // _WIDE_TW                   # [J, W] tile column of every footprint column
        ty = ((y0[:, None] + k) % G) // _WIDE_TH
        key = (ty[:, :, None] * ntx + tx[:, None, :]).astype(np.int64) * J + np.arange(J)[:, None, None]
        key = np.unique(key.ravel())                                # by tile, then by sample id
        tile, j = key // J, key % J
        cnt = np.bincount(tile, minlength=ntiles)
        padded = cnt + (-cnt) % _WIDE_PAD
        ptr = np.zeros(ntiles + 1, np.int64)
        ptr[1:] = np.cumsum(padded)
        pos = ptr[tile] + np.arange(len(key)) - (np.cumsum(cnt) - cnt)[tile]
        ent = np.zeros(ptr[-1], np.int32)
        wt = np.zeros((ptr[-1], _WIDE_TH, _WIDE_TW), np.float32)
        ent[pos] = j
        dy = ((tile // ntx)[:, None] * _WIDE_TH + np.arange(_WIDE_TH) - y0[j][:, None]) % G      # [E, 2]
        dx = ((tile % ntx)[:, None] * _WIDE_TW + np.arange(_WIDE_TW) - x0[j][:, None]) % G       # [E, 16]
        wy = np.where(dy < W, np.take_along_axis(w32[t, j, 1], np.minimum(dy, W - 1), axis=1), np.float32(0))
        wx = np.where(dx < W, np.take_along_axis(w32[t, j, 0], np.minimum(dx, W - 1), axis=1), np.float32(0))
        wt[pos] = wy[:, :, None] * wx[:, None, :]                   # fp32 x fp32 -> fp32, as the other kernel rounds it
        # per tile column the bounds on the centred tile-row coordinate outside which every list is empty
        # (nufft_adjoint neither writes nor reads those tiles)
        nty = ntiles // ntx
        cr = np.arange(nty)
        cr = np.where(cr < nty // 2, cr, cr - nty)
        full = cnt.reshape(nty, ntx) > 0
        lo = np.where(full.any(0), np.where(full, cr[:, None], nty).min(0), 1)
        hi = np.where(full.any(0), np.where(full, cr[:, None], -nty).max(0), 0)
        outside = ((cr[:, None] < lo[None, :]) | (cr[:, None] > hi[None, :])).reshape(-1)
        ranges.append(np.stack([lo, hi], axis=1).astype(np.int32))
        # launch order: longest list first (the four tiles around k = 0 hold ~200 entries against a median of
        # 6; pairing long with short lists as neighbours measured slower), the tiles outside the bounds last
        inside = np.flatnonzero(~outside)
        order = np.concatenate([inside[np.argsort(-cnt[inside], kind="stable")], np.flatnonzero(outside)])
        rec = np.zeros((ntiles, 16), np.int32)
        rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = order, base + ptr[:-1][order], padded[order], outside[order]
        first = ptr[:-1][order][:, None] + np.arange(8)                     # a list's first 8 entries (0 beyond its end)
        rec[:, 4:12] = np.where(np.arange(8) < padded[order][:, None], ent[np.minimum(first, len(ent) - 1)], 0)
        recs.append(rec)
        ents.append(ent)
        wts.append(wt)
        base += len(ent)
    return dict(wide_rec=np.stack(recs), wide_ent=np.concatenate(ents), wide_w=np.concatenate(wts),
                wide_range=np.ascontiguousarray(np.stack(ranges)))


def kb_plan(kappa, R, W=6):
    """Host tables of the gridding plan for a trajectory ``kappa`` [Tt, J, 2] (fp64 arithmetic):
    first grid index and W separable Kaiser-Bessel weights per sample and axis on the G = 2 R
    grid (beta from Beatty et al. 2005 for oversampling 2), the samples binned by 16 x 4 grid tile,
    and the de-apodisation 1 / phi_hat at the pixel positions."""
    kappa = np.asarray(kappa, dtype=np.float64)
    Tt, J, _ = kappa.shape
    G, sigma = 2 * R, 2.0
    beta = math.pi * math.sqrt((W / sigma) ** 2 * (sigma - 0.5) ** 2 - 0.8)
    if np.abs(kappa).max() > R / 2:
        raise ValueError("sample frequencies must lie in [-R/2, R/2] cycles per field of view")
    u = sigma * kappa                                               # grid coordinates
    i0 = np.ceil(u - W / 2.0).astype(np.int64)                      # [Tt, J, 2]
    off = i0[..., None] + np.arange(W) - u[..., None]               # [Tt, J, 2, W] in [-W/2, W/2)
    w = _kb(off, W, beta)
    if G % _TW:
        raise ValueError("the oversampled grid G = %d must be a multiple of %d" % (G, _TW))
    ntx, nty = G // _TW, G // _TH
    ptrs, ents, base = [], [], [0]
    for t in range(Tt):
        lists = [[] for _ in range(ntx * nty)]
        for j in range(J):
            xs = sorted({((i0[t, j, 0] + k) % G) // _TW for k in range(W)})
            ys = sorted({((i0[t, j, 1] + k) % G) // _TH for k in range(W)})
            for ty in ys:
                for tx in xs:
                    lists[ty * ntx + tx].append(j)
        ptr = np.zeros(ntx * nty + 1, np.int32)
        ptr[1:] = np.cumsum([len(l) for l in lists])
        ptrs.append(ptr)
        ents.append(np.array([j for l in lists for j in l], dtype=np.int32))
        base.append(base[-1] + len(ents[-1]))
    # every sample once, by the 64 x 32 tile in which its footprint starts (the interpolation kernel
    # stages that tile and the W - 1 points beyond it in LDS); needs whole tiles
    own_ptr = own_ent = own_x0 = own_y0 = own_wx = own_wy = None
    if G % _OWN_TW == 0 and G % _OWN_TH == 0:
        nox, no = G // _OWN_TW, (G // _OWN_TW) * (G // _OWN_TH)
        key = ((i0[..., 1] % G) // _OWN_TH) * nox + (i0[..., 0] % G) // _OWN_TW          # [Tt, J]
        own_ent = np.argsort(key, axis=1, kind="stable").astype(np.int32)
        own_ptr = np.zeros((Tt, no + 1), np.int32)
        for t in range(Tt):
            own_ptr[t, 1:] = np.cumsum(np.bincount(key[t], minlength=no))
        # the tables once more in bin order: the kernel reads a tile's samples from consecutive memory
        take = lambda a: np.take_along_axis(a, own_ent.astype(np.int64).reshape(own_ent.shape + (1,) * (a.ndim - 2)), axis=1)
        own_x0, own_y0 = take((i0[..., 0] % G).astype(np.int32)), take((i0[..., 1] % G).astype(np.int32))
        own_wx, own_wy = take(w[:, :, 0].astype(np.float32)), take(w[:, :, 1].astype(np.float32))
    # every sample once more, by the strip of 16 grid columns in which its footprint starts (the fused column
    # pass + interpolation of nufft_fft.hip keeps that strip and the W - 1 columns beyond it in LDS)
    strips = dict(strip_ptr=None, strip_ent=None, strip_x0=None, strip_y0=None, strip_wx=None, strip_wy=None)
    if G % _STRIP == 0:
        ns = G // _STRIP
        skey = (i0[..., 0] % G) // _STRIP                                                  # [Tt, J]
        sent = np.argsort(skey, axis=1, kind="stable").astype(np.int32)
        sptr = np.zeros((Tt, ns + 1), np.int32)
        for t in range(Tt):
            sptr[t, 1:] = np.cumsum(np.bincount(skey[t], minlength=ns))
        stake = lambda a: np.take_along_axis(a, sent.astype(np.int64).reshape(sent.shape + (1,) * (a.ndim - 2)), axis=1)
        strips = dict(strip_ptr=sptr, strip_ent=sent,
                      strip_x0=stake((i0[..., 0] % G).astype(np.int32)), strip_y0=stake((i0[..., 1] % G).astype(np.int32)),
                      strip_wx=stake(w[:, :, 0].astype(np.float32)), strip_wy=stake(w[:, :, 1].astype(np.float32)))
    # continuous transform of the kernel at nu = (n - R/2) / G cycles per grid sample
    d = np.linspace(-W / 2.0, W / 2.0, 4001)
    nu = (np.arange(R) - R / 2) / G
    phat = np.trapezoid(_kb(d, W, beta)[None, :] * np.cos(2 * math.pi * nu[:, None] * d[None, :]), d, axis=1)
    wide = _wide_lists(i0, w.astype(np.float32), G, W) if G % _WIDE_TW == 0 else \
        dict(wide_rec=None, wide_ent=None, wide_w=None, wide_range=None)
    return dict(**wide, **strips, R=R, G=G, W=W, J=J, traj_frames=Tt,
                ix0=i0[..., 0].astype(np.int32), iy0=i0[..., 1].astype(np.int32),
                wx=w[:, :, 0].astype(np.float32), wy=w[:, :, 1].astype(np.float32),
                tile_ptr=np.stack(ptrs), tile_ent=np.concatenate(ents), tile_base=np.array(base[:-1], np.int64),
                apod=(1.0 / phat).astype(np.float32), own_ptr=own_ptr, own_ent=own_ent,
                own_x0=own_x0, own_y0=own_y0, own_wx=own_wx, own_wy=own_wy)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, y, unit_upstream):
        need = ctx.needs_input_grad[0]
        loss, _, grad = op._loss_grad(x.detach(), y, need, want_total=False)
        ctx.dtype, ctx.unit = x.dtype, bool(unit_upstream)
        if need:
            ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        if not ctx.unit:
            grad = grad * grad_loss.to(grad.dtype).view(-1, 1, 1, 1, 1)
        return grad.to(ctx.dtype), None, None, None


class _ForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        ctx.op, ctx.dtype = op, x.dtype
        return op._type2(x.detach())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return ctx.op._type1(g.detach(), 1.0).to(ctx.dtype), None


@register_operator('mri_radial')
class RadialMRI(Operator):
    def __init__(self, spokes=32, resolution=192, frames=12, sigma=0, kernel_width=6, trajectory=None,
                 max_chunk=64):
        super().__init__(sigma)
        self.resolution, self.frames, self.spokes = resolution, frames, spokes
        kappa = golden_angle_radial(frames, spokes, resolution) if trajectory is None \
            else np.asarray(trajectory, dtype=np.float64)
        if kappa.ndim == 2:
            kappa = kappa[None]
        if kappa.shape[0] not in (1, frames) or kappa.shape[-1] != 2:
            raise ValueError("trajectory must be [J, 2] or [1 or frames, J, 2], got %s" % (kappa.shape,))
        self.trajectory = kappa
        self._host = kb_plan(kappa, resolution, kernel_width)
        self.J = self._host["J"]
        self.max_chunk = max_chunk        # chains per pass: the oversampled grids are 8 G^2 bytes per frame
        self.grid_fft = 'auto'            # 'auto' | 'fused' (nufft_fft.hip, R = 192) | 'generic' (nufft_any.hip) | 'library' (torch.fft)
        self.adjoint = 'auto'             # 'auto' | 'full' (nufft_spread + nufft_inv_grid over the whole grid)
        self.fwd_interp = 'auto'          # 'auto' | 'fused' (column pass + interpolation in one kernel, no grid) | 'grid'
        self._dev = {}

    _prof = None

    def _plan(self, device):
        key = str(device)
        if key not in self._dev:
            h = self._host
            names = ("ix0", "iy0", "wx", "wy", "tile_ptr", "tile_ent", "tile_base", "apod", "own_ptr", "own_ent",
                     "own_x0", "own_y0", "own_wx", "own_wy", "wide_rec", "wide_ent", "wide_w",
                     "strip_ptr", "strip_ent", "strip_x0", "strip_y0", "strip_wx", "strip_wy")
            t = {k: torch.from_numpy(np.ascontiguousarray(h[k])).to(device) for k in names if h[k] is not None}
            plan = _lib.nufft_plan_t(h["R"], h["G"], h["W"], h["J"], h["traj_frames"],
                                     *[t[k].data_ptr() if k in t else None for k in names])
            self._dev[key] = (plan, t)
        return self._dev[key][0]

    def _check(self, x):
        R = self.resolution
        if x.dim() != 5 or tuple(x.shape[1:]) != (self.frames, 2, R, R):
            raise ValueError("mri_radial expects [B, %d, 2, %d, %d], got %s" % (self.frames, R, R, tuple(x.shape)))
        return x.to(torch.float32).contiguous()

    def _own_fft(self):
        """The fused, pruned grid transforms (nufft_fft.hip) exist for R = 192; other sizes take
        pad / crop kernels around a library FFT2."""
        return self.grid_fft == 'fused' or (self.grid_fft == 'auto'
                                            and bool(_lib.lib().nufft_grid_fft_supported(self.resolution, 2 * self.resolution)))

    def _any_fft(self):
        """The library's own FFT2 at any grid size up to 1024 (nufft_fft2_any: mixed-radix passes over LDS) for
        the resolutions the fused transforms are not built for; ``grid_fft = 'library'`` forces torch.fft."""
        return (not self._own_fft() and self.grid_fft in ('auto', 'generic')
                and bool(_lib.lib().nufft_fft2_any_supported(2 * self.resolution)))

    def _fused_fwd(self, device):
        """nufft_fwd_interp (nufft_fft.hip): the forward column pass interpolates from its own LDS strip;
        needs the own transforms (R = 192) and the plan's strip bins."""
        if self.fwd_interp == 'grid' or not self._own_fft():
            return 0
        parts = _lib.lib().nufft_fwd_interp_parts(C.byref(self._plan(device)))
        if self.fwd_interp == 'fused' and not parts:
            raise RuntimeError("mri_radial: the fused forward needs R = 192 and the plan's strip bins")
        return parts

    def _samples(self, xc, yc, res_scale, out, part, loss):
        """type 2 of a chunk: samples (yc None) or res_scale * (samples - yc) into ``out``, the per-chain
        loss into ``loss``; -> the oversampled grid when one was made (else None)."""
        n, T, dev = xc.shape[0], self.frames, xc.device
        L, plan, st = _lib.lib(), self._plan(dev), _lib.stream_of(xc)
        ya = (None, 1) if yc is None else (yc.data_ptr(), yc.shape[0])
        if self._fused_fwd(dev):
            work = torch.empty(n, T, self.resolution, 2 * self.resolution, device=dev, dtype=torch.complex64)
            rc = L.nufft_fwd_interp(xc.data_ptr(), n * T, T, C.byref(plan), ya[0], ya[1], res_scale, work.data_ptr(),
                                    out.data_ptr(), _lib.ptr(part), _lib.ptr(loss), None, st)
            _lib.check(rc, "nufft_fwd_interp")
            return None
        Xg = self._grid_fft(xc)
        rc = L.nufft_interp(Xg.data_ptr(), n * T, T, C.byref(plan), ya[0], ya[1], res_scale, out.data_ptr(),
                            _lib.ptr(part), _lib.ptr(loss), None, st)
        _lib.check(rc, "nufft_interp")
        return Xg

    def _parts(self, device):
        return self._fused_fwd(device) or _lib.lib().nufft_interp_parts(C.byref(self._plan(device)))

    def _grid_fft(self, x):
        """pad + de-apodise -> FFT2 on the oversampled grid: [B, T, G, G] complex64."""
        B, T, R, G = x.shape[0], self.frames, self.resolution, 2 * self.resolution
        grid = torch.empty(B, T, G, G, device=x.device, dtype=torch.complex64)
        if self._own_fft():
            work = torch.empty(B, T, R, G, device=x.device, dtype=torch.complex64)
            rc = _lib.lib().nufft_fwd_grid(x.data_ptr(), B * T, T, C.byref(self._plan(x.device)), work.data_ptr(),
                                           grid.data_ptr(), _lib.stream_of(x))
            _lib.check(rc, "nufft_fwd_grid")
            return grid
        rc = _lib.lib().nufft_pad_apod(x.data_ptr(), B * T, T, C.byref(self._plan(x.device)), grid.data_ptr(),
                                       _lib.stream_of(x))
        _lib.check(rc, "nufft_pad_apod")
        if self._any_fft():
            _lib.check(_lib.lib().nufft_fft2_any(grid.data_ptr(), B * T, G, -1, _lib.stream_of(x)), "nufft_fft2_any")
            return grid
        return torch.fft.fft2(grid)

    def _type2(self, x):
        """-> samples [B, T, J, 2] (Re, Im)."""
        x = self._check(x)
        outs = []
        for i in range(0, x.shape[0], self.max_chunk):
            xc = x[i:i + self.max_chunk]
            out = torch.empty(xc.shape[0], self.frames, self.J, 2, device=x.device, dtype=torch.float32)
            self._samples(xc, None, 1.0, out, None, None)
            outs.append(out)
        return torch.cat(outs) if len(outs) > 1 else outs[0]

    def _type1(self, vals, out_scale, chain_scale=None, out=None):
        """samples [B, T, J, 2] -> image [B, T, 2, R, R] (the exact adjoint of _type2); written
        into ``out`` (contiguous, same shape) when given."""
        vals = vals.to(torch.float32).contiguous()
        B, T, R, G = vals.shape[0], self.frames, self.resolution, 2 * self.resolution
        if out is None:
            out = torch.empty(B, T, 2, R, R, device=vals.device, dtype=torch.float32)
        L, plan, st = _lib.lib(), self._plan(vals.device), _lib.stream_of(vals)
        for i in range(0, B, self.max_chunk):
            vc = vals[i:i + self.max_chunk]
            n = vc.shape[0]
            grid = torch.empty(n, T, G, G, device=vals.device, dtype=torch.complex64)
            cs = None if chain_scale is None else chain_scale[i:i + n].contiguous()
            rng = self._host["wide_range"]
            if self.adjoint == 'auto' and rng is not None and self._own_fft() and L.nufft_adjoint_supported(C.byref(plan), n):
                work = torch.empty(n, T, R, G, device=vals.device, dtype=torch.complex64)
                _lib.check(L.nufft_adjoint(vc.data_ptr(), n * T, T, C.byref(plan), rng.ctypes.data, _lib.ptr(cs), out_scale,
                                           grid.data_ptr(), work.data_ptr(), out[i:i + n].data_ptr(), st), "nufft_adjoint")
                continue
            _lib.check(L.nufft_spread(vc.data_ptr(), n * T, T, C.byref(plan), grid.data_ptr(), st), "nufft_spread")
            if self._own_fft():
                work = torch.empty(n, T, R, G, device=vals.device, dtype=torch.complex64)
                _lib.check(L.nufft_inv_grid(grid.data_ptr(), n * T, T, C.byref(plan), _lib.ptr(cs), out_scale,
                                            work.data_ptr(), out[i:i + n].data_ptr(), st), "nufft_inv_grid")
                continue
            if self._any_fft():
                _lib.check(L.nufft_fft2_any(grid.data_ptr(), n * T, G, 1, st), "nufft_fft2_any")
                Z = grid
            else:
                Z = torch.fft.ifft2(grid, norm="forward")           # unnormalised inverse (in place measured slower)
            _lib.check(L.nufft_crop_apod(Z.data_ptr(), n * T, T, C.byref(plan), _lib.ptr(cs), out_scale,
                                         out[i:i + n].data_ptr(), st), "nufft_crop_apod")
        return out

    def _loss_grad(self, x, y, need_grad, want_total=True):
        x = self._check(x)
        B, T, dev = x.shape[0], self.frames, x.device
        y = y.detach().to(device=dev, dtype=torch.float32).contiguous()
        if y.dim() != 4 or tuple(y.shape[1:]) != (T, self.J, 2) or y.shape[0] not in (1, B):
            raise ValueError("measurement shape %s, expected [1 or %d, %d, %d, 2]" % (tuple(y.shape), B, T, self.J))
        nblk = self._parts(dev)
        loss = torch.empty(B, device=dev, dtype=torch.float32)
        grad = torch.empty_like(x) if need_grad else None
        t0 = None
        if self._prof is not None:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(torch.cuda.current_stream(dev))
        for i in range(0, B, self.max_chunk):
            xc = x[i:i + self.max_chunk]
            n = xc.shape[0]
            res = torch.empty(n, T, self.J, 2, device=dev, dtype=torch.float32)
            part = torch.empty(n * T * nblk, device=dev, dtype=torch.float32)
            yc = y if y.shape[0] == 1 else y[i:i + n]
            Xg = self._samples(xc, yc, 2.0, res, part, loss[i:i + n])
            if need_grad:
                del Xg
                self._type1(res, 1.0, out=grad[i:i + n])
        if t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream(dev))
            self._prof.setdefault("nufft_loss_grad", []).append((t0, t1))
        return loss, (loss.sum() if want_total else None), grad

    # ------------------------------------------------------------------ operator API
    def __call__(self, x):
        return _ForwardFn.apply(x, self)

    def loss(self, x, y, unit_upstream=False):
        return _LossFn.apply(x, self, y, unit_upstream)

    def gradient(self, x, y, return_loss=False):
        loss, total, grad = self._loss_grad(x, y, True, want_total=return_loss)
        grad = grad.to(x.dtype)
        return (grad, total) if return_loss else grad
