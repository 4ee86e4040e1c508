"""Black-hole interferometric imaging operator on MI355X kernels, behind the reference's
interface (reference forward_operator/blackhole.py): registry name ``'blackhole'``, the same
constructor kwargs, ``__call__`` / ``loss`` / ``gradient`` / ``evaluate_chisq`` and the
``BlackHoleObservation`` measurement container.

Three kernel paths compute the same contraction vis = A_vis x:
  * ``sep3``  -- A_vis kept as its rank-1 factors (what ehtim's ftmatrix produces), never
                 materialised; fp32 operands split exactly into three bf16 planes for the bf16
                 matrix cores (HBM-bound).  Default for square 256 / 128 / 64 images;
  * ``sep``   -- the same factors on the fp32 matrix cores (any H, W multiple of 32 up to 512);
  * ``dense`` -- an arbitrary complex A_vis tensor (``from_tensors``), packed once for MFMA.
The set-up that the reference delegates to ehtim (array file -> uv, sigma, closure sets) is
done by ``step_amd.eht_lite`` (PARITY UNPINNED, see there); pass ``obs_table=`` to supply it.

``cphase_form`` names how the closure-phase chi^2 evaluates ``1 - cos(d)`` (blackhole.py:472-474):
``'stable'`` (default) as ``2 sin^2(d/2)``, which keeps fp32 accuracy when a chain already fits the
data (the reference's own fp32 value is rounding noise there, ~1e-2 relative once chi^2 << 1);
``'reference'`` evaluates the reference's literal ``1 - cos(d)`` in fp32.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib, eht_lite
from .base import Operator, register_operator

__all__ = ["BlackHoleObservation", "BlackHoleImagingTorch", "save_observation", "load_observation"]

_NEG_INF = float("-inf")
# how chi2_cphase evaluates 1 - cos(d) (bh_weights_t::cphase_form in include/step_amd.h):
# 'stable' = 2 sin^2(d/2), 'reference' = the reference's literal fp32 1 - cos(d)
_CPHASE_FORMS = {"stable": 0, "reference": 1}


class BlackHoleObservation(object):
    """Nine-field measurement (reference blackhole.py:23-132); ``to_list`` order is
    (vis, sigma_vis, amp, sigma_amp, cphase, sigma_cphase, logcamp, sigma_logcamp, flux)."""

    FIELDS = ("vis", "sigma_vis", "amp", "sigma_amp", "cphase", "sigma_cphase", "logcamp",
              "sigma_logcamp", "flux")

    def __init__(self, vis=None, sigma_vis=None, amp=None, sigma_amp=None, cphase=None,
                 sigma_cphase=None, logcamp=None, sigma_logcamp=None, flux=None):
        self.vis, self.sigma_vis = vis, sigma_vis
        self.amp, self.sigma_amp = amp, sigma_amp
        self.cphase, self.sigma_cphase = cphase, sigma_cphase
        self.logcamp, self.sigma_logcamp = logcamp, sigma_logcamp
        self.flux = flux

    def _map(self, fn):
        return BlackHoleObservation(**{k: (None if getattr(self, k) is None else fn(getattr(self, k)))
                                       for k in self.FIELDS})

    def __len__(self):
        return self.vis.shape[0]

    def __getitem__(self, idx):
        return self._map(lambda t: t[idx])

    def to(self, device):
        return self._map(lambda t: t.to(device))

    def detach(self):
        for k in self.FIELDS:
            if getattr(self, k) is not None:
                setattr(self, k, getattr(self, k).detach())

    def to_dict(self):
        return {k: getattr(self, k) for k in self.FIELDS if getattr(self, k) is not None}

    def to_list(self):
        return tuple(getattr(self, k) for k in self.FIELDS)

    def from_dict(self, results):
        for k in self.FIELDS:
            if k in results:
                setattr(self, k, results[k])
        return self

    @classmethod
    def merge(cls, obs_list):
        return cls(**{k: torch.cat([getattr(o, k) for o in obs_list], dim=0) for k in cls.FIELDS})


class _RefObservation(BlackHoleObservation):
    """The same container under the reference's pickle identity
    (``forward_operator.blackhole.BlackHoleObservation``, what ``sample.py:62`` writes to
    ``observations.pth``): instance state is the nine tensors under the reference's attribute names."""


_RefObservation.__module__ = "forward_operator.blackhole"
_RefObservation.__qualname__ = _RefObservation.__name__ = "BlackHoleObservation"


def save_observation(obs, path):
    """Write ``obs`` the way the reference does (``torch.save(observations.to('cpu'), ...)``,
    sample.py:62) so that the reference's ``torch.load`` returns its own class."""
    import sys
    import types
    name = "forward_operator.blackhole"
    ref_mod = sys.modules.get(name)
    if ref_mod is not None and hasattr(ref_mod, "BlackHoleObservation"):
        torch.save(ref_mod.BlackHoleObservation(*obs.to("cpu").to_list()), path)   # inside the reference
        return
    added = []
    try:
        for m in ("forward_operator", name):
            if m not in sys.modules:
                sys.modules[m] = types.ModuleType(m)
                added.append(m)
        sys.modules[name].BlackHoleObservation = _RefObservation
        torch.save(_RefObservation(*obs.to("cpu").to_list()), path)
    finally:
        for m in added:
            sys.modules.pop(m, None)


def load_observation(path, map_location="cpu"):
    """Read an ``observations.pth`` written by the reference or by ``save_observation`` with
    torch's weights-only unpickler: the only non-tensor global it may resolve is the
    observation container, so nothing from the file is executed."""
    with torch.serialization.safe_globals([_RefObservation]):
        o = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(o, BlackHoleObservation):
        raise TypeError("%s does not hold a BlackHoleObservation (got %s)" % (path, type(o).__name__))
    return BlackHoleObservation(*o.to_list())


def factor_A_vis(A, H, W, tol=1e-6):
    """Rank-1 factors of a dense DFT matrix: ``A[t,f].reshape(H,W) = outer(ey[t,f], ex[t,f])``.

    ehtim's ftmatrix rows are separable (pulse * e^{2 pi i y v} * e^{2 pi i x u}),
    but the reference only keeps the product, rounded to complex64.  Per row: pivot on the largest
    entry, ``ex = A[yc, :]``, ``ey = A[:, xc] / A[yc, xc]`` (fp64), and accept only if the outer
    product reproduces every entry to ``tol`` of the row's largest magnitude (a few fp32 ulps: the
    entries were rounded separately).  The free complex constant between the two factors is then
    chosen so that ``ey[H-1-y] = conj(ey[y])`` when the grid allows it (centred pixel grid), and the
    remaining asymmetry -- rounding only, below ``tol`` -- is averaged out, which is what lets the
    folded kernels run.  Returns ``(ey [T,F,H], ex [T,F,W])`` complex128 or None."""
    T, F, HW = A.shape
    ey = np.empty((T, F, H), np.complex128)
    ex = np.empty((T, F, W), np.complex128)
    for t in range(T):
        At = A[t].detach().cpu().numpy().astype(np.complex128).reshape(F, H, W)
        for f in range(F):
            M = At[f]
            peak = np.abs(M).max()
            if not np.isfinite(peak) or peak == 0.0:
                return None
            yc, xc = np.unravel_index(np.argmax(np.abs(M)), M.shape)
            fx = M[yc, :].copy()
            fy = M[:, xc] / M[yc, xc]
            if np.abs(np.outer(fy, fx) - M).max() > tol * peak:
                return None
            # ey -> c ey, ex -> ex / c with c^2 ey[y] ey[H-1-y] real positive
            prod = (fy * fy[::-1]).sum()
            if np.abs(prod) > 0:
                c = np.exp(-0.5j * np.angle(prod))
                gy, gx = fy * c, fx / c
                asym = np.abs(gy[::-1] - np.conj(gy)).max()
                if asym <= tol * np.abs(gy).max():
                    fy, fx = 0.5 * (gy + np.conj(gy[::-1])), gx
            ey[t, f], ex[t, f] = fy, fx
    return ey, ex


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class _LossFn(torch.autograd.Function):
    """loss[b] with the fused adjoint as its backward (Appendix B of SURVEY.md).

    ``unit_upstream``: the caller promises that the gradient arriving at the loss is exactly one
    per chain (the loss is only summed: LatentWrapper.gradient).  The forward then runs the whole
    fused step and the backward hands the stored gradient on; otherwise the backward is the adjoint
    kernel scaled per chain by the upstream gradient."""

    @staticmethod
    def forward(ctx, x, op, y, unit_upstream):
        need = ctx.needs_input_grad[0]
        ctx.op, ctx.shape, ctx.dtype = op, x.shape, x.dtype
        ctx.whole = bool(need and unit_upstream and op._fused_ok(y, x.shape[0]))
        if ctx.whole:
            loss, _, grad = op._fused_step(x.detach(), y, want_total=False)
            ctx.save_for_backward(grad)
            return loss
        loss, gvis, gflux = op._loss_forward(x.detach(), y, need)
        if need:
            ctx.save_for_backward(gvis, gflux)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        if ctx.whole:
            (grad,) = ctx.saved_tensors
            return grad.view(ctx.shape).to(ctx.dtype), None, None, None
        gvis, gflux = ctx.saved_tensors
        g = ctx.op._adjoint(gvis, gflux, _f32(grad_loss, gvis.device), 0.5)
        return g.view(ctx.shape).to(ctx.dtype), None, None, None


class _VisFn(torch.autograd.Function):
    """forward_vis (x already in [0,1]) with the DFT adjoint as its backward."""

    @staticmethod
    def forward(ctx, x, op):
        vis, _ = op._dft(x.detach(), 1.0, 0.0, _NEG_INF)
        ctx.op, ctx.shape, ctx.dtype = op, x.shape, x.dtype
        return vis

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gvis):
        op = ctx.op
        gv = torch.view_as_real(gvis.detach().to(torch.complex64).contiguous())
        gflux = torch.zeros(gv.shape[0], op.T, device=gv.device, dtype=torch.float32)
        g = op._adjoint(gv, gflux, None, 1.0)
        return g.view(ctx.shape).to(ctx.dtype), None


@register_operator('blackhole')
class BlackHoleImagingTorch(Operator):
    def __init__(self, array="configs/task/EHT2017.txt", imsize=256, w_vis=0, w_amp=0,
                 w_cphase=1, w_logcamp=1, w_flux=0.5, loss_normalize=True, num_frames=64,
                 ref_multiplier=1.0, device='cuda', obs_table=None, force_dense=False,
                 dft='auto', cphase_form='stable'):
        super().__init__(0)
        assert num_frames == 64, "num_frames should be 64"
        table = obs_table if obs_table is not None else eht_lite.eht_table(array)
        self.ref_multiplier = ref_multiplier
        self.device = torch.device(device)
        self.psize = eht_lite.PSIZE
        uv = np.asarray(table["uv"], dtype=np.float64)
        T, F = uv.shape[0], uv.shape[1]
        assert T == num_frames, "observation table has %d frames, num_frames=%d" % (T, num_frames)
        self.time_list = np.asarray(table["frame_times"])
        sigma = torch.from_numpy(np.asarray(table["sigma"], dtype=np.float64)).float() / ref_multiplier
        cp_i, cp_c, ca_i, ca_c = self._closure_tables(table)
        # station ids of every visibility row (for measure()'s station-based gains) and, when the
        # array has an ALMA-APEX baseline, its row in every frame (estimate_flux, blackhole.py:303-312)
        names = sorted(set(table["t1"]) | set(table["t2"]))
        sid = {n: i for i, n in enumerate(names)}
        self._row_sites = ([sid[s] for s in table["t1"]], [sid[s] for s in table["t2"]], names)
        zs = [i for i, (a, b) in enumerate(zip(table["t1"], table["t2"])) if {a, b} == {"ALMA", "APEX"}]
        self._zero_spacing_rows = zs if len(zs) == T else None
        ey, ex = eht_lite.ft_factors(uv, imsize, imsize, self.psize)
        self._init_common(T, F, imsize, imsize, sigma, cp_i, cp_c, ca_i, ca_c, w_vis, w_amp,
                          w_cphase, w_logcamp, w_flux, loss_normalize, cphase_form)
        sep_ok = (F <= 31 and imsize % 32 == 0 and imsize <= 512)
        assert dft in ("auto", "sep3", "sep", "dense"), dft
        if force_dense:
            dft = "dense"
        if sep_ok and dft != "dense":
            self._init_sep(ey, ex)
            L = _lib.lib()
            sep3_ok = bool(L.bh_sep3_supported(imsize, imsize)) and \
                bool(L.bh_sep3_tables_ok(_lib.ptr(self._ey_ri_host), T, imsize))
            if dft == "sep3" and not sep3_ok:
                raise ValueError("dft='sep3' needs a square 256/128/64 image and ey tables that "
                                 "are conjugate-symmetric about the mid-line (centred pixel grid)")
            if dft in ("auto", "sep3") and sep3_ok:
                self.path = "sep3"
        else:
            A = torch.from_numpy(eht_lite.dense_from_factors(ey, ex)).to(torch.complex64)
            self._init_dense(A.to(self.device))

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_tensors(cls, A_vis, sigma, cp_index, cp_conjugate, camp_index, camp_conjugate,
                     imsize, w_vis=0, w_amp=0, w_cphase=1, w_logcamp=1, w_flux=0.5,
                     loss_normalize=True, device='cuda', dft='auto', cphase_form='stable'):
        """Operator around a dense ``A_vis`` ([1,T,F,H*W] or [T,F,H*W] complex) and the
        reference's index tensors -- the attributes the reference keeps (blackhole.py:155-163).
        ``sigma`` is the per-visibility sigma already divided by ref_multiplier.

        ``dft='auto'`` looks for the rank-1 structure ehtim's ftmatrix gives every row
        (``A[t,f] = ey (x) ex`` as an H x W matrix): if every row factors to fp32 rounding the
        separable kernels are used (``factor_A_vis``), otherwise -- or with ``dft='dense'`` -- the
        matrix is streamed as given."""
        assert dft in ("auto", "sep3", "sep", "dense"), dft
        self = cls.__new__(cls)
        Operator.__init__(self, 0)
        self.device = torch.device(device)
        self.ref_multiplier, self.psize, self.time_list = 1.0, eht_lite.PSIZE, None
        self._row_sites = self._zero_spacing_rows = None
        A = A_vis[0] if A_vis.dim() == 4 else A_vis
        T, F, HW = A.shape
        H, W = (imsize, imsize) if isinstance(imsize, int) else imsize
        assert H * W == HW
        self._init_common(T, F, H, W, sigma.float(), cp_index, cp_conjugate, camp_index,
                          camp_conjugate, w_vis, w_amp, w_cphase, w_logcamp, w_flux, loss_normalize,
                          cphase_form)
        factors = None
        if dft != "dense" and F <= 31 and H % 32 == 0 and W % 32 == 0 and max(H, W) <= 512:
            factors = factor_A_vis(A, H, W)
        if factors is None:
            if dft in ("sep", "sep3"):
                raise ValueError("dft=%r: A_vis is not rank-1 per baseline to fp32 rounding" % dft)
            self._init_dense(A.to(device=self.device, dtype=torch.complex64).contiguous())
            return self
        self._init_sep(*factors)
        self._A_vis = A.to(device=self.device, dtype=torch.complex64).contiguous().unsqueeze(0)
        L = _lib.lib()
        sep3_ok = bool(L.bh_sep3_supported(H, W)) and \
            bool(L.bh_sep3_tables_ok(_lib.ptr(self._ey_ri_host), T, H))
        if dft == "sep3" and not sep3_ok:
            raise ValueError("dft='sep3' needs a square 256/128/64 image and a mirror-symmetric grid")
        if dft in ("auto", "sep3") and sep3_ok:
            self.path = "sep3"
        return self

    @staticmethod
    def _closure_tables(table):
        """Index mapping of blackhole.py:235-264 through the C ABI (bh_build_closure_tables)."""
        names = {}
        ids = lambda seq: np.array([names.setdefault(s, len(names)) for s in seq], dtype=np.int32)
        t1, t2 = ids(table["t1"]), ids(table["t2"])
        rt = np.ascontiguousarray(table["times"], dtype=np.float64)
        tri, quad = table["tri"], table["quad"]
        tt = np.array([t[0] for t in tri], dtype=np.float64)
        ts = ids([s for t in tri for s in t[1:]]).reshape(-1, 3)
        qt = np.array([q[0] for q in quad], dtype=np.float64)
        qs = ids([s for q in quad for s in q[1:]]).reshape(-1, 4)
        cp_i = np.zeros((len(tri), 3), np.int64)
        cp_c = np.zeros_like(cp_i)
        ca_i = np.zeros((len(quad), 4), np.int64)
        ca_c = np.zeros_like(ca_i)
        rc = _lib.lib().bh_build_closure_tables(_lib.ptr(rt), _lib.ptr(t1), _lib.ptr(t2), len(rt),
                                                _lib.ptr(tt), _lib.ptr(ts), len(tri),
                                                _lib.ptr(qt), _lib.ptr(qs), len(quad),
                                                _lib.ptr(cp_i), _lib.ptr(cp_c), _lib.ptr(ca_i),
                                                _lib.ptr(ca_c))
        _lib.check(rc, "bh_build_closure_tables")
        return tuple(torch.from_numpy(a) for a in (cp_i, cp_c, ca_i, ca_c))

    def _init_common(self, T, F, H, W, sigma, cp_i, cp_c, ca_i, ca_c, w_vis, w_amp, w_cphase,
                     w_logcamp, w_flux, loss_normalize, cphase_form='stable'):
        if cphase_form not in _CPHASE_FORMS:
            raise ValueError("cphase_form must be 'stable' or 'reference', got %r" % (cphase_form,))
        self.cphase_form = cphase_form
        dev = self.device
        self.T, self.F, self.H, self.W = T, F, H, W
        self.sigma = sigma.to(dev).contiguous()                           # [T*F]
        self.cp_index = cp_i.long().to(dev)
        self.cp_conjugate = cp_c.long().to(dev)
        self.camp_index = ca_i.long().to(dev)
        self.camp_conjugate = ca_c.long().to(dev)
        # dimensions and weights, blackhole.py:166-180
        self.amp_dim = F * T
        self.cphase_dim = self.cp_index.shape[0]
        self.logcamp_dim = self.camp_index.shape[0]
        self.flux_dim = T
        self.weight_vis = w_vis * self.amp_dim
        self.weight_amp = w_amp * self.amp_dim
        self.weight_cp = w_cphase * self.cphase_dim
        self.weight_camp = w_logcamp * self.logcamp_dim
        self.weight_flux = w_flux * self.flux_dim
        self.loss_normalize = loss_normalize
        # device tables for the closure kernels (int32) + transposed incidence (host-built)
        n_vis = T * F
        cpi = cp_i.cpu().long().contiguous().numpy()
        cpc = cp_c.cpu().long().contiguous().numpy()
        cai = ca_i.cpu().long().contiguous().numpy()
        cac = ca_c.cpu().long().contiguous().numpy()
        cp_ptr = np.zeros(n_vis + 1, np.int32)
        ca_ptr = np.zeros(n_vis + 1, np.int32)
        cp_ent = np.zeros(max(cpi.size, 1), np.int32)
        ca_ent = np.zeros(max(cai.size, 1), np.int32)
        rc = _lib.lib().bh_build_incidence(_lib.ptr(cpi), _lib.ptr(cpc), cpi.shape[0],
                                           _lib.ptr(cai), _lib.ptr(cac), cai.shape[0], n_vis,
                                           _lib.ptr(cp_ptr), _lib.ptr(cp_ent), _lib.ptr(ca_ptr),
                                           _lib.ptr(ca_ent))
        _lib.check(rc, "bh_build_incidence")
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(dev)
        self._tab_tensors = dict(cp_index=i32(cpi), cp_conj=i32(cpc), camp_index=i32(cai),
                                 cp_ptr=i32(cp_ptr), cp_ent=i32(cp_ent), ca_ptr=i32(ca_ptr),
                                 ca_ent=i32(ca_ent))
        tt = self._tab_tensors
        self._tab = _lib.bh_tables_t(n_vis, cpi.shape[0], cai.shape[0], T,
                                     tt["cp_index"].data_ptr(), tt["cp_conj"].data_ptr(),
                                     tt["camp_index"].data_ptr(), tt["cp_ptr"].data_ptr(),
                                     tt["cp_ent"].data_ptr(), tt["ca_ptr"].data_ptr(),
                                     tt["ca_ent"].data_ptr())
        # per-frame closure lists of the fused two-launch step (None: some closure spans frames or
        # a frame has more than 64 of a kind -> the three-kernel path)
        ftab = np.zeros((T, 2, 64, 2), np.int32)
        rc = _lib.lib().bh_build_frame_closures(_lib.ptr(cpi), _lib.ptr(cpc), cpi.shape[0], _lib.ptr(cai),
                                                cai.shape[0], T, F, _lib.ptr(ftab)) if F <= 31 else -1
        self._frame_tab = torch.from_numpy(ftab).to(dev) if rc == 0 else None
        self._meas_cache = None
        self.fused_step = True           # set False to force the three-kernel path (tests, A/B)
        self._weights = _lib.bh_weights_t(float(self.weight_vis), float(self.weight_amp),
                                          float(self.weight_cp), float(self.weight_camp),
                                          float(self.weight_flux), int(bool(loss_normalize)),
                                          _CPHASE_FORMS[cphase_form])

    def _init_sep(self, ey, ex):
        T, F, H, W, dev = self.T, self.F, self.H, self.W, self.device
        self.path = "sep"
        self._ey, self._ex = ey, ex                       # fp64 factors (host), for A_vis on demand
        ey_ri = np.zeros((T, 2, 32, H), np.float32)
        ex_ri = np.zeros((T, 2, 32, W), np.float32)
        ey_ri[:, 0, :F], ey_ri[:, 1, :F] = ey.real, ey.imag
        ex_ri[:, 0, :F], ex_ri[:, 1, :F] = ex.real, ex.imag
        ey_ri[:, 0, F], ex_ri[:, 0, F] = 1.0, 1.0         # flux row
        ey_t = np.ascontiguousarray(ey_ri.reshape(T, 64, H).transpose(0, 2, 1))   # [T][H][64]
        self._ey_t = torch.from_numpy(ey_t).to(dev)
        self._ey_ri_host = ey_ri
        self._ey_ri = torch.from_numpy(ey_ri).to(dev)
        self._ex_ri = torch.from_numpy(ex_ri).to(dev)
        self._A_vis = None

    def _init_dense(self, A):
        T, F, HW = A.shape
        self.path = "dense"
        self._A_vis = A.unsqueeze(0)                       # [1, T, F, HW] like the reference
        L = _lib.lib()
        n = L.bh_dense_packed_floats(T, F, HW)
        self._a_frag = torch.empty(n, device=self.device, dtype=torch.float32)
        self._a_rows = torch.empty(n, device=self.device, dtype=torch.float32)
        rc = L.bh_dense_pack(A.data_ptr(), T, F, HW, self._a_frag.data_ptr(),
                             self._a_rows.data_ptr(), _lib.stream_of(A))
        _lib.check(rc, "bh_dense_pack")

    @property
    def A_vis(self):
        """[1, T, F, H*W] complex64 (materialised on first access on the separable path)."""
        if self._A_vis is None:
            A = eht_lite.dense_from_factors(self._ey, self._ex)
            self._A_vis = torch.from_numpy(A).to(torch.complex64).unsqueeze(0).to(self.device)
        return self._A_vis

    # ------------------------------------------------------------------ kernel launches
    _prof = None      # bench.py sets this to {} to collect (start, end) event pairs per kernel

    def _tick(self):
        if self._prof is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.device))
        return e

    def _tock(self, name, start):
        if start is not None:
            self._prof.setdefault(name, []).append((start, self._tick()))

    def _video(self, x):
        x = x.detach().to(device=self.device, dtype=torch.float32)
        return x.reshape(-1, self.T, self.H * self.W).contiguous()

    def _dft(self, x, scale, shift, clip_min):
        """-> vis [B, T*F] complex64, flux [B, T] of x' = clip(x*scale+shift)."""
        xv = self._video(x)
        B = xv.shape[0]
        vis = torch.empty(B, self.T * self.F, 2, device=self.device, dtype=torch.float32)
        flux = torch.empty(B, self.T, device=self.device, dtype=torch.float32)
        L, st = _lib.lib(), _lib.stream_of(xv)
        t0 = self._tick()
        if self.path in ("sep", "sep3"):
            fn = L.bh_dft_fwd_sep3 if self.path == "sep3" else L.bh_dft_fwd_sep
            rc = fn(self._ey_t.data_ptr(), self._ex_ri.data_ptr(), self.T, self.F,
                    self.H, self.W, xv.data_ptr(), B, scale, shift, clip_min,
                    vis.data_ptr(), flux.data_ptr(), st)
        else:
            HW = self.H * self.W
            ws = torch.empty(L.bh_dense_fwd_workspace_floats(self.T, self.F, HW, B),
                             device=self.device, dtype=torch.float32)
            rc = L.bh_dft_fwd_dense(self._a_frag.data_ptr(), self.T, self.F, HW, xv.data_ptr(), B,
                                    scale, shift, clip_min, vis.data_ptr(), flux.data_ptr(),
                                    ws.data_ptr(), st)
        self._tock("dft_fwd", t0)
        _lib.check(rc, "bh_dft_fwd_" + self.path)
        return torch.view_as_complex(vis), flux

    def _adjoint(self, gvis, gflux, chain_scale, out_scale, sum_of=None, update=None):
        """gvis [B, T*F, 2] fp32, gflux [B, T] -> grad [B, T, H*W] fp32.
        sum_of: a 1-d fp32 tensor whose sum the caller also wants (-> (grad, 0-d sum));
        on the sep3 path it rides along in the adjoint launch.
        update = a _lib.bh_update_t (sep3 only): no gradient is written, the adjoint applies the
        state update in place on update.x_inout and grad comes back None."""
        B = gvis.shape[0]
        L, st = _lib.lib(), _lib.stream_of(gvis)
        cs = None if chain_scale is None else chain_scale.data_ptr()
        total = None
        if update is not None:
            if sum_of is not None:
                total = torch.empty(1, device=self.device, dtype=torch.float32)
            t0 = self._tick()
            rc = L.bh_dft_adj_sep3_update(self._ey_ri.data_ptr(), self._ex_ri.data_ptr(), self.T, self.F,
                                          self.H, self.W, gvis.data_ptr(), gflux.data_ptr(), cs, B,
                                          out_scale, C.byref(update), _lib.ptr(sum_of), 0 if sum_of is None else sum_of.numel(),
                                          _lib.ptr(total), st)
            self._tock("dft_adj", t0)
            _lib.check(rc, "bh_dft_adj_sep3_update")
            return None, (total[0] if total is not None else None)
        grad = torch.empty(B, self.T, self.H * self.W, device=self.device, dtype=torch.float32)
        t0 = self._tick()
        if self.path == "sep3":
            if sum_of is not None:
                total = torch.empty(1, device=self.device, dtype=torch.float32)
            rc = L.bh_dft_adj_sep3(self._ey_ri.data_ptr(), self._ex_ri.data_ptr(), self.T, self.F,
                                   self.H, self.W, gvis.data_ptr(), gflux.data_ptr(), cs, B,
                                   out_scale, grad.data_ptr(), _lib.ptr(sum_of),
                                   0 if sum_of is None else sum_of.numel(), _lib.ptr(total), st)
        elif self.path == "sep":
            rc = L.bh_dft_adj_sep(self._ey_ri.data_ptr(), self._ex_ri.data_ptr(), self.T, self.F,
                                  self.H, self.W, gvis.data_ptr(), gflux.data_ptr(), cs, B,
                                  out_scale, grad.data_ptr(), st)
        else:
            rc = L.bh_dft_adj_dense(self._a_rows.data_ptr(), self.T, self.F, self.H * self.W,
                                    gvis.data_ptr(), gflux.data_ptr(), cs, B, out_scale,
                                    grad.data_ptr(), st)
        self._tock("dft_adj", t0)
        _lib.check(rc, "bh_dft_adj_" + self.path)
        if sum_of is None:
            return grad
        return grad, (total[0] if total is not None else sum_of.sum())

    def _meas(self, y):
        """bh_meas_t for a BlackHoleObservation (+ the tensors that must outlive the launch).
        The struct of the last measurement is kept: a sampler passes the same observation on every
        step, and rebuilding nine tensors per step costs more host time than a small-batch step
        takes on the GPU.  The cache is keyed on the object and on its tensors' storage."""
        key = [id(y)]
        for k in BlackHoleObservation.FIELDS:
            t = getattr(y, k)
            if t is not None:
                key += (t.data_ptr(), t._version, t.shape)
        cache = self._meas_cache
        if cache is not None and cache[0] == key:
            return cache[1], cache[2]
        meas, keep = self._meas_build(y)
        self._meas_cache = (key, meas, keep, y)
        return meas, keep

    def _meas_build(self, y):
        keep, ptrs = [], []
        for k in BlackHoleObservation.FIELDS:
            t = getattr(y, k)
            if t is None:
                ptrs.append(None)
                continue
            if k == "vis":
                t = torch.view_as_real(t.detach().to(self.device, torch.complex64).contiguous())
            else:
                t = _f32(t, self.device)
            keep.append(t)
            ptrs.append(t.data_ptr())
        batch = next(getattr(y, k).shape[0] for k in ("cphase", "logcamp", "flux", "vis", "amp")
                     if getattr(y, k) is not None)
        return _lib.bh_meas_t(*ptrs, batch), keep

    def _loss_forward(self, x, y, need_grad):
        vis, flux = self._dft(x, 0.5, 0.5, _NEG_INF)          # unnormalize fused, blackhole.py:738
        B = vis.shape[0]
        dev = self.device
        loss = torch.empty(B, device=dev, dtype=torch.float32)
        chi2 = torch.empty(B, 5, device=dev, dtype=torch.float32)
        gvis = torch.empty(B, self.T * self.F, 2, device=dev, dtype=torch.float32) if need_grad else None
        gflux = torch.empty(B, self.T, device=dev, dtype=torch.float32) if need_grad else None
        meas, keep = self._meas(y)
        t0 = self._tick()
        rc = _lib.lib().bh_closure_loss(vis.data_ptr(), flux.data_ptr(), B, C.byref(self._tab),
                                        C.byref(meas), C.byref(self._weights), loss.data_ptr(),
                                        chi2.data_ptr(), _lib.ptr(gvis), _lib.ptr(gflux),
                                        _lib.stream_of(vis))
        self._tock("closure", t0)
        _lib.check(rc, "bh_closure_loss")
        self.last_chi2 = chi2        # raw (vis, amp, cphase, logcamp, flux) per chain
        return loss, gvis, gflux

    # The two-launch step saves the closure kernel and a launch gap (5-6 us per step) but its DFT
    # kernels carry the closure work per chain (+0.2 us per chain and launch pair, rocprofv3,
    # profiles/r02_*): it wins below ~32 chains per GPU, the three-kernel path above.
    fused_max_batch = 31
    _last_chi2 = None
    _last_fused = None

    def _fused_ok(self, y, B=None):
        if B is not None and B > self.fused_max_batch:
            return False
        return (self.fused_step and self.path == "sep3" and self._frame_tab is not None
                and self.weight_vis == 0 and self.weight_amp == 0
                and y.flux is not None and y.flux.shape[0] == 1
                and (self.weight_cp == 0 or (y.cphase is not None and y.cphase.shape[0] == 1))
                and (self.weight_camp == 0 or (y.logcamp is not None and y.logcamp.shape[0] == 1)))

    def _fused_step(self, x, y, want_total=True, chain_scale=None, update=None):
        """The whole data step in two launches (bh_loss_grad_sep3): -> loss [B], total (0-d) or
        None, grad [B, T, H*W] = d loss / d x.  update = a _lib.bh_update_t on x: no gradient is
        written (grad comes back None); the adjoint applies the state update to x, which must
        then be the caller's own contiguous fp32 storage."""
        xv = self._video(x)
        B, M, T, dev = xv.shape[0], self.T * self.F, self.T, self.device
        Kcp, Kca = self.cphase_dim, self.logcamp_dim
        # one allocation for every small buffer, S (16-byte aligned) first, addressed by offset: at
        # one chain the step is 26 us on the GPU and every tensor view made here costs about one
        o_vis = 4 * B * T
        o_flux = o_vis + 2 * B * M
        o_dcp = o_flux + B * T
        o_dca = o_dcp + B * Kcp
        o_loss = o_dca + B * Kca
        o_chi2 = o_loss + B
        o_total = o_chi2 + 5 * B
        buf = torch.empty(o_total + 4, device=dev, dtype=torch.float32)
        base = buf.data_ptr()
        grad = None
        if update is None:
            grad = torch.empty(B, T, self.H * self.W, device=dev, dtype=torch.float32)
        elif xv.data_ptr() != x.data_ptr() or update.x_inout != x.data_ptr():
            raise ValueError("in-place update needs a contiguous fp32 state on the operator's device")
        ws = _lib.bh_step_ws_t(base + 4 * o_vis, base + 4 * o_flux, base + 4 * o_dcp, base + 4 * o_dca, base)
        meas, keep = self._meas(y)
        t0 = self._tick()
        rc = _lib.lib().bh_loss_grad_sep3(
            self._ey_t.data_ptr(), self._ey_ri.data_ptr(), self._ex_ri.data_ptr(), T, self.F, self.H,
            self.W, xv.data_ptr(), B, C.byref(self._tab), self._frame_tab.data_ptr(), C.byref(meas),
            C.byref(self._weights), C.byref(ws), _lib.ptr(chain_scale), 0.5, base + 4 * o_loss,
            base + 4 * o_chi2, base + 4 * o_total if want_total else None, _lib.ptr(grad),
            None if update is None else C.byref(update), _lib.stream_of(xv))
        self._tock("fused_step", t0)
        _lib.check(rc, "bh_loss_grad_sep3")
        self._last_chi2 = None
        self._last_fused = (buf, o_vis, o_chi2, B, M)         # last_chi2 / last_vis are made on request
        return buf[o_loss:o_chi2], (buf[o_total] if want_total else None), grad

    @property
    def last_chi2(self):
        """Raw (vis, amp, cphase, logcamp, flux) chi^2 per chain of the last loss evaluation."""
        if self._last_chi2 is None and self._last_fused is not None:
            buf, _, o_chi2, B, _ = self._last_fused
            return buf[o_chi2:o_chi2 + 5 * B].view(B, 5)
        return self._last_chi2

    @last_chi2.setter
    def last_chi2(self, value):
        self._last_chi2 = value
        self._last_fused = None

    @property
    def last_vis(self):
        """Visibilities of the last two-launch step (complex64 [B, T * F])."""
        if self._last_fused is None:
            return None
        buf, o_vis, _, B, M = self._last_fused
        return torch.view_as_complex(buf[o_vis:o_vis + 2 * B * M].view(B, M, 2))

    def _observe(self, vis):
        vis = vis.detach().to(self.device, torch.complex64).contiguous()
        B, M, dev = vis.shape[0], self.T * self.F, self.device
        f = lambda n: torch.empty(B, n, device=dev, dtype=torch.float32)
        amp, samp = f(M), f(M)
        cp, scp = f(self.cphase_dim), f(self.cphase_dim)
        lc, slc = f(self.logcamp_dim), f(self.logcamp_dim)
        rc = _lib.lib().bh_closure_observe(vis.data_ptr(), self.sigma.data_ptr(), B,
                                           C.byref(self._tab), amp.data_ptr(), samp.data_ptr(),
                                           cp.data_ptr(), scp.data_ptr(), lc.data_ptr(),
                                           slc.data_ptr(), _lib.stream_of(vis))
        _lib.check(rc, "bh_closure_observe")
        return amp, samp, cp, scp, lc, slc

    # ------------------------------------------------------------------ reference API
    # 1. visibility and flux from image x in range [0,1]  (blackhole.py:314-329)
    def forward_vis(self, x):
        return _VisFn.apply(x, self)

    def forward_amp(self, x):
        return self.forward_amp_from_vis(self.forward_vis(x))

    def forward_flux(self, x):
        return x.flatten(2).sum(-1)

    # 2. forward from visibilities  (blackhole.py:331-401)
    def correct_vis_direction(self, vis, conj):
        return vis * (1 - conj) + vis.conj() * conj

    def forward_amp_from_vis(self, vis):
        amp = vis.abs()
        return amp, self.sigma + 0 * amp

    def forward_bisepectra_from_vis(self, vis):
        """The three direction-corrected visibilities of every triangle (blackhole.py:341-353;
        the reference's spelling).  Index arithmetic in torch: the kernels gather in place and
        never materialise these; callers of the reference API get them here."""
        legs = (vis[:, self.cp_index[:, k]] for k in range(3))
        return tuple(self.correct_vis_direction(v, self.cp_conjugate[None, :, k]) for k, v in enumerate(legs))

    def forward_logcamp_bispectra_from_vis(self, vis):
        """The four amplitudes of every quadrangle (blackhole.py:365-380; the conjugate of a
        modulus is the modulus, the correction is kept for the dtype it gives the reference)."""
        legs = (vis[:, self.camp_index[:, k]].abs() for k in range(4))
        return tuple(self.correct_vis_direction(v, self.camp_conjugate[None, :, k]) for k, v in enumerate(legs))

    def forward_cp_from_vis(self, vis):
        _, _, cp, scp, _, _ = self._observe(vis)
        return cp, scp

    def forward_logcamp_from_vis(self, vis):
        _, _, _, _, lc, slc = self._observe(vis)
        return lc, slc

    def forward_from_vis(self, x):
        vis, flux = self._dft(x, 1.0, 0.0, _NEG_INF)
        amp, samp, cp, scp, lc, slc = self._observe(vis)
        return BlackHoleObservation(vis, samp, amp, samp, cp, scp, lc, slc, flux)

    # 4. chi-square evaluation  (blackhole.py:447-491)
    def chi2_vis(self, x, y_vis, y_vis_sigma):
        return self.chi2_vis_from_meas(self.forward_vis(x), y_vis, y_vis_sigma)

    @staticmethod
    def chi2_vis_from_meas(y_vis_meas, y_vis, y_vis_sigma):
        return torch.mean(torch.square(torch.abs(y_vis_meas - y_vis) / y_vis_sigma), dim=1) / 2

    def chi2_amp(self, x, y_amp, y_amp_sigma):
        return self.chi2_amp_from_meas(self.forward_amp(x)[0], y_amp, y_amp_sigma)

    @staticmethod
    def chi2_amp_from_meas(y_amp_meas, y_amp, y_amp_sigma):
        return torch.mean(torch.square((y_amp_meas - y_amp) / y_amp_sigma), dim=1)

    def chi2_cphase(self, x, y_cphase, y_cphase_sigma):
        cp, _ = self.forward_cp_from_vis(self.forward_vis(x))
        return self.chi2_cphase_from_meas(cp, y_cphase, y_cphase_sigma)

    @staticmethod
    def chi2_cphase_from_meas(y_cphase_meas, y_cphase, y_cphase_sigma):
        return 2. * torch.mean((1 - torch.cos(y_cphase - y_cphase_meas)) / torch.square(y_cphase_sigma), dim=1)

    def chi2_logcamp(self, x, y_camp, y_logcamp_sigma):
        lc, _ = self.forward_logcamp_from_vis(self.forward_vis(x))
        return self.chi2_logcamp_from_meas(lc, y_camp, y_logcamp_sigma)

    @staticmethod
    def chi2_logcamp_from_meas(y_logcamp_meas, y_logcamp, y_logcamp_sigma):
        return torch.mean(torch.abs((y_logcamp_meas - y_logcamp) / y_logcamp_sigma) ** 2, dim=1)

    def chi2_flux(self, x, y_flux):
        return self.chi2_flux_from_meas(self.forward_flux(x), y_flux)

    @staticmethod
    def chi2_flux_from_meas(y_flux_meas, y_flux):
        return torch.mean(torch.square((y_flux_meas - y_flux) / 2), dim=1)

    # 8. evaluating chi-square  (blackhole.py:664-699)
    @staticmethod
    def normalize_chisq(chisq):
        overfit = chisq < 1.0
        return chisq * (~overfit) + 1 / chisq * overfit

    def evaluate_chisq(self, x, y, normalize=False, chi_sq_list=['vis', 'amp', 'cphase', 'logcamp']):
        # clip(min=0) fused into the DFT load; the per-frame flux alignment is linear in the DFT,
        # so it is applied to the visibilities (vis_scale) instead of re-running it on x * ratio
        vis, x_flux = self._dft(x, 0.5, 0.5, 0.0)
        B = vis.shape[0]
        scale = (_f32(y.flux, self.device) / x_flux).contiguous()
        chi2 = torch.empty(B, 4, device=self.device, dtype=torch.float32)
        meas, keep = self._meas(y)
        rc = _lib.lib().bh_closure_chisq(torch.view_as_real(vis).data_ptr(), scale.data_ptr(), B,
                                         self.F, C.byref(self._tab), C.byref(meas),
                                         self._weights.cphase_form, chi2.data_ptr(),
                                         _lib.stream_of(vis))
        _lib.check(rc, "bh_closure_chisq")
        results = {}
        for i, k in enumerate(('vis', 'amp', 'cphase', 'logcamp')):
            if k in chi_sq_list:
                v = chi2[:, i]
                results[k] = self.normalize_chisq(v) if normalize else v
        return results

    # 7. comparing two observations  (blackhole.py:613-649)
    def cosine_similarity(self, a1, a2):
        """min over the batch of |<a1, a2>|-like similarity: sum |a1 conj(a2)| / (|a1| |a2|)."""
        a1, a2 = a1.flatten(1), a2.flatten(1)
        return ((a1 * a2.conj()).abs().sum(1) / (torch.norm(a1, dim=1) * torch.norm(a2, dim=1))).min().item()

    def compare(self, y1, y2, verbose=False):
        """Field-by-field cosine similarity of two observations; returns the LARGEST of the nine,
        as the reference does (blackhole.py:645-649)."""
        names = ("vis", "sigmavis", "amp", "sigmaamp", "cphase", "sigmacp", "logcamp", "sigmacamp", "flux")
        sims = {n: self.cosine_similarity(a, b) for n, a, b in zip(names, y1.to_list(), y2.to_list())}
        if verbose:
            for n in ("vis", "amp", "cphase", "logcamp", "flux", "sigmavis", "sigmaamp", "sigmacp", "sigmacamp"):
                print("{} similarity: {:.3f} %".format(n, sims[n] * 100))
        return max(sims.values())

    # 9. public interface  (blackhole.py:719-771)
    def unnormalize(self, inputs):
        return inputs * 0.5 + 0.5

    def normalize(self, inputs):
        return inputs * 2 - 1

    def __call__(self, x, **kwargs):
        vis, flux = self._dft(x, 0.5, 0.5, _NEG_INF)
        amp, samp, cp, scp, lc, slc = self._observe(vis)
        return BlackHoleObservation(vis, samp, amp, samp, cp, scp, lc, slc, flux)

    # station calibration statistics the reference hands to ehtim (blackhole.py:515-548, "measured
    # from the 2017 campaign"): std of the constant gain offset and of the per-scan gain fluctuation
    GAIN_OFFSET = {'ALMA': 0.029, 'APEX': 0.028, 'SMT': 0.045, 'JCMT': 0.020, 'LMT': 0.147, 'PV': 0.050,
                   'SMA': 0.019, 'SPT': 0.052, 'SR': 0.0}
    GAINP = {'ALMA': 0.054, 'APEX': 0.045, 'SMT': 0.056, 'JCMT': 0.030, 'LMT': 0.124, 'PV': 0.075,
             'SMA': 0.028, 'SPT': 0.095, 'SR': 0.0}

    def measure(self, x, noise='eht', seed=24, **kwargs):
        """Noisy measurement of a video in [-1, 1] -> BlackHoleObservation [B, ...], the role of
        ``measure`` -> ``measure_eht`` in the reference (blackhole.py:494-586, 732-734, called at
        sample.py:60).  The reference delegates the corruption to ehtim's ``observe_same(jones=True,
        ampcal=False, phasecal=False, add_th_noise=True, seed=24)``; ehtim is not available to this
        build, so the model is RESTATED from its published behaviour -- PARITY UNPINNED:

          'eht'      station-based complex gains g_s(t) = (1 + o_s)(1 + p_s n_s(t)) e^{i phi_s(t)} with
                     o_s ~ N(0, GAIN_OFFSET_s) fixed per station, n_s(t) ~ N(0, 1) and
                     phi_s(t) ~ U(0, 2 pi) per frame (no amplitude / phase calibration), applied as
                     v_ij -> g_i conj(g_j) v_ij, plus complex thermal noise of std sigma per
                     visibility (D-term leakage, second order in Stokes I, is not modelled);
          'thermal'  thermal noise only;   'none'  the noiseless observation ``op(x)``.

        The data products follow the reference's container and scaling: vis, sigma and amp in units
        of 1 / ref_multiplier (sigma is already stored that way), closure phases in radians
        (ehtim's degrees x eh.DEGREE) with sigma_cp and log closure amplitudes with sigma_ca
        propagated from the noisy visibilities' SNR (what ehtim's make_cphase_df / make_camp_df do
        to first order), flux = the amplitude on the ALMA-APEX baseline per frame when the array has
        one (estimate_flux, blackhole.py:303-312), else the image's total flux.  Closure quantities
        are unaffected by the station gains by construction -- the reason the task fits them.
        A measurement written by the reference itself (``observations.pth``, sample.py:62) can be
        used instead: ``load_observation(path)``."""
        if noise not in ('eht', 'thermal', 'none'):
            raise ValueError("noise must be 'eht', 'thermal' or 'none'")
        clean = self(x)
        if noise == 'none':
            return clean
        dev = clean.vis.device
        B, M = clean.vis.shape
        gen = torch.Generator(device='cpu').manual_seed(int(seed))
        vis = clean.vis
        if noise == 'eht':
            if getattr(self, "_row_sites", None) is None:
                raise ValueError("noise='eht' needs the station names of every baseline (an operator "
                                 "built from an array / obs_table, not from_tensors); use 'thermal'")
            t1, t2, names = self._row_sites
            S = len(names)
            # a station without a measured value gets the average of those that have one (the
            # advice in the reference's own comment, blackhole.py:506-510)
            avg = lambda d: sum(v for k, v in d.items() if k != 'SR') / (len(d) - 1)
            go = torch.tensor([self.GAIN_OFFSET.get(n, avg(self.GAIN_OFFSET)) for n in names], dtype=torch.float64)
            gp = torch.tensor([self.GAINP.get(n, avg(self.GAINP)) for n in names], dtype=torch.float64)
            off = 1.0 + go * torch.randn(S, generator=gen, dtype=torch.float64)
            fluct = 1.0 + gp * torch.randn(B, self.T, S, generator=gen, dtype=torch.float64)
            phase = 2 * math.pi * torch.rand(B, self.T, S, generator=gen, dtype=torch.float64)
            gain = (off * fluct) * torch.exp(1j * phase)                       # [B, T, S]
            frame = torch.arange(M) // self.F
            gi = gain[:, frame, torch.as_tensor(t1)]
            gj = gain[:, frame, torch.as_tensor(t2)]
            vis = vis * (gi * gj.conj()).to(torch.complex64).to(dev)
        n = torch.randn(B, M, 2, generator=gen, dtype=torch.float32).to(dev)
        vis = vis + self.sigma * torch.view_as_complex(n)
        amp, samp, cp, scp, lc, slc = self._observe(vis)
        flux = clean.flux
        zs = getattr(self, "_zero_spacing_rows", None)
        if zs is not None:
            flux = amp[:, torch.as_tensor(zs, device=dev)]
        return BlackHoleObservation(vis, samp, amp, samp, cp, scp, lc, slc, flux)

    def loss(self, x, y, unit_upstream=False):
        return _LossFn.apply(x, self, y, unit_upstream)

    def gradient(self, x, y, return_loss=False):
        """Fused loss forward + adjoint (no autograd graph); same result as the base class's
        autograd.grad of loss(x, y).sum()  (base.py:43-49).  Two launches on the default path
        (closure stage inside the DFT kernels), three otherwise."""
        if self._fused_ok(y, x.shape[0]):
            loss, total, grad = self._fused_step(x, y, want_total=return_loss)
            grad = grad.view(x.shape).to(x.dtype)
            return (grad, total) if return_loss else grad
        loss, gvis, gflux = self._loss_forward(x, y, True)
        if return_loss:
            grad, total = self._adjoint(gvis, gflux, None, 0.5, sum_of=loss)
            return grad.view(x.shape).to(x.dtype), total
        return self._adjoint(gvis, gflux, None, 0.5).view(x.shape).to(x.dtype)

    @property
    def has_descent_step(self):
        """descent_step lives in the folded 3xbf16 adjoint: the default path only."""
        return self.path == "sep3"

    def _state_and_anchor(self, x, x0hat, who):
        if self.path != "sep3":
            raise NotImplementedError("%s is built on the sep3 kernels only" % who)
        if (x.dtype != torch.float32 or not x.is_contiguous() or x.device.type != self.device.type
                or (self.device.index is not None and x.device.index != self.device.index)
                or x.requires_grad):
            raise ValueError("%s needs a contiguous fp32 state on %s without autograd history" % (who, self.device))
        x0 = x0hat.detach()
        if x0.dtype != torch.float32 or not x0.is_contiguous() or x0.shape != x.shape or x0.device != x.device:
            x0 = x0.to(device=x.device, dtype=torch.float32).expand_as(x).contiguous()
        if x0.data_ptr() == x.data_ptr():
            raise ValueError("%s: x and x0hat must not share storage" % who)
        return x0

    def _step_with_update(self, x, y, upd, return_loss):
        if self._fused_ok(y, x.shape[0]):
            _, total, _ = self._fused_step(x, y, want_total=return_loss, update=upd)
            return total
        loss, gvis, gflux = self._loss_forward(x, y, True)
        _, total = self._adjoint(gvis, gflux, None, 0.5, sum_of=loss if return_loss else None, update=upd)
        return total

    def descent_step(self, x, y, x0hat, lr, inv_sigma2, return_loss=False):
        """One step of MCMCSampler.optimize (daps.py:210-223) in place on x:
        ``x -= lr * (gradient(x, y) + (x - x0hat) * inv_sigma2)`` with the update applied by the
        adjoint kernel where it would have stored the gradient, which therefore never reaches HBM
        (per step and chain 4 video-sized streams instead of 6 + 2).  x: contiguous fp32 on the
        operator's device, modified; -> the summed loss at the old x (0-d) or None."""
        x0 = self._state_and_anchor(x, x0hat, "descent_step")
        upd = _lib.bh_update_t(x.data_ptr(), x0.data_ptr(), float(lr), float(inv_sigma2))
        return self._step_with_update(x, y, upd, return_loss)

    has_langevin_step = has_descent_step

    def prefers_langevin_step(self, batch):
        """Whether langevin_step beats gradient + mcmc_update_philox at this batch: the adjoint's
        waves draw the noise of their tiles one generator call after the other, which only pays
        once the workgroups fill the chip (measured at T = 64: 2 chains 31 k vs 34 k
        chain-steps/s, 4 chains 54 k vs 45 k, 64 chains 74 k vs 43 k)."""
        return self.has_langevin_step and batch * self.T >= 256

    def langevin_step(self, x, y, x0hat, params, seeds, step=0, step_dev=None, nan_flag=None,
                      return_loss=False, eps=None):
        """One step of MCMCSampler.sample (daps.py:184-200) in place on x for the 'gaussian' prior
        solver, whose score terms add up to ``-g / tau^2 + (x0hat - x) / sigma^2``: langevin
        (params.algo 0) or hmc with momentum 0 (algo 1), the update applied by the adjoint kernel
        where it would have stored the gradient and the noise drawn there (Philox keyed by the
        chain's seed, column-quad layout) -- neither gradient nor noise reaches HBM.  Equals
        ``gradient`` + ``mcmc_update_philox(x, None, g, xt=x0hat, prior=None, ..., quad_w=W)``.
        With ``eps`` (a noise tensor shaped like x, e.g. the reference's ``torch.randn_like``) nothing
        is drawn: the adjoint reads it, ``seeds`` / ``step`` are unused, and the result equals
        ``gradient`` + ``mcmc_update(x, None, g, xt=x0hat, prior=None, eps, ...)``.
        params: _lib.mcmc_params_t; seeds: int64 [B] on the device; step_dev: optional device
        int64 read instead of ``step`` (graph replay); nan_flag: optional device int32."""
        x0 = self._state_and_anchor(x, x0hat, "langevin_step")
        if eps is not None:
            # the noise as a tensor (torch's draw, daps.py:192): read by the adjoint, nothing is generated
            if eps.dtype != torch.float32 or eps.shape != x.shape or eps.device != x.device or not eps.is_contiguous():
                raise ValueError("langevin_step: eps must be contiguous float32 %s on %s" % (tuple(x.shape), x.device))
            upd = _lib.bh_update_t(x.data_ptr(), x0.data_ptr(), 0.0, 0.0, C.pointer(params), None, 0, None,
                                   _lib.ptr(nan_flag), eps.data_ptr())
            return self._step_with_update(x, y, upd, return_loss)
        if seeds.dtype != torch.int64 or seeds.numel() != x.shape[0] or seeds.device != x.device or not seeds.is_contiguous():
            raise ValueError("langevin_step: seeds must be contiguous int64 [%d] on %s" % (x.shape[0], x.device))
        upd = _lib.bh_update_t(x.data_ptr(), x0.data_ptr(), 0.0, 0.0, C.pointer(params), seeds.data_ptr(),
                               int(step), _lib.ptr(step_dev), _lib.ptr(nan_flag))
        return self._step_with_update(x, y, upd, return_loss)


def thermal_measure(op, x, generator=None):
    """Noisy observation WITHOUT ehtim (no reference counterpart for the noise model): complex
    Gaussian thermal noise of std ``op.sigma`` on the visibilities, data products recomputed."""
    clean = op(x)
    n = torch.randn(clean.vis.shape + (2,), device=clean.vis.device, generator=generator)
    vis = clean.vis + op.sigma * torch.view_as_complex(n)
    amp, samp, cp, scp, lc, slc = op._observe(vis)
    return BlackHoleObservation(vis, samp, amp, samp, cp, scp, lc, slc, clean.flux)
