"""Oracle (TEST INFRASTRUCTURE, not product): PyTorch-CPU restatement of the
black-hole interferometry operator arithmetic, reference
``forward_operator/blackhole.py``.  Pinned by ``tests/golden/bh_*.npz``.

Observation field order everywhere: (vis, sigma_vis, amp, sigma_amp, cphase,
sigma_cphase, logcamp, sigma_logcamp, flux) -- blackhole.py:107-108.
"""
from collections import namedtuple

import torch

Obs = namedtuple(
    "Obs", "vis sigma_vis amp sigma_amp cphase sigma_cphase logcamp sigma_logcamp flux")


def to01(x):
    """[-1,1] -> [0,1], blackhole.py:720-722."""
    return x * 0.5 + 0.5


def chisq_fold(chisq):
    """e(chi)=chi if chi>=1 else 1/chi, blackhole.py:665-669 (same bool arithmetic)."""
    under = chisq < 1.0
    return chisq * (~under) + 1 / chisq * under


class BlackholeOracle:
    """State = the tensors the reference keeps as attributes (blackhole.py:155-180)."""

    def __init__(self, A_vis, sigma, cp_index, cp_conjugate, camp_index, camp_conjugate,
                 T, H, W, w_vis=0.0, w_amp=0.0, w_cphase=1.0, w_logcamp=1.0, w_flux=0.5,
                 loss_normalize=True):
        self.A_vis = A_vis            # [1, T, F, H*W] complex64
        self.sigma = sigma            # [T*F] float32 (already / ref_multiplier)
        self.cp_index, self.cp_conjugate = cp_index, cp_conjugate        # int64 [Kcp,3]
        self.camp_index, self.camp_conjugate = camp_index, camp_conjugate  # int64 [Kca,4]
        self.T, self.H, self.W = T, H, W
        F = A_vis.shape[-2]
        # blackhole.py:166-179
        self.w = dict(vis=w_vis * F * T, amp=w_amp * F * T,
                      cp=w_cphase * cp_index.shape[0],
                      camp=w_logcamp * camp_index.shape[0], flux=w_flux * T)
        self.loss_normalize = loss_normalize

    # -- blackhole.py:315-321: broadcast multiply then sum (materialises [B,T,F,HW]) --
    def vis(self, x01):
        xc = x01.to(self.A_vis)
        xv = xc.reshape(-1, self.T, 1, self.H * self.W)
        return (self.A_vis * xv).sum(-1).flatten(1)

    def flux(self, x01):                      # blackhole.py:328-329
        return x01.flatten(2).sum(-1)

    @staticmethod
    def _dir(v, c):                           # blackhole.py:332-334
        return v * (1 - c) + v.conj() * c

    def amp(self, vis):                       # blackhole.py:336-339
        a = vis.abs()
        return a, self.sigma + 0 * a

    def cphase(self, vis):                    # blackhole.py:341-363
        legs = [self._dir(vis[:, self.cp_index[:, j]], self.cp_conjugate[None, :, j])
                for j in range(3)]
        ang = torch.angle(legs[0] * legs[1] * legs[2])
        s = [self.sigma[self.cp_index[:, j]] for j in range(3)]
        sig = (s[0] ** 2 / legs[0].abs() ** 2 + s[1] ** 2 / legs[1].abs() ** 2
               + s[2] ** 2 / legs[2].abs() ** 2).sqrt()
        return ang, sig

    def logcamp(self, vis):                   # blackhole.py:365-391
        a = [self._dir(vis[:, self.camp_index[:, j]].abs(), self.camp_conjugate[None, :, j])
             for j in range(4)]
        val = a[0].log() + a[1].log() - a[2].log() - a[3].log()
        s = [self.sigma[self.camp_index[:, j]] for j in range(4)]
        sig = (s[0] ** 2 / a[0] ** 2 + s[1] ** 2 / a[1] ** 2 + s[2] ** 2 / a[2] ** 2
               + s[3] ** 2 / a[3] ** 2).sqrt()
        return val, sig

    def observe01(self, x01):                 # blackhole.py:393-401
        v = self.vis(x01)
        a, sa = self.amp(v)
        cp, scp = self.cphase(v)
        lc, slc = self.logcamp(v)
        return Obs(v, sa, a, sa, cp, scp, lc, slc, self.flux(x01))

    def __call__(self, x):                    # blackhole.py:728-730
        return self.observe01(to01(x))

    # -- chi^2 terms, blackhole.py:448-491 --
    @staticmethod
    def chi2_vis_meas(v, y, s):
        return torch.mean(torch.square(torch.abs(v - y) / s), dim=1) / 2

    @staticmethod
    def chi2_amp_meas(a, y, s):
        return torch.mean(torch.square((a - y) / s), dim=1)

    @staticmethod
    def chi2_cp_meas(p, y, s):
        return 2. * torch.mean((1 - torch.cos(y - p)) / torch.square(s), dim=1)

    @staticmethod
    def chi2_lc_meas(l, y, s):
        return torch.mean(torch.abs((l - y) / s) ** 2, dim=1)

    @staticmethod
    def chi2_flux_meas(f, y):
        return torch.mean(torch.square((f - y) / 2), dim=1)

    def loss01_terms(self, x01, y):
        """Per-term chi^2 exactly as `loss` evaluates them: one forward_vis per term
        (blackhole.py:448-491; the double DFT at :467 and :477 is kept on purpose)."""
        out = {}
        if self.w['vis'] > 0:
            out['vis'] = self.chi2_vis_meas(self.vis(x01), y.vis, y.sigma_vis)
        if self.w['amp'] > 0:
            out['amp'] = self.chi2_amp_meas(self.vis(x01).abs(), y.amp, y.sigma_amp)
        if self.w['cp'] > 0:
            out['cp'] = self.chi2_cp_meas(self.cphase(self.vis(x01))[0], y.cphase, y.sigma_cphase)
        if self.w['camp'] > 0:
            out['camp'] = self.chi2_lc_meas(self.logcamp(self.vis(x01))[0], y.logcamp,
                                            y.sigma_logcamp)
        if self.w['flux'] > 0:
            out['flux'] = self.chi2_flux_meas(self.flux(x01), y.flux)
        return out

    def loss(self, x, y):                     # blackhole.py:736-771
        terms = self.loss01_terms(to01(x), y)
        total = 0
        for k in ('vis', 'amp', 'cp', 'camp'):
            if k in terms:
                t = chisq_fold(terms[k]) if self.loss_normalize else terms[k]
                total = total + self.w[k] * t
        if 'flux' in terms:
            total = total + self.w['flux'] * terms['flux']
        return total

    def gradient(self, x, y, return_loss=False):   # base.py:43-49
        xt = x.clone().detach().requires_grad_(True)
        l = self.loss(xt, y).sum()
        g = torch.autograd.grad(l, xt)[0]
        return (g, l) if return_loss else g

    def evaluate_chisq(self, x, y, normalize=False,
                       chi_sq_list=('vis', 'amp', 'cphase', 'logcamp')):   # blackhole.py:671-699
        x01 = to01(x).clip(min=0)
        xf = self.flux(x01)
        xa = x01 * (y.flux / xf)[:, :, None, None, None]
        res = {}
        if 'vis' in chi_sq_list:
            res['vis'] = self.chi2_vis_meas(self.vis(xa), y.vis, y.sigma_vis)
        if 'amp' in chi_sq_list:
            res['amp'] = self.chi2_amp_meas(self.vis(xa).abs(), y.amp, y.sigma_amp)
        if 'cphase' in chi_sq_list:
            res['cphase'] = self.chi2_cp_meas(self.cphase(self.vis(xa))[0], y.cphase,
                                              y.sigma_cphase)
        if 'logcamp' in chi_sq_list:
            res['logcamp'] = self.chi2_lc_meas(self.logcamp(self.vis(xa))[0], y.logcamp,
                                               y.sigma_logcamp)
        if normalize:
            res = {k: chisq_fold(v) for k, v in res.items()}
        return res


def closure_tables(times, t1, t2, tri, quad):
    """Reference-owned index mapping, blackhole.py:235-264.

    times/t1/t2: per visibility row i (row order = flattened (frame, baseline)).
    tri:  iterable of (time, a, b, c)   -> rows of (a,b),(b,c),(c,a)
    quad: iterable of (time, a, b, c, d)-> rows of (a,b),(c,d),(a,d),(b,c)
    A row stored as (t1,t2) is looked up conj=0 as (t1,t2) and conj=1 as (t2,t1);
    later rows overwrite earlier ones with the same key (dict semantics).
    Returns python lists of [K][3], [K][3], [K][4], [K][4] ints.
    """
    row, flip = {}, {}
    for i, (tm, a, b) in enumerate(zip(times, t1, t2)):
        row[(tm, a, b)] = i
        flip[(tm, a, b)] = 0
        row[(tm, b, a)] = i
        flip[(tm, b, a)] = 1
    cp_i, cp_c, ca_i, ca_c = [], [], [], []
    for tm, a, b, c in tri:
        keys = [(tm, a, b), (tm, b, c), (tm, c, a)]
        cp_i.append([row[k] for k in keys])
        cp_c.append([flip[k] for k in keys])
    for tm, a, b, c, d in quad:
        keys = [(tm, a, b), (tm, c, d), (tm, a, d), (tm, b, c)]
        ca_i.append([row[k] for k in keys])
        ca_c.append([flip[k] for k in keys])
    return cp_i, cp_c, ca_i, ca_c
