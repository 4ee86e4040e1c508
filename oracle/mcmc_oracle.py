"""Oracle (TEST INFRASTRUCTURE, not product): PyTorch-CPU restatement of the DAPS
MCMC inner loop (reference ``sampler/daps.py:76-251``) and of the latent wrapper
(``forward_operator/latent_wrapper.py``).  Pinned by ``tests/golden/mcmc_*.npz``.

All noise is drawn through ``torch.randn_like`` / ``torch.rand_like`` in the same
order as the reference so that a shared generator state reproduces its stream.
"""
import numpy as np
import torch


class LatentWrapOracle:
    """latent_wrapper.py:5-30: decode -> op.loss -> autograd -> cast -> clamp(+-1)."""

    def __init__(self, op, model):
        self.op, self.model, self.sigma = op, model, op.sigma

    def __call__(self, z):
        return self.op(self.model.decode(z))

    def loss(self, z, y):
        return self.op.loss(self.model.decode(z).float(), y)

    def gradient(self, z, y, return_loss=False):
        zt = z.clone().detach().requires_grad_(True)
        l = self.loss(zt, y).sum()
        g = torch.autograd.grad(l, zt)[0].to(z.dtype)
        g = torch.clamp(g, -1, 1)
        return (g, l) if return_loss else g


def lr_at(lr, lr_min_ratio, ratio):
    """daps.py:233-239 with p = 1."""
    p = 1
    mult = (1 ** (1 / p) + ratio * (lr_min_ratio ** (1 / p) - 1 ** (1 / p))) ** p
    return mult * lr


class MCMCOracle:
    def __init__(self, num_steps, lr, tau=0.01, lr_min_ratio=0.01, prior_solver="gaussian",
                 prior_sigma_min=1e-2, mc_algo="langevin", momentum=0.9):
        self.num_steps, self.lr, self.tau = num_steps, lr, tau
        self.lr_min_ratio, self.prior_solver = lr_min_ratio, prior_solver
        self.prior_sigma_min, self.mc_algo, self.momentum = prior_sigma_min, mc_algo, momentum
        self.trace = None

    # daps.py:106-129
    def _cache_prior(self, x0hat, xt, model, sigma):
        s = self.prior_solver
        if s == "score-min":
            self._prior = model.score(x0hat, self.prior_sigma_min).detach()
        elif s == "score-t":
            self._prior = model.score(xt, sigma).detach()
        elif s == "gaussian":
            self._prior = (x0hat - xt).detach() / sigma ** 2
        elif s != "exact":
            raise NotImplementedError

    def _prior_score(self, x, model):
        if self.prior_solver == "exact":
            return model.score(x, torch.tensor(self.prior_sigma_min).to(x.device)).detach()
        return self._prior

    # daps.py:93-104
    def score(self, x, x0hat, model, xt, op, y, sigma):
        g, l = op.gradient(x, y, return_loss=True)
        return -g / self.tau ** 2 + (xt - x) / sigma ** 2 + self._prior_score(x, model), l

    # daps.py:135-144
    def update(self, x, s, lr, eps):
        if self.mc_algo == "langevin":
            return x + lr * s + np.sqrt(2 * lr) * eps
        if self.mc_algo == "hmc":
            h = np.sqrt(lr)
            self.velocity = self.momentum * self.velocity + h * s \
                + np.sqrt(2 * (1 - self.momentum)) * eps
            return x + self.velocity * h
        raise NotImplementedError

    # daps.py:178-208
    def sample(self, xt, model, x0hat, op, y, sigma, ratio, record=False):
        if self.mc_algo == "MH":
            return self.sample_mh(xt, model, x0hat, op, y, sigma, ratio)
        self.trace = [] if record else None
        lr = lr_at(self.lr, self.lr_min_ratio, ratio)
        if self.mc_algo == "hmc":
            self.velocity = torch.randn_like(x0hat)
        self._cache_prior(x0hat, xt, model, sigma)
        x = x0hat.clone().detach()
        for _ in range(self.num_steps):
            s, _l = self.score(x, x0hat, model, xt, op, y, sigma)
            eps = torch.randn_like(x)
            x = self.update(x, s, lr, eps)
            if torch.isnan(x).any():
                return torch.zeros_like(x)
            if record:
                self.trace.append(x.detach().clone())
        return x.detach()

    # daps.py:210-223
    def optimize(self, x0hat, op, y, sigma, ratio):
        lr = lr_at(self.lr, self.lr_min_ratio, ratio)
        x = x0hat.clone().detach()
        for _ in range(self.num_steps):
            g, _l = op.gradient(x, y, return_loss=True)
            x = x - lr * (g + (x - x0hat.detach()) / sigma ** 2)
        return x.detach()

    # daps.py:146-176
    def sample_mh(self, xt, model, x0hat, op, y, sigma, ratio):
        lr = lr_at(self.lr, self.lr_min_ratio, ratio)
        x = x0hat.clone().detach()
        for _ in range(self.num_steps):
            xn = x + torch.randn_like(x) * np.sqrt(lr)
            ln, lo = op.loss(xn, y), op.loss(x, y)
            d_data = (lo - ln) / (2 * self.tau ** 2)
            pn = (xn - x0hat).pow(2).flatten(1).sum(-1)
            po = (x - x0hat).pow(2).flatten(1).sum(-1)
            logp = d_data + (po - pn) / (2 * sigma ** 2)
            acc = torch.rand_like(logp).log() < logp
            acc = acc.view(-1, *[1] * len(x.shape[1:]))
            x = torch.where(acc, xn, x)
        return x.detach()


def edm_sigmas(num_steps, sigma_max=100, sigma_min=1e-2, timestep="poly-7"):
    """Annealing grid, cores/scheduler.py:266-315 + :38-44: num_steps+1 entries,
    last forced to 0; float64 like the reference (torch.from_numpy of float64)."""
    p = int(timestep.split("-")[1])
    r = np.linspace(0, 1, num_steps + 1)
    t = np.array([(sigma_max ** (1 / p) + s * (sigma_min ** (1 / p) - sigma_max ** (1 / p))) ** p
                  for s in r])
    t = torch.from_numpy(t).float()
    return torch.cat([t[:-1], torch.zeros_like(t[:1])])


def edm_time_steps(num_steps, sigma_max, sigma_min, timestep="poly-7"):
    """EDMScheduler.get_discrete_time_steps(self.num_steps), cores/scheduler.py:266-315: the
    scheduler stores num_steps + 1 (:38) and the grid keeps numpy's float64 (torch.from_numpy
    without .float(), :313-315); sigma_max may be a 0-d tensor (daps.py:300)."""
    p = int(timestep.split("-")[1])
    fn = lambda r: (sigma_max ** (1 / p) + r * (sigma_min ** (1 / p) - sigma_max ** (1 / p))) ** p
    return torch.from_numpy(np.array([fn(s) for s in np.linspace(0, 1, num_steps + 1)]))


def pfode_euler_oracle(model, ts, xT):
    """DiffusionPFODE.sample with method='euler' (cores/scheduler.py:378-408) for the EDM
    schedule (s = 1, sigma(t) = t): dx/dt = -t score(x, sigma=t); fixed-grid Euler on ts (what
    torchdiffeq's 'euler' does -- restated from its published behaviour, unpinned)."""
    x = xT.flatten(1)
    for i in range(len(ts) - 1):
        t = ts[i]
        d = -1.0 * torch.ones_like(t.float()) * t.float() * model.score(x.view(xT.shape), sigma=t.float())
        x = x + (ts[i + 1] - t) * d.flatten(1)
    return x.view(xT.shape)


class LatentDAPSOracle:
    """LatentDAPS.sample, sampler/daps.py:254-382 (no evaluator / recorder)."""

    def __init__(self, ann, dif, mc, latent_wrapper=True):
        self.sigmas = edm_sigmas(ann["num_steps"], ann["sigma_max"], ann["sigma_min"], ann["timestep"])
        self.n = ann["num_steps"] + 1
        self.dif, self.mc, self.wrap = dif, MCMCOracle(**mc), latent_wrapper

    def sample(self, model, z_start, op, y):
        oper = LatentWrapOracle(op, model) if self.wrap else op
        zt = z_start
        for step in range(self.n - 1):
            sigma = self.sigmas[step]
            with torch.no_grad():
                ts = edm_time_steps(self.dif["num_steps"], sigma, self.dif["sigma_min"], self.dif["timestep"])
                z0hat = pfode_euler_oracle(model, ts, zt)
                x0hat = model.decode(z0hat)
            ratio = step / self.n
            if self.wrap:
                z0y = self.mc.sample(zt, model, z0hat, oper, y, sigma, ratio)
            else:
                z0y = model.encode(self.mc.optimize(x0hat, op, y, sigma, ratio))
            zt = z0y + torch.randn_like(z0y) * self.sigmas[step + 1]
        return model.decode(zt)
