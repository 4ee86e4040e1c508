"""DAPS sampling loop behind the reference's interface (reference sampler/daps.py).

``MCMCSampler`` is the hot inner loop (daps.py:76-251): per step one fused operator
gradient (HIP kernels via ``operator.gradient``) and one fused state-update kernel
(``mcmc_update``) replacing ~10 elementwise launches and the per-step NaN host sync.
``LatentDAPS`` is the annealing driver around it (daps.py:254-382); the denoiser / VAE it
calls stay stock PyTorch.  Noise is drawn with ``torch.randn_like`` in the reference's order,
so a shared generator state reproduces the reference's random stream.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..forward_operator.latent_wrapper import LatentWrapper
from .base import register_sampler
from .schedule import get_diffusion_scheduler, pfode_euler

__all__ = ["Trajectory", "MCMCSampler", "LatentDAPS"]


class Trajectory(nn.Module):
    """Recorder for intermediate states (daps.py:10-73)."""

    def __init__(self):
        super().__init__()
        self.tensor_data, self.value_data, self._compile = {}, {}, False

    def add_tensor(self, name, images):
        self.tensor_data.setdefault(name, []).append(images.detach().cpu())

    def add_value(self, name, values):
        self.value_data.setdefault(name, []).append(values)

    def compile(self):
        if not self._compile:
            self._compile = True
            self.tensor_data = {k: torch.stack(v, dim=0) for k, v in self.tensor_data.items()}
            self.value_data = {k: torch.tensor(v) for k, v in self.value_data.items()}
        return self

    @classmethod
    def merge(cls, trajs):
        out = cls()
        for name in trajs[0].tensor_data.keys():
            out.tensor_data[name] = torch.cat([t.tensor_data[name] for t in trajs], dim=1)
        for name in trajs[0].value_data.keys():
            out.value_data[name] = trajs[0].value_data[name]
        return out


def chain_randn_like(ref, generators=None):
    """``torch.randn_like(ref)`` (the reference's draw, from the device's default generator) or,
    with one generator per chain, chain b's slice from its own generator: the result for a chain
    then does not depend on which other chains share the batch."""
    if generators is None:
        return torch.randn_like(ref)
    if len(generators) != ref.shape[0]:
        raise ValueError("%d chain generators for a batch of %d" % (len(generators), ref.shape[0]))
    return torch.stack([torch.randn(ref.shape[1:], device=ref.device, dtype=ref.dtype, generator=g)
                        for g in generators])


class MCMCSampler(nn.Module):
    """Langevin / damped-HMC / Metropolis-Hastings inner loop.

    ``nan_check``: 'deferred' (default) keeps the NaN guard of daps.py:202-203 on the device and
    reads it once after the last step -- the returned tensor is identical (a NaN state stays NaN,
    the whole batch comes back as zeros) but no host synchronisation happens per step;
    'step' reads it every step and returns early exactly like the reference.  Two differences of
    'deferred' once a NaN HAS appeared: the loop runs on, so later
    ``operator.gradient`` / ``model.score`` / noise draws still happen and the global RNG stream
    seen by later annealing steps differs from the reference's; and with hmc at momentum 0
    ``self.velocity`` keeps its initial draw (the kernel neither reads nor writes a velocity of
    weight zero) where the reference leaves the last one.  Use 'step' for stream-exact replays.

    ``nan_guard``: 'batch' (default, the reference: one NaN anywhere zeroes the whole batch) or
    'chain' (only the chains that hold a NaN come back as zeros -- for batched / sharded drivers,
    where unrelated chains share a batch; a behaviour change, off by default).

    ``noise``: 'torch' (default) draws epsilon with torch.randn_like as the reference does
    (daps.py:192), or chain by chain from ``set_chains(generators=...)``; 'philox' draws it inside
    the update kernel from per-chain seeds (``set_chains(seeds=...)``): no noise tensor in HBM,
    per-chain reproducible, statistical parity only.  With 'philox', the 'gaussian' prior solver,
    no carried velocity (langevin, or hmc at momentum 0) and an operator that offers
    ``langevin_step`` (the black-hole operator called on pixel-space frames), the whole step --
    forward, chi^2, adjoint, score, noise, update -- is the operator's two or three launches and
    neither the gradient nor the noise reaches HBM (``fuse_langevin``, on by default).

    ``graph``: with ``noise='philox'``, a prior score that is constant over the inner loop (every
    solver but 'exact') and an operator that is called directly (pixel-space state), the first step
    runs eagerly, the step is then captured into a HIP graph -- operator step, update kernel with
    the step number read from a device word, the increment of that word -- and the rest of the loop
    is ``num_steps - 1`` replays: no Python, no allocator and no launch set-up per step.  Same kernels,
    same arguments, bit-identical result (tests/test_gpu_mcmc.py).  Off by default."""

    def __init__(self, num_steps, lr, tau=0.01, lr_min_ratio=0.01, prior_solver='gaussian',
                 prior_sigma_min=1e-2, mc_algo='langevin', momentum=0.9, nan_check='deferred',
                 nan_guard='batch', noise='torch', graph=False):
        super().__init__()
        self.num_steps = num_steps
        self.lr = lr
        self.tau = tau
        self.lr_min_ratio = lr_min_ratio
        self.prior_solver = prior_solver
        self.prior_sigma_min = prior_sigma_min
        self.mc_algo = mc_algo
        self.momentum = momentum
        assert nan_check in ('deferred', 'step')
        self.nan_check = nan_check
        assert nan_guard in ('batch', 'chain')
        self.nan_guard = nan_guard
        assert noise in ('torch', 'philox')
        self.noise = noise
        self.graph = bool(graph)
        self.chain_generators = None     # list of torch.Generator, one per chain, or None
        self.chain_seeds = None          # int64 [B] on the device ('philox'), or None
        self.noise_step = 0              # Philox counter: steps taken since set_chains
        self.fuse_descent = True         # optimize(): use operator.descent_step where offered
        # sample(), noise='philox': use operator.langevin_step where offered and where the operator
        # prefers it at this batch (True), at any batch ('always'), or never (False)
        self.fuse_langevin = True

    def set_chains(self, generators=None, seeds=None):
        """Per-chain randomness for the next sample() calls: torch generators (start velocity,
        'torch' noise) and/or 64-bit seeds ('philox' noise); resets the Philox step counter."""
        self.chain_generators = list(generators) if generators is not None else None
        self.chain_seeds = seeds
        self.noise_step = 0

    def _nan_result(self, x):
        """The guard of daps.py:202-203 once the flag is up."""
        if self.nan_guard == 'batch':
            return torch.zeros_like(x)
        bad = torch.isnan(x).flatten(1).any(dim=1)
        x = x.clone()
        x[bad] = 0
        return x

    # ------------------------------------------------------------------ reference-shaped pieces
    def score_fn(self, x, x0hat, model, xt, operator, measurement, sigma):
        """score of p(x_0 = x | x_t, y) as separate torch ops (daps.py:93-104); the sampling loop
        uses the fused kernel instead, this stays for callers of the reference API."""
        g, loss = operator.gradient(x, measurement, return_loss=True)
        data_term = -g / self.tau ** 2
        xt_term = (xt - x) / sigma ** 2
        return data_term + xt_term + self.get_prior_score(x, x0hat, xt, model, sigma), loss

    def get_prior_score(self, x, x0hat, xt, model, sigma):
        if self.prior_solver in ('score-min', 'score-t', 'gaussian'):
            return self.prior_score
        if self.prior_solver == 'exact':
            return model.score(x, torch.tensor(self.prior_sigma_min).to(x.device)).detach()
        raise NotImplementedError

    def prepare_prior_score(self, x0hat, xt, model, sigma):
        if self.prior_solver == 'score-min':
            self.prior_score = model.score(x0hat, self.prior_sigma_min).detach()
        elif self.prior_solver == 'score-t':
            self.prior_score = model.score(xt, sigma).detach()
        elif self.prior_solver == 'gaussian':
            self.prior_score = (x0hat - xt).detach() / sigma ** 2
        elif self.prior_solver != 'exact':
            raise NotImplementedError

    def mc_prepare(self, x0hat, xt, model, operator, measurement, sigma):
        if self.mc_algo == 'hmc':
            self.velocity = chain_randn_like(x0hat, self.chain_generators)

    def mc_update(self, x, cur_score, lr, epsilon):
        """Unfused update (daps.py:135-144), kept for callers of the reference API."""
        if self.mc_algo == 'langevin':
            return x + lr * cur_score + np.sqrt(2 * lr) * epsilon
        if self.mc_algo == 'hmc':
            step_size = np.sqrt(lr)
            self.velocity = self.momentum * self.velocity + step_size * cur_score \
                + np.sqrt(2 * (1 - self.momentum)) * epsilon
            return x + self.velocity * step_size
        raise NotImplementedError

    def get_lr(self, ratio):
        p = 1
        multiplier = (1 ** (1 / p) + ratio * (self.lr_min_ratio ** (1 / p) - 1 ** (1 / p))) ** p
        return multiplier * self.lr

    # ------------------------------------------------------------------ fused step
    def _params(self, lr, sigma):
        algo = {'langevin': 0, 'hmc': 1}.get(self.mc_algo)
        if algo is None:
            raise NotImplementedError
        return _lib.mcmc_params_t(1.0 / self.tau ** 2, 1.0 / float(sigma) ** 2, lr, np.sqrt(lr),
                                  np.sqrt(2 * lr), self.momentum,
                                  np.sqrt(2 * (1 - self.momentum)), algo)

    @staticmethod
    def _kernel_tensor(t, like):
        if t.device.type != 'cuda':
            raise RuntimeError("step_amd MCMCSampler runs on the GPU only (tensor on %s)" % t.device)
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != like.shape:
            t = t.to(torch.float32).expand_as(like).contiguous()
        return t

    def _fused_update(self, x, grad, xt, prior, eps, params, nan_flag):
        """In place on x (and self.velocity): score_fn + mc_update + NaN flag in one launch."""
        vel = self.velocity if self.mc_algo == 'hmc' else None
        rc = _lib.lib().mcmc_update(x.data_ptr(), _lib.ptr(vel), grad.data_ptr(), xt.data_ptr(),
                                    _lib.ptr(prior), eps.data_ptr(), x.numel(), C.byref(params),
                                    nan_flag.data_ptr(), _lib.stream_of(x))
        _lib.check(rc, "mcmc_update")

    # ------------------------------------------------------------------ samplers
    def sample(self, xt, model, x0hat, operator, measurement, sigma, ratio, record=False,
               verbose=False):
        if self.mc_algo == 'MH':
            return self.sample_MH(xt, model, x0hat, operator, measurement, sigma, ratio, record, verbose)
        if record:
            self.trajectory = Trajectory()
        lr = self.get_lr(ratio)
        self.mc_prepare(x0hat, xt, model, operator, measurement, sigma)
        self.prepare_prior_score(x0hat, xt, model, sigma)

        x = x0hat.clone().detach()
        if x.dtype != torch.float32:
            raise TypeError("step_amd MCMCSampler needs a float32 state, got %s" % x.dtype)
        x = x.contiguous()
        xt_k = self._kernel_tensor(xt.detach(), x)
        if self.mc_algo == 'hmc':
            self.velocity = self._kernel_tensor(self.velocity, x)
        params = self._params(lr, sigma)
        nan_flag = torch.zeros(1, device=x.device, dtype=torch.int32)
        philox = self.noise == 'philox'
        step_fn = None
        if philox:
            seeds = self.chain_seeds
            if seeds is None or seeds.numel() != x.shape[0]:
                raise ValueError("noise='philox' needs set_chains(seeds=int64[%d])" % x.shape[0])
            seeds = seeds.to(device=x.device, dtype=torch.int64).contiguous()
            step_fn = self._philox_step_fn(x, xt_k, x0hat, operator, measurement, params, seeds, nan_flag,
                                           record)
        # torch noise, 'gaussian' solver, no carried velocity, an operator that applies the update in
        # its adjoint: the step of the loop below in the operator's two or three launches, with the
        # reference's own draw handed to the adjoint as a tensor (gradient and score never reach
        # HBM; the 'gaussian' score's two xt terms are evaluated as their sum (x0hat - x) / sigma^2)
        eps_adjoint = (not philox and self.fuse_langevin and self.prior_solver == 'gaussian'
                       and (self.mc_algo == 'langevin' or self.momentum == 0)
                       and getattr(operator, 'has_langevin_step', False))
        x0_k = self._kernel_tensor(x0hat.detach(), x) if eps_adjoint else None
        steps_left = self.num_steps
        # graph replay is for an operator called directly on a pixel-space state: through a
        # LatentWrapper a step is the decoder's forward and autograd backward, which is not a
        # fixed sequence of this library's launches -- those run the eager loop
        from ..forward_operator.latent_wrapper import LatentWrapper
        if (self.graph and philox and not record and self.prior_solver != 'exact'
                and self.nan_check == 'deferred' and self.num_steps > 1
                and not isinstance(operator, LatentWrapper)):
            steps_left -= self._graph_loop(x, step_fn)
            if steps_left == 0:
                if nan_flag.item():
                    return self._nan_result(x)
                return x.detach()
        for _ in range(steps_left):
            if philox:
                prior = self.get_prior_score(x, x0hat, xt, model, sigma) if step_fn.needs_prior else None
                epsilon, fitting_loss = step_fn(self.noise_step, None, prior)
                self.noise_step += 1
            elif eps_adjoint:
                epsilon = chain_randn_like(x, self.chain_generators)
                fitting_loss = operator.langevin_step(x, measurement, x0_k, params, None, eps=epsilon,
                                                      nan_flag=nan_flag, return_loss=record)
            else:
                grad, fitting_loss = operator.gradient(x, measurement, return_loss=True)
                prior = self.get_prior_score(x, x0hat, xt, model, sigma)
                epsilon = chain_randn_like(x, self.chain_generators)
                self._fused_update(x, self._kernel_tensor(grad, x), xt_k,
                                   self._kernel_tensor(prior, x), epsilon, params, nan_flag)
            if self.nan_check == 'step' and nan_flag.item():
                return self._nan_result(x)
            if record:
                self._record(x.clone(), epsilon, fitting_loss.sqrt())
        if self.nan_check == 'deferred' and nan_flag.item():
            return self._nan_result(x)
        return x.detach()

    def _philox_step_fn(self, x, xt_k, x0hat, operator, measurement, params, seeds, nan_flag, record):
        """-> step(step, step_dev, prior) taking one in-place step on x with in-kernel noise;
        returns (epsilon or None, summed loss or None).  Two routes:
        * the operator applies the update itself (``langevin_step``: the black-hole adjoint) when
          the score is the 'gaussian' solver's -- its x-dependent terms collapse to
          (x0hat - x) / sigma^2 -- and no velocity is carried (langevin, or hmc at momentum 0):
          neither gradient nor noise reaches HBM;
        * otherwise ``operator.gradient`` followed by ``mcmc_update_philox``."""
        vel = self.velocity if self.mc_algo == 'hmc' else None
        n, chain_len = x.numel(), x[0].numel()
        in_adjoint = (self.fuse_langevin and not record and self.prior_solver == 'gaussian'
                      and (self.mc_algo == 'langevin' or self.momentum == 0)
                      and getattr(operator, 'has_langevin_step', False)
                      and getattr(operator, 'langevin_draws_noise', True)      # the MRI kernel only reads `eps`
                      and (self.fuse_langevin == 'always' or operator.prefers_langevin_step(x.shape[0])))
        if in_adjoint:
            x0 = self._kernel_tensor(x0hat.detach(), x)

            def step(step, step_dev, prior=None):
                operator.langevin_step(x, measurement, x0, params, seeds, step=step, step_dev=step_dev,
                                       nan_flag=nan_flag)
                return None, None
            step.needs_prior = False
            return step

        # 'gaussian': (xt - x) / sigma^2 + (x0hat - xt) / sigma^2 is passed as (x0hat - x) / sigma^2
        # (x0hat in the kernel's xt slot, no prior tensor): one video-sized stream less per step,
        # and without the cancellation of the two xt terms (the noise is not the reference's
        # stream on this route anyway; the torch-noise route keeps the reference's two terms)
        x0_k = self._kernel_tensor(x0hat.detach(), x) if self.prior_solver == 'gaussian' else None

        def step(step, step_dev, prior):
            grad, loss = operator.gradient(x, measurement, return_loss=True)
            epsilon = torch.empty_like(x) if record else None
            anchor = xt_k if x0_k is None else x0_k
            rc = _lib.lib().mcmc_update_philox(
                x.data_ptr(), _lib.ptr(vel), self._kernel_tensor(grad, x).data_ptr(), anchor.data_ptr(),
                None if x0_k is not None else self._kernel_tensor(prior, x).data_ptr(), n, C.byref(params),
                seeds.data_ptr(), chain_len,
                step, _lib.ptr(step_dev), 0, _lib.ptr(epsilon), nan_flag.data_ptr(), _lib.stream_of(x))
            _lib.check(rc, "mcmc_update_philox")
            return epsilon, loss
        step.needs_prior = x0_k is None
        return step

    def _graph_loop(self, x, step_fn):
        """num_steps steps in place on x: one eager, one captured, num_steps - 1 replays.
        Returns the number of steps taken: num_steps, or 1 when the capture failed (the eager step
        has advanced x and the step counter by one; the caller finishes in the eager loop)."""
        dev = x.device
        step_dev = torch.full((1,), self.noise_step, dtype=torch.int64, device=dev)
        prior = self._kernel_tensor(self.prior_score, x) if step_fn.needs_prior else None

        def one_step():
            step_fn(0, step_dev, prior)
            step_dev.add_(1)

        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        taken = 1
        with torch.cuda.stream(side):
            one_step()                                   # eager: also warms every allocation size
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    one_step()                           # recorded, not run
            except Exception as e:                       # a step that cannot be captured
                import warnings
                warnings.warn("MCMCSampler(graph=True): capture failed (%s); finishing eagerly" % (e,))
                graph = None
            if graph is not None:
                for _ in range(self.num_steps - 1):
                    graph.replay()
                taken = self.num_steps
        cur.wait_stream(side)
        self.noise_step += taken
        return taken

    def sample_MH(self, xt, model, x0hat, operator, measurement, sigma, ratio, record=False,
                  verbose=False):
        if record:
            self.trajectory = Trajectory()
        lr = self.get_lr(ratio)
        x = x0hat.clone().detach()
        for _ in range(self.num_steps):
            x_new = x + torch.randn_like(x) * np.sqrt(lr)
            loss_new = operator.loss(x_new, measurement)
            loss_old = operator.loss(x, measurement)
            log_data_ratio = (loss_old - loss_new) / (2 * self.tau ** 2)
            prior_new = (x_new - x0hat).pow(2).flatten(1).sum(-1)
            prior_old = (x - x0hat).pow(2).flatten(1).sum(-1)
            log_accept = log_data_ratio + (prior_old - prior_new) / (2 * sigma ** 2)
            accept = torch.rand_like(log_accept).log() < log_accept
            x = torch.where(accept.view(-1, *[1] * len(x.shape[1:])), x_new, x)
        return x.detach()

    def optimize(self, x0hat, operator, measurement, sigma, ratio, record=False, verbose=False):
        """Pixel-space gradient descent on decoded frames (daps.py:210-223).  An operator that
        offers ``descent_step`` (the black-hole operator) applies the update
        ``x -= lr (g + (x - x0hat) / sigma^2)`` inside its adjoint kernel, so the gradient never
        reaches HBM; otherwise the update is one fused kernel after ``operator.gradient``."""
        lr = self.get_lr(ratio)
        x = x0hat.clone().detach()
        fused = x.device.type == 'cuda' and x.dtype == torch.float32
        if fused:
            x = x.contiguous()
            x0 = self._kernel_tensor(x0hat.detach(), x)
            params = _lib.mcmc_params_t(1.0, 1.0 / float(sigma) ** 2, lr, 0.0, 0.0, 0.0, 0.0, 2)
            if not record and self.fuse_descent and getattr(operator, 'has_descent_step', False):
                for _ in range(self.num_steps):
                    operator.descent_step(x, measurement, x0, lr, 1.0 / float(sigma) ** 2)
                return x.detach()
        for _ in range(self.num_steps):
            g, loss = operator.gradient(x, measurement, return_loss=True)
            if fused:
                rc = _lib.lib().mcmc_update(x.data_ptr(), None, self._kernel_tensor(g, x).data_ptr(),
                                            x0.data_ptr(), None, None, x.numel(), C.byref(params),
                                            None, _lib.stream_of(x))
                _lib.check(rc, "mcmc_update")
            else:
                x = x - lr * (g + (x - x0hat.detach()) / sigma ** 2)
            if record:
                self._record(x.clone() if fused else x, g, loss)
        return x.detach()

    def _record(self, x, epsilon, loss):
        self.trajectory.add_tensor('xi', x)
        self.trajectory.add_tensor('epsilon', epsilon)
        self.trajectory.add_value('loss', loss)

    def summary(self):
        print('+' * 50)
        print('MCMC Sampler Summary (step_amd fused kernels)')
        print('+' * 50)
        for k, v in (("Prior Solver", self.prior_solver), ("MCMC Algorithm", self.mc_algo),
                     ("Num Steps", self.num_steps), ("Learning Rate", self.lr), ("Tau", self.tau),
                     ("LR Min Ratio", self.lr_min_ratio)):
            print(f"{k:<16}: {v}")
        print('+' * 50)


@register_sampler('latent_daps')
class LatentDAPS(nn.Module):
    """Latent decoupled annealing posterior sampling (daps.py:254-382): per annealing step a
    PF-ODE denoise (stock PyTorch prior), the MCMC data-consistency loop above, and re-noising."""

    def __init__(self, annealing_scheduler_config, diffusion_scheduler_config, mcmc_sampler_config,
                 latent_wrapper=True, batch_consistency=False):
        super().__init__()
        diffusion_scheduler_config = dict(diffusion_scheduler_config)
        diffusion_scheduler_config.pop('sigma_max', None)      # changes every step (daps.py:355-363)
        self.annealing_scheduler = get_diffusion_scheduler(**annealing_scheduler_config)
        self.diffusion_scheduler_config = diffusion_scheduler_config
        self.mcmc_sampler = MCMCSampler(**mcmc_sampler_config)
        self.batch_consistency = batch_consistency
        self.latent_wrapper = latent_wrapper
        self.chain_generators = None

    def set_chains(self, generators=None, seeds=None):
        """Per-chain randomness (step_amd.driver): every draw of the next sample() call -- start
        and re-noising here, velocity and Langevin noise in the MCMC sampler -- comes from the
        chain's own generator / seed, so a chain's result does not depend on the batch it is in."""
        self.chain_generators = list(generators) if generators is not None else None
        self.mcmc_sampler.set_chains(generators, seeds)

    def get_randn_like(self, ref):
        if self.batch_consistency:
            return chain_randn_like(ref[:, 0], self.chain_generators).unsqueeze(1).expand_as(ref)
        return chain_randn_like(ref, self.chain_generators)

    def sample(self, model, z_start, operator, measurement, evaluator=None, record=False,
               verbose=False, **kwargs):
        if record:
            self.trajectory = Trajectory()
        sched = self.annealing_scheduler
        wrapped = LatentWrapper(operator, model) if self.latent_wrapper else operator
        zt = z_start
        for step in range(sched.num_steps - 1):
            sigma = sched.sigma_steps[step]
            with torch.no_grad():
                ds = get_diffusion_scheduler(**self.diffusion_scheduler_config, sigma_max=sigma)
                z0hat = pfode_euler(model, ds, zt)
                x0hat = model.decode(z0hat)
            ratio = step / sched.num_steps
            if self.latent_wrapper:
                z0y = self.mcmc_sampler.sample(zt, model, z0hat, wrapped, measurement, sigma, ratio)
                with torch.no_grad():
                    x0y = model.decode(z0y)
            else:
                x0y = self.mcmc_sampler.optimize(x0hat, operator, measurement, sigma, ratio)
                z0y = model.encode(x0y)
            # step < num_steps - 1 always holds here, as in the reference (daps.py:316-319)
            zt = z0y + self.get_randn_like(z0y) * sched.sigma_steps[step + 1]
            x0hat_results = x0y_results = {}
            if evaluator and 'gt' in kwargs:
                with torch.no_grad():
                    x0hat_results = evaluator(kwargs['gt'], measurement, x0hat)
                    x0y_results = evaluator(kwargs['gt'], measurement, x0y)
            if record:
                self._record(model.decode(zt), x0y, x0hat, sigma, x0hat_results, x0y_results)
        return model.decode(zt)

    def _record(self, xt, x0y, x0hat, sigma, x0hat_results, x0y_results):
        self.trajectory.add_tensor('xt', xt)
        self.trajectory.add_tensor('x0y', x0y)
        self.trajectory.add_tensor('x0hat', x0hat)
        self.trajectory.add_value('sigma', sigma)
        for name, v in x0hat_results.items():
            self.trajectory.add_value(f'x0hat_{name}', v)
        for name, v in x0y_results.items():
            self.trajectory.add_value(f'x0y_{name}', v)

    def get_start(self, ref):
        return self.get_randn_like(ref) * self.annealing_scheduler.sigma_max

    def summary(self):
        print('Annealing Scheduler: edm, %d steps, sigma %g -> %g'
              % (self.annealing_scheduler.num_steps - 1, self.annealing_scheduler.sigma_max,
                 self.annealing_scheduler.sigma_min))
        print('MCMC Sampler:')
        self.mcmc_sampler.summary()
