"""Noise schedules and the Euler PF-ODE used by the annealing driver (stock PyTorch; these
belong to the denoiser side of the reference, cores/scheduler.py, and are not kernel work).

Only the 'edm' schedule is provided: it is the one both shipped task configs select for the
annealing and the diffusion grids (configs/task/blackhole.yaml:42-53, mri.yaml:35-46)."""
import numpy as np
import torch

__all__ = ["EDMScheduler", "get_diffusion_scheduler", "pfode_euler"]


class EDMScheduler:
    """sigma(t) = t, s(t) = 1, poly-p time grid (cores/scheduler.py:266-315, :38-44)."""

    def __init__(self, num_steps, sigma_max=100, sigma_min=1e-2, timestep='poly-7'):
        self.num_steps = num_steps + 1                      # include the initial step
        # sigma_max is used as given: LatentDAPS passes a float32 0-d tensor per annealing step
        # (daps.py:301), and the reference's grid is then float32 tensor arithmetic, not float64
        # ... but PyYAML reads the reference configs' `sigma_min: 1e-2` as a STRING (no dot in the
        # mantissa), so strings and plain Python numbers are made floats; tensors stay what they are
        if isinstance(sigma_max, (str, int)):
            sigma_max = float(sigma_max)
        if isinstance(sigma_min, (str, int)):
            sigma_min = float(sigma_min)
        self.sigma_max, self.sigma_min = sigma_max, sigma_min
        p = int(timestep.split('-')[1])
        self._fn = lambda r: (sigma_max ** (1 / p) + r * (sigma_min ** (1 / p) - sigma_max ** (1 / p))) ** p
        t = self.get_discrete_time_steps(self.num_steps)
        s = t[:-1].float()
        self.sigma_steps = torch.cat([s, torch.zeros_like(s[:1])])   # last entry forced to 0

    def get_discrete_time_steps(self, num_steps):
        return torch.from_numpy(np.array([self._fn(s) for s in np.linspace(0, 1, num_steps)]))     # scheduler.py:312-315


def get_diffusion_scheduler(name: str, **kwargs):
    if name != 'edm':
        raise NameError(f"Name {name} is not defined.")
    return EDMScheduler(**kwargs)


def pfode_euler(model, scheduler, xT):
    """Reverse probability-flow ODE with fixed-grid Euler steps on the scheduler's time grid
    (what torchdiffeq's method='euler' does for DiffusionPFODE.sample, cores/scheduler.py:378-408).
    For EDM: dx/dt = -t * score(x, sigma=t)."""
    ts = scheduler.get_discrete_time_steps(scheduler.num_steps).to(xT.device)
    x = xT
    for i in range(len(ts) - 1):
        t = ts[i]
        x = x + (ts[i + 1] - t) * (-t * model.score(x, sigma=t)).to(x.dtype)
    return x
