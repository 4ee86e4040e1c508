"""ms per step of MCMCSampler.optimize / .sample (torch noise, 'gaussian' solver, hmc at momentum 0) on
the MRI operator, with the state update inside the kernel (descent_step / langevin_step(eps=...)) and
as gradient + update kernel.  usage: python scripts/bench_sampler_mri.py [B ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.forward_operator import get_operator
from step_amd.sampler.daps import MCMCSampler
dev = 'cuda:0'
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
for B in [int(a) for a in sys.argv[1:]] or [32, 256]:
    x0 = torch.randn(B, 12, 2, 192, 192, device=dev)
    xt = x0 + 0.3 * torch.randn_like(x0)
    tau, steps = 0.1, 30
    lr = float(1e-3 * tau ** 2 / op.gradient(x0, y).abs().max())
    for name in ("optimize", "sample"):
        for fuse in (False, True):
            mc = MCMCSampler(num_steps=steps, lr=lr, tau=tau, lr_min_ratio=0.1, prior_solver="gaussian", mc_algo="hmc", momentum=0.0)
            mc.fuse_langevin = mc.fuse_descent = fuse
            run = (lambda: mc.optimize(x0, op, y, 0.5, 0.0)) if name == "optimize" else (lambda: mc.sample(xt, None, x0, op, y, 0.5, 0.0))
            run(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): r = run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / (3 * steps) * 1e3
            assert torch.isfinite(r).all()
            print("B=%d %-8s %-22s %.4f ms per step (%.0f k chain-steps/s)" % (B, name, "update in the kernel" if fuse else "gradient + update", ms, B / ms), flush=True)
