"""GPU: nufft_spread per chain count with and without the lane-per-chain lists (where is the crossover?)."""
import ctypes as C
import torch
from step_amd import _lib
from step_amd.forward_operator import get_operator

dev = torch.device("cuda:0")
a = get_operator("mri_radial", spokes=32, resolution=192, frames=12)
b = get_operator("mri_radial", spokes=32, resolution=192, frames=12)
for k in ("wide_rec", "wide_ent", "wide_w"):
    b._host[k] = None
L = _lib.lib()
for B in (8, 12, 16, 24, 32, 48, 64):
    v = torch.randn(B, 12, a.J, 2, device=dev)
    grid = torch.empty(B, 12, 384, 384, 2, device=dev)
    res = []
    for op in (a, b):
        plan = op._plan(dev)
        # bypass the chain threshold for the wide lists by calling at this B directly
        def run():
            _lib.check(L.nufft_spread(v.data_ptr(), B * 12, 12, C.byref(plan), grid.data_ptr(), _lib.stream_of(v)), "spread")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    print("B = %2d: with lists %7.1f us, without %7.1f us" % (B, res[0], res[1]), flush=True)
