"""Event-timed mcmc_update vs mcmc_update_philox at 1, 8, 64 chains of 64x256x256."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd import _lib
dev = torch.device('cuda:0')
L = _lib.lib()
for B in (1, 8, 64):
    n = B * 64 * 256 * 256
    x = torch.randn(n, device=dev); v = torch.zeros_like(x); g = torch.randn_like(x); xt = torch.randn_like(x); pr = torch.randn_like(x); eps = torch.randn_like(x)
    seeds = torch.arange(B, dtype=torch.int64, device=dev); flag = torch.zeros(1, dtype=torch.int32, device=dev)
    for algo, mom in ((0, 0.0), (1, 0.0), (1, 0.8)):
        p = _lib.mcmc_params_t(1.0, 1.0, 1e-4, 1e-2, 1.4e-2, mom, 1.0, algo)
        st = torch.cuda.current_stream().cuda_stream
        def plain(): assert L.mcmc_update(x.data_ptr(), v.data_ptr(), g.data_ptr(), xt.data_ptr(), pr.data_ptr(), eps.data_ptr(), n, C.byref(p), flag.data_ptr(), st) == 0
        def phil(): assert L.mcmc_update_philox(x.data_ptr(), v.data_ptr(), g.data_ptr(), xt.data_ptr(), pr.data_ptr(), n, C.byref(p), seeds.data_ptr(), n // B, 5, None, 0, None, flag.data_ptr(), st) == 0
        for name, fn in (("plain", plain), ("philox", phil)):
            for _ in range(5): fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in ev: a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            print('B=%-2d algo %d momentum %.1f %-6s %.1f us' % (B, algo, mom, name, ts[len(ts) // 2] * 1e3), flush=True)
