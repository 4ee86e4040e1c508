import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1: os.environ['STEP_AMD_LIB'] = os.path.abspath(sys.argv[1])
import torch
from step_amd import _lib
L = _lib.lib()
n = 64 * 64 * 256 * 256
dev = 'cuda:0'
x = torch.randn(n, device=dev); g = torch.randn(n, device=dev); xt = torch.randn(n, device=dev)
pr = torch.randn(n, device=dev); eps = torch.randn(n, device=dev); vel = torch.zeros(n, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn, k=50):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for name, algo, mom, pp in (('hmc momentum 0, prior', 1, 0.0, pr), ('hmc momentum 0.9, prior', 1, 0.9, pr), ('gd (optimize)', 2, 0.0, None)):
    p = _lib.mcmc_params_t(1.0, 1e-4, 1e-5, 3.2e-3, 4.5e-3, mom, 1.0, algo)
    e = None if algo == 2 else eps.data_ptr()
    fn = lambda: L.mcmc_update(x.data_ptr(), vel.data_ptr(), g.data_ptr(), xt.data_ptr(), None if pp is None else pp.data_ptr(), e, n, C.byref(p), None, st)
    ms = t(fn)
    nstream = (3 + (pp is not None) + (e is not None) + 1) + (2 if mom else 0)
    print('%-28s %.3f ms  %.2f TB/s (%d streams)' % (name, ms, nstream * n * 4 / ms / 1e9, nstream), flush=True)
