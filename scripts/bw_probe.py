"""What this box's HBM delivers to simple torch kernels (yardstick for the DFT kernels)."""
import torch
dev = 'cuda:0'
x = torch.rand(64, 64, 256, 256, device=dev)
y = torch.empty_like(x)
def t(fn, n=200):
    for _ in range(50): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
gb = x.numel() * 4 / 1e9
m = t(lambda: x.sum()); print('sum (read %.2f GB): %.3f ms  %.0f GB/s' % (gb, m, gb / m * 1e3))
m = t(lambda: x.view(-1, 256).sum(0)); print('column sum: %.3f ms  %.0f GB/s' % (m, gb / m * 1e3))
m = t(lambda: y.fill_(1.0)); print('fill (write): %.3f ms  %.0f GB/s' % (m, gb / m * 1e3))
m = t(lambda: y.copy_(x)); print('copy (read+write): %.3f ms  %.0f GB/s' % (m, 2 * gb / m * 1e3))
m = t(lambda: torch.mul(x, 2.0, out=y)); print('mul: %.3f ms  %.0f GB/s' % (m, 2 * gb / m * 1e3))
