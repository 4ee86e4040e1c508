import torch, ctypes as C
L = C.CDLL('variants/store_probe.so')
B = 64
out = torch.empty(B, 64, 256, 256, device='cuda:0')
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=300):
    for _ in range(100): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
gb = out.numel() * 4 / 1e9
for mode, name in ((0, 'adjoint pattern (2 x 128 B per instr)'), (1, 'rows as float4 (1 KB per instr)'), (2, 'rows as dword (256 B per instr)'), (3, '4 waves x 64 columns, adjoint row order'), (4, 'adjoint pattern, one tile of every chain'), (5, 'adjoint pattern, 16 waves per CU'), (6, 'adjoint pattern, 4 waves per CU'), (7, 'adjoint pattern, tiles staggered'), (8, 'adjoint pattern, tiles + rows staggered'), (9, 'adjoint pattern, a workgroup per tile'), (10, 'adjoint pattern, a workgroup per frame'), (11, 'the same, one workgroup per CU'), (12, 'persistent, chains dealt round-robin'), (13, 'persistent, barrier per frame'), (14, 'persistent, barrier per tile'), (15, 'persistent, XCD c writes t % 8 == c'), (16, 'persistent, XCD c writes t / 8 == c'), (17, 'persistent, XCD c: t % 4, chain-group parity'), (18, 'XCD-mapped, frame written linearly'), (19, 'XCD-mapped, tile rows then mirror rows linearly'), (20, 'XCD-mapped adjoint pattern, no pause'), (21, 'the same, 768 idle cycles per tile'), (22, 'the same, 1536 idle cycles per tile'), (23, 'the same, 3072 idle cycles per tile')):
    m = t(lambda: L.store_probe(mode, C.c_void_p(out.data_ptr()), B, C.c_void_p(st)))
    print('%-40s %.3f ms  %.0f GB/s' % (name, m, gb / m * 1e3))
m = t(lambda: out.fill_(1.0)); print('%-40s %.3f ms  %.0f GB/s' % ('torch fill_', m, gb / m * 1e3))
