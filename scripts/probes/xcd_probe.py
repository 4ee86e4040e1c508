import torch, ctypes as C
L = C.CDLL('variants/xcd_probe.so')
L.xcd_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
n = 1 << 28
buf = torch.zeros(n, device='cuda:0'); sink = torch.zeros(1, device='cuda:0')
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=100):
    for _ in range(30): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rd in (0, 1):
    print('reads' if rd else 'writes', '(TB/s; rows: chunk size, columns: k = 0..7)')
    for S in (11, 12, 13, 14, 16, 18, 20, 22):
        r = []
        for k in range(8):
            m = t(lambda: L.xcd_probe(buf.data_ptr(), n * 4, S, k, rd, sink.data_ptr(), st))
            r.append(n * 4 / m / 1e9)
        print('%8d B ' % (1 << S) + ' '.join('%5.2f' % v for v in r), flush=True)
