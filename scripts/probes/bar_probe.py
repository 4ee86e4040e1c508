import os, time, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl")
x = torch.zeros(1, device='cuda')
for i in range(3): dist.barrier()
torch.cuda.synchronize()
for name, fn in (("dist.barrier()", lambda: dist.barrier()),
                 ("barrier(device_ids)", lambda: dist.barrier(device_ids=[0])),
                 ("all_reduce tiny + sync", lambda: (dist.all_reduce(x), torch.cuda.synchronize()))):
    ts = []
    for i in range(10):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, ' '.join('%.3f' % t for t in ts), flush=True)
dist.destroy_process_group()
