"""Does gloo all-reduce CUDA tensors on this box (sync, async, async on a side stream)?  2 ranks on cuda:0."""
import os, sys, faulthandler, torch, torch.distributed as dist
faulthandler.dump_traceback_later(50, exit=True)
dist.init_process_group('gloo')
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.full((1 << 20,), float(r + 1), device='cuda')
dist.all_reduce(x); torch.cuda.synchronize(); print(r, 'sync ok', float(x[0]), flush=True)
h = dist.all_reduce(x, async_op=True); h.wait(); torch.cuda.synchronize(); print(r, 'async ok', float(x[0]), flush=True)
s = torch.cuda.Stream()
ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); s.wait_event(ev)
with torch.cuda.stream(s):
    h = dist.all_reduce(x, async_op=True)
h.wait(); torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize(); print(r, 'side-stream async ok', float(x[0]), flush=True)
big = torch.ones(120 << 20, device='cuda')
hs = []
for i in range(0, big.numel(), 8 << 20):
    with torch.cuda.stream(s):
        hs.append(dist.all_reduce(big[i:i + (8 << 20)], async_op=True))
for h in hs: h.wait()
torch.cuda.synchronize(); print(r, 'bucketed ok', float(big[0]), float(big[-1]), flush=True)
dist.destroy_process_group()
