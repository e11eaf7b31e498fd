"""Where does a gloo all-reduce of CUDA tensors stop on this box?  2 ranks on cuda:0, stage by stage, each stage printed.
Run: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/exp/gloo_cuda_check.py [stages]"""
import os, sys, time, faulthandler, torch, torch.distributed as dist
faulthandler.dump_traceback_later(int(os.environ.get('WATCHDOG', '45')), exit=True)
import threading
def _dump_tasks():
    # what every thread of this process is blocked in, from /proc (no debugger): comm, kernel wait channel, current syscall
    time.sleep(int(os.environ.get('WATCHDOG', '45')) - 8)
    pid = os.getpid()
    for tid in sorted(os.listdir('/proc/%d/task' % pid), key=int):
        def rd(f):
            try: return open('/proc/%d/task/%s/%s' % (pid, tid, f)).read().strip()
            except Exception as e: return 'n/a'
        print('[r%s task %s] comm=%s wchan=%s syscall=%s' % (os.environ.get('RANK'), tid, rd('comm'), rd('wchan'), rd('syscall')[:60]), flush=True)
threading.Thread(target=_dump_tasks, daemon=True).start()
dist.init_process_group('gloo')
r = dist.get_rank()
torch.cuda.set_device(0)
stages = sys.argv[1].split(',') if len(sys.argv) > 1 else ['sync', 'async', 'side', 'bucketed', 'busy', 'host']
def say(*a):
    print('[r%d %.2f]' % (r, time.time() % 1000), *a, flush=True)
x = torch.full((1 << 20,), float(r + 1), device='cuda')
if 'sync' in stages:
    dist.all_reduce(x); torch.cuda.synchronize(); say('sync ok', float(x[0]))
if 'async' in stages:
    h = dist.all_reduce(x, async_op=True); h.wait(); torch.cuda.synchronize(); say('async ok', float(x[0]))
s = torch.cuda.Stream()
if 'side' in stages:
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); s.wait_event(ev)
    with torch.cuda.stream(s):
        h = dist.all_reduce(x, async_op=True)
    h.wait(); torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize(); say('side-stream async ok', float(x[0]))
big = torch.ones(120 << 20, device='cuda')
if 'bucketed' in stages:
    hs = []
    for n, i in enumerate(range(0, big.numel(), 2 << 20)):          # 60 handles of 8 MB in flight, like the 55 of the LUT step
        with torch.cuda.stream(s):
            hs.append(dist.all_reduce(big[i:i + (2 << 20)], async_op=True))
        if n % 10 == 0: say('bucketed: launched', n)
    for n, h in enumerate(hs):
        h.wait()
        if n % 10 == 0: say('bucketed: waited', n)
    torch.cuda.synchronize(); say('bucketed ok', float(big[0]), float(big[-1]))
for st in stages:
    if st.startswith('inflight'):
        # at most k async all-reduces of device tensors in flight
        k = int(st[len('inflight'):]); hs = []
        for n, i in enumerate(range(0, big.numel(), 2 << 20)):
            with torch.cuda.stream(s):
                hs.append(dist.all_reduce(big[i:i + (2 << 20)], async_op=True))
            if len(hs) >= k:
                hs.pop(0).wait()
        for h in hs: h.wait()
        torch.cuda.synchronize(); say('inflight', k, 'ok', float(big[0]))
if 'busy' in stages:
    # the same while the GPU is busy on the main stream and the side stream waits for main-stream events (the bucketer's pattern)
    a = torch.randn(4096, 4096, device='cuda'); hs = []
    for n, i in enumerate(range(0, big.numel(), 2 << 20)):
        for _ in range(4): a = (a @ a).clamp_(-1, 1)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); s.wait_event(ev)
        with torch.cuda.stream(s):
            hs.append(dist.all_reduce(big[i:i + (2 << 20)], async_op=True))
        if n % 10 == 0: say('busy: launched', n)
    for n, h in enumerate(hs):
        h.wait()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize(); say('busy ok', float(big[0]))
if 'host' in stages:
    # staging by hand: gloo only ever sees host tensors
    pin = torch.empty(2 << 20, dtype=torch.float32).pin_memory()
    for n, i in enumerate(range(0, big.numel(), 2 << 20)):
        with torch.cuda.stream(s):
            pin.copy_(big[i:i + (2 << 20)], non_blocking=True)
        s.synchronize()
        dist.all_reduce(pin)
        with torch.cuda.stream(s):
            big[i:i + (2 << 20)].copy_(pin, non_blocking=True)
        s.synchronize()
    torch.cuda.synchronize(); say('host-staged ok', float(big[0]))
dist.destroy_process_group()
say('done')
