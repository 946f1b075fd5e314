"""Streaming write / copy bandwidth of the device at the cost volumes' sizes (what bounds a volume writer):
torch fill_ (pure write) and copy_ (read + write) at 84 ... 1344 MB."""
import torch
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for mb in (42, 84, 168, 252, 336, 504, 672, 1344):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    tf = timed(lambda: a.fill_(1.0))
    tc = timed(lambda: b.copy_(a))
    print(f"{mb:5d} MB  fill {tf*1e3:7.1f} us {mb*1.048576/tf/1e3:6.2f} TB/s   copy {tc*1e3:7.1f} us {2*mb*1.048576/tc/1e3:6.2f} TB/s (r+w)", flush=True)
