import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for mb in (413, 826, 2000):
    x=torch.empty(mb*1000*1000//4, dtype=torch.float32, device="cuda"); y=torch.empty_like(x)
    us=t(lambda: x.fill_(1.0)); print(f"fill {mb} MB: {us:.1f} us = {mb/us*1e3/1e3:.2f} TB/s write")
    us=t(lambda: y.copy_(x)); print(f"copy {mb} MB: {us:.1f} us = {2*mb/us*1e3/1e3:.2f} TB/s r+w")
    us=t(lambda: x.sum()); print(f"sum  {mb} MB: {us:.1f} us = {mb/us*1e3/1e3:.2f} TB/s read")
