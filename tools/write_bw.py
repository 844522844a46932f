import torch, time
x = torch.empty(1024*3*53215, dtype=torch.float32, device='cuda')
y = torch.empty_like(x)
for name, fn in (('fill', lambda: x.fill_(1.0)), ('copy', lambda: y.copy_(x)), ('mul', lambda: torch.mul(x, 2.0, out=y))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(name, f'{ms*1e3:.1f} us', f'{x.numel()*4/ms/1e9:.2f} TB/s written')
