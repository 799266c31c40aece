import torch
x = torch.empty(1<<30, device='cuda')   # 4 GB
y = torch.empty(1<<30, device='cuda')
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
tf = t(lambda: x.fill_(1.0)); print('fill 4 GB: %.3f ms %.2f TB/s' % (tf, 4.295/tf))
tz = t(lambda: x.zero_()); print('zero 4 GB: %.3f ms %.2f TB/s' % (tz, 4.295/tz))
tc = t(lambda: y.copy_(x)); print('copy 4+4 GB: %.3f ms %.2f TB/s' % (tc, 8.59/tc))
ts = t(lambda: x.sum()); print('sum 4 GB: %.3f ms %.2f TB/s' % (ts, 4.295/ts))
