"""HBM fill / copy / read bandwidth of the box with plain torch ops on a 1.67 GB buffer (the size of the record
tables): the yardstick for sr_prep_kernel's stores (fill 6.2, copy 5.0, read 6.0 TB/s)."""
import torch, time
x = torch.empty(1670000000 // 8, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for name, fn, nbytes in (("fill", lambda: x.fill_(1.0), x.numel()*8), ("copy", lambda: y.copy_(x), 2*x.numel()*8), ("read(sum)", lambda: x.sum(), x.numel()*8)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%s: %.3f ms  %.2f TB/s" % (name, dt*1e3, nbytes/dt/1e12))
