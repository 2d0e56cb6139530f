import time, torch
torch.cuda.init(); x = torch.randn(1 << 20, device="cuda")
def t(f, n=20000):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("current_device idle us", t(torch.cuda.current_device))
print("_cuda_getDevice idle us", t(torch._C._cuda_getDevice))
print("current_stream us", t(lambda: torch.cuda.current_stream()))
print("_cuda_getCurrentRawStream us", t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("torch.empty us", t(lambda: torch.empty((30, 4), device="cuda"), 5000))
for _ in range(200): y = x * 2
print("current_device busy us", t(torch.cuda.current_device, 2000))
torch.cuda.synchronize()
