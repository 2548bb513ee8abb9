import torch, time
d=torch.device("cuda:0")
n=1<<30   # 1 GiB of bf16 = 2 GiB? use uint8 4 GiB
x=torch.empty(4*(1<<30), dtype=torch.uint8, device=d)
for name,fn in (("fill", lambda: x.fill_(1)), ("zero", lambda: x.zero_())):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(name, "%.2f TB/s"%(x.numel()/dt/1e12))
y=torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print("copy r+w %.2f TB/s"%(2*x.numel()/dt/1e12))
