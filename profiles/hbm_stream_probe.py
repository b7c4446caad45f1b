"""What does HBM streaming give on this box?  torch copy (read + write), fill (write only).  MI355X: 4.79 / 6.88 TB/s."""
import torch, time
d=torch.device('cuda:0')
x=torch.empty(2_000_000_000, dtype=torch.uint8, device=d).random_(0,255)
y=torch.empty_like(x)
for name, fn, bytes_ in (("copy", lambda: y.copy_(x), 4e9), ("read-sum(int32 view)", lambda: x.view(torch.int32).sum(), 2e9), ("fill", lambda: y.fill_(1), 2e9)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(f"{name}: {bytes_/dt/1e12:.2f} TB/s")
