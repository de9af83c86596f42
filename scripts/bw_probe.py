import torch, time
d = torch.device("cuda:0")
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for mb in (77.5, 310, 1240):
    n = int(mb * 1e6 / 2)
    x = torch.empty(n, dtype=torch.bfloat16, device=d); y = torch.empty_like(x)
    us = t(lambda: x.zero_()); print(f"{mb:7.1f} MB zero_   : {us:8.1f} us  {mb*1e6/us/1e6:7.1f} GB/s write")
    us = t(lambda: x.fill_(1.5)); print(f"{mb:7.1f} MB fill_   : {us:8.1f} us  {mb*1e6/us/1e6:7.1f} GB/s write")
    us = t(lambda: y.copy_(x)); print(f"{mb:7.1f} MB copy_   : {us:8.1f} us  {2*mb*1e6/us/1e6:7.1f} GB/s read+write")
    us = t(lambda: torch.sum(x)); print(f"{mb:7.1f} MB sum     : {us:8.1f} us  {mb*1e6/us/1e6:7.1f} GB/s read")
    z = x.view(-1, 768)[:, :192]  # strided rows
