import torch
def med(fn, it=20, warm=5):
    for _ in range(warm): fn()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a,b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in ev); return t[len(t)//2]
for gb in (1, 4, 12.85, 51.4, 102.8):
    n=int(gb*1e9/4)
    x=torch.empty(n, dtype=torch.float32, device="cuda")
    t=med(lambda: x.fill_(1.0))
    print(f"torch fill_ {gb} GB: {t:.3f} ms {n*4/t/1e9:.2f} TB/s", flush=True)
    t=med(lambda: x.zero_())
    print(f"torch zero_ {gb} GB: {t:.3f} ms {n*4/t/1e9:.2f} TB/s", flush=True)
    del x
