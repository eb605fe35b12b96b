"""HBM / Infinity-Cache probe with stock torch streaming kernels (calibrates what 'achievable' means on this box).
For a range of buffer sizes S: fill (write S), copy (read S + write S), read-after-write (sum of a buffer the
previous kernel just wrote), read-cold (sum of a buffer after 1 GB of other traffic)."""
import torch, sys
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
big = torch.empty(1 << 28, device=dev)   # 1 GB thrash buffer
print("size_MB fill_TBs copy_TBs(R+W) sum_hot_TBs sum_after_write_TBs sum_cold_TBs")
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    fill = t(lambda: b.fill_(1.0))
    copy = t(lambda: torch.add(a, 1.0, out=b))
    hot = t(lambda: a.sum())
    # read right after the buffer was written (producer -> consumer through the memory-side cache?)
    def raw():
        torch.add(a, 1.0, out=b); return b.sum()
    rw = t(raw) - copy
    def cold():
        big.fill_(0.0); return a.sum()
    cd = t(cold) - t(lambda: big.fill_(0.0))
    S = mb * (1 << 20) / 1e9
    print(f"{mb:5d} {S/fill:8.2f} {2*S/copy:8.2f} {S/hot:8.2f} {S/max(rw,1e-6):8.2f} {S/max(cd,1e-6):8.2f}", flush=True)
