"""Dev probe: does the 256 MB Infinity Cache serve a second streaming read of a buffer another kernel has just read (or written)?
For several sizes: flush (read 2 GB of other data), read the buffer (cold), read it again (warm); and write it, then read it."""
import torch

dev = torch.device("cuda:0")
flush = torch.empty(2 << 30, dtype=torch.uint8, device=dev).view(torch.float32)
flush.fill_(1.0)


def t_ms(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for mb in (64, 128, 192, 256, 384, 704):
    x = torch.empty(mb << 20, dtype=torch.uint8, device=dev).view(torch.float32)
    x.fill_(2.0)
    res = []
    for _ in range(3):
        flush.sum()
        torch.cuda.synchronize()
        cold = t_ms(lambda: x.sum())
        warm = t_ms(lambda: x.sum())
        flush.sum()
        torch.cuda.synchronize()
        x.mul_(1.0)  # read + write
        torch.cuda.synchronize()
        after_write = t_ms(lambda: x.sum())
        res.append((cold, warm, after_write))
    c, w, a = (sorted(r[i] for r in res)[1] for i in range(3))
    gb = mb / 1024
    print(f"{mb:4d} MB: cold read {c * 1e3:7.1f} us ({gb / c * 1e3:6.0f} GB/s) | second read {w * 1e3:7.1f} us ({gb / w * 1e3:6.0f} GB/s) | read after write {a * 1e3:7.1f} us ({gb / a * 1e3:6.0f} GB/s)")
