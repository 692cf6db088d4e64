"""Does the 256 MB Infinity Cache keep what a kernel wrote LAST?  Fill a buffer of S MB front to back, then read either its first or
its last R MB: if the memory-side cache retains the most recent writes, the tail reads faster than the head (and a consumer that walks a
producer's output BACKWARDS would take part of it out of the cache instead of out of HBM).   python tools/probe/mall_recency_probe.py"""
import torch
dev = "cuda"
for S in (192, 400, 800):
    for R in (64, 128, 192):
        if R > S:
            continue
        n, r = S * 1024 * 1024 // 4, R * 1024 * 1024 // 4
        x = torch.empty(n, dtype=torch.float32, device=dev)
        junk = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=dev)
        res = {}
        for which in ("head", "tail", "head", "tail"):
            ts = []
            for rep in range(5):
                x.fill_(1.0)                       # the producer: front to back
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                v = x[:r] if which == "head" else x[n - r:]
                e0.record()
                s = v.sum()                        # the consumer: R MB
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res.setdefault(which, []).append(sorted(ts)[2])
        print("buffer %4d MB written front to back, then %3d MB read: head %7.1f us (%.2f TB/s)   tail %7.1f us (%.2f TB/s)" % (
            S, R, min(res["head"]), R * 1.048576 / min(res["head"]), min(res["tail"]), R * 1.048576 / min(res["tail"])))

# do READS allocate in the cache?  write X (192 MB: fits), stream-read an unrelated 512 MB buffer, then read X again
n = 192 * 1024 * 1024 // 4
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.ones(512 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
for label, between in (("nothing in between", None), ("512 MB READ in between", "read"), ("512 MB WRITTEN in between", "write")):
    ts = []
    for rep in range(5):
        x.fill_(1.0)
        if between == "read":
            y.sum()
        elif between == "write":
            y.fill_(2.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s = x.sum(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("192 MB written, %-26s then read: %7.1f us (%.2f TB/s)" % (label + ",", sorted(ts)[2], 192 * 1.048576 / sorted(ts)[2]))
# read-after-read: is a buffer that was only READ retained?
for S in (192, 400):
    n = S * 1024 * 1024 // 4
    x = torch.ones(n, dtype=torch.float32, device=dev)
    y.fill_(3.0); torch.cuda.synchronize()            # flush the cache with writes
    ts = []
    for rep in range(5):
        y.fill_(3.0); x.sum(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r = 128 * 1024 * 1024 // 4
        e0.record(); s = x[n - r:].sum(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("%d MB READ front to back, then its last 128 MB read again: %7.1f us (%.2f TB/s)" % (S, sorted(ts)[2], 128 * 1.048576 / sorted(ts)[2]))
