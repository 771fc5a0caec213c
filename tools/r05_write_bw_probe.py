"""Pure-write, pure-read and copy bandwidth of this device with torch's own elementwise kernels (1 GiB fp32 buffers): is a kernel
that only WRITES (the volume build's feature kernel: 401 MB out, 37 MB in) held by a write ceiling below the copy rate?"""
import torch

dev = torch.device("cuda:0")
n = 1 << 28                                   # 1 GiB of fp32
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: x.fill_(1.0))
print("fill  (write 1 GiB):        %.3f ms = %.2f TB/s written" % (ms, n * 4 / ms / 1e9))
ms = timed(lambda: x.zero_())
print("zero  (write 1 GiB):        %.3f ms = %.2f TB/s written" % (ms, n * 4 / ms / 1e9))
ms = timed(lambda: y.copy_(x))
print("copy  (read + write 1 GiB): %.3f ms = %.2f TB/s moved" % (ms, 2 * n * 4 / ms / 1e9))
ms = timed(lambda: torch.sum(x))
print("sum   (read 1 GiB):         %.3f ms = %.2f TB/s read" % (ms, n * 4 / ms / 1e9))
m = 100 * (1 << 20) // 4                      # 100 MiB: the size class of the volume (401 MB) is between the two
xs = torch.empty(4 * m, device=dev)
ms = timed(lambda: xs.fill_(1.0))
print("fill  (write 400 MiB):      %.3f ms = %.2f TB/s written" % (ms, 4 * m * 4 / ms / 1e9))
