"""Round 3: do kernels of two streams run SIDE BY SIDE on this stack?  A compute-bound kernel (fp32 matmul, MFMA) on one stream and a memory-bound one (a large copy) on another,
alone, one after the other in one stream, and on two streams at once; and the same with each kernel limited to fewer workgroups than the chip holds (so that there IS room)."""
import json
import time

import torch

dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev)
b = torch.randn(8192, 8192, device=dev)
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def mm():
    with torch.cuda.stream(s1):
        for _ in range(4):
            torch.matmul(a, b)


def cp():
    with torch.cuda.stream(s2):
        for _ in range(4):
            dst.copy_(src)


def both_one_stream():
    with torch.cuda.stream(s1):
        for _ in range(4):
            torch.matmul(a, b)
        for _ in range(4):
            dst.copy_(src)


def both_two_streams():
    mm(); cp()


print(json.dumps({"matmul_alone_ms": round(timed(mm), 3), "copy_alone_ms": round(timed(cp), 3), "one_stream_ms": round(timed(both_one_stream), 3), "two_streams_ms": round(timed(both_two_streams), 3)}))
