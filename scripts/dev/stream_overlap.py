#!/usr/bin/env python3
"""Development: do kernels of two HIP streams overlap on this box?  torch.cuda._sleep (one spinning thread) on one stream, on two
streams, and a GPU-filling elementwise kernel beside a sleeping one."""
import time
import torch

dev = torch.device("cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cycles = 20_000_000


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def one():
    with torch.cuda.stream(s1):
        torch.cuda._sleep(cycles)


def two():
    with torch.cuda.stream(s1):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(s2):
        torch.cuda._sleep(cycles)


x = torch.randn(256 * 1024 * 1024 // 4, device=dev)


def big():
    with torch.cuda.stream(s2):
        for _ in range(8):
            x.mul_(1.0001)


def both():
    with torch.cuda.stream(s1):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(s2):
        for _ in range(8):
            x.mul_(1.0001)


print("sleep on one stream        %.2f ms" % timed(one))
print("sleep on two streams       %.2f ms  (overlap: same as one; serialised: twice)" % timed(two))
print("8 streaming kernels        %.2f ms" % timed(big))
print("sleep + streaming kernels  %.2f ms  (overlap: the max of the two; serialised: the sum)" % timed(both))
