#!/usr/bin/env python3
"""Keeps the GPU busy from a second process (short bursts of copies and matrix products with
pauses in between) while another command runs - a timing perturbation for the tests: kernels
whose workgroups depend on one another's progress only show it when CUs are taken away from
them.  usage: tools_gpu_background_load.py <seconds>"""
import sys
import time

import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda", 0)
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
big = torch.empty(1 << 28, device=dev)  # 1 GiB of floats
t_end = time.time() + secs
i = 0
while time.time() < t_end:
    for _ in range(4):
        c = a @ b
    big.add_(1.0)
    torch.cuda.synchronize()
    i += 1
    if i % 7 == 0:
        time.sleep(0.003)
print("background load done, %d bursts" % i)
