#!/usr/bin/env python3
"""What a tuned dense bf16 GEMM reaches on this box (torch.matmul -> hipBLASLt): the practical MFMA ceiling under the chip's power
limit, to read the convolution kernel's TFLOP/s against (the 2.5 PF figure is the data-sheet peak at the maximum clock)."""
import torch

for (m, n, k) in ((8192, 8192, 8192), (16384, 8192, 4096), (1081600 // 2, 64, 576), (278784, 128, 1152), (73984, 256, 2304)):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(k, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"M={m:7d} N={n:5d} K={k:5d}: {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:7.0f} TFLOP/s", flush=True)
