"""Correctness + timing of alo_hip.conv3x3 against MIOpen on the ResNet-50 bottleneck shapes (B=8, 640x640 -> padded 667)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
import alo_hip

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

torch.manual_seed(0)
shapes = [(8, 128, 100, 167), (8, 256, 50, 84), (8, 512, 25, 42), (2, 128, 7, 9), (1, 256, 5, 70)]
for (n, c, h, w) in shapes:
    x = torch.randn(n, c, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(c, c, 3, 3, device="cuda") * (1.0 / (9 * c) ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(c, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        ref = F.relu(F.conv2d(x.float(), wt.float(), b.float(), 1, 1))
        got = alo_hip.conv3x3(x, wt, b, relu=True)
        stock = F.relu(F.conv2d(x, wt, b, 1, 1))
        err = (got.float() - ref).abs().max().item()
        err_stock = (stock.float() - ref).abs().max().item()
        t_mine = timeit(lambda: alo_hip.conv3x3(x, wt, b, relu=True))
        t_stock = timeit(lambda: F.conv2d(x, wt, None, 1, 1))
    gf = 2.0 * 9 * c * c * n * h * w
    print(f"{(n,c,h,w)}: err {err:.4f} (stock {err_stock:.4f})  mine {t_mine:.1f} us ({gf/t_mine/1e6:.0f} TF/s)  miopen {t_stock:.1f} us", flush=True)
