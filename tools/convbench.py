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
shapes = [(8, 64, 200, 334), (8, 128, 100, 167), (8, 256, 50, 84), (8, 512, 25, 42), (2, 128, 7, 9), (1, 256, 5, 70)]
import itertools
cases = [(s, 1, None) for s in shapes] + [((8, 128, 200, 334), 2, None), ((8, 256, 100, 167), 2, None), ((8, 512, 50, 84), 2, None), ((8, 2048, 25, 42), 2, 256), ((2, 1024, 9, 12), 1, 128), ((2, 128, 9, 11), 2, None), ((1, 64, 8, 8), 2, None)]
for (n, c, h, w), st, co in cases:
    x = torch.randn(n, c, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co or c, c, 3, 3, device="cuda") * (1.0 / (9 * c) ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co or c, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        ref = F.relu(F.conv2d(x.float(), wt.float(), b.float(), st, 1))
        got = alo_hip.conv3x3(x, wt, b, relu=True, stride=st)
        stock = F.relu(F.conv2d(x, wt, b, st, 1))
        err = (got.float() - ref).abs().max().item()
        err_stock = (stock.float() - ref).abs().max().item()
        t_mine = timeit(lambda: alo_hip.conv3x3(x, wt, b, relu=True, stride=st))
        t_stock = timeit(lambda: F.conv2d(x, wt, None, st, 1))
    gf = 2.0 * 9 * c * got.numel()
    print(f"{(n,c,h,w)} s{st}: err {err:.4f} (stock {err_stock:.4f})  mine {t_mine:.1f} us ({gf/t_mine/1e6:.0f} TF/s)  miopen {t_stock:.1f} us", flush=True)

# ---- the stem: conv7x7/s2 + bias + relu + maxpool3x3/s2 -------------------------------------------------------------------
x = torch.randn(8, 3, 800, 1333, device="cuda").to(torch.bfloat16)
wt = (torch.randn(64, 3, 7, 7, device="cuda") / 147 ** 0.5).to(torch.bfloat16)
b = torch.randn(64, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    ref = F.max_pool2d(F.relu(F.conv2d(x[:1].float(), wt.float(), b.float(), 2, 3)), 3, 2, 1)
    got = alo_hip.stem_conv_pool(x, wt, b)
    print("stem err", (got[:1].float() - ref).abs().max().item(), flush=True)
    t_mine = timeit(lambda: alo_hip.stem_conv_pool(x, wt, b))
    xl = x.contiguous(memory_format=torch.channels_last); wl = wt.contiguous(memory_format=torch.channels_last)
    t_stock = timeit(lambda: F.max_pool2d(F.relu_(F.conv2d(xl, wl, b, 2, 3)), 3, 2, 1))
print(f"stem: mine {t_mine:.1f} us  stock (conv+relu+pool, channels_last) {t_stock:.1f} us")
