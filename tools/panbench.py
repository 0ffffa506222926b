"""PanopticHead's FPN-style decoder convolutions (B*Q = 128 maps): MIOpen vs alo_hip.conv3x3 with channels zero-padded to 64."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
import alo_hip

def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

pad64 = lambda c: (c + 63) // 64 * 64
torch.manual_seed(0)
for (cin, cout, h, w) in [(264, 264, 25, 42), (264, 128, 25, 42), (128, 64, 50, 84), (64, 32, 100, 167), (32, 16, 200, 334), (16, 1, 200, 334)]:
    n = 128
    x = torch.randn(n, cin, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda") / (9 * cin) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        t_stock = timeit(lambda: F.conv2d(x, wt, b, 1, 1))
        cip, cop = pad64(cin), pad64(cout)
        xp = torch.zeros(n, cip, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last); xp[:, :cin] = x
        wp = torch.zeros(cop, cip, 3, 3, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last); wp[:cout, :cin] = wt
        bp = torch.zeros(cop, device="cuda", dtype=torch.bfloat16); bp[:cout] = b
        got = alo_hip.conv3x3(xp, wp, bp)[:, :cout]
        ref = F.conv2d(x.float(), wt.float(), b.float(), 1, 1)
        err = (got.float() - ref).abs().max().item()
        t_mine = timeit(lambda: alo_hip.conv3x3(xp, wp, bp))
        t_pad = timeit(lambda: xp[:, :cin].copy_(x))
    print(f"{cin}->{cout} @ {h}x{w}: miopen {t_stock:.0f} us   padded conv3x3 {t_mine:.0f} us (+ pad copy {t_pad:.0f} us)  err {err:.4f}", flush=True)
