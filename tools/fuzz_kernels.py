"""Randomised parity sweep of the convolution / norm / geometry kernels against PyTorch (dev tool; run on the GPU box)."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
import torch
import torch.nn.functional as F
import alo_hip

random.seed(0); torch.manual_seed(0)
dev = "cuda"
bad = 0
def check(name, got, ref, tol):
    global bad
    err = (got.float() - ref.float()).abs().max().item() if got.numel() else 0.0
    lim = tol * max(1.0, ref.float().abs().max().item() if ref.numel() else 1.0)
    if not (err <= lim) or got.shape != ref.shape:
        bad += 1
        print("MISMATCH", name, "err", err, "lim", lim, tuple(got.shape), tuple(ref.shape), flush=True)

with torch.no_grad():
    for it in range(60):
        n = random.choice([1, 2, 3]); cin = random.choice([64, 128, 192, 256]); cout = random.choice([64, 128, 320])
        h = random.choice([1, 2, 3, 5, 8, 17, 33]); w = random.choice([1, 2, 3, 4, 7, 31, 64, 65, 130]); st = random.choice([1, 2])
        x = torch.randn(n, cin, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5).bfloat16()
        b = torch.randn(cout, device=dev).bfloat16()
        relu = random.random() < 0.5
        ref = F.conv2d(x.float(), wt.float(), b.float(), st, 1)
        ref = F.relu(ref) if relu else ref
        check(f"conv3x3 {n,cin,cout,h,w,st}", alo_hip.conv3x3(x, wt, b, relu=relu, stride=st), ref, 2 ** -8)
    for it in range(30):
        n = random.choice([1, 2]); h = random.choice([1, 2, 6, 7, 8, 15, 31, 64, 97]); w = random.choice([1, 3, 7, 8, 29, 56, 57, 130])
        x = torch.randn(n, 3, h, w, device=dev).bfloat16()
        if random.random() < 0.5: x = x.contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(64, 3, 7, 7, device=dev) / 147 ** 0.5).bfloat16(); b = torch.randn(64, device=dev).bfloat16()
        ref = F.max_pool2d(F.relu(F.conv2d(x.float(), wt.float(), b.float(), 2, 3)), 3, 2, 1)
        check(f"stem {n,h,w}", alo_hip.stem_conv_pool(x, wt, b), ref, 2 ** -8)
    for it in range(30):
        B = random.choice([1, 2, 5]); HW = random.choice([1, 2, 255, 256, 257, 1000]); C, G = random.choice([(256, 32), (128, 8), (64, 8), (512, 32), (256, 16)])
        x = (torch.randn(B, HW, C, device=dev) * 2 + 0.3).bfloat16(); wt = torch.randn(C, device=dev).bfloat16(); b = torch.randn(C, device=dev).bfloat16()
        ref = F.group_norm(x.float().transpose(1, 2), G, wt.float(), b.float(), 1e-5).transpose(1, 2)
        check(f"groupnorm {B,HW,C,G}", alo_hip.groupnorm_rows(x, wt, b, G, 1e-5), ref, 2 ** -7)
    for it in range(30):
        M = random.choice([1, 63, 64, 65, 1000]); K = random.choice([512, 768, 1024, 2048]); N = random.choice([128, 256, 384, 1024])
        x = torch.randn(M, K, device=dev).bfloat16(); wt = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        r = torch.randn(M, N, device=dev).bfloat16() if random.random() < 0.5 else None
        ref = F.linear(x.float(), wt.float(), b.float())
        if r is not None: ref = ref.bfloat16().float() + r.float()
        check(f"linear_packed {M,K,N}", alo_hip.linear_packed(x, wt, b, False, residual=r), ref, 2 ** -7)
    for it in range(20):
        n = random.choice([1, 2]); cin = random.choice([64, 128, 256, 512, 1024]); cout = random.choice([128, 256, 1024]); h = random.choice([1, 2, 5, 9, 16]); w = random.choice([1, 3, 8, 13]); st = random.choice([2, 3])
        x = torch.randn(n, cin, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, device=dev) / cin ** 0.5).bfloat16(); b = torch.randn(cout, device=dev).bfloat16()
        if alo_hip.conv1x1_strided_supported(x, wt):
            check(f"conv1x1_strided {n,cin,cout,h,w,st}", alo_hip.conv1x1_strided(x, wt, b, st), F.conv2d(x.float(), wt.float()[:, :, None, None], b.float(), st), 2 ** -8)
print("fuzz done, mismatches:", bad)
