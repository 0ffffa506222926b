# dev only: is the correlation build power-bound?  Same launch, operands of different switching activity.
import sys, torch
sys.path.insert(0, "aloception-oss_amd"); sys.path.insert(0, "tools")
import alo_hip, kbench
f1, f2 = kbench.corr_inputs(4)
for name, (a, b) in {"randn": (f1, f2), "zeros": (torch.zeros_like(f1), torch.zeros_like(f2)),
                     "ones": (torch.ones_like(f1), torch.ones_like(f2)),
                     "pow2": (torch.full_like(f1, 0.5), torch.full_like(f2, 2.0))}.items():
    t = kbench.time_launches(lambda: alo_hip.corr_build(a, b, 4), 30)
    print(name, round(t * 1e3, 4), "ms")
