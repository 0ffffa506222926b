# dev only: per-stage cycle counts of msda_bwd_tiled_kernel from a build that writes s_memtime deltas into grad_attn (tools/exp/v_trace.so)
import sys, torch
sys.path.insert(0, "aloception-oss_amd"); sys.path.insert(0, "tools")
import alo_hip, kbench
_, _, S = kbench.detr_geometry()
value, shapes, start, loc, attn = kbench.msda_inputs(4, S, "encoder", torch.float32)
go = torch.randn(4, S, 8 * 32, device="cuda")
for _ in range(3):
    gv, gl, ga = alo_hip.msda_backward(value, shapes, start, loc, attn, go)
torch.cuda.synchronize()
ga = ga.reshape(4, S, 8, 4, 4)
t = ga[:, :, :, 0, :]      # head, A-build, stage 2, stage 3 (summed over passes)
u = ga[:, :, :, 1, :]      # last pass, whole tile, passes, corner-route flag
names = ["head (loads, taps, windows, plan)", "zero + row table + A build", "stage 2 (dV, atomics)", "stage 3 (value rows, D)"]
for i, n in enumerate(names):
    print(f"{n:40s} {t[..., i].mean().item():10.0f} cycles")
print(f"{'whole tile (before the per-corner route)':40s} {u[..., 1].mean().item():10.0f} cycles;  passes {u[..., 2].mean().item():.2f};  tiles with a per-corner level {u[..., 3].mean().item():.3f}")
lv0 = slice(0, 16700)
print("level-0 queries only: ", [round(t[:, lv0, :, i].mean().item()) for i in range(4)], round(u[:, lv0, :, 1].mean().item()))
