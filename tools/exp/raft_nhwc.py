# dev only: RAFT forward in NCHW (product) vs channels_last, fused glue kernels on / off (inference, fp32, B=4, 1280x720, 32 iters)
import sys, time, torch
sys.path.insert(0, "aloception-oss_amd"); sys.path.insert(0, ".")
import alo_hip
import alonet.raft.update as upd
from alonet.raft import RAFT
import bench

def run(tag, cl, fused):
    torch.manual_seed(0)
    model = RAFT().eval().cuda()
    f1, f2 = bench.flow_inputs(4, 0, torch.device("cuda"))
    orig = upd._fusable
    if not fused:
        upd._fusable = lambda *t: False
    if cl:
        model = model.to(memory_format=torch.channels_last)
        f1, f2 = f1.contiguous(memory_format=torch.channels_last), f2.contiguous(memory_format=torch.channels_last)
    try:
        with torch.no_grad():
            for _ in range(2):
                model(f1, f2, iters=32, only_last=True)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                model(f1, f2, iters=32, only_last=True)
            torch.cuda.synchronize()
            print(tag, round((time.time() - t0) / 3 * 1e3, 2), "ms per forward", flush=True)
    except Exception as exc:
        print(tag, "failed:", type(exc).__name__, str(exc)[:200], flush=True)
    upd._fusable = orig

run("NCHW fused (product)", False, True)
run("NCHW stock glue", False, False)
run("channels_last stock glue", True, False)
run("channels_last fused", True, True)
