"""How much of the detection step is inference() (device-to-host hand-over + host-side object construction)?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    model = bench.build_detector(device, torch.bfloat16)
    frames = bench.detection_inputs(8, 0, device, torch.bfloat16)
    from alonet.common import GraphedForward
    graphed = GraphedForward(model, adopt_inputs=True)
    with torch.no_grad():
        graphed(frames)

        def t(fn, n=40):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        def fwd_only():
            graphed(frames)
            torch.cuda.synchronize()

        def fwd_inf():
            return model.inference(graphed(frames))

        out = graphed(frames)
        torch.cuda.synchronize()

        def inf_only():
            return model.inference(out)

        print("forward (graph replay + sync)   ms", round(t(fwd_only), 3))
        print("forward + inference()           ms", round(t(fwd_inf), 3))
        print("inference() alone on ready outs ms", round(t(inf_only), 3))
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            inf_only()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
