"""Two training steps under DistributedDataParallel with world_size 2 on gloo (CPU): gradient all-reduce, the
criterion's all_reduce(num_boxes); the DDP-averaged gradients must equal the single-process gradients of the CONCATENATED
batch (both ranks' frames in one forward), and both ranks must hold identical parameters after the step.  The attention runs through the
reference's differentiable pure-torch branch (``is_tracing``) because the HIP op has no CPU implementation."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(%r, "aloception-oss_amd"))
    import aloscene
    from alonet.deformable_detr import DeformableDETR, DeformableTransformer
    from alonet.deformable_detr.backbone import Joiner, Backbone
    from alonet.deformable_detr.training import build_criterion, configure_optimizers
    from alonet.transformers import PositionEmbeddingSine

    dist.init_process_group("gloo")
    rank = dist.get_rank()

    def build():
        torch.manual_seed(0)  # same initial weights everywhere
        backbone = Joiner(Backbone("resnet50", True, True, False), PositionEmbeddingSine(32, normalize=True, center=True))
        transformer = DeformableTransformer(d_model=64, nhead=4, num_encoder_layers=1, num_decoder_layers=2,
                                            dim_feedforward=64, dropout=0.0, return_intermediate_dec=True)
        return DeformableDETR(backbone, transformer, num_classes=5, num_queries=12, aux_loss=True, device=None).train()

    names = [f"c{i}" for i in range(5)]

    def shard(r):   # the frame rank r owns
        gen = torch.Generator().manual_seed(100 + r)
        n_boxes = 1 + r
        lab = aloscene.Labels(torch.arange(n_boxes).float(), encoding="id", labels_names=names)
        bx = aloscene.BoundingBoxes2D(torch.rand(n_boxes, 4, generator=gen) * 0.3 + 0.3, "xcyc", False, labels=lab)
        return aloscene.Frame(torch.rand(3, 64, 96, generator=gen) * 255, normalization="255", boxes2d=bx).norm_resnet()

    model = build()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    frames = aloscene.Frame.batch_list([shard(rank)])
    crit, opt = build_criterion(aux_loss_stage=2), configure_optimizers(model)
    opt.zero_grad()
    total, parts = crit(ddp(frames, is_tracing=None), frames)
    total.backward()
    ddp_grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    # single-process reference: BOTH frames in one batch through an identical, un-wrapped model.  (Every rank runs it, so the
    # criterion's all_reduce(num_boxes) sums 3 + 3 and divides by the world size: the batch's own 3 boxes.)
    ref = build()
    both = aloscene.Frame.batch_list([shard(0), shard(1)])
    ref_total, _ = build_criterion(aux_loss_stage=2)(ref(both, is_tracing=None), both)
    ref_total.backward()
    worst = 0.0
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert n not in ddp_grads, n
            continue
        scale = max(1e-6, float(p.grad.abs().max()))
        worst = max(worst, float((ddp_grads[n] - p.grad).abs().max()) / scale)
    opt.step()
    # a SECOND step through the same DDP wrapper: with find_unused_parameters=False the reducer raises at this forward if any
    # trainable parameter went without a gradient in the first one
    opt.zero_grad()
    total2, _ = crit(ddp(frames, is_tracing=None), frames)
    total2.backward()
    opt.step()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(2)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        print("SAME" if torch.equal(gathered[0], gathered[1]) else "DIFFERENT", float(total))
        print("GRADDIFF", worst)
    dist.destroy_process_group()
''') % ROOT


def test_ddp_training_step_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29633", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith(("SAME", "DIFFERENT"))]
    assert line and line[0].startswith("SAME"), out.stdout[-2000:]
    diff = [float(ln.split()[1]) for ln in out.stdout.splitlines() if ln.startswith("GRADDIFF")]
    # DDP's mean of the per-rank gradients == gradient of the concatenated batch (relative to each tensor's largest entry)
    assert diff and diff[0] <= 2e-4, out.stdout[-2000:]
