"""The training step under DistributedDataParallel with world_size 2 on gloo (CPU): gradient all-reduce, the
criterion's all_reduce(num_boxes), identical parameters on both ranks afterwards.  The attention runs through the
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
    torch.manual_seed(0)  # same initial weights on both ranks
    backbone = Joiner(Backbone("resnet50", True, True, False), PositionEmbeddingSine(32, normalize=True, center=True))
    transformer = DeformableTransformer(d_model=64, nhead=4, num_encoder_layers=1, num_decoder_layers=2,
                                        dim_feedforward=64, dropout=0.0, return_intermediate_dec=True)
    model = DeformableDETR(backbone, transformer, num_classes=5, num_queries=12, aux_loss=True, device=None).train()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    names = [f"c{i}" for i in range(5)]
    gen = torch.Generator().manual_seed(100 + rank)  # different shard per rank
    n_boxes = 1 + rank
    lab = aloscene.Labels(torch.arange(n_boxes).float(), encoding="id", labels_names=names)
    bx = aloscene.BoundingBoxes2D(torch.rand(n_boxes, 4, generator=gen) * 0.3 + 0.3, "xcyc", False, labels=lab)
    fr = aloscene.Frame(torch.rand(3, 64, 96, generator=gen) * 255, normalization="255", boxes2d=bx).norm_resnet()
    frames = aloscene.Frame.batch_list([fr])
    crit, opt = build_criterion(aux_loss_stage=2), configure_optimizers(model)
    opt.zero_grad()
    total, parts = crit(ddp(frames, is_tracing=None), frames)
    total.backward()
    opt.step()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(2)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        print("SAME" if torch.equal(gathered[0], gathered[1]) else "DIFFERENT", float(total))
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
