"""One-pass epilogues (alo_add_layernorm, alo_bias_act) against the stock PyTorch ops they replace.  GPU only."""
import pytest
import torch
import torch.nn.functional as F

import alo_hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,C", [(1, 256), (7, 256), (300 * 8, 256), (22223, 256), (33, 64), (5, 768), (4, 1024), (9, 260)])
@pytest.mark.parametrize("with_res,with_pos", [(True, False), (True, True), (False, False), (False, True)])
def test_add_layernorm_fp32(rows, C, with_res, with_pos):
    g = torch.Generator(device=DEV).manual_seed(rows * 131 + C)
    x = torch.randn(rows, C, device=DEV, generator=g) * 3 + 0.5
    res = torch.randn(rows, C, device=DEV, generator=g) if with_res else None
    pos = torch.randn(rows, C, device=DEV, generator=g) if with_pos else None
    w, b = torch.randn(C, device=DEV, generator=g), torch.randn(C, device=DEV, generator=g)
    ref = F.layer_norm((x + res if with_res else x).double(), (C,), w.double(), b.double(), 1e-5)
    got = alo_hip.add_layernorm(x, res, w, b, 1e-5, pos=pos)
    out = got[0] if with_pos else got
    assert (out.double() - ref).abs().max().item() <= 2e-5
    if with_pos:
        assert torch.equal(got[1], out + pos)


def test_add_layernorm_bf16_tracks_fp32_and_allows_aliasing():
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(8, 1000, 256, device=DEV, generator=g).bfloat16()
    res = torch.randn(8, 1000, 256, device=DEV, generator=g).bfloat16()
    pos = torch.randn(8, 1000, 256, device=DEV, generator=g).bfloat16()
    w = (torch.rand(256, device=DEV, generator=g) + 0.5).bfloat16()
    b = torch.randn(256, device=DEV, generator=g).bfloat16()
    ref = F.layer_norm(x.float() + res.float(), (256,), w.float(), b.float(), 1e-5)
    out, out_pos = alo_hip.add_layernorm(x, res, w, b, 1e-5, pos=pos)
    assert out.dtype == torch.bfloat16 and out.shape == x.shape
    # fp32 arithmetic, one rounding at the end: within half a bf16 ulp of the fp32 result
    assert ((out.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-6).all()
    assert torch.equal(out_pos, out + pos)  # computed from the rounded `out`, exactly like the unfused `out + pos`
    stock = F.layer_norm(x + res, (256,), w, b, 1e-5)
    assert (out.float() - stock.float()).abs().max().item() <= 0.0625
    # `out` aliasing `x` through the raw ABI (each row is read completely before it is written)
    x2 = x.clone()
    rc = alo_hip.lib().alo_add_layernorm(alo_hip._ptr(x2), alo_hip._ptr(res), alo_hip._ptr(w), alo_hip._ptr(b),
                                         alo_hip._ptr(x2), None, None, x.numel() // 256, 256, 1e-5, alo_hip.ALO_BF16,
                                         alo_hip._stream(x.device))
    assert rc == 0 and torch.equal(x2, out)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (False, False), (True, False)])
def test_bias_act_channels_last(dtype, with_res, relu):
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(3, 64, 25, 42, device=DEV, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    res = torch.randn(3, 64, 25, 42, device=DEV, generator=g).to(dtype).contiguous(memory_format=torch.channels_last) if with_res else None
    bias = torch.randn(64, device=DEV, generator=g).to(dtype)
    ref = x.float() + bias.float().view(1, -1, 1, 1) + (res.float() if with_res else 0)
    ref = (ref.relu() if relu else ref).to(dtype)
    y = alo_hip.bias_act_(x.clone(memory_format=torch.preserve_format), bias, res, relu)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, ref)  # one rounding of the exact fp32 sum


def test_bias_act_rejects_nchw_and_fusable_gates_autograd():
    x = torch.randn(2, 8, 4, 4, device=DEV)
    with pytest.raises(RuntimeError, match="channels_last"):
        alo_hip.bias_act_(x, torch.zeros(8, device=DEV))
    p = torch.randn(4, 256, device=DEV, requires_grad=True)
    assert not alo_hip.fusable(p)
    with torch.no_grad():
        assert alo_hip.fusable(p)
    assert not alo_hip.fusable(torch.randn(4, 256))  # CPU


def test_resnet_bottleneck_fused_vs_stock():
    """The fused conv epilogue is what the backbone runs at inference: compare with the autograd (stock-op) path."""
    from alonet.detr.backbone import ResNetBody

    torch.manual_seed(0)
    body = ResNetBody("resnet50", return_layers={"layer1": "0", "layer2": "1", "layer3": "2", "layer4": "3"}).to(DEV).eval()
    for m in body.modules():  # non-trivial frozen statistics
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 2.0); m.running_mean.normal_(0, 0.2); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    x = torch.randn(2, 3, 96, 128, device=DEV)
    with torch.no_grad():
        fused = body(x)
    for p in body.parameters():
        p.requires_grad_(True)
    stock = body(x.requires_grad_(True))  # grad mode with trainable weights: stock ops
    for k in fused:
        scale = stock[k].abs().max().item()
        assert (fused[k] - stock[k]).abs().max().item() <= 2e-4 * scale, k


def test_raft_update_block_fused_glue_matches_stock_ops():
    """BasicUpdateBlock at inference (merged z/r convolution, HIP gate / update / bias passes, 0.25 folded into the mask
    head) against the same module evaluated through the stock PyTorch ops (autograd path)."""
    from alonet.raft.update import BasicUpdateBlock

    torch.manual_seed(3)
    blk = BasicUpdateBlock(corr_levels=4, corr_radius=4).to(DEV).eval()
    B, H, W = 2, 12, 18
    net = torch.tanh(torch.randn(B, 128, H, W, device=DEV))
    inp = torch.relu(torch.randn(B, 128, H, W, device=DEV))
    corr = torch.randn(B, 324, H, W, device=DEV)
    flow = torch.randn(B, 2, H, W, device=DEV) * 3
    with alo_hip.LaunchTimer() as t, torch.no_grad():
        n1, m1, d1 = blk(net, inp, corr, flow)
    tags = set(t.summary())
    assert {"gru_gate/C=128", "gru_update/C=128"} <= tags and any(k.startswith("bias_act_nchw") for k in tags)
    n0, m0, d0 = blk(net, inp, corr, flow)  # parameters require grad: stock ops
    assert n0.requires_grad
    for a, b in ((n1, n0), (m1, m0), (d1, d0)):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    # odd H*W: the stock path is used (the kernels want H*W % 4 == 0)
    with alo_hip.LaunchTimer() as t2, torch.no_grad():
        blk(*(t[..., :11, :17].contiguous() for t in (net, inp, corr, flow)))  # 11 * 17 = 187
    assert not t2.summary()


@pytest.mark.parametrize("dtype,normalize,center", [(torch.float32, True, True), (torch.float32, True, False),
                                                     (torch.float32, False, False), (torch.bfloat16, True, True)])
def test_pos_sine_flat_matches_the_module_plus_level_embed(dtype, normalize, center):
    """alo_pos_sine_flat against PositionEmbeddingSine per level -> flatten -> + level_embed -> cat (the stock chain)."""
    from alonet.transformers import PositionEmbeddingSine

    torch.manual_seed(1)
    shapes = [(13, 21), (7, 11), (4, 6), (2, 3)]
    B, F = 3, 128
    enc = PositionEmbeddingSine(F, normalize=normalize, center=center)
    level_embed = torch.randn(len(shapes), 2 * F, device=DEV).to(dtype).float()  # a parameter of the model's dtype
    masks, want = [], []
    for lvl, (h, w) in enumerate(shapes):
        m = torch.zeros(B, 1, h, w, dtype=torch.bool, device=DEV)
        m[1, :, :, (2 * w) // 3:] = True   # right padding
        m[2, :, h // 2:, :] = True         # bottom padding
        masks.append(m)
        pos = enc((torch.empty(B, 1, h, w, device=DEV), m))
        want.append(pos.flatten(2).transpose(1, 2) + level_embed[lvl].view(1, 1, -1))
    want = torch.cat(want, 1)
    mask_flat = torch.cat([m[:, 0].flatten(1) for m in masks], 1)
    sh = torch.tensor(shapes, dtype=torch.int32, device=DEV)
    start = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1]), dtype=torch.int32, device=DEV)
    got = alo_hip.pos_sine_flat(mask_flat, sh, start, enc.dim_t(torch.device(DEV)), level_embed, normalize, center, enc.scale, dtype)
    assert got.shape == want.shape and got.dtype == dtype
    if dtype == torch.float32:
        # sin / cos of arguments up to 2*pi (un-normalised: up to the map size): fp32 rounding of the argument only
        assert (got - want).abs().max().item() <= (3e-6 if normalize else 3e-5)
    else:
        assert ((got.float() - want).abs() <= want.abs() * 2.0 ** -8 + 1e-6).all()


@pytest.mark.parametrize("M,N,K,relu,with_bias", [(177784, 256, 256, False, True), (5000, 128, 256, False, True),
                                                   (4099, 1024, 256, True, True), (33, 64, 64, True, False),
                                                   (1, 320, 128, False, True), (777, 512, 128, True, True)])
def test_linear_shortk_matches_fp32_matmul(M, N, K, relu, with_bias):
    """alo_linear_shortk (weights resident in registers, bf16 MFMA, fp32 accumulation) against an fp32 matmul of the same
    bf16 operands: only the final rounding to bf16 may differ."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.1).bfloat16()
    b = torch.randn(N, device=DEV, generator=g).bfloat16() if with_bias else None
    ref = x.float() @ w.float().t() + (b.float() if with_bias else 0)
    ref = ref.relu() if relu else ref
    got = alo_hip.linear_shortk(x, w, b, relu)
    assert got.shape == (M, N) and got.dtype == torch.bfloat16
    assert ((got.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-5 * K ** 0.5).all()
    assert alo_hip.linear_shortk_supported(x, w) and not alo_hip.linear_shortk_supported(x.float(), w.float())
    # leading dims are kept, linear_auto picks the same kernel
    got3 = alo_hip.linear_auto(x[: (M // 3) * 3].view(3, M // 3, K), w, b, relu) if M >= 3 else None
    if got3 is not None:
        assert torch.equal(got3.reshape(-1, N), got[: (M // 3) * 3])


def test_linear_shortk_residual_epilogue():
    """y = relu(x W^T + b + identity): the bottleneck's last 1x1 convolution with its identity in the GEMM epilogue."""
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(5000, 128, device=DEV, generator=g).bfloat16()
    w = (torch.randn(512, 128, device=DEV, generator=g) * 0.1).bfloat16()
    b = torch.randn(512, device=DEV, generator=g).bfloat16()
    res = torch.randn(5000, 512, device=DEV, generator=g).bfloat16()
    pre = (x.float() @ w.float().t() + b.float()).bfloat16().float()  # the convolution's own output is rounded first
    for relu in (True, False):
        ref = pre + res.float()
        ref = ref.relu() if relu else ref
        got = alo_hip.linear_shortk(x, w, b, relu, residual=res)
        assert ((got.float() - ref).abs() <= ref.abs() * 2.0 ** -7 + 0.02).all()  # pre may round the other way by one ulp


def test_linear_auto_falls_back_to_the_stock_gemm():
    x = torch.randn(10, 1024, device=DEV).bfloat16()  # K = 1024: not a short-K problem
    w, b = torch.randn(256, 1024, device=DEV).bfloat16() * 0.05, torch.randn(256, device=DEV).bfloat16()
    with alo_hip.LaunchTimer() as t:
        y = alo_hip.linear_auto(x, w, b, relu=True)
    assert not t.summary() and (y.float() - torch.relu(x.float() @ w.float().t() + b.float())).abs().max().item() < 0.5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-2)])
def test_decoder_self_attention_fast_path_matches_multihead_attention(dtype, tol):
    from alonet.deformable_detr.deformable_transformer import _self_attention

    torch.manual_seed(2)
    mha = torch.nn.MultiheadAttention(256, 8, dropout=0.1).to(DEV).to(dtype).eval()
    tgt = torch.randn(3, 300, 256, device=DEV).to(dtype)
    pos = torch.randn(3, 300, 256, device=DEV).to(dtype)
    with torch.no_grad():
        q = tgt + pos
        want = mha(q.transpose(0, 1), q.transpose(0, 1), tgt.transpose(0, 1))[0].transpose(0, 1)
        got = _self_attention(mha, q, tgt)
    assert got.shape == want.shape and (got.float() - want.float()).abs().max().item() <= tol


@pytest.mark.parametrize("M,Fh,with_bias", [(177784, 1024, True), (2400, 1024, True), (65, 512, False), (1, 256, True), (6401, 2048, True)])
def test_ffn256_matches_two_step_reference(M, Fh, with_bias):
    """alo_ffn256 (hidden activation kept in LDS, packed weights) against fp32 matmuls of the same bf16 operands with the
    hidden activation rounded to bf16 in between, exactly as the two-GEMM path stores it."""
    g = torch.Generator(device=DEV).manual_seed(M + Fh)
    x = torch.randn(M, 256, device=DEV, generator=g).bfloat16()
    w1 = (torch.randn(Fh, 256, device=DEV, generator=g) * 0.06).bfloat16()
    w2 = (torch.randn(256, Fh, device=DEV, generator=g) * 0.03).bfloat16()
    b1 = torch.randn(Fh, device=DEV, generator=g).bfloat16() if with_bias else None
    b2 = torch.randn(256, device=DEV, generator=g).bfloat16() if with_bias else None
    h = (x.float() @ w1.float().t() + (b1.float() if with_bias else 0)).relu().bfloat16().float()
    ref = h @ w2.float().t() + (b2.float() if with_bias else 0)
    got = alo_hip.ffn256(x, w1, b1, w2, b2)
    assert got.shape == x.shape and got.dtype == torch.bfloat16
    # one bf16 rounding of the result + the hidden activations that round the other way by one ulp (fp32 summation order)
    assert ((got.float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 4e-3).all()
    # the packed copy follows in-place weight updates
    with torch.no_grad():
        w1.mul_(0.5)
    again = alo_hip.ffn256(x, w1, b1, w2, b2)
    h2 = (x.float() @ w1.float().t() + (b1.float() if with_bias else 0)).relu().bfloat16().float()
    ref2 = h2 @ w2.float().t() + (b2.float() if with_bias else 0)
    assert ((again.float() - ref2).abs() <= ref2.abs() * 2.0 ** -8 + 4e-3).all()


@pytest.mark.parametrize("N,S,heads,K,with_mask", [(2, 22223, 8, 256, True), (3, 301, 4, 128, True), (1, 64, 2, 64, False)])
def test_value_proj_head_major_equals_linear_then_relayout(N, S, heads, K, with_mask):
    """value_proj + padding mask + head-major layout in the GEMM epilogue == alo_linear_shortk followed by alo_value_head_major."""
    g = torch.Generator(device=DEV).manual_seed(N + S + K)
    x = torch.randn(N, S, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(heads * 32, K, device=DEV, generator=g) * 0.1).bfloat16()
    b = torch.randn(heads * 32, device=DEV, generator=g).bfloat16()
    mask = (torch.rand(N, S, device=DEV, generator=g) < 0.25) if with_mask else None
    got = alo_hip.value_proj_head_major(x, w, b, mask, heads)
    two_step = alo_hip.value_head_major(alo_hip.linear_shortk(x, w, b).view(N, S, heads, 32), mask)
    assert got.shape == (N, heads, S, 32) and torch.equal(got, two_step)
    if with_mask:
        assert float(got.permute(0, 2, 1, 3)[mask].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 128, 25, 42), (1, 256, 7, 9), (3, 512, 5, 70), (1, 128, 1, 1), (2, 128, 64, 3),
                                   (2, 64, 30, 41), (1, 64, 2, 2), (2, 1024, 13, 11)])   # the last one takes the split-K route
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("relu,with_bias", [(True, True), (False, False)])
def test_conv3x3_matches_fp32_convolution(shape, stride, relu, with_bias):
    """alo_conv3x3_nhwc against F.conv2d in fp32 on the same bf16 inputs: fp32 accumulation, one bf16 rounding of the result."""
    n, c, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + c + h)
    x = torch.randn(n, c, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    cout = c // 2 if c >= 256 else (2 * c if c == 64 and h > 2 else c)
    wt = (torch.randn(cout, c, 3, 3, device="cuda", generator=g) / (9 * c) ** 0.5).to(torch.bfloat16)
    wt = wt.contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda", generator=g).to(torch.bfloat16) if with_bias else None
    with torch.no_grad():
        ref = F.conv2d(x.float(), wt.float(), None if b is None else b.float(), stride, 1)
        if relu:
            ref = F.relu(ref)
        got = alo_hip.conv3x3(x, wt, b, relu=relu, stride=stride)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    # |result| is O(1): half a bf16 ulp of the largest value plus fp32 summation-order noise
    assert (got.float() - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_conv3x3_refuses_what_it_does_not_cover():
    x = torch.randn(1, 32, 8, 8, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(64, 32, 3, 3, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        assert not alo_hip.conv3x3_supported(x, wt)
        with pytest.raises(RuntimeError):
            alo_hip.conv3x3(x, wt)
        x128 = torch.randn(1, 128, 8, 8, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w128 = torch.randn(128, 128, 3, 3, device="cuda", dtype=torch.bfloat16)
        assert alo_hip.conv3x3_supported(x128, w128) and alo_hip.conv3x3_supported(x128, w128, stride=(2, 2))
        assert not alo_hip.conv3x3_supported(x128, w128, stride=(3, 3))
        assert not alo_hip.conv3x3_supported(x128, w128, dilation=(2, 2))
        assert not alo_hip.conv3x3_supported(x128.contiguous(), w128)
    assert not alo_hip.conv3x3_supported(x128, w128)   # autograd on: the kernel has no backward


@pytest.mark.gpu
@pytest.mark.parametrize("shape,layout", [((2, 3, 64, 96), "nchw"), ((1, 3, 37, 53), "nhwc"), ((3, 3, 8, 8), "nchw"),
                                          ((1, 3, 101, 30), "nchw"), ((1, 3, 5, 200), "nhwc")])
def test_stem_conv_pool_matches_fp32_stem(shape, layout):
    """alo_stem_conv_pool against max_pool2d(relu(conv2d(...))) in fp32 on the same bf16 inputs (one bf16 rounding, which
    commutes with the max)."""
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(*shape, device="cuda", generator=g).to(torch.bfloat16)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(64, 3, 7, 7, device="cuda", generator=g) / 147 ** 0.5).to(torch.bfloat16)
    b = torch.randn(64, device="cuda", generator=g).to(torch.bfloat16)
    with torch.no_grad():
        ref = F.max_pool2d(F.relu(F.conv2d(x.float(), wt.float(), b.float(), 2, 3)), 3, 2, 1)
        got = alo_hip.stem_conv_pool(x, wt, b)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.float() - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_resnet_stem_fused_matches_unfused():
    from alonet.detr.backbone import ResNetBody
    torch.manual_seed(3)
    net = ResNetBody("resnet50").cuda().to(torch.bfloat16).eval()
    for m in net.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 1.5); m.running_mean.normal_(0, 0.1); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 3, 96, 130, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        assert net._stem_fusable(x)
        fused = net(x)["0"]
        net._stem_fusable = lambda _x: False
        plain = net(x)["0"]
    scale = plain.float().abs().max().item()
    assert (fused.float() - plain.float()).abs().max().item() <= 0.03 * max(scale, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,HW,C,groups", [(2, 1000, 256, 32), (1, 7, 256, 32), (3, 300, 128, 8), (1, 513, 64, 8), (2, 257, 256, 16)])
def test_groupnorm_rows_matches_group_norm(B, HW, C, groups):
    """alo_groupnorm_rows against F.group_norm in fp32 on the same bf16 rows, written into a slice of a larger flat buffer."""
    g = torch.Generator(device="cuda").manual_seed(B * HW + C)
    x = (torch.randn(B, HW, C, device="cuda", generator=g) * 3 + 0.7).to(torch.bfloat16)
    wt = torch.randn(C, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn(C, device="cuda", generator=g).to(torch.bfloat16)
    flat = torch.full((B, HW + 21, C), 7.0, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        ref = F.group_norm(x.float().transpose(1, 2), groups, wt.float(), b.float(), 1e-5).transpose(1, 2)
        out = alo_hip.groupnorm_rows(x, wt, b, groups, 1e-5, out=flat[:, 8:8 + HW])
    assert out.data_ptr() == flat[:, 8:].data_ptr()
    assert (out.float() - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item()) + 1e-3
    assert (flat[:, :8] == 7.0).all() and (flat[:, 8 + HW:] == 7.0).all()   # nothing written outside the slot
    with torch.no_grad():
        alone = alo_hip.groupnorm_rows(x, wt, b, groups, 1e-5)
    assert torch.equal(alone, out)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W,groups", [(3, 16, 21, 34, 8), (2, 32, 13, 17, 8), (5, 64, 9, 11, 8), (2, 128, 7, 5, 8), (1, 16, 300, 301, 8),
                                            (4, 32, 1, 3, 16)])
@pytest.mark.parametrize("relu", [False, True])
def test_groupnorm_nhwc_with_narrow_groups_matches_group_norm(N, C, H, W, groups, relu):
    """alo_groupnorm_rows_act (2, 4, 8 or 16 channels per group, optional ReLU) on channels-last maps against F.group_norm in fp32 on
    the same bf16 values: the GroupNorm(8, 16..128) + ReLU layers of PanopticHead's mask decoder (FPNstyle.py:24-36)."""
    g = torch.Generator(device="cuda").manual_seed(N * C + H * W)
    x = (torch.randn(N, C, H, W, device="cuda", generator=g) * 2 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    norm = torch.nn.GroupNorm(groups, C).cuda().to(torch.bfloat16)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, device="cuda", generator=g))
        norm.bias.copy_(torch.randn(C, device="cuda", generator=g))
        ref = F.group_norm(x.float(), groups, norm.weight.float(), norm.bias.float(), norm.eps)
        ref = F.relu(ref) if relu else ref
        assert alo_hip.groupnorm_nhwc_supported(x, norm)
        out = alo_hip.groupnorm_nhwc(x, norm, relu=relu)
    assert out.shape == x.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert (out.float() - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item()) + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(3, 32, 16, 21, 34), (2, 64, 32, 13, 17), (2, 16, 1, 9, 50), (1, 16, 4, 8, 16), (5, 32, 32, 1, 1),
                                            (2, 64, 8, 7, 3), (1, 32, 16, 200, 334)])
@pytest.mark.parametrize("bias", [True, False])
def test_conv3x3_small_matches_fp32_convolution(N, Cin, Cout, H, W, bias):
    """alo_conv3x3_small_nhwc (few-channel 3x3 convolutions of the mask decoder: FPNstyle.py lay4 / lay5 / out_lay) against F.conv2d in
    fp32 on the same bf16 operands: one rounding of difference; tiles that hang over the map's edges, 1 x 1 maps, Cout = 1."""
    g = torch.Generator(device="cuda").manual_seed(N * Cin + Cout + H * W)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=bias).cuda().to(torch.bfloat16).to(memory_format=torch.channels_last)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (9 * Cin) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(Cout, device="cuda", generator=g))
        assert alo_hip.conv3x3_small_supported(x, conv)
        ref = F.conv2d(x.float(), conv.weight.float(), conv.bias.float() if bias else None, padding=1)
        out = alo_hip.conv3x3_small(x, conv)
        again = alo_hip.conv3x3_small(x, conv)          # cached operands
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out, again)
    assert (out.float() - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item()) + 1e-3
    with torch.no_grad():                               # weights edited in place: the packed copy follows the version counter
        conv.weight.mul_(2.0)
        out2 = alo_hip.conv3x3_small(x, conv)
        ref2 = F.conv2d(x.float(), conv.weight.float(), conv.bias.float() if bias else None, padding=1)
    assert (out2.float() - ref2).abs().max().item() <= 2.0 ** -8 * max(1.0, ref2.abs().max().item()) + 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,Q,C,h,w,H,W", [(2, 3, 32, 5, 7, 10, 13), (1, 16, 64, 25, 42, 50, 84), (3, 1, 8, 4, 4, 9, 7), (2, 5, 128, 3, 4, 3, 4),
                                           (1, 2, 16, 100, 167, 200, 334)])
def test_upsample_add_is_the_stock_expand_interpolate_add_bit_for_bit(B, Q, C, h, w, H, W):
    """alo_upsample_add_nhwc == `_expand(fpn, Q) + F.interpolate(x, size=(H, W), mode="nearest")` of the mask decoder's FPN steps
    (FPNstyle.py:60-84), odd ratios included: same source pixels as ATen's nearest kernel, fp32 add, one rounding."""
    g = torch.Generator(device="cuda").manual_seed(B * Q + C + H)
    x = torch.randn(B * Q, C, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fpn = torch.randn(B, C, H, W, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want = fpn.unsqueeze(1).repeat(1, Q, 1, 1, 1).flatten(0, 1) + F.interpolate(x, size=(H, W), mode="nearest")
        got = alo_hip.upsample_add(x, fpn)
    assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(1000, 512, 128), (77, 1024, 256), (130, 2048, 512), (64, 512, 2048), (1, 768, 384)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_linear_packed_matches_fp32_linear(M, K, N, relu, res):
    """alo_linear_packed against F.linear in fp32 on the same bf16 inputs (the identity is added to the bf16-rounded product,
    as the unfused conv -> add -> relu sequence does)."""
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if res else None
    with torch.no_grad():
        ref = F.linear(x.float(), w.float(), b.float())
        if res:
            ref = ref.to(torch.bfloat16).float() + r.float()
        if relu:
            ref = F.relu(ref)
        got = alo_hip.linear_packed(x, w, b, relu, residual=r)
    assert (got.float() - ref).abs().max().item() <= 2.0 ** -7 * max(1.0, ref.abs().max().item())


def _stock_masks(frame_mask, shapes, n_bilinear):
    """What the reference computes per level (detr/backbone.py:127-128, deformable_detr.py:147, deformable_transformer.py:318-323)."""
    masks, ratios = [], []
    for lvl, (h, w) in enumerate(shapes):
        if lvl < n_bilinear:
            m = F.interpolate(frame_mask.float(), size=(h, w), mode="bilinear", align_corners=False).to(torch.bool)[:, 0]
        else:
            m = F.interpolate(frame_mask.float(), size=(h, w)).to(torch.bool)[:, 0]
        masks.append(m.flatten(1))
        valid_h = torch.sum((~m).float()[:, :, 0], 1, keepdim=True)
        valid_w = torch.sum((~m).float()[:, 0, :], 1, keepdim=True)
        ratios.append(torch.cat([valid_w / w, valid_h / h], 1))
    return torch.cat(masks, 1), torch.stack(ratios, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(800, 1333), (641, 487), (96, 130), (33, 65), (512, 512)])
@pytest.mark.parametrize("as_float", [True, False])
def test_mask_pyramid_matches_interpolate(H, W, as_float):
    """alo_mask_pyramid == F.interpolate(...).to(bool) + get_valid_ratio, bit for bit, for rectangular paddings of every size and
    for random masks."""
    def down(v, k):
        for _ in range(k):
            v = (v - 1) // 2 + 1
        return v
    shapes = [(down(H, k), down(W, k)) for k in (3, 4, 5, 6)]
    g = torch.Generator(device="cuda").manual_seed(H * 7 + W)
    B = 6
    mask = torch.zeros(B, 1, H, W, device="cuda")
    for b in range(B - 1):   # bottom / right padding of various extents, as batched frames have
        ph = int(torch.randint(0, H // 2, (1,), generator=g, device="cuda"))
        pw = int(torch.randint(0, W // 2, (1,), generator=g, device="cuda"))
        if ph:
            mask[b, :, H - ph:, :] = 1
        if pw:
            mask[b, :, :, W - pw:] = 1
    mask[B - 1] = (torch.rand(1, H, W, device="cuda", generator=g) < 0.3).float()
    arg = mask if as_float else mask.bool()
    want_m, want_r = _stock_masks(mask, shapes, 3)
    got_m, got_r = alo_hip.mask_pyramid(arg, shapes, nearest_levels=[3])
    assert got_m.dtype == torch.bool and torch.equal(got_m, want_m)
    assert torch.equal(got_r, want_r)


@pytest.mark.gpu
def test_encoder_reference_points_match_the_torch_construction():
    from alonet.deformable_detr.deformable_transformer import DeformableTransformerEncoder, _level_geometry
    shapes = ((100, 167), (50, 84), (25, 42), (13, 21))
    spatial_shapes, _ = _level_geometry(shapes, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(5)
    ratios = torch.rand(8, 4, 2, device="cuda", generator=g) * 0.6 + 0.4
    got = DeformableTransformerEncoder.get_reference_points(spatial_shapes, ratios, device=ratios.device)
    want = DeformableTransformerEncoder.get_reference_points(spatial_shapes, ratios, device=ratios.device, is_tracing=True)
    assert got.shape == want.shape == (8, sum(h * w for h, w in shapes), 4, 2)
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout", [((2, 256, 17, 23), 512), ((1, 512, 10, 12), 1024), ((3, 64, 9, 9), 128), ((1, 1024, 5, 8), 2048)])
@pytest.mark.parametrize("stride", [2, 3])
def test_conv1x1_strided_matches_convolution(shape, cout, stride):
    """alo_conv1x1_nhwc (stride folded into the GEMM's tile loader) against F.conv2d in fp32."""
    n, c, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(c + h + stride)
    x = torch.randn(n, c, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, c, device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda", generator=g).to(torch.bfloat16)
    with torch.no_grad():
        assert alo_hip.conv1x1_strided_supported(x, wt)
        got = alo_hip.conv1x1_strided(x, wt, b, stride, relu=True)
        ref = F.relu(F.conv2d(x.float(), wt.float()[:, :, None, None], b.float(), stride))
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.float() - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("B,Q,h,w,H,W", [(2, 16, 50, 84, 200, 334), (1, 3, 13, 21, 97, 160), (3, 1, 8, 8, 64, 64), (1, 5, 10, 10, 10, 10)])
def test_panoptic_onehot_matches_the_torch_chain(B, Q, h, w, H, W):
    """alo_panoptic_onehot against interpolate -> sigmoid -> threshold -> argmax one-hot (detr_panoptic.py:96-110).  A pixel can
    only differ where two queries' probabilities (or a probability and the threshold) are within float rounding of each other."""
    g = torch.Generator(device="cuda").manual_seed(B * 100 + Q)
    logits = torch.randn(B, Q, h, w, device="cuda", generator=g) * 3
    logits[:, :, : h // 3] -= 6.0     # a region where no query passes the threshold
    got = alo_hip.panoptic_onehot(logits, (H, W), 0.5)
    m = F.threshold(F.interpolate(logits, size=(H, W), mode="bilinear", align_corners=False).sigmoid(), 0.5, 0.0)
    want = []
    for masks in m:
        nothing = (~masks.bool()).all(dim=0, keepdim=True)
        onehot = torch.zeros_like(masks)
        onehot.scatter_(0, masks.argmax(dim=0, keepdim=True), 1)
        want.append(onehot.long() * (~nothing))
    want = torch.stack(want)
    assert got.shape == want.shape and got.dtype == torch.long
    assert int(got.sum(1).max()) <= 1 and bool((want.sum(1) == 0).any())
    assert (got != want).float().mean().item() <= 1e-5
