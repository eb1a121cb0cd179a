"""Standalone ``MSRB`` / ``ResBlock`` modules (reference model/tactileSR_model.py:157-225 are callable ``nn.Module``s) and
the device-resident loader driving the trainer step on the GPU."""
import pytest
import torch
import torch.nn.functional as F

from oracle import tactilesr_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _randomise(block, seed):
    """Trained-like parameters: random BN affine / running statistics, biases (the seeded init has gamma = beta = 0.1)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in block.state_dict().items():
            if k.endswith("running_var"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith("running_mean"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.2)
            elif k.endswith(".1.weight"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith("bias"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    return {k: v.detach().clone() for k, v in block.state_dict().items()}


@pytest.mark.parametrize("impl", ["fp16x3", "f32", "bf16x6"])
@pytest.mark.parametrize("B,H,W", [(3, 40, 40), (2, 13, 21)])
def test_standalone_msrb_eval_forward(impl, B, H, W):
    import tactilesr_amd
    torch.manual_seed(5)
    blk = tactilesr_amd.MSRB()
    sd = _randomise(blk, 6)
    x = torch.randn(B, 64, H, W, generator=torch.Generator().manual_seed(7)).clamp_(min=0) * 2
    with torch.no_grad():
        ref = O.msrb_forward({f"b.{k}": v for k, v in sd.items()}, "b", x)
    blk = blk.cuda().eval()
    blk.conv_impl = impl
    y = blk(x.cuda())
    assert y.shape == ref.shape and relerr(y, ref) < TOL
    with torch.no_grad():                       # weights changed in place -> the packed plan is rebuilt
        blk.confusion.bias.add_(0.5)
    sd2 = {f"b.{k}": v.detach().cpu() for k, v in blk.state_dict().items()}
    with torch.no_grad():
        ref2 = O.msrb_forward(sd2, "b", x)
    assert relerr(blk(x.cuda()), ref2) < TOL


@pytest.mark.parametrize("impl", ["fp16x3", "f32"])
def test_standalone_resblock_eval_forward(impl):
    import tactilesr_amd
    torch.manual_seed(8)
    blk = tactilesr_amd.ResBlock()
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    x = torch.randn(5, 64, 24, 40, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = O.resblock_forward({f"b.{k}": v for k, v in sd.items()}, "b", x)
    blk = blk.cuda().eval()
    blk.conv_impl = impl
    assert relerr(blk(x.cuda()), ref) < TOL
    with pytest.raises(Exception, match="no CPU fallback"):
        blk(x)


def _block_train_check(kind, B, H, W, seed):
    """Train-mode forward (batch statistics, running-stat update) and backward (input + every parameter) of a standalone
    block against torch autograd on the CPU oracle in fp64, gradients evaluated on the device's own ReLU pattern
    (tests/_gradcheck.py explains the split); bars as for the whole network: output / loss 1e-5, gradients 2e-5."""
    import tactilesr_amd
    torch.manual_seed(seed)
    blk = tactilesr_amd.MSRB() if kind == "msrb" else tactilesr_amd.ResBlock()
    sd = _randomise(blk, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(B, 64, H, W, generator=g).clamp_(min=0) * 2 + 0.01 * torch.randn(B, 64, H, W, generator=g)
    dy = torch.randn(B, 64, H, W, generator=g)
    blk = blk.cuda().train()
    eng = blk.block_engine()
    eng.keep_ctx = True
    xg = x.cuda().requires_grad_(True)
    y = blk(xg)
    (y * dy.cuda()).sum().backward()
    # device ReLU pattern of this forward -> oracle masks
    c = eng.last_ctx
    from tactilesr_amd.model.tactileSR_model import from_cb16
    s = c.s
    masks = {}

    def bn_mask(buf, ctot, coff, C, scale, shift):
        z = from_cb16(buf, B, C, H, W, ctot, coff).double().cpu()
        return (z * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)) > 0

    if kind == "msrb":
        masks["b.conv_3_1.1"] = bn_mask(s.cat1, 128, 0, 64, s.bn_c1[0, :64], s.bn_c1[1, :64])
        masks["b.conv_5_1.1"] = bn_mask(s.cat1, 128, 64, 64, s.bn_c1[0, 64:], s.bn_c1[1, 64:])
        masks["b.conv_3_2.1"] = bn_mask(s.cat2, 256, 0, 128, s.bn_c2[0, :128], s.bn_c2[1, :128])
        masks["b.conv_5_2.1"] = bn_mask(s.cat2, 256, 128, 128, s.bn_c2[0, 128:], s.bn_c2[1, 128:])
    else:
        masks["b.conv1"] = from_cb16(s.F1.buf, B, 64, H, W).cpu() > 0
    masks["b.out"] = y.detach().cpu() > 0
    p64 = {f"b.{k}": (v.double().requires_grad_(O.is_trainable(k)) if v.is_floating_point() else v) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    ns = {}
    tap = O.ReluTap(masks=masks)
    if kind == "msrb":
        ref = O.msrb_forward(p64, "b", x64, training=True, new_stats=ns, tap=tap)
    else:
        ref = O.resblock_forward(p64, "b", x64, tap=tap)
    leaves = [v for k, v in p64.items() if v.is_floating_point() and v.requires_grad]
    names = [k for k, v in p64.items() if v.is_floating_point() and v.requires_grad]
    grads = torch.autograd.grad((ref * dy.double()).sum(), [x64] + leaves)
    assert relerr(y, ref) < TOL
    assert relerr(xg.grad, grads[0]) < 2e-5
    named = dict(blk.named_parameters())
    worst = 0.0
    for n, gr in zip(names, grads[1:]):
        got = named[n[2:]].grad
        assert got is not None, n
        if float(gr.abs().max()) < 1e-9 * float(dy.abs().max()) * B * H * W:      # conv bias in front of a train-mode BN
            assert float(got.abs().max()) < 1e-3, n
            continue
        e = relerr(got, gr)
        worst = max(worst, e)
        assert e < 2e-5, (n, e)
    new_sd = blk.state_dict()
    for k, v in ns.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert relerr(new_sd[k[2:]], v) < TOL, k
    print(f"[standalone {kind} train] out {relerr(y, ref):.2e}, dx {relerr(xg.grad, grads[0]):.2e}, worst param grad {worst:.2e}")
    # no-grad train-mode call: forward only, statistics still move
    with torch.no_grad():
        y2 = blk(x.cuda())
    assert relerr(y2, ref) < TOL and not y2.requires_grad


@pytest.mark.parametrize("kind,B,H,W", [("msrb", 3, 40, 40), ("msrb", 2, 13, 21), ("res", 4, 40, 40), ("res", 1, 9, 17)])
def test_standalone_block_train_forward_backward(kind, B, H, W):
    _block_train_check(kind, B, H, W, 31)


@pytest.mark.parametrize("form", ["sum", "chain"])
def test_shared_resblock_applied_twice_in_one_graph(form):
    """ADVICE r03 (ddp.py:133) on the real engine: ONE ResBlock instance applied twice inside a single autograd graph --
    `f(blk(a)) + f(blk(b))` and the weight-shared chain `blk(blk(x))`.  Once the block engine's gradient arena exists both
    backward passes used to be handed the same arena slots (the second overwrote the first's gradient before autograd
    summed them).  Checked against torch autograd on the CPU oracle in fp64 with the shared parameters, each application
    on the ReLU pattern the device took for it: input gradients and every parameter gradient within 2e-5."""
    import tactilesr_amd
    from tactilesr_amd.model.tactileSR_model import from_cb16
    torch.manual_seed(12)
    blk = tactilesr_amd.ResBlock()
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(13)
    B, H, W = 2, 24, 16
    xa, xb = torch.randn(B, 64, H, W, generator=g), torch.randn(B, 64, H, W, generator=g)
    da, db = torch.randn(B, 64, H, W, generator=g), torch.randn(B, 64, H, W, generator=g)
    blk = blk.cuda().train()
    eng = blk.block_engine()
    eng.keep_ctx = True
    named = dict(blk.named_parameters())

    def masks_of(c, y):
        return {"b.conv1": from_cb16(c.s.F1.buf, B, 64, H, W).cpu() > 0, "b.out": y.detach().cpu() > 0}

    def run():
        for p in blk.parameters():
            p.grad = None
        a = xa.cuda().requires_grad_(True)
        b = xb.cuda().requires_grad_(True)
        y1 = blk(a)
        c1 = eng.last_ctx
        y2 = blk(b if form == "sum" else y1)
        c2 = eng.last_ctx
        loss = (y1 * da.cuda()).sum() + (y2 * db.cuda()).sum() if form == "sum" else (y2 * db.cuda()).sum()
        loss.backward()
        return a, b, y1, y2, masks_of(c1, y1), masks_of(c2, y2)

    # step 1 builds nothing shared-safe by accident: run a PLAIN step first so that the arena exists, then the shared graph
    (blk(xa.cuda()) * da.cuda()).sum().backward()
    assert eng.arena is not None
    a, b, y1, y2, m1, m2 = run()
    p64 = {f"b.{k}": v.double().requires_grad_(True) for k, v in sd.items()}
    a64, b64 = xa.double().requires_grad_(True), xb.double().requires_grad_(True)
    r1 = O.resblock_forward(p64, "b", a64, tap=O.ReluTap(masks=m1))
    r2 = O.resblock_forward(p64, "b", b64 if form == "sum" else r1, tap=O.ReluTap(masks=m2))
    ref_loss = (r1 * da.double()).sum() + (r2 * db.double()).sum() if form == "sum" else (r2 * db.double()).sum()
    leaves = list(p64.values())
    ins = [a64, b64] if form == "sum" else [a64]
    grads = torch.autograd.grad(ref_loss, ins + leaves)
    assert relerr(y1, r1) < TOL and relerr(y2, r2) < TOL
    assert relerr(a.grad, grads[0]) < 2e-5
    if form == "sum":
        assert relerr(b.grad, grads[1]) < 2e-5
    for n, gr in zip(p64, grads[len(ins):]):
        e = relerr(named[n[2:]].grad, gr)
        assert e < 2e-5, (form, n, e)
    # the step after a shared graph is a plain arena step again
    for p in blk.parameters():
        p.grad = None
    (blk(xa.cuda()) * da.cuda()).sum().backward()
    arena = eng.arena
    assert all(named[n].grad.data_ptr() == arena.flat.data_ptr() + 4 * arena.offsets[n] for n in arena.names)


def test_device_resident_loader_on_cuda_drives_the_trainer_step():
    """DeviceSRLoader(device="cuda"): two shuffled epochs of `train_one_iter` straight from device-resident tensors
    (no host batch, train/tactileSR_train.py:43 removed from the step), every sample seen once per epoch, the loss
    finite and decreasing on a learnable target, then eval_func from the same loader protocol."""
    import tactilesr_amd
    from tactilesr_amd import optim
    from tactilesr_amd.data.device_loader import DeviceSRLoader
    from tactilesr_amd.train import tactileSR_train as TR
    g = torch.Generator().manual_seed(3)
    n = 22
    LR = torch.rand(n, 3, 4, 4, generator=g) * 8
    HR = F.interpolate(LR.mean(1, keepdim=True), size=(100, 100), mode="bilinear") * 30
    ld = DeviceSRLoader(LR, HR, batch_size=8, shuffle=True, seed=5, device="cuda")
    assert ld.LR.is_cuda and ld.HR.is_cuda and len(ld) == 3
    torch.manual_seed(11)
    m = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1).cuda().train()
    opt = optim.Adam(m.parameters(), lr=3e-5, weight_decay=1e-2)
    conf = TR.default_config()
    losses, orders = [], []
    for epoch in range(2):
        seen = []
        for it, (lr_b, hr_b) in enumerate(ld):
            assert lr_b.is_cuda and hr_b.is_cuda and lr_b.shape[0] in (8, 6)
            # which samples are in this batch (rows of LR are distinct)
            d = (lr_b.flatten(1)[:, None, :] - ld.LR.flatten(1)[None]).abs().amax(-1)
            seen += d.argmin(1).tolist()
            losses.append(float(TR.train_one_iter(m, opt, (lr_b, hr_b), conf, check_finite=True, cur_iter=it)["total_loss"]))
        assert sorted(seen) == list(range(n))
        orders.append(seen)
    assert orders[0] != orders[1] and orders[0] != list(range(n))        # a fresh permutation per epoch
    assert all(torch.isfinite(torch.tensor(losses))) and sum(losses[3:]) < sum(losses[:3])
    mse, ssim, psnr = TR.eval_func(m, DeviceSRLoader(LR, HR, batch_size=8, device="cuda"), conf)
    assert mse > 0 and -1 <= ssim <= 1 and psnr == psnr
    assert ld.epoch == 2


def test_graphed_eval_forward_small_batch_replay():
    """HIP-graph replay of the eval forward at the reference's eval batch (8; config/default.py:53): bit-identical to the
    plain forward on fresh inputs, follows a parameter update (re-captures), refuses train mode / other shapes; prints the
    host-side latency of both forms."""
    import time
    import tactilesr_amd
    from tactilesr_amd.model.graph import GraphedForward
    torch.manual_seed(42)
    m = tactilesr_amd.TactileSR().cuda().eval()
    g = torch.Generator().manual_seed(1)
    xs = [(torch.rand(8, 3, 4, 4, generator=g) * 8).cuda() for _ in range(3)]
    gf = GraphedForward(m, xs[0])
    for x in xs:
        assert torch.equal(gf(x).clone(), m(x))
    with torch.no_grad():
        m.output_layer[2].weight.mul_(2.0)
    y = gf(xs[1]).clone()
    assert gf.captures == 2 and torch.equal(y, m(xs[1]))

    def lat(fn, n=50):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_plain, t_graph = lat(lambda: m(xs[0])), lat(lambda: gf(xs[0]))
    print(f"[graph replay] B=8 eval forward: plain {t_plain:.3f} ms, HIP-graph replay {t_graph:.3f} ms")
    assert t_graph < t_plain * 1.1
    with pytest.raises(Exception, match="input shape"):
        gf(torch.zeros(4, 3, 4, 4, device="cuda"))
    m.train()
    with pytest.raises(Exception, match="train mode"):
        gf(xs[0])
