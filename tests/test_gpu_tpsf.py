"""GPU parity of the batched tPSFNet path against the reference's own outputs (tests/golden/tpsf.npz:
forward 4-tuple, trainer loss, MLP gradients)."""
import numpy as np
import pytest
import torch

from oracle import tactilesr_oracle as O

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_tpsf_forward_backward_vs_reference_golden(golden):
    import tactilesr_amd
    from tactilesr_amd.train import tPSFNet_train as TP
    g = golden("tpsf")
    sd = O.random_state_dict(O.tpsf_state_shapes(), int(g["seed"]))
    net = tactilesr_amd.tPSFNet(gama=1.4, perception_scale=None, device="cuda")
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    LR_raw, depth = torch.from_numpy(g["LR_raw"]).cuda(), torch.from_numpy(g["depth"]).cuda()
    HR, LRd, psf, ab = net(LR_raw / 100, depth.unsqueeze(1))
    assert HR.shape == (4, 1, 100, 100) and LRd.shape == (4, 1, 4, 4) and psf.shape == (4, 1, 99, 99) \
        and ab.shape == (4, 1, 3)
    assert relerr(ab, torch.from_numpy(g["alphaBeta"])) < 1e-5
    assert relerr(HR, torch.from_numpy(g["HR"])) < 1e-5
    assert relerr(LRd, torch.from_numpy(g["LR_deg"])) < 1e-5
    assert relerr(psf[:, 0, ::7, ::7], torch.from_numpy(g["psf_probe"])) < 1e-5
    assert np.abs(psf.double().sum(dim=(1, 2, 3)).cpu().numpy() - g["psf_sum"]).max() < 1e-5 * np.abs(g["psf_sum"]).max()
    loss, _ = TP.train_cal_loss(net, (LR_raw, depth), 100.0)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    for k, p in net.named_parameters():
        ref = torch.from_numpy(g[f"grad/{k}"])
        err = relerr(p.grad, ref)
        print(f"[tpsf grad] {k}: {err:.2e}")
        assert err < 2e-5, (k, err)


def test_tpsf_large_batch_properties():
    """Batch independence at a size the oracle cannot run in seconds: any sample's outputs equal the same
    sample processed alone; psf is symmetric; plateau pixels all hold one value."""
    import tactilesr_amd
    torch.manual_seed(0)
    net = tactilesr_amd.tPSFNet(1.4, None).cuda()
    B = 257
    g = torch.Generator().manual_seed(1)
    depth = (torch.rand(B, 1, 100, 100, generator=g) > 0.7).float().cuda()
    x = (torch.rand(B, 3, 4, 4, generator=g) * 8).cuda()
    with torch.no_grad():
        HR, LRd, psf, ab = net(x, depth)
        HR1, LRd1, psf1, ab1 = net(x[200:201], depth[200:201])
    assert torch.equal(HR[200:201], HR1) and torch.equal(LRd[200:201], LRd1) and torch.equal(psf[200:201], psf1)
    assert torch.equal(psf, psf.transpose(2, 3))
    m = depth[5, 0] > depth[5, 0].max() - 1e-3
    assert HR[5, 0][m].unique().numel() == 1


def test_inplace_edit_of_returned_HR_before_backward_is_caught():
    """The backward's reductions read the STORED forward output (model/tPSFNet.py:58): normalising / clamping
    the returned HR in place between forward and backward must trip autograd's version check instead of silently changing
    d(alpha, beta, gamma); an out-of-place edit leaves the gradients untouched."""
    import tactilesr_amd
    torch.manual_seed(7)
    net = tactilesr_amd.tPSFNet(1.4, None).cuda()
    g = torch.Generator().manual_seed(8)
    depth = (torch.rand(3, 1, 100, 100, generator=g) > 0.7).float().cuda()
    x = (torch.rand(3, 3, 4, 4, generator=g) * 8).cuda()
    HR, LRd, _, _ = net(x, depth)
    HR2 = HR / HR.amax()                       # out of place: fine
    LRd.sum().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    HR, LRd, _, _ = net(x, depth)
    HR.div_(HR.amax())
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        LRd.sum().backward()
    net.zero_grad()
    HR, LRd, _, _ = net(x, depth)
    LRd.sum().backward()
    assert all(torch.equal(p.grad, r) for p, r in zip(net.parameters(), ref)) and HR2.isfinite().all()


def test_mlp_gradients_live_in_one_flat_buffer_and_adam_builds_its_table_once():
    """ADVICE r03 (optim.py:60): tPSFNet's parameter gradients used to be fresh autograd tensors every step, so the fused
    Adam re-built its device table (a blocking host-to-device copy) per step.  They now land in one flat buffer that
    `p.grad` aliases: over several zero_grad / backward / step rounds the table is built ONCE; gradient accumulation
    (no zero_grad) and two applications of the module inside one graph still sum correctly."""
    import tactilesr_amd
    from tactilesr_amd import optim
    from tactilesr_amd.train import tPSFNet_train as TP
    torch.manual_seed(9)
    net = tactilesr_amd.tPSFNet(1.4, None).cuda()
    opt = optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(10)
    LR = torch.rand(6, 3, 4, 4, generator=g) * 800
    depth = (torch.rand(6, 100, 100, generator=g) > 0.7).float()
    for _ in range(4):
        loss, _ = TP.train_cal_loss(net, (LR.cuda(), depth.cuda()), 100.0)
        opt.zero_grad()
        loss.backward()
        flat = net._grad_plan.flat
        assert all(flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + 4 * flat.numel() for p in net.parameters())
        opt.step()
    assert opt.table_builds == 1 and opt.launches == 4
    # accumulation: backward twice without zero_grad == 2 x the gradient; shared graph: f(net(a)) + f(net(b))
    def grads(fn):
        for p in net.parameters():
            p.grad = None
        fn()
        return [p.grad.clone() for p in net.parameters()]
    def one(lo, hi):
        return TP.train_cal_loss(net, (LR[lo:hi].cuda(), depth[lo:hi].cuda()), 100.0)[0]
    g_a, g_b = grads(lambda: one(0, 3).backward()), grads(lambda: one(3, 6).backward())
    def twice():
        one(0, 3).backward()
        one(0, 3).backward()
    for x, y in zip(grads(twice), g_a):
        assert relerr(x, 2 * y) < 1e-6
    for x, y, z in zip(grads(lambda: (one(0, 3) + one(3, 6)).backward()), g_a, g_b):
        assert relerr(x, y + z) < 1e-6
    for x, y in zip(grads(lambda: one(0, 3).backward()), g_a):      # and a plain step afterwards is a plain step
        assert torch.equal(x, y)


def test_dataset_generator_matches_batch1_loop(tmp_path):
    """The batched generator (data/SRdataset/depth2tactile.py:104-160 rewritten without the batch-1 loop)
    writes, per sample, exactly what a batch-1 forward produces, in the reference's file format -- and every WRITTEN
    entry (read back from the .npy like utility/load_tactile_dataset.py:39-47 reads it) holds what the CPU oracle's
    tPSFNet forward (direct 99x99 conv per sample, reference :107-119) computes for that sample: HR, LR_degrade,
    alphaBeta within 1e-5, LR = LR_raw / scale_num and depth bit-exact."""
    import os
    import tactilesr_amd
    from tactilesr_amd.data import depth2tactile as D
    torch.manual_seed(2)
    net = tactilesr_amd.tPSFNet(1.4, None).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    LR_raw = torch.rand(9, 3, 4, 4, generator=g) * 800
    depth = (torch.rand(9, 100, 100, generator=g) > 0.7).float()
    entries = D.synthesize(net, LR_raw, depth, scale_num=100.0, batch_size=4)
    path = os.path.join(tmp_path, "SRdataset_test.npy")
    D.save_dataset(path, entries)
    ds = np.load(path, allow_pickle=True)
    assert len(ds) == 9
    with torch.no_grad():
        for i in (0, 4, 8):
            HR, LRd, _, ab = net((LR_raw[i:i + 1] / 100).cuda(), depth[i:i + 1].unsqueeze(1).cuda())
            it = ds[i].item()
            assert torch.equal(it["HR"], HR[0].cpu()) and torch.equal(it["LR_degrade"], LRd[0].cpu())
            assert torch.equal(it["alphaBeta"], ab[0, 0].cpu()) and it["depth"].shape == (1, 100, 100)
            assert torch.allclose(it["LR"], LR_raw[i] / 100)
        rHR, rLRd, _, rab = O.tpsf_forward(sd, LR_raw / 100, depth.unsqueeze(1))
    for i in range(9):
        it = ds[i].item()
        assert set(it) == {"LR", "depth", "HR", "LR_degrade", "alphaBeta"}
        assert relerr(it["HR"], rHR[i]) < 1e-5 and relerr(it["LR_degrade"], rLRd[i]) < 1e-5, i
        assert relerr(it["alphaBeta"], rab[i, 0]) < 1e-5 and it["alphaBeta"].shape == (3,), i
        # (LR is divided on the device: its fp32 division may differ from the host's in the last bit)
        assert relerr(it["LR"], LR_raw[i] / 100) < 2e-7 and torch.equal(it["depth"], depth[i].unsqueeze(0)), i


@pytest.mark.parametrize("M,N,K,act,ta,tb", [
    (37, 3, 256, 2, False, True), (130, 256, 48, 1, False, True), (64, 64, 16, 0, False, False),
    (100, 70, 33, 0, True, False), (1, 129, 300, 0, False, False), (257, 65, 1030, 1, True, True),
    # the MLP's own shapes at a batch that runs whole 128-row tiles and the 16-B load paths: forward (k-fast x k-fast),
    # dx (k-fast x n-fast), dW (m-fast x n-fast, K = batch), and ragged edges on every path
    (640, 1024, 256, 1, False, True), (600, 256, 1024, 0, False, False), (1024, 256, 2048, 0, True, False),
    (3, 256, 2048, 0, True, False), (515, 260, 132, 2, False, True), (260, 516, 1028, 0, True, False),
])
def test_sgemm_mfma_strides_activations_and_splitk(M, N, K, act, ta, tb):
    """tsr_sgemm (fp32 matrix cores) with every stride form the MLP uses, ragged tiles, the three epilogues,
    and the split-K form + tsr_reduce_splits -- against torch fp64."""
    from tactilesr_amd._lib import call, ptr, stream, c_int as I, c_longlong as L, c_float as Fl
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(K, M, generator=g).cuda() if ta else torch.randn(M, K, generator=g).cuda()
    Bm = torch.randn(N, K, generator=g).cuda() if tb else torch.randn(K, N, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    sa = (1, M) if ta else (K, 1)
    sb = (1, K) if tb else (N, 1)
    A64 = (A.t() if ta else A).double().cpu()
    B64 = (Bm.t() if tb else Bm).double().cpu()
    ref = A64 @ B64 + bias.double().cpu()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = torch.nn.functional.softplus(ref)
    C = torch.empty(M, N, device="cuda")
    call("tsr_sgemm", ptr(A), L(sa[0]), L(sa[1]), ptr(Bm), L(sb[0]), L(sb[1]), ptr(bias), ptr(C), I(M), I(N), I(K), I(act),
         stream())
    assert relerr(C, ref) < 2e-6
    for ns in (1, 3, 8):
        slab = torch.full((ns, M, N), float("nan"), device="cuda")
        call("tsr_sgemm_splitk", ptr(A), L(sa[0]), L(sa[1]), ptr(Bm), L(sb[0]), L(sb[1]), ptr(slab), I(M), I(N), I(K), I(ns),
             stream())
        out = torch.empty(M, N, device="cuda")
        call("tsr_reduce_splits", ptr(slab), ptr(out), L(M * N), I(ns), Fl(1.0), stream())
        assert relerr(out, A64 @ B64) < 2e-6, ns
    # fused ReLU-backward mask (tsr_sgemm_masked), a strided slab shared by two results + column sums (tsr_sgemm_splitk_strided,
    # tsr_colsum_splitk): what the tPSFNet backward issues
    ref_act = torch.randn(M, N, generator=g).cuda()
    Cm = torch.empty(M, N, device="cuda")
    call("tsr_sgemm_masked", ptr(A), L(sa[0]), L(sa[1]), ptr(Bm), L(sb[0]), L(sb[1]), ptr(ref_act), ptr(Cm), I(M), I(N), I(K),
         stream())
    assert relerr(Cm, (A64 @ B64) * (ref_act.cpu() > 0)) < 2e-6
    ns, tot = 5, M * N + 64 + N
    slab = torch.full((ns * tot,), float("nan"), device="cuda")
    call("tsr_sgemm_splitk_strided", ptr(A), L(sa[0]), L(sa[1]), ptr(Bm), L(sb[0]), L(sb[1]), ptr(slab), L(tot), I(M), I(N), I(K),
         I(ns), stream())
    Y = torch.randn(K, N, generator=g).cuda()                 # column sums over K rows, same ranges
    call("tsr_colsum_splitk", ptr(Y), ptr(slab[M * N + 64:]), L(tot), I(K), I(N), I(ns), stream())
    sl = slab.view(ns, tot)
    assert relerr(sl[:, :M * N].double().sum(0).view(M, N), A64 @ B64) < 2e-6
    assert relerr(sl[:, M * N + 64:].double().sum(0), Y.double().cpu().sum(0)) < 2e-6
    assert torch.isnan(sl[:, M * N:M * N + 64]).all()          # the gap between the two results is not touched


def test_tpsf_kernels_wide_dynamic_range_batch():
    """MFMA Toeplitz-GEMM forward/backward (two scaled fp16 planes) against an fp64 torch restatement on inputs the
    golden fixture does not reach: signed depth, tiny and large magnitudes, narrow and wide PSFs, 300 samples so that
    the persistent workgroups loop."""
    from tactilesr_amd._lib import call, ptr, stream, c_int as I
    B = 300
    g = torch.Generator().manual_seed(5)
    depth = torch.rand(B, 100, 100, generator=g) * 10
    depth[1] = depth[1] * 1e-3
    depth[2] = (depth[2] - 5) * 40                  # signed, large
    depth[3] = 0
    depth[3, 40:60, 40:60] = 7.5                    # a real plateau
    ab = torch.rand(B, 3, generator=g) * torch.tensor([1.0, 3.0, 2.0]) + torch.tensor([0.2, 0.25, 0.6])
    dl = torch.randn(B, 16, generator=g)
    d, a_, dl_ = depth.cuda(), ab.cuda(), dl.cuda()
    HR = torch.empty(B, 1, 100, 100, device="cuda")
    LRd = torch.empty(B, 16, device="cuda")
    psf = torch.empty(B, 1, 99, 99, device="cuda")
    dab = torch.empty(B, 3, device="cuda")
    call("tpsf_forward", ptr(d), ptr(a_), ptr(HR), ptr(LRd), ptr(psf), I(B), stream())
    work = torch.empty(B * 10000, device="cuda")
    call("tpsf_backward", ptr(d), ptr(a_), ptr(HR), ptr(dl_), ptr(dab), ptr(work), I(B), stream())
    n = 24                                          # fp64 restatement (oracle functions) on a subset
    ab64 = ab[:n].double().requires_grad_(True)
    HR64, LR64, psf64 = O.tpsf_forward_from_ab(ab64, depth[:n].double())
    (LR64.reshape(n, 16) * dl[:n].double()).sum().backward()
    for i in range(n):
        assert relerr(HR[i], HR64[i]) < 1e-5, i
        assert relerr(psf[i], psf64[i]) < 1e-5, i
    assert relerr(LRd[:n], LR64.reshape(n, 16)) < 1e-5
    assert relerr(dab[:n], ab64.grad) < 1e-5
    assert torch.isfinite(HR).all() and torch.isfinite(dab).all()


def test_tpsf_B8192_forward_backward_tiling_invariance():
    """BASELINE configs[2] size: tPSFNet forward + backward at B = 8192 = 16 distinct samples tiled 512x.  Samples are
    independent, so every replica is bit-identical to the first; the 16 distinct outputs match the CPU oracle (direct
    99x99 conv) to 1e-5, and the MLP gradients of the trainer loss (a mean over the batch) equal the oracle's on the
    16 base samples."""
    import tactilesr_amd
    from tactilesr_amd.train import tPSFNet_train as TP
    sd = O.random_state_dict(O.tpsf_state_shapes(), 8192)
    net = tactilesr_amd.tPSFNet(gama=1.4, perception_scale=None, device="cuda")
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    g = torch.Generator().manual_seed(8193)
    depth = (torch.rand(16, 100, 100, generator=g) > 0.7).float()
    LR_raw = torch.rand(16, 3, 4, 4, generator=g) * 800
    reps = 512
    LRb, db = LR_raw.repeat(reps, 1, 1, 1).cuda(), depth.repeat(reps, 1, 1).cuda()
    assert LRb.shape[0] == 8192
    with torch.no_grad():
        HR, LRd, psf, ab = net(LRb / 100, db.unsqueeze(1))
    for t in (HR, LRd, psf, ab):
        v = t.view(reps, 16, -1)
        assert torch.equal(v, v[:1].expand_as(v))
    with torch.no_grad():
        rHR, rLRd, rpsf, rab = O.tpsf_forward(sd, LR_raw / 100, depth.unsqueeze(1))
    assert relerr(HR[:16], rHR) < 1e-5 and relerr(LRd[:16], rLRd) < 1e-5
    assert relerr(psf[:16], rpsf) < 1e-5 and relerr(ab[:16], rab) < 1e-5
    loss, _ = TP.train_cal_loss(net, (LRb, db), 100.0)
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo = O.tpsf_train_cal_loss(leaves, LR_raw, depth)
    gl = torch.autograd.grad(lo, list(leaves.values()))
    assert abs(loss.item() - lo.item()) < 1e-5 * lo.item()
    for (k, p), ref in zip(net.named_parameters(), gl):
        assert relerr(p.grad, ref) < 2e-5, k


def test_seqs_dataset_generator_matches_the_batch1_loop(tmp_path):
    """The batched Seqs generator against a literal batch-1 restatement of the reference loop
    (data/SeqsDataset/seqsDepth2Tactile.py:47-98): per item the seven LR frames newest first -> LR (21,4,4), tPSFNet on
    the 30-degree tap only, entries {LR, depth, HR}, translation 0 -> test, 1 -> validation, rest -> train; and the
    file reads back like utility/load_tactile_dataset.py:52-57."""
    import os
    import tactilesr_amd
    from tactilesr_amd.data import seqs_depth2tactile as S
    torch.manual_seed(4)
    net = tactilesr_amd.tPSFNet(1.4, None).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    nc, nt, sc = 2, 3, 4
    N = nc * 81 * sc
    g = torch.Generator().manual_seed(5)
    LR_raw = torch.rand(N, 3, 4, 4, generator=g) * 800
    depth = (torch.rand(N, 100, 100, generator=g) > 0.7).float()
    out = S.synthesize_seqs(net, LR_raw, depth, n_contacts=nc, n_trans=nt, sample_cnt=sc, batch_size=7)
    assert (len(out["train"]), len(out["validation"]), len(out["test"])) == (nc * 1 * sc, nc * sc, nc * sc)
    it = {"train": 0, "validation": 0, "test": 0}
    with torch.no_grad():
        for c in range(nc):
            for t in range(nt):
                for s in range(sc):
                    taps = [sc - 1 + sc * (r + t * 9) + sc * 81 * c for r in range(6)] + [s + sc * (6 + t * 9) + sc * 81 * c]
                    lr = [LR_raw[i] / 100 for i in taps]                                  # 0, 5, ..., 30 degrees
                    HR, _, _, _ = net(lr[6].unsqueeze(0).cuda(), depth[taps[6]].view(1, 1, 100, 100).cuda())
                    split = "validation" if t == 1 else ("test" if t == 0 else "train")
                    e = out[split][it[split]][0]
                    it[split] += 1
                    assert set(e) == {"LR", "depth", "HR"} and e["LR"].shape == (21, 4, 4)
                    assert torch.equal(e["LR"], torch.cat(lr[::-1], dim=0))               # newest first
                    assert torch.equal(e["depth"], depth[taps[6]].unsqueeze(0))
                    assert torch.equal(e["HR"], HR[0].cpu())
                    # ... and the stored HR is what the CPU oracle computes for the 30-degree tap (direct 99x99 conv)
                    rHR, _, _, _ = O.tpsf_forward(sd, lr[6].unsqueeze(0), depth[taps[6]].view(1, 1, 100, 100))
                    assert relerr(e["HR"], rHR[0]) < 1e-5, (c, t, s)
    path = os.path.join(tmp_path, "SRdataset_train_32.npy")
    S.save_seqs_dataset(path, out["train"])
    ds = np.load(path, allow_pickle=True)
    assert len(ds) == len(out["train"]) and np.ascontiguousarray(ds[1].item()["LR"]).shape == (21, 4, 4)
    assert np.ascontiguousarray(ds[1].item()["HR"]).shape == (1, 100, 100)
