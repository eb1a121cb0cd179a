"""CPU-side checks: C-ABI library loads and exports every declared symbol, the host
module mirrors the reference interface (state_dict names, init RNG order, error
behaviour).  No compute calls (there is no GPU in the build container)."""
import hashlib
import os
import re

import numpy as np
import pytest
import torch

import tactilesr_amd
from tactilesr_amd import _lib
from oracle import tactilesr_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "tactilesr_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|long long)\s+(tsr_\w+|tpsf_\w+)\s*\(", hdr, re.M))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.tsr_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("tag,cfg", [("t1", dict()), ("t7", dict(seqsCnt=7))])
def test_init_matches_reference_rng_order(golden, tag, cfg):
    g = golden("init")
    torch.manual_seed(42)
    m = tactilesr_amd.TactileSR(**cfg)
    sd = m.state_dict()
    assert len(sd) == int(g[f"{tag}/nkeys"])
    assert sum(p.numel() for p in m.parameters()) == int(g[f"{tag}/nparams"])
    assert list(sd.keys()) == list(O.tactilesr_state_shapes(**cfg).keys())
    assert sd_hash(sd) == str(g[f"{tag}/sha256"])


def test_interface_attributes_and_errors():
    m = tactilesr_amd.TactileSR(scale_factor=10, seqsCnt=2, axisCnt=3, patternFeatureExtraLayerCnt=1,
                                forceFeatureExtraLayerCnt=1)
    assert (m.scale_factor, m.seqsCnt, m.axisCnt, m.taxel_cnt) == (10, 2, 3, 4)
    assert isinstance(m.patternFeatureExtra_layer, torch.nn.Sequential)
    m.eval()
    with pytest.raises(AssertionError, match="input channel should be same"):
        m(torch.zeros(1, 3, 4, 4))
    with pytest.raises(_lib.TactileSRHipError, match="no CPU fallback"):
        m(torch.zeros(1, 6, 4, 4))          # CPU tensor: must fail loudly, never fall back


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = ("import sys; import tactilesr_amd; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; "
            "assert not bad, bad")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=REPO)
    for root, _, files in os.walk(os.path.join(REPO, "tactilesr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No fallback: without the HIP library every product entry point raises (it never routes to the oracle)."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(tmp_path, "nope", "libtactilesr_hip.so"))
    with pytest.raises(_lib.TactileSRHipError, match="no CPU fallback"):
        _lib.load()
    with pytest.raises(_lib.TactileSRHipError):
        _lib.call("tsr_abi_version")
    monkeypatch.undo()
    assert _lib.load().tsr_abi_version() == _lib.ABI_VERSION


def test_tpsfnet_interface_and_cpu_refusal():
    net = tactilesr_amd.tPSFNet(gama=1.4, perception_scale=None, device="cpu")
    assert (net.gama, net.perception_scale) == (1.4, None)
    assert list(net.state_dict()) == [f"MLP_layer.{i}.{p}" for i in (1, 3, 5, 7) for p in ("weight", "bias")]
    assert net.PSF_sdf.shape == (1, 1, 99, 99) and net.LR_masking_sdf.shape == (4, 4, 100, 100)
    assert abs(float(net.PSF_sdf.max()) - 10) < 1e-5 and abs(float(net.LR_masking_sdf.max()) - 10) < 1e-5
    with pytest.raises(AssertionError, match="Batch size of LR tactile and depth"):
        net(torch.zeros(2, 3, 4, 4), torch.zeros(3, 1, 100, 100))
    with pytest.raises(_lib.TactileSRHipError, match="no CPU fallback"):
        net(torch.zeros(2, 3, 4, 4), torch.zeros(2, 1, 100, 100))


def test_reference_import_line_and_tactilesrcnn_init(golden):
    """The reference trainers' import line works verbatim against the drop-in module (train/tactileSR_train.py:24,
    train/tactileSRSeqs_train.py:24): `TactileSRCNN` exists with the reference's constructor, state_dict key order and
    seeded init (model/tactileSR_model.py:101-145; sha256 of the reference's own seed-42 construction in
    tests/golden/srcnn.npz); it refuses CPU tensors and train mode loudly."""
    from tactilesr_amd.model.tactileSR_model import TactileSR, TactileSRCNN   # noqa: F401  (the reference's line)
    g = golden("srcnn")
    torch.manual_seed(42)
    m = TactileSRCNN()
    sd = m.state_dict()
    assert list(sd.keys()) == list(O.tactilesrcnn_state_shapes().keys())
    assert sd_hash(sd) == bytes(g["init_sha"]).hex()
    assert "conv_impl='fp16x3'" in repr(m)
    with pytest.raises(_lib.TactileSRHipError, match="no CPU fallback"):
        m.eval()(torch.zeros(1, 3, 4, 4))


def test_default_environment_selects_the_parity_arithmetic(monkeypatch):
    """the arithmetic is an explicit constructor / attribute choice printed by repr(model); the
    defaults are the parity paths and NO environment variable changes them (the former TSR_CONV_IMPL / TSR_TRAIN_IMPL /
    TSR_HEAD_IMPL / TSR_FUSE* switches are gone; the library holds no getenv at all)."""
    for k, v in (("TSR_CONV_IMPL", "bf16"), ("TSR_TRAIN_IMPL", "bf16"), ("TSR_HEAD_IMPL", "bf16"), ("TSR_FUSE1X1", "0"),
                 ("TSR_FUSE_PAIR", "0"), ("TSR_CONV_M32", "1"), ("TSR_WGRAD_OLD", "1")):
        monkeypatch.setenv(k, v)
    m = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    assert (m.conv_impl, m.train_impl, m.head_impl, m.fuse_1x1, m.fuse_pair) == ("fp16x3", "fp16x3", None, True, True)
    assert m.train_engine().impl == "fp16x3" and m.train_engine().nsplit == -2
    r = repr(m).splitlines()[1]
    assert "conv_impl='fp16x3'" in r and "train_impl='fp16x3'" in r and "fp32-grade" in r
    blk = tactilesr_amd.MSRB()
    assert (blk.conv_impl, blk.train_impl) == ("fp16x3", "fp16x3")
    # explicit choices: constructor keywords (after the reference's positional arguments) or attributes
    m2 = tactilesr_amd.TactileSR(10, 1, 3, 1, 1, conv_impl="bf16", train_impl="bf16")
    assert "REDUCED precision" in repr(m2) and m2.train_engine().io16
    m2.train_impl = "f32"
    assert m2.train_engine().impl == "f32" and not m2.train_engine().io16
    with pytest.raises(_lib.TactileSRHipError, match="conv_impl"):
        tactilesr_amd.TactileSR(conv_impl="fp8")
    m2.train_impl = "int4"
    with pytest.raises(_lib.TactileSRHipError, match="train_impl"):
        m2.train_engine()
    blk.train_impl = "bf16"                  # only pinned for the whole network (bf16-emulating oracle): refused here
    with pytest.raises(_lib.TactileSRHipError, match="train_impl"):
        blk.block_engine()
    # the shipped library reads no environment variable and is not a variant build
    src = "".join(open(os.path.join(REPO, "tactilesr_amd", "csrc", f)).read()
                  for f in os.listdir(os.path.join(REPO, "tactilesr_amd", "csrc")))
    assert "getenv" not in src and not re.search(r"TSR_ABL_|TSR_EXP_", src)
    assert _lib.build_flags() == 0
