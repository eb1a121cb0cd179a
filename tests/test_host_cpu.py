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
