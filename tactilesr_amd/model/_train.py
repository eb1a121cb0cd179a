"""Train-mode forward (batch-statistics BatchNorm) and full backward of TactileSR on the HIP
kernels, exposed to autograd as ONE ``torch.autograd.Function`` so that the reference's
step protocol -- ``loss = criterion(model(LR), HR); loss.backward(); optimizer.step()``
(train/tactileSR_train.py:41-51, cpu/trainer.py:346-362) -- works unchanged and ``.grad``
appears on the registered ``nn.Parameter``s.

Forward keeps, per BN layer, only the raw bias-free conv output ``z`` (CB16) plus four
per-channel vectors; ``relu(bn(z))`` is never materialised -- consumers apply it while
staging their input tiles.  Backward = dgrad (the same MFMA conv kernel with flipped /
transposed weights and a ReLU-mask + BN-reduction epilogue) + wgrad (MFMA split-K over the
batch) + a small elementwise BN-backward pass per BN layer.
"""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_void_p, c_float, c_double, c_longlong, POINTER, byref

import torch

from .. import _lib
from .._lib import call, ptr, stream

_I, _F, _D, _L = c_int, c_float, c_double, c_longlong


class ConvDesc(ctypes.Structure):
    """Mirror of ``tsr_conv_desc`` (include/tactilesr_hip.h)."""
    _fields_ = [
        ("in_", c_void_p), ("in_ctot", c_int), ("in_coff", c_int), ("cin", c_int),
        ("w_packed", c_void_p), ("cout", c_int), ("ks", c_int),
        ("scale", c_void_p), ("shift", c_void_p),
        ("res", c_void_p), ("res_ctot", c_int), ("res_coff", c_int),
        ("out", c_void_p), ("out_ctot", c_int), ("out_coff", c_int), ("relu", c_int),
        ("B", c_int), ("H", c_int), ("W", c_int),
        ("in_scale", c_void_p), ("in_shift", c_void_p),
        ("res_scale", c_void_p), ("res_shift", c_void_p),
        ("epi_mode", c_int),
        ("mask", c_void_p), ("mask_ctot", c_int), ("mask_coff", c_int),
        ("mask_scale", c_void_p), ("mask_shift", c_void_p),
        ("bn_a", c_void_p), ("bn_b", c_void_p),
        ("slab", c_void_p), ("slab_cnt", c_void_p),
        ("nsplit", c_int),
        ("in_amax", c_void_p), ("w_inv_scale", c_float), ("out_amax", c_void_p), ("w_amax", c_void_p),
    ]


def _p(t):
    return None if t is None else ptr(t).value


class Act:
    """A CB16 activation slice, optionally 'virtual': value = relu(buf*scale+shift)."""
    __slots__ = ("buf", "ctot", "coff", "c", "scale", "shift", "xa", "xb", "amax", "mat")

    def __init__(self, buf, ctot, coff, c, scale=None, shift=None, xa=None, xb=None, amax=None):
        self.buf, self.ctot, self.coff, self.c = buf, ctot, coff, c
        self.scale, self.shift, self.xa, self.xb = scale, shift, xa, xb
        self.amax = amax     # fp16x3 only: 1-element device tensor holding max|buf| (of the raw stored values)
        self.mat = None      # bf16 storage: the materialised relu(buf*scale+shift) once a kernel needed it plain


def conv_ex(*, B, H, W, src: Act, w, cout, ks, out, out_ctot, out_coff, scale=None, shift=None, relu=0,
            res: Act = None, epi_mode=0, mask: Act = None, bn=False, slab=None, slab_cnt=None, nsplit=0,
            w_inv_scale=1.0, out_amax=None, w_amax=None):
    d = ConvDesc()
    d.nsplit = nsplit
    d.in_amax, d.w_inv_scale, d.out_amax, d.w_amax = _p(src.amax), w_inv_scale, _p(out_amax), _p(w_amax)
    d.in_, d.in_ctot, d.in_coff, d.cin = _p(src.buf), src.ctot, src.coff, src.c
    d.w_packed, d.cout, d.ks = _p(w), cout, ks
    d.scale, d.shift = _p(scale), _p(shift)
    if res is not None:
        d.res, d.res_ctot, d.res_coff = _p(res.buf), res.ctot, res.coff
        d.res_scale, d.res_shift = _p(res.scale), _p(res.shift)
    d.out, d.out_ctot, d.out_coff, d.relu = _p(out), out_ctot, out_coff, int(relu)
    d.B, d.H, d.W = B, H, W
    d.in_scale, d.in_shift = _p(src.scale), _p(src.shift)
    d.epi_mode = epi_mode
    if mask is not None:
        d.mask, d.mask_ctot, d.mask_coff = _p(mask.buf), mask.ctot, mask.coff
        d.mask_scale, d.mask_shift = _p(mask.scale), _p(mask.shift)
        if bn:
            d.bn_a, d.bn_b = _p(mask.xa), _p(mask.xb)
    d.slab, d.slab_cnt = _p(slab), _p(slab_cnt)
    st = _lib.load().tsr_conv2d_ex(byref(d), stream())
    if st != 0:
        raise _lib.TactileSRHipError(f"tsr_conv2d_ex failed: status {st}")


def _pack(w, cout, cin, ks, nsplit=0, w_amax=None):
    if nsplit == -1:          # bf16 activation storage: the weights are the one-plane bf16 pack
        nsplit = 1
    if nsplit == -2:
        n = _lib.load().tsr_conv_weight_bf16s_elems(cout, cin, ks, 2)
        wp = torch.empty(n, dtype=torch.float16, device=w.device)
        call("tsr_pack_conv_weight_f16s_dev", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), ptr(w_amax), stream())
        return wp
    if nsplit:
        n = _lib.load().tsr_conv_weight_bf16s_elems(cout, cin, ks, nsplit)
        wp = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        call("tsr_pack_conv_weight_bf16s", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), _I(nsplit), stream())
        return wp
    wp = torch.empty(cout * cin * ks * ks, dtype=torch.float32, device=w.device)
    call("tsr_pack_conv_weight", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), stream())
    return wp


def _pack_dgrad(w, cout, cin, ks, ci0, nprime, nsplit=0, w_amax=None):
    if nsplit == -3:          # bf16 activation storage, the launch runs csrc/conv_b16k.hip: that kernel's slab layout
        wp = torch.empty(_lib.load().tsr_conv_weight_b16k_elems(nprime, cout, ks), dtype=torch.bfloat16, device=w.device)
        call("tsr_pack_conv_weight_dgrad_b16k", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), _I(ci0), _I(nprime), stream())
        return wp
    if nsplit == -1:
        nsplit = 1
    if nsplit == -2:
        n = _lib.load().tsr_conv_weight_bf16s_elems(nprime, cout, ks, 2)
        wp = torch.empty(n, dtype=torch.float16, device=w.device)
        call("tsr_pack_conv_weight_dgrad_f16s_dev", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), _I(ci0), _I(nprime),
             ptr(w_amax), stream())
        return wp
    if nsplit:
        n = _lib.load().tsr_conv_weight_bf16s_elems(nprime, cout, ks, nsplit)
        wp = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        call("tsr_pack_conv_weight_dgrad_bf16s", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), _I(ci0), _I(nprime),
             _I(nsplit), stream())
        return wp
    wp = torch.empty(nprime * cout * ks * ks, dtype=torch.float32, device=w.device)
    call("tsr_pack_conv_weight_dgrad", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), _I(ci0), _I(nprime), stream())
    return wp


TRAIN_IMPLS = {"f32": 0, "bf16x6": 3, "fp16x3": -2, "bf16": -1, "bf16op": 1}


class _Ctx:
    """Everything backward needs (device buffers stay alive through this object)."""


class TrainEngine:
    """Runs one train-mode forward / backward of a ``TactileSR`` module."""

    def __init__(self, model, impl: str = "fp16x3"):
        self.m = model
        self.debug = None        # tools may set a dict: backward then stores clones of dz tensors in it
        self.keep_ctx = False    # tests: keep the last forward's context alive in `last_ctx` (activation_masks)
        self.last_ctx = None
        # gradient arena (tactilesr_amd.ddp): one flat buffer in backward-production order; weight gradients are
        # reduced straight into it, so p.grad addresses are stable across steps (the fused Adam's table is built
        # once) and a GradSync can put finished buckets on the wire from inside backward
        self.arena = None
        self.grad_sync = None
        self.n_buckets = 8
        self.profile = None      # bench.py: dict -> HIP-event brackets per launch family, on the launch stream
        # conv arithmetic of the train path (chosen EXPLICITLY: `TactileSR(train_impl=...)` / `model.train_impl = ...`;
        # no environment variable changes it): 0 = fp32 MFMA, 3 = split-bf16 (six products, fp32-equivalent),
        # -2 = two scaled fp16 planes (three products; operand scales from device-side max|.| scalars),
        # -1 = "bf16": BASELINE's "bf16" configurations -- every stored activation / gradient tensor is bf16 CB16 (saved
        #      pre-activations z, dz, dgrad outputs), plain bf16 MFMA operands, fp32 accumulation, fp32 master weights,
        #      BatchNorm statistics, weight gradients and Adam (the reference's reduced-precision switch is the unused
        #      fp16 autocast of cpu/trainer.py:96,203,346-362); 1 = "bf16op": bf16 operands on fp32 tensors (A/B only)
        if impl not in TRAIN_IMPLS:
            raise _lib.TactileSRHipError(f"train_impl {impl!r}: expected one of {sorted(TRAIN_IMPLS)}")
        self.impl = impl
        self.nsplit = TRAIN_IMPLS[impl]
        self.f16 = self.nsplit == -2
        self.io16 = self.nsplit == -1
        self.act_dtype = torch.bfloat16 if self.io16 else torch.float32

    def _timed(self, key):
        """Context manager: bracket the launches inside with two HIP events when profiling is on."""
        eng = self

        class _T:
            def __enter__(self_):
                self_.on = eng.profile is not None
                if self_.on:
                    self_.e0, self_.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    self_.e0.record()

            def __exit__(self_, *exc):
                if self_.on:
                    self_.e1.record()
                    eng.profile.setdefault(key, []).append((self_.e0, self_.e1))
                return False
        return _T()

    def _mfma_convs(self):
        m = self.m
        cv = [m.inputContact_layer[0], m.output_layer[0]]
        cv += [seq[4] for seq in m.inputLayer_pattern_list]
        for blk in m.patternFeatureExtra_layer:
            cv += [blk.conv_3_1[0], blk.conv_5_1[0], blk.conv_3_2[0], blk.conv_5_2[0], blk.confusion]
        for rb in m.forceFeatureExtra_layer:
            cv += [rb.conv1, rb.conv2]
        return cv

    def _weight_scales(self, c):
        """fp16x3: max|w| per conv weight as DEVICE scalars (one batched launch, no host round trip); the pack kernels
        and the convolutions derive the same power-of-two scale (max|w|*wscale in [2^13, 2^14)) from them."""
        c.wamax = {}
        if not self.f16:
            return
        cv = self._mfma_convs()
        mx = torch.stack(torch._foreach_norm([k.weight.detach() for k in cv], float("inf")))
        for i, k in enumerate(cv):
            c.wamax[id(k)] = mx[i:i + 1]

    def _amax_pool(self, c, dev):
        n = 64 + 8 * (len(self.m.patternFeatureExtra_layer) + len(self.m.forceFeatureExtra_layer) + self.m.seqsCnt)
        pool = torch.zeros(n, dtype=torch.float32, device=dev) if self.f16 else None
        state = [0]

        def new():
            if pool is None:
                return None
            i = state[0]
            state[0] += 1
            return pool[i:i + 1]
        return new

    # ------------------------------------------------------------------ helpers
    def _bn_finalize(self, c: _Ctx, conv_bias, bn, slab, cnt, entries, C):
        dev = slab.device
        vec = torch.empty(4, C, dtype=torch.float32, device=dev)
        call("tsr_bn_stats_finalize", ptr(slab), ptr(cnt), _I(entries), _I(C),
             ptr(conv_bias.detach() if conv_bias is not None else None), ptr(bn.weight.detach()),
             ptr(bn.bias.detach()), ptr(bn.running_mean), ptr(bn.running_var), _F(bn.momentum), _F(bn.eps),
             ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(c.work), stream())
        c.nbt.append(bn.num_batches_tracked)      # (+1 for all of them in one launch at the end of forward)
        return vec   # rows: scale, shift, xhat_a, xhat_b

    @staticmethod
    def _bump_nbt(c):
        """num_batches_tracked += 1 of every BatchNorm layer the forward ran, as ONE launch (27 one-element launches before)."""
        if c.nbt:
            torch._foreach_add_(c.nbt, 1)
            c.nbt = []

    def _stem(self, x, ctot_in, coff, A, hin, win, sf, w, dst, relu, B, am):
        """bilinear x sf + conv 3 -> 64 (raw or ReLU'd) into a 64-channel CB16 buffer of the engine's storage type."""
        if self.io16:
            call("tsr_stem_fwd_b16", ptr(x), _I(ctot_in), _I(coff), _I(A), _I(hin), _I(win), _I(sf), ptr(w), ptr(None),
                 ptr(None), ptr(dst), _I(64), _I(0), _I(relu), _I(B), stream())
        else:
            call("tsr_stem_fwd", ptr(x), _I(ctot_in), _I(coff), _I(A), _I(hin), _I(win), _I(sf), ptr(w), ptr(None),
                 ptr(None), ptr(dst), _I(64), _I(0), _I(relu), _I(B), ptr(am), stream())

    def _entries(self, c, cout, ks, nsplit=None):
        """Statistics-slab entries the conv launch (cout, ks) of this engine's arithmetic (or of the form `nsplit`) writes."""
        return _lib.load().tsr_conv2d_slab_entries_ex(c.B, c.H, c.W, cout, ks, self.nsplit if nsplit is None else nsplit)

    def _packw(self, c, conv):
        """(packed weight, device scalar max|w| or None)"""
        w = conv.weight.detach().contiguous()
        wa = c.wamax.get(id(conv))
        return _pack(w, w.shape[0], w.shape[1], w.shape[2], self.nsplit, wa), wa

    def _plain(self, c, a: Act) -> Act:
        """bf16 storage: `a` as a tensor the LDS-DMA kernels (conv_b16k / wgrad_b16k: operands go from HBM to LDS as stored)
        can read -- a virtual activation relu(buf*scale+shift) is materialised ONCE (tsr_bn_relu_b16) and kept on the Act:
        the forward convs of an MSRB's second stage and, in backward, their weight gradients all read the same tensor."""
        if a.scale is None:
            return a
        if a.mat is None:
            a.mat = torch.empty(c.B * a.c * c.HW, dtype=torch.bfloat16, device=a.buf.device)
            call("tsr_bn_relu_b16", ptr(a.buf), _I(a.ctot), _I(a.coff), _I(a.c), ptr(a.scale), ptr(a.shift), ptr(a.mat),
                 _I(c.B), _I(c.HW), stream())
        return Act(a.mat, a.c, 0, a.c)

    def _b16k(self, cout, cin, ks):
        """bf16 storage: this forward conv shape runs csrc/conv_b16k.hip (tsr_conv2d_ex, nsplit = -3)."""
        return self.io16 and ks > 1 and bool(_lib.load().tsr_conv2d_ex_dgrad_b16k(cout, cin, ks))

    def _packw_b16k(self, conv):
        w = conv.weight.detach().contiguous()
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
        wp = torch.empty(_lib.load().tsr_conv_weight_b16k_elems(cout, cin, ks), dtype=torch.bfloat16, device=w.device)
        call("tsr_pack_conv_weight_b16k", ptr(w), ptr(wp), _I(cout), _I(cin), _I(ks), stream())
        return wp

    def _pair_bn(self, c: _Ctx, X: Act, blk, cat1):
        """bf16 storage: conv_3_1 || conv_5_1 of an MSRB (reference model/tactileSR_model.py:198-200) as ONE 5x5 launch with
        128 output channels on conv_b16k (tsr_conv2d_ex, nsplit = -4: the 3x3 weight sits in the inner taps of its half) and
        ONE statistics pass over the 128 channels; returns the 4x128 BN vectors [conv_3_1 | conv_5_1]."""
        c3, b3, c5, b5 = blk.conv_3_1[0], blk.conv_3_1[1], blk.conv_5_1[0], blk.conv_5_1[1]
        if b3.momentum != b5.momentum or b3.eps != b5.eps or (c3.bias is None) != (c5.bias is None):
            raise _lib.TactileSRHipError("stage-1 pair: the two conv + BatchNorm layers differ in momentum / eps / bias")
        cin = c3.weight.shape[1]
        w = torch.cat([torch.nn.functional.pad(c3.weight.detach(), (1, 1, 1, 1)), c5.weight.detach()], 0).contiguous()
        wp = torch.empty(_lib.load().tsr_conv_weight_b16k_pair_elems(cin), dtype=torch.bfloat16, device=w.device)
        call("tsr_pack_conv_weight_b16k_pair", ptr(w), ptr(wp), _I(cin), stream())
        with self._timed(("fwd", 5, 128, cin)):
            conv_ex(B=c.B, H=c.H, W=c.W, src=self._plain(c, X), w=wp, cout=128, ks=5, out=cat1, out_ctot=128, out_coff=0,
                    epi_mode=1, slab=c.slab, slab_cnt=c.slab_cnt, nsplit=-4)
        cat = lambda a, b: torch.cat([a.detach(), b.detach()])
        vec = torch.empty(4, 128, dtype=torch.float32, device=w.device)
        rm, rv = cat(b3.running_mean, b5.running_mean), cat(b3.running_var, b5.running_var)
        bias = cat(c3.bias, c5.bias) if c3.bias is not None else None
        gamma, beta = cat(b3.weight, b5.weight), cat(b3.bias, b5.bias)     # (named: a temporary's block would be reused)
        call("tsr_bn_stats_finalize", ptr(c.slab), ptr(c.slab_cnt), _I(self._entries(c, 128, 5)), _I(128), ptr(bias),
             ptr(gamma), ptr(beta), ptr(rm), ptr(rv), _F(b3.momentum), _F(b3.eps),
             ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(c.work), stream())
        b3.running_mean.copy_(rm[:64]); b5.running_mean.copy_(rm[64:])
        b3.running_var.copy_(rv[:64]); b5.running_var.copy_(rv[64:])
        c.nbt += [b3.num_batches_tracked, b5.num_batches_tracked]
        return vec

    def _conv_bn(self, c: _Ctx, src: Act, conv, bn, out, out_ctot, out_coff, out_amax=None):
        """conv (bias-free raw output) + batch statistics; returns the 4xC BN vectors."""
        w = conv.weight
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
        if self._b16k(cout, cin, ks):
            src, wp, wis, ns = self._plain(c, src), self._packw_b16k(conv), None, -3
        else:
            (wp, wis), ns = self._packw(c, conv), self.nsplit
        with self._timed(("fwd", ks, cout, cin)):
            conv_ex(B=c.B, H=c.H, W=c.W, src=src, w=wp, cout=cout, ks=ks, out=out, out_ctot=out_ctot,
                    out_coff=out_coff, epi_mode=1, slab=c.slab, slab_cnt=c.slab_cnt, nsplit=ns,
                    w_amax=wis, out_amax=out_amax)
        return self._bn_finalize(c, conv.bias, bn, c.slab, c.slab_cnt, self._entries(c, cout, ks), cout)

    # ------------------------------------------------------------------ forward
    def _new_ctx(self, B, H, W, dev):
        """Per-forward context: geometry + the statistics slabs / workspace every BatchNorm layer reuses."""
        c = _Ctx()
        HW = H * W
        c.B, c.H, c.W, c.HW = B, H, W, HW
        c.nbt = []          # BatchNorm num_batches_tracked counters this forward updates (bumped together: _bump_nbt)
        lib = _lib.load()
        c.entries = lib.tsr_conv2d_slab_entries(B, H, W)
        c.st_entries = st_entries = lib.tsr_cb16_stats_entries(B, HW)
        # statistics slabs are indexed by (workgroup, image slot): the entry count depends on how many images the
        # kernel variant of (C_out, k, arithmetic) puts in a workgroup -- ask the library for every shape in use
        e64 = max(lib.tsr_conv2d_slab_entries_ex(B, H, W, 64, k, self.nsplit) for k in (1, 3, 5))
        e128 = max(c.entries, max(lib.tsr_conv2d_slab_entries_ex(B, H, W, 128, k, self.nsplit) for k in (1, 3, 5)))
        if self.io16:
            e128 = max(e128, lib.tsr_conv2d_slab_entries_ex(B, H, W, 128, 1, -3))
        c.slab = torch.empty(max(e128 * 128 * 2, e64 * 64 * 2, st_entries * 64 * 2), dtype=torch.float32, device=dev)
        c.slab_cnt = torch.empty(max(e128, e64, st_entries), dtype=torch.float32, device=dev)
        c.work = torch.empty(512 * 128 * 3, dtype=torch.float64, device=dev)
        return c

    # ------------------------------------------------------------------ blocks (shared with the standalone modules)
    def _msrb_fwd(self, c, blk, X: Act, out, octot, ocoff, am_o, new_amax, buf):
        """One MSRB in train mode (reference model/tactileSR_model.py:196-206): X -> `out[ocoff:ocoff+64]`."""
        dev = X.buf.device
        s = _Ctx()
        s.X = X
        s.cat1, s.cat2 = buf(128), buf(256)
        s.bn_c1 = torch.empty(4, 128, dtype=torch.float32, device=dev)
        s.bn_c2 = torch.empty(4, 256, dtype=torch.float32, device=dev)
        am_c1, am_c2 = new_amax(), new_amax()
        if self._b16k(128, X.c, 5) and blk.conv_3_1[0].weight.shape[0] == 64:
            s.bn_c1[:] = self._pair_bn(c, X, blk, s.cat1)
        else:
            s.bn_c1[:, 0:64] = self._conv_bn(c, X, blk.conv_3_1[0], blk.conv_3_1[1], s.cat1, 128, 0, am_c1)
            s.bn_c1[:, 64:128] = self._conv_bn(c, X, blk.conv_5_1[0], blk.conv_5_1[1], s.cat1, 128, 64, am_c1)
        A1 = Act(s.cat1, 128, 0, 128, s.bn_c1[0], s.bn_c1[1], s.bn_c1[2], s.bn_c1[3], amax=am_c1)
        s.bn_c2[:, 0:128] = self._conv_bn(c, A1, blk.conv_3_2[0], blk.conv_3_2[1], s.cat2, 256, 0, am_c2)
        s.bn_c2[:, 128:256] = self._conv_bn(c, A1, blk.conv_5_2[0], blk.conv_5_2[1], s.cat2, 256, 128, am_c2)
        A2 = Act(s.cat2, 256, 0, 256, s.bn_c2[0], s.bn_c2[1], s.bn_c2[2], s.bn_c2[3], amax=am_c2)
        s.A1, s.A2 = A1, A2
        if self.io16 and _lib.load().tsr_conv2d_ex_fwd1x1_b16k(64, 256):
            # bf16 storage: the virtual 256-channel input is transformed in LDS behind the DMA (csrc/conv1x1_b16k.hip)
            wp, wis, nsc = self._packw_b16k(blk.confusion), None, -3
        else:
            (wp, wis), nsc = self._packw(c, blk.confusion), self.nsplit
        with self._timed(("fwd", 1, 64, 256)):
            conv_ex(B=c.B, H=c.H, W=c.W, src=A2, w=wp, cout=64, ks=1, out=out, out_ctot=octot, out_coff=ocoff,
                    shift=blk.confusion.bias.detach(), relu=1, res=X, nsplit=nsc, w_amax=wis,
                    out_amax=am_o)
        s.Y = Act(out, octot, ocoff, 64, amax=am_o)
        return s

    def _res_fwd(self, c, rb, F0: Act, out, octot, ocoff, am_o, new_amax, buf):
        """One ResBlock (reference model/tactileSR_model.py:222-225): relu(x + conv2(relu(conv1(x))))."""
        s = _Ctx()
        s.X = F0
        s.f1 = buf(64)
        s.F1 = Act(s.f1, 64, 0, 64, amax=new_amax())
        if self._b16k(64, 64, 3) and F0.scale is None:       # bf16 storage: both convs on conv_b16k's plain mode
            (w1, wis1), (w2, wis2), nsr = (self._packw_b16k(rb.conv1), None), (self._packw_b16k(rb.conv2), None), -3
        else:
            w1, wis1 = self._packw(c, rb.conv1)
            w2, wis2 = self._packw(c, rb.conv2)
            nsr = self.nsplit
        with self._timed(("fwd", 3, 64, 64)):
            conv_ex(B=c.B, H=c.H, W=c.W, src=F0, w=w1, cout=64, ks=3, out=s.f1, out_ctot=64, out_coff=0,
                    shift=rb.conv1.bias.detach(), relu=1, nsplit=nsr, w_amax=wis1, out_amax=s.F1.amax)
            conv_ex(B=c.B, H=c.H, W=c.W, src=s.F1, w=w2, cout=64, ks=3, out=out, out_ctot=octot,
                    out_coff=ocoff, shift=rb.conv2.bias.detach(), relu=1, res=F0, nsplit=nsr,
                    w_amax=wis2, out_amax=am_o)
        s.Y = Act(out, octot, ocoff, 64, amax=am_o)
        return s

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor):
        m = self.m
        dev = x.device
        B, hin, win = x.shape[0], x.shape[2], x.shape[3]
        sf, T, A = m.scale_factor, m.seqsCnt, m.axisCnt
        H, W = hin * sf, win * sf
        HW = H * W
        c = self._new_ctx(B, H, W, dev)
        c.x, c.hin, c.win = x, hin, win
        st_entries = c.st_entries

        def buf(ch):
            return torch.empty(B * ch * HW, dtype=self.act_dtype, device=dev)

        self._weight_scales(c)
        new_amax = self._amax_pool(c, dev)
        ctot_in = x.shape[1]
        # ---- pattern stems
        c.z1, c.bn1, c.am_z1 = [], [], []
        c.catT = buf(64 * T)
        c.am_catT = new_amax()
        c.bn2 = torch.empty(4, 64 * T, dtype=torch.float32, device=dev)
        for t, seq in enumerate(m.inputLayer_pattern_list):
            z1 = buf(64)
            am = new_amax()
            self._stem(x, ctot_in, A * t, A, hin, win, sf, seq[1].weight.detach(), z1, 0, B, am)
            call("tsr_cb16_stats_b16" if self.io16 else "tsr_cb16_stats", ptr(z1), _I(64), _I(0), _I(B), _I(HW),
                 ptr(c.slab), ptr(c.slab_cnt), stream())
            v1 = self._bn_finalize(c, None, seq[2], c.slab, c.slab_cnt, st_entries, 64)
            c.z1.append(z1)
            c.bn1.append(v1)
            c.am_z1.append(am)
            v2 = self._conv_bn(c, Act(z1, 64, 0, 64, v1[0], v1[1], amax=am), seq[4], seq[5], c.catT, 64 * T, 64 * t,
                               out_amax=c.am_catT)
            c.bn2[:, 64 * t:64 * (t + 1)] = v2
        # ---- fuse conv
        c.zf = buf(64)
        am_zf = new_amax()
        c.bnf = self._conv_bn(c, Act(c.catT, 64 * T, 0, 64 * T, c.bn2[0], c.bn2[1], amax=c.am_catT),
                              m.inputContact_layer[0], m.inputContact_layer[1], c.zf, 64, 0, out_amax=am_zf)
        X = Act(c.zf, 64, 0, 64, c.bnf[0], c.bnf[1], c.bnf[2], c.bnf[3], amax=am_zf)
        # ---- MSRB chain
        c.hcat = buf(128)
        c.am_hcat = new_amax()
        c.blocks = []
        n_msrb = len(m.patternFeatureExtra_layer)
        for i, blk in enumerate(m.patternFeatureExtra_layer):
            if i == n_msrb - 1:
                out, octot, ocoff, am_o = c.hcat, 128, 64, c.am_hcat
            else:
                out, octot, ocoff, am_o = buf(64), 64, 0, new_amax()
            s = self._msrb_fwd(c, blk, X, out, octot, ocoff, am_o, new_amax, buf)
            X = s.Y
            c.blocks.append(s)
        if n_msrb == 0:
            raise _lib.TactileSRHipError("train path needs patternFeatureExtraLayerCnt >= 1")
        # ---- force branch
        c.f0 = buf(64)
        am_f0 = new_amax()
        self._stem(x, ctot_in, 0, A, hin, win, sf, m.input_layer_force[1].weight.detach(), c.f0, 1, B, am_f0)
        F0 = Act(c.f0, 64, 0, 64, amax=am_f0)
        c.res = []
        n_res = len(m.forceFeatureExtra_layer)
        if n_res == 0:
            raise _lib.TactileSRHipError("train path needs forceFeatureExtraLayerCnt >= 1")
        for i, rb in enumerate(m.forceFeatureExtra_layer):
            if i == n_res - 1:
                out, octot, ocoff, am_o = c.hcat, 128, 0, c.am_hcat
            else:
                out, octot, ocoff, am_o = buf(64), 64, 0, new_amax()
            s = self._res_fwd(c, rb, F0, out, octot, ocoff, am_o, new_amax, buf)
            F0 = s.Y
            c.res.append(s)
        # ---- head
        c.h0 = buf(128)
        if self._b16k(128, 128, 3):
            wh, wish, nsh = self._packw_b16k(m.output_layer[0]), None, -3
        else:
            (wh, wish), nsh = self._packw(c, m.output_layer[0]), self.nsplit
        conv_ex(B=B, H=H, W=W, src=Act(c.hcat, 128, 0, 128, amax=c.am_hcat), w=wh, cout=128, ks=3, out=c.h0,
                out_ctot=128, out_coff=0, relu=1, nsplit=nsh, w_amax=wish)
        out = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        call("tsr_head_fwd_b16" if self.io16 else "tsr_head_fwd", ptr(c.h0), _I(128), _I(128),
             ptr(m.output_layer[2].weight.detach()), ptr(out), _I(1), _I(B), _I(H), _I(W), stream())
        c.out = out
        self._bump_nbt(c)
        self.last_ctx = c if self.keep_ctx else None
        return out, c

    @torch.no_grad()
    def activation_masks(self, c: _Ctx, img0: int = 0, nimg: int = None):
        """The ReLU activation pattern this forward took for images [img0, img0+nimg) (default: all), as
        {name: bool (nimg,C,H,W) tensor} keyed like the CPU
        oracle's ``ReluTap`` (test plumbing: the parity tests evaluate the fp64 gradient on exactly this pattern).
        BN layers store the raw conv output z plus per-channel (scale, shift); every kernel evaluates
        ``fmaf(z, scale, shift) > 0``, whose sign equals that of the exactly evaluated z*scale+shift in fp64."""
        from .tactileSR_model import from_cb16
        m = self.m
        H, W = c.H, c.W
        B = c.B - img0 if nimg is None else nimg
        assert 0 <= img0 and B > 0 and img0 + B <= c.B

        def sub(buf, ctot):          # CB16 is image-major: a batch slice is a contiguous range
            return buf[img0 * ctot * H * W:(img0 + B) * ctot * H * W]

        def bn_mask(buf, ctot, coff, C, scale, shift):
            z = from_cb16(sub(buf, ctot), B, C, H, W, ctot, coff).double()
            return (z * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)) > 0

        def pos(a: Act):
            return from_cb16(sub(a.buf, a.ctot), B, a.c, H, W, a.ctot, a.coff) > 0

        out = {}
        T = m.seqsCnt
        for t in range(T):
            v1 = c.bn1[t]
            out[f"inputLayer_pattern_list.{t}.2"] = bn_mask(c.z1[t], 64, 0, 64, v1[0], v1[1])
            o = 64 * t
            out[f"inputLayer_pattern_list.{t}.5"] = bn_mask(c.catT, 64 * T, o, 64, c.bn2[0, o:o + 64], c.bn2[1, o:o + 64])
        out["inputContact_layer.1"] = bn_mask(c.zf, 64, 0, 64, c.bnf[0], c.bnf[1])
        for i, s in enumerate(c.blocks):
            pre = f"patternFeatureExtra_layer.{i}"
            out[pre + ".conv_3_1.1"] = bn_mask(s.cat1, 128, 0, 64, s.bn_c1[0, :64], s.bn_c1[1, :64])
            out[pre + ".conv_5_1.1"] = bn_mask(s.cat1, 128, 64, 64, s.bn_c1[0, 64:], s.bn_c1[1, 64:])
            out[pre + ".conv_3_2.1"] = bn_mask(s.cat2, 256, 0, 128, s.bn_c2[0, :128], s.bn_c2[1, :128])
            out[pre + ".conv_5_2.1"] = bn_mask(s.cat2, 256, 128, 128, s.bn_c2[0, 128:], s.bn_c2[1, 128:])
            out[pre + ".out"] = pos(s.Y)
        out["force_in"] = pos(Act(c.f0, 64, 0, 64))
        for i, s in enumerate(c.res):
            pre = f"forceFeatureExtra_layer.{i}"
            out[pre + ".conv1"] = pos(s.F1)
            out[pre + ".out"] = pos(s.Y)
        out["head0"] = pos(Act(c.h0, 128, 0, 128))
        out["out"] = c.out[img0:img0 + B] > 0
        return out

    # ------------------------------------------------------------------ backward pieces
    def _nsplit(self, B, tiles, ks, cout, cin):
        """Batch splits of the wgrad launch: slices*nsplit ~ 2 x (3 workgroups x 256 CUs) resident slots;
        work items are (image, 8x8 patch) pairs, so splits may outnumber images."""
        slices = ks * (cout // 64) * (cin // 64)
        return max(1, min(B * tiles, (1024 if self.nsplit else 1536) // slices))   # 2 (16-bit) / 3 (f32) WGs per CU

    def _wgrad(self, c, a: Act, dz: Act, conv, grads, name, with_bias):
        w = conv.weight
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
        if self.io16 and _lib.load().tsr_conv2d_wgrad_b16k(cout, cin, ks):
            a = self._plain(c, a)      # the LDS-DMA weight-gradient kernel takes its operands as they are stored
        if self.nsplit:       # 16-bit MFMA form: the library sizes the batch split for its tile shape
            ns = _lib.load().tsr_conv2d_wgrad_splits(cout, cin, ks, self.nsplit, c.B, c.H, c.W)
        else:
            ns = self._nsplit(c.B, ((c.H + 7) // 8) * ((c.W + 7) // 8), ks, cout, cin)
        n = cout * cin * ks * ks
        slab = torch.empty(ns * n, dtype=torch.float32, device=w.device)
        bslab = torch.empty(ns * cout, dtype=torch.float32, device=w.device) if with_bias else None
        with self._timed(("wgrad", ks, cout, cin)):
            if self.nsplit:
                call("tsr_conv2d_wgrad_bf16s", ptr(a.buf), _I(a.ctot), _I(a.coff), _I(cin), ptr(a.scale), ptr(a.shift),
                     ptr(dz.buf), _I(dz.ctot), _I(dz.coff), _I(cout), _I(ks), _I(self.nsplit), ptr(a.amax),
                     ptr(dz.amax), ptr(slab), ptr(bslab), _I(ns), _I(c.B), _I(c.H), _I(c.W), stream())
            else:
                call("tsr_conv2d_wgrad", ptr(a.buf), _I(a.ctot), _I(a.coff), _I(cin), ptr(a.scale), ptr(a.shift),
                     ptr(dz.buf), _I(dz.ctot), _I(dz.coff), _I(cout), _I(ks), ptr(slab), ptr(bslab), _I(ns),
                     _I(c.B), _I(c.H), _I(c.W), stream())
        gw = grads.dest(name + ".weight", w.shape)
        call("tsr_reduce_splits", ptr(slab), ptr(gw), _L(n), _I(ns), _F(1.0), stream())
        grads.put(name + ".weight", gw)
        if with_bias:
            gb = grads.dest(name + ".bias", (cout,))
            call("tsr_reduce_splits", ptr(bslab), ptr(gb), _L(cout), _I(ns), _F(1.0), stream())
            grads.put(name + ".bias", gb)

    def _dgrad(self, c, dz: Act, conv, ci0, nprime, out, out_ctot, out_coff, res: Act = None, mask: Act = None,
               bn=False, out_amax=None):
        """d(input)[ci0:ci0+nprime] of conv given dz; optional + res, ReLU mask, BN-backward sums."""
        w = conv.weight.detach().contiguous()
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
        wa = c.wamax.get(id(conv))
        # bf16 storage: the 128-channel 3x3 / 5x5 dgrads (masked or not) run conv_b16k (nsplit -3; same tensors, its own pack)
        ns = self.nsplit
        if self.io16 and dz.scale is None and (res is None or res.scale is None) and \
                (ks > 1 or (mask is not None and res is None)) and _lib.load().tsr_conv2d_ex_dgrad_b16k(nprime, cout, ks):
            ns = -3
        wp = _pack_dgrad(w, cout, cin, ks, ci0, nprime, ns, wa)
        with self._timed(("dgrad", ks, nprime, cout)):
            conv_ex(B=c.B, H=c.H, W=c.W, src=dz, w=wp, cout=nprime, ks=ks, out=out, out_ctot=out_ctot,
                    out_coff=out_coff, res=res, epi_mode=2 if mask is not None else 0, mask=mask, bn=bn,
                    slab=c.slab if bn else None, slab_cnt=None, nsplit=ns, w_amax=wa,
                    out_amax=out_amax)
        c.last_entries = self._entries(c, nprime, ks, ns)      # what a following _bn_bwd reduces

    def _bn_bwd(self, c, g_buf, g_ctot, g_coff, z: Act, zoff, C, bn_vec, bn_mod, grads, name, out_amax=None, gnames=None):
        """Finish BatchNorm backward for C channels whose masked gradient g sits in g_buf (slab sums
        were just produced by the dgrad epilogue over the same C channels).  `gnames` = (weight, bias) parameter names:
        dgamma / dbeta are then written straight into their gradient slots (no copy launches)."""
        dev = g_buf.device
        out = torch.empty(5, C, dtype=torch.float32, device=dev)
        dg, db = (grads.dest(gnames[0], (C,)), grads.dest(gnames[1], (C,))) if gnames else (out[0], out[1])
        call("tsr_bn_bwd_finalize", ptr(c.slab), _I(c.last_entries), _I(C), _D(float(c.B * c.HW)),
             ptr(bn_vec[0]), ptr(bn_vec[2]), ptr(bn_vec[3]), ptr(dg), ptr(db), ptr(out[2]), ptr(out[3]),
             ptr(out[4]), ptr(c.work), stream())
        if gnames:
            grads.put(gnames[0], dg)
            grads.put(gnames[1], db)
        if self.io16:
            call("tsr_bn_bwd_apply_b16", ptr(g_buf), _I(g_ctot), _I(g_coff), ptr(z.buf), _I(z.ctot), _I(z.coff + zoff),
                 ptr(out[2]), ptr(out[3]), ptr(out[4]), _I(C), _I(c.B), _I(c.HW), stream())
        else:
            call("tsr_bn_bwd_apply", ptr(g_buf), _I(g_ctot), _I(g_coff), ptr(z.buf), _I(z.ctot), _I(z.coff + zoff),
                 ptr(out[2]), ptr(out[3]), ptr(out[4]), _I(C), _I(c.B), _I(c.HW), ptr(out_amax), stream())
        return out   # rows 0,1 = dgamma, dbeta

    def _res_bwd(self, c, s, rb, name, dpre: Act, grads, new_amax, buf, mask_input=True):
        """Backward of one ResBlock.  `dpre` = gradient w.r.t. the block output BEFORE its ReLU; returns the gradient
        w.r.t. the block input -- masked by the input's own ReLU pattern (`mask_input`, the chained engine: the input is
        the previous block's post-ReLU output) or plain (standalone module)."""
        name = name + "." if name else ""          # (standalone module: parameter names carry no prefix)
        F1 = s.F1
        self._wgrad(c, F1, dpre, rb.conv2, grads, name + "conv2", True)
        d1 = buf(64)
        D1 = Act(d1, 64, 0, 64, amax=new_amax())
        self._dgrad(c, dpre, rb.conv2, 0, 64, d1, 64, 0, mask=F1, out_amax=D1.amax)
        self._wgrad(c, s.X, D1, rb.conv1, grads, name + "conv1", True)
        d0 = buf(64)
        am = new_amax()
        self._dgrad(c, D1, rb.conv1, 0, 64, d0, 64, 0, res=dpre, mask=s.X if mask_input else None, out_amax=am)
        return Act(d0, 64, 0, 64, amax=am)

    def _msrb_bwd(self, c, s, blk, name, dpre: Act, grads, new_amax, buf, tag="msrb", mask_input=True):
        """Backward of one MSRB.  `dpre` = gradient w.r.t. the block output BEFORE its ReLU; returns the gradient w.r.t.
        the block input: masked by the input's ReLU pattern / with the BatchNorm-backward sums of a virtual input
        (`mask_input`, the chained engine) or plain (standalone module)."""
        name = name + "." if name else ""          # (standalone module: parameter names carry no prefix)
        # confusion 1x1: a = relu(bn(cat2)), dz = dpre
        self._wgrad(c, s.A2, dpre, blk.confusion, grads, name + "confusion", True)
        g2 = buf(256)
        am_g2 = [new_amax(), new_amax()]
        for half, (cv, bnm, nm) in enumerate(((blk.conv_3_2[0], blk.conv_3_2[1], "conv_3_2"),
                                              (blk.conv_5_2[0], blk.conv_5_2[1], "conv_5_2"))):
            o = 128 * half
            mk = Act(s.cat2, 256, o, 128, s.bn_c2[0, o:o + 128], s.bn_c2[1, o:o + 128], s.bn_c2[2, o:o + 128],
                     s.bn_c2[3, o:o + 128])
            self._dgrad(c, dpre, blk.confusion, o, 128, g2, 256, o, mask=mk, bn=True)
            self._bn_bwd(c, g2, 256, o, Act(s.cat2, 256, 0, 256), o, 128, s.bn_c2[:, o:o + 128], bnm, grads,
                         nm, out_amax=am_g2[half], gnames=(f"{name}{nm}.1.weight", f"{name}{nm}.1.bias"))
        if self.debug is not None:
            self.debug[f"{tag}.dz2"] = g2.clone()
        DZ32, DZ52 = Act(g2, 256, 0, 128, amax=am_g2[0]), Act(g2, 256, 128, 128, amax=am_g2[1])
        self._wgrad(c, s.A1, DZ32, blk.conv_3_2[0], grads, f"{name}conv_3_2.0", True)
        self._wgrad(c, s.A1, DZ52, blk.conv_5_2[0], grads, f"{name}conv_5_2.0", True)
        g1 = buf(128)
        self._dgrad(c, DZ32, blk.conv_3_2[0], 0, 128, g1, 128, 0)
        mk = Act(s.cat1, 128, 0, 128, s.bn_c1[0], s.bn_c1[1], s.bn_c1[2], s.bn_c1[3])
        self._dgrad(c, DZ52, blk.conv_5_2[0], 0, 128, g1, 128, 0, res=Act(g1, 128, 0, 128), mask=mk, bn=True)
        am_g1 = new_amax()
        r = self._bn_bwd(c, g1, 128, 0, Act(s.cat1, 128, 0, 128), 0, 128, s.bn_c1, None, grads, "", out_amax=am_g1)
        grads.put_copy(f"{name}conv_3_1.1.weight", r[0, :64])
        grads.put_copy(f"{name}conv_3_1.1.bias", r[1, :64])
        grads.put_copy(f"{name}conv_5_1.1.weight", r[0, 64:])
        grads.put_copy(f"{name}conv_5_1.1.bias", r[1, 64:])
        del g2
        if self.debug is not None:
            self.debug[f"{tag}.dz1"] = g1.clone()
        DZ31, DZ51 = Act(g1, 128, 0, 64, amax=am_g1), Act(g1, 128, 64, 64, amax=am_g1)
        self._wgrad(c, s.X, DZ31, blk.conv_3_1[0], grads, f"{name}conv_3_1.0", True)
        self._wgrad(c, s.X, DZ51, blk.conv_5_1[0], grads, f"{name}conv_5_1.0", True)
        dx = buf(64)
        self._dgrad(c, DZ31, blk.conv_3_1[0], 0, 64, dx, 64, 0, res=dpre)
        virtual = mask_input and s.X.scale is not None
        am = new_amax()
        self._dgrad(c, DZ51, blk.conv_5_1[0], 0, 64, dx, 64, 0, res=Act(dx, 64, 0, 64),
                    mask=s.X if mask_input else None, bn=virtual, out_amax=None if virtual else am)
        return Act(dx, 64, 0, 64, amax=am)

    # ------------------------------------------------------------------ backward
    def backward(self, c: _Ctx, dout: torch.Tensor):
        m = self.m
        dev = dout.device
        B, H, W, HW = c.B, c.H, c.W, c.HW
        from ..ddp import GradSink
        grads = GradSink(self, dict(m.named_parameters()), dev, token=c)

        def buf(ch):
            return torch.empty(B * ch * HW, dtype=self.act_dtype, device=dev)

        new_amax = self._amax_pool(c, dev)      # a gradient tensor consumed by an MFMA launch carries max|.|
        dout = dout.contiguous().float()
        # ---- head: out = relu(conv(h0)), h0 = relu(conv(hcat))
        ns = max(1, min(B * (8 if H > 64 else 1), 2048))     # (image split, row band) entries: ~8 resident workgroups per CU
        dz_h0 = buf(128)
        am_dzh0 = new_amax()
        wslab = torch.empty(ns * 128 * 9, dtype=torch.float32, device=dev)
        if self.io16:
            call("tsr_head_bwd_b16", ptr(dout), ptr(c.out), ptr(c.h0), _I(128), _I(128),
                 ptr(m.output_layer[2].weight.detach()), ptr(dz_h0), _I(128), ptr(wslab), _I(ns), _I(B), _I(H), _I(W),
                 stream())
        else:
            call("tsr_head_bwd", ptr(dout), ptr(c.out), ptr(c.h0), _I(128), _I(128),
                 ptr(m.output_layer[2].weight.detach()), ptr(dz_h0), _I(128), ptr(wslab), _I(ns), _I(B), _I(H), _I(W),
                 ptr(am_dzh0), stream())
        gw = grads.dest("output_layer.2.weight", m.output_layer[2].weight.shape)
        call("tsr_reduce_splits", ptr(wslab), ptr(gw), _L(128 * 9), _I(ns), _F(1.0), stream())
        grads.put("output_layer.2.weight", gw)
        DZ = Act(dz_h0, 128, 0, 128, amax=am_dzh0)
        HC = Act(c.hcat, 128, 0, 128, amax=c.am_hcat)
        self._wgrad(c, HC, DZ, m.output_layer[0], grads, "output_layer.0", False)
        g_hcat = buf(128)
        am_ghcat = new_amax()
        self._dgrad(c, DZ, m.output_layer[0], 0, 128, g_hcat, 128, 0, mask=HC, out_amax=am_ghcat)
        if self.debug is not None:
            self.debug["dz_h0"] = dz_h0.clone()
            self.debug["g_hcat"] = g_hcat.clone()
        del dz_h0

        # ---- force branch (ResBlocks, reversed); gradient w.r.t. block output pre-ReLU in `dpre`
        dpre = Act(g_hcat, 128, 0, 64, amax=am_ghcat)
        for i in reversed(range(len(c.res))):
            dpre = self._res_bwd(c, c.res[i], m.forceFeatureExtra_layer[i], f"forceFeatureExtra_layer.{i}", dpre, grads,
                                 new_amax, buf)
        ns = max(1, min(B, 2048))     # image splits: ~8 resident workgroups per CU hide the load latency
        sslab = torch.empty(ns * 64 * 27, dtype=torch.float32, device=dev)
        call("tsr_stem_wgrad_b16" if self.io16 else "tsr_stem_wgrad", ptr(c.x), _I(c.x.shape[1]), _I(0), _I(c.hin),
             _I(c.win), _I(m.scale_factor), ptr(dpre.buf), _I(dpre.ctot), _I(dpre.coff), ptr(sslab), _I(ns), _I(B), stream())
        gw = grads.dest("input_layer_force.1.weight", m.input_layer_force[1].weight.shape)
        call("tsr_reduce_splits", ptr(sslab), ptr(gw), _L(64 * 27), _I(ns), _F(1.0), stream())
        grads.put("input_layer_force.1.weight", gw)

        # ---- pattern branch: MSRB blocks reversed
        dpre = Act(g_hcat, 128, 64, 64, amax=am_ghcat)
        for i in reversed(range(len(c.blocks))):
            dpre = self._msrb_bwd(c, c.blocks[i], m.patternFeatureExtra_layer[i], f"patternFeatureExtra_layer.{i}", dpre,
                                  grads, new_amax, buf, tag=f"msrb{i}")
        # X of block 0 is the fuse conv's relu(bn(zf)): finish its BN backward -> dzf
        self._bn_bwd(c, dpre.buf, 64, 0, Act(c.zf, 64, 0, 64), 0, 64, c.bnf, None, grads, "", out_amax=dpre.amax,
                     gnames=("inputContact_layer.1.weight", "inputContact_layer.1.bias"))
        T = m.seqsCnt
        AT = Act(c.catT, 64 * T, 0, 64 * T, c.bn2[0], c.bn2[1], c.bn2[2], c.bn2[3], amax=c.am_catT)
        self._wgrad(c, AT, dpre, m.inputContact_layer[0], grads, "inputContact_layer.0", False)
        gT = buf(64 * T)
        for t, seq in enumerate(m.inputLayer_pattern_list):
            name = f"inputLayer_pattern_list.{t}"
            o = 64 * t
            mk = Act(c.catT, 64 * T, o, 64, c.bn2[0, o:o + 64], c.bn2[1, o:o + 64], c.bn2[2, o:o + 64],
                     c.bn2[3, o:o + 64])
            self._dgrad(c, dpre, m.inputContact_layer[0], o, 64, gT, 64 * T, o, mask=mk, bn=True)
            am_gT = new_amax()
            self._bn_bwd(c, gT, 64 * T, o, Act(c.catT, 64 * T, 0, 64 * T), o, 64, c.bn2[:, o:o + 64], None, grads,
                         "", out_amax=am_gT, gnames=(name + ".5.weight", name + ".5.bias"))
            DZ2 = Act(gT, 64 * T, o, 64, amax=am_gT)
            v1 = c.bn1[t]
            A1 = Act(c.z1[t], 64, 0, 64, v1[0], v1[1], v1[2], v1[3], amax=c.am_z1[t])
            self._wgrad(c, A1, DZ2, seq[4], grads, name + ".4", False)
            g1 = buf(64)
            self._dgrad(c, DZ2, seq[4], 0, 64, g1, 64, 0, mask=A1, bn=True)
            self._bn_bwd(c, g1, 64, 0, Act(c.z1[t], 64, 0, 64), 0, 64, v1, None, grads, "",
                         gnames=(name + ".2.weight", name + ".2.bias"))
            ns = max(1, min(B * (8 if H > 64 else 1), 2048))     # (image split, row band) entries
            sslab = torch.empty(ns * 64 * 27, dtype=torch.float32, device=dev)
            call("tsr_stem_wgrad_b16" if self.io16 else "tsr_stem_wgrad", ptr(c.x), _I(c.x.shape[1]), _I(m.axisCnt * t),
                 _I(c.hin), _I(c.win), _I(m.scale_factor), ptr(g1), _I(64), _I(0), ptr(sslab), _I(ns), _I(B), stream())
            gw = grads.dest(name + ".1.weight", seq[1].weight.shape)
            call("tsr_reduce_splits", ptr(sslab), ptr(gw), _L(64 * 27), _I(ns), _F(1.0), stream())
            grads.put(name + ".1.weight", gw)
        return grads.finalize()


class TactileSRTrainFn(torch.autograd.Function):
    """out = TactileSR_train_forward(x; params).  ``params`` are passed only so that autograd
    routes their gradients; the engine reads them from the module."""

    @staticmethod
    def forward(ctx, engine: TrainEngine, names, x, *params):
        out, c = engine.forward(x)
        ctx.engine, ctx.c, ctx.names = engine, c, names
        if any(ctx.needs_input_grad):
            from ..ddp import note_forward
            note_forward(engine, c)          # the arena is handed out only to the sole outstanding application
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = ctx.engine.backward(ctx.c, dout)
        ctx.c = None
        missing = [n for n in ctx.names if n not in grads]
        if missing:
            raise _lib.TactileSRHipError(f"backward produced no gradient for {missing[:4]}...")
        return (None, None, None) + tuple(grads[n] for n in ctx.names)


class BlockEngine(TrainEngine):
    """Train-mode forward / backward of ONE standalone ``MSRB`` or ``ResBlock`` module on an NCHW tensor (reference
    model/tactileSR_model.py:196-206,222-225 are callable modules): the same kernels and the same per-block code as the
    whole-network engine, with NCHW <-> CB16 conversion at the boundary and the gradient w.r.t. the block input
    returned unmasked."""

    def __init__(self, block, kind: str, impl: str = "fp16x3"):
        super().__init__(None, impl)
        assert kind in ("msrb", "res")
        self.block, self.kind = block, kind

    def _mfma_convs(self):
        b = self.block
        if self.kind == "msrb":
            return [b.conv_3_1[0], b.conv_5_1[0], b.conv_3_2[0], b.conv_5_2[0], b.confusion]
        return [b.conv1, b.conv2]

    def _amax_pool(self, c, dev):
        pool = torch.zeros(64, dtype=torch.float32, device=dev) if self.f16 else None
        state = [0]

        def new():
            if pool is None:
                return None
            i = state[0]
            state[0] += 1
            return pool[i:i + 1]
        return new

    def forward(self, x: torch.Tensor):
        from .tactileSR_model import to_cb16, from_cb16
        B, C, H, W = x.shape
        dev = x.device
        c = self._new_ctx(B, H, W, dev)
        self._weight_scales(c)
        new_amax = self._amax_pool(c, dev)

        def buf(ch):
            return torch.empty(B * ch * H * W, dtype=self.act_dtype, device=dev)

        am_x = new_amax()
        if am_x is not None:
            am_x.copy_(x.abs().amax())          # device-side: the operand scale of the first convs
        X = Act(to_cb16(x).to(self.act_dtype), 64, 0, 64, amax=am_x)
        out, am_o = buf(64), new_amax()
        fwd = self._msrb_fwd if self.kind == "msrb" else self._res_fwd
        c.s = fwd(c, self.block, X, out, 64, 0, am_o, new_amax, buf)
        self._bump_nbt(c)
        self.last_ctx = c if self.keep_ctx else None
        return from_cb16(out, B, 64, H, W), c

    def backward(self, c: _Ctx, dout: torch.Tensor):
        from ..ddp import GradSink
        from .tactileSR_model import to_cb16, from_cb16
        dev = dout.device
        B, H, W = c.B, c.H, c.W
        grads = GradSink(self, dict(self.block.named_parameters()), dev, token=c)
        new_amax = self._amax_pool(c, dev)

        def buf(ch):
            return torch.empty(B * ch * H * W, dtype=self.act_dtype, device=dev)

        # gradient w.r.t. the block output BEFORE its ReLU (in the whole-network engine the consumer's dgrad epilogue
        # applies this mask; here it is one elementwise pass over the boundary tensor)
        d = (to_cb16(dout.contiguous().float()) * (c.s.Y.buf > 0)).to(self.act_dtype)
        am = new_amax()
        if am is not None:
            am.copy_(d.abs().amax())
        dpre = Act(d, 64, 0, 64, amax=am)
        if self.kind == "msrb":
            dx = self._msrb_bwd(c, c.s, self.block, "", dpre, grads, new_amax, buf, mask_input=False)
        else:
            dx = self._res_bwd(c, c.s, self.block, "", dpre, grads, new_amax, buf, mask_input=False)
        return from_cb16(dx.buf, B, 64, H, W), grads.finalize()


class BlockTrainFn(torch.autograd.Function):
    """y = block(x) for a standalone MSRB / ResBlock in train mode; gradients for x and the block's parameters."""

    @staticmethod
    def forward(ctx, engine: BlockEngine, names, x, *params):
        out, c = engine.forward(x)
        ctx.engine, ctx.c, ctx.names = engine, c, names
        if any(ctx.needs_input_grad):
            from ..ddp import note_forward
            note_forward(engine, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, grads = ctx.engine.backward(ctx.c, dout)
        ctx.c = None
        missing = [n for n in ctx.names if n not in grads]
        if missing:
            raise _lib.TactileSRHipError(f"backward produced no gradient for {missing[:4]}...")
        return (None, None, dx) + tuple(grads[n] for n in ctx.names)
