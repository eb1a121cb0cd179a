#!/usr/bin/env python3
"""Headline benchmark: TactileSR 4x4 -> 40x40 SR samples/s on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

Workload (configs[1] of BASELINE.json): `tactileSR_model batch=4096 fp32 on 1xMI355X`:
one step = one eval-mode forward of TactileSR(scale_factor=10, seqsCnt=1) over a
synthetic batch of 4096 taxel frames per GPU, inputs resident in HBM, fp32 MFMA.
N>1: one process per GPU (torch.distributed / RCCL rendezvous); inference shards by
sample with no data-path collective ("weak" scaling, 4096 frames per GPU).

`--gpus N` without a torchrun environment starts the N ranks itself (child processes, one per GPU).

The default run (`python bench.py`, N = 1, mode infer) prints the headline line AND, under "legs", short runs of every
other BASELINE configuration with the same fields (value, ms_per_step, roofline of the leg's dominant kernel,
cpu_baseline): train B=2048 (x5 steps) and B=8192 (x2, configs[3] per-GPU batch), bf16-storage eval (x10) and bf16
train B=8192 (x2; configs[2]), tPSFNet train + forward (x20; configs[2]), tactileSRSeqs eval / train (x3; configs[4]
shape).  `--no-legs` prints the headline only.

One JSON line on rank 0 carries the contract fields plus
  roofline     - the dominant kernel (5x5 128->128 conv, 54 % of all FLOPs), timed live
                 with HIP events on the launch stream inside the timed region;
                 `achieved` / `frac` are ALGORITHMIC FLOP/s (direct-conv MACs x 2) against the peak of the
                 pipe the kernel runs on; the executed-product rate (x3 for fp16x3) is `mfma_pipe_util`;
  cpu_baseline - the CPU oracle (a port of the reference's torch CPU path) on the host
                 cores, bounded sample (configs[0]: B=32): the trainer STEP (train_cal_loss + backward +
                 Adam, cpu/trainer.py:346-362) and the eval forward; rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FWD_FLOP_PER_SAMPLE = 14_642_380_800          # SURVEY.md section 8(d), sf=10, T=1, direct conv
C5_FLOP_PER_SAMPLE = 2 * 1600 * 128 * 128 * 25  # one 5x5 128->128 conv launch, per sample
PEAK_F32_MFMA = 157.3e12                      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MFMA = 2500e12                      # MI355X_MICROARCH.md: dense bf16 MFMA peak
LAYER_BYTES_PER_SAMPLE = 48.38e6              # SURVEY.md section 8(d), layer-wise fp32 bytes


def make_model(args):
    """(model, LR channels, HR side, config overrides, fwd FLOP/sample): every conv runs at the output resolution,
    so forward MACs/sample = H*W * (number of conv weights) -- 14,642,380,800 FLOP for the default model."""
    import tactilesr_amd
    kw = dict(scale_factor=25, seqsCnt=8) if args.seqs else {}
    model = tactilesr_amd.TactileSR(**kw)
    sf, T = model.scale_factor, model.seqsCnt
    nw = sum(p.numel() for n, p in model.named_parameters() if p.dim() == 4)
    flop = 2 * (4 * sf) ** 2 * nw
    assert args.seqs or flop == FWD_FLOP_PER_SAMPLE
    return model, 3 * T, 4 * sf, dict(scale_factor=sf, seqsCnt=T), flop


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks here, one child process per GPU, before
    anything touches the GPU in this process (children are fresh interpreters: subprocess, never exec), wait, and
    exit with the worst return code.  Under torchrun (WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def host_cores():
    # the GPU box exposes all host cores but a 1-GPU job's CPU share is 16 of them
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(16, avail))


def pmc_traffic(kernel_substr, mode="eval"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN[_train]_pmc_summary.json, made by tools/collect_profiles.sh + tools/summarize_prof.py: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 FETCH x2 correction).  PMC
    counters cannot be read from inside the process, so bench.py cites the latest pass."""
    import glob
    pat = "r[0-9][0-9]_pmc_summary.json" if mode == "eval" else f"r[0-9][0-9]_{mode}_pmc_summary.json"      # eval | train | bf16 | trainbf16
    files = sorted(glob.glob(os.path.join(REPO, "profiles", pat)))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], REPO)
    from tactilesr_amd import build as _b
    took = d.get("_meta", {}).get("files")
    if not took:
        return None, f"stale: {rel} predates the source hashes -- re-collect (tools/collect_profiles.sh)"
    # the counters are quoted only while the file that defines the kernel and every shared header are what they were
    # when the pass was taken
    now = _b.kernel_source_state(kernel_substr)
    changed = sorted(f for f, v in now.items() if took.get(f) != v)
    if changed:
        return None, f"stale: {', '.join(changed)} changed since {rel} was collected -- re-collect (tools/collect_profiles.sh)"
    for k, v in d.items():
        if not k.startswith("_") and kernel_substr in k and "hbm_bytes_per_launch" in v:
            return v["hbm_bytes_per_launch"], f"{rel} (kernel '{k}', sources unchanged, commit {d['_meta'].get('summarized_at_commit')})"
    return None, f"no kernel matching '{kernel_substr}' in {rel}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline only (skip the short legs of the other configurations)")
    ap.add_argument("--broadcast-buffers", action="store_true",
                    help="train, N > 1: rank 0's BatchNorm statistics before every forward (torch-DDP default semantics)")
    ap.add_argument("--impl", choices=["fp16x3", "bf16x6", "f32", "bf16x3", "bf16"], default="fp16x3",
                    help="conv arithmetic of the timed path (all accumulate in fp32): fp16x3 = fp32 operands scaled by "
                         "powers of two and split into 2 fp16 planes, 3 f16-MFMA products (fp32-grade, default); "
                         "bf16x6 = 3 bf16 planes, 6 products (fp32-equivalent); f32 = fp32 MFMA; "
                         "bf16x3 / bf16 = reduced precision (not the headline)")
    ap.add_argument("--no-fuse-pair", action="store_true",
                    help="eval: two launches per MSRB stage 1 instead of the one-launch pair form (same arithmetic; A/B)")
    ap.add_argument("--seqs", action="store_true",
                    help="tactileSRSeqs shape (BASELINE configs[4]): TactileSR(scale_factor=25, seqsCnt=8), 4x4x24 -> "
                         "100x100, for --mode infer / train (default batch 512 / 256 per GPU)")
    ap.add_argument("--mode", choices=["infer", "train", "tpsf"], default="infer",
                    help="infer = BASELINE configs[1] (headline); train = data-parallel train step "
                         "(train_cal_loss + backward + RCCL grad all-reduce + Adam), configs[3] shape")
    args = ap.parse_args()
    args.impl_given = any(x == "--impl" or x.startswith("--impl=") for x in sys.argv[1:])
    spawn_ranks_if_needed(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != max(1, args.gpus) and "WORLD_SIZE" in os.environ:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's world size is used",
              file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local = local % torch.cuda.device_count()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TSR_BENCH_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo = rehearsal only
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group(backend=backend, init_method="env://")
    run = {"infer": run_infer, "train": run_train, "tpsf": run_tpsf}[args.mode]
    res = run(args, world, rank, dev)
    if rank == 0 and world == 1 and args.mode == "infer" and not args.no_legs and not args.seqs and not args.impl_given \
            and args.batch == 4096:
        res["legs"] = run_legs(args, dev)
    if rank == 0:
        from tactilesr_amd import _lib
        res["library"] = {"abi": _lib.ABI_VERSION, "variant_build_flags": _lib.build_flags(),        # 0 = the shipped build
                          "path": os.path.relpath(_lib.LIB_PATH, REPO)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_legs(args, dev):
    """Short runs of the other BASELINE configurations, each reported like its own bench line (see module docstring)."""
    import copy
    import gc
    legs = {}

    def leg(name, fn, **over):
        a = copy.copy(args)
        a.seqs, a.impl, a.impl_given, a.batch = False, "fp16x3", False, 4096
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            r = fn(a, 1, 0, dev)
            r["leg_wall_s"] = round(time.perf_counter() - t0, 2)
            legs[name] = r
        except Exception as e:        # a leg must never take the headline down with it
            legs[name] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            gc.collect()
            torch.cuda.empty_cache()

    leg("train_b32", run_train, mode="train", batch=32, steps=20, warmup=3)      # the reference's train_batch_size (config/default.py:46)
    leg("train_b2048", run_train, mode="train", steps=5, warmup=2, no_cpu_baseline=True)
    leg("train_b8192", run_train, mode="train", batch=8192, steps=2, warmup=1, no_cpu_baseline=True)
    leg("eval_bf16_storage", run_infer, impl="bf16", impl_given=True, steps=10, warmup=2, no_cpu_baseline=True)
    leg("train_bf16_b8192", run_train, mode="train", impl="bf16", impl_given=True, batch=8192, steps=2, warmup=1,
        no_cpu_baseline=True)
    leg("tpsf_b8192", run_tpsf, mode="tpsf", steps=20, warmup=3)
    leg("seqs_eval_b512", run_infer, seqs=True, steps=3, warmup=1)
    leg("seqs_eval_bf16_b512", run_infer, seqs=True, impl="bf16", impl_given=True, steps=3, warmup=1,    # configs[4] is a bf16 configuration
        no_cpu_baseline=True)
    leg("seqs_train_b256", run_train, mode="train", seqs=True, steps=3, warmup=1)
    leg("seqs_train_bf16_b256", run_train, mode="train", seqs=True, impl="bf16", impl_given=True, steps=3, warmup=1,   # configs[4] as worded
        no_cpu_baseline=True)
    try:
        legs["eval_b8_latency"] = small_batch_latency(dev)
    except Exception as e:
        legs["eval_b8_latency"] = {"error": f"{type(e).__name__}: {e}"}
    return legs


def small_batch_latency(dev, B=8, n=100):
    """Latency of ONE eval forward at the reference's eval batch (test_batch_size = 8, config/default.py:53;
    train/tactileSR_train.py:66-101), host call to device completion, plain launches vs HIP-graph replay
    (tactilesr_amd.model.graph.GraphedForward).  Answers whether the host's launch path bounds small batches."""
    import tactilesr_amd
    from tactilesr_amd.model.graph import GraphedForward
    torch.manual_seed(42)
    m = tactilesr_amd.TactileSR().to(dev).eval()
    x = (torch.rand(B, 3, 4, 4, generator=torch.Generator().manual_seed(42)) * 8).to(dev)
    gf = GraphedForward(m, x)

    def lat(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    plain, replay = lat(lambda: m(x)), lat(lambda: gf(x))
    return {"metric": "SR samples/sec (4x4->40x40) at the reference's eval batch (test_batch_size = 8, config/default.py:47)",
            "value": round(B / plain * 1e3, 1), "unit": "samples/s", "ms_per_step": round(plain, 4), "batch": B,
            "plain_launches_ms": round(plain, 4), "hip_graph_replay_ms": round(replay, 4),
            "samples_per_s_plain": round(B / plain * 1e3, 1), "forwards_timed": n,
            "diagnosis": "latency-bound on the device, not launch-bound: ~45 dependent launches of ~20 us, each a single "
                         "partial wave of workgroups (100 on 256 CUs) running its whole serial K loop; HIP-graph replay "
                         "(hip_graph_replay_ms) removes the host but not that chain"}


def run_infer(args, world, rank, dev):
    if world > 1:
        import torch.distributed as dist
    import tactilesr_amd
    torch.manual_seed(42)
    model, cin_lr, side, _, fwd_flop = make_model(args)
    model = model.to(dev).eval()
    model.conv_impl = args.impl
    if args.no_fuse_pair:
        model.fuse_pair = False
    B = args.batch if not (args.seqs and args.batch == 4096) else 512
    model.max_images_per_pass = B
    g = torch.Generator().manual_seed(42 + rank)
    LR = (torch.rand(B, cin_lr, 4, 4, generator=g) * 8).to(dev)
    c5_flop = 2 * side * side * 128 * 128 * 25      # one 5x5 128->128 conv launch, per sample
    fused = model.fuse_1x1 and args.impl in ("fp16x3", "bf16")
    if fused:                                       # + its half of the MSRB's 1x1 confusion (64 x 128), same launch
        c5_flop += 2 * side * side * 128 * 64

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model(LR)
    barrier()
    model._profile = {}          # HIP-event brackets around every conv launch, keyed (ks, cout)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = model(LR)
    barrier()
    dt = time.perf_counter() - t0
    prof = model._profile
    model._profile = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # secondary measurement: the strict fp32-MFMA path (v_mfma_f32_32x32x2_f32), 2 steps, same inputs
    f32_ref = None
    if args.impl != "f32" and not args.seqs and not args.impl_given:
        model.conv_impl = "f32"
        model(LR)
        barrier()
        model._profile = {}
        t1 = time.perf_counter()
        for _ in range(2):
            model(LR)
        barrier()
        dt32 = (time.perf_counter() - t1) / 2
        ev32 = model._profile.get((5, 128), [])
        model._profile = None
        model.conv_impl = args.impl
        c5 = sum(a.elapsed_time(b) for a, b in ev32) / max(1, len(ev32))
        f32_ref = {"value": round(B / dt32, 2), "unit": "samples/s per GPU", "ms_per_step": round(dt32 * 1e3, 3),
                   "kernel": "conv_mfma_f32_kernel<5,128>", "kernel_tflops": round(B * c5_flop / (c5 * 1e-3) / 1e12, 2),
                   "kernel_frac_of_f32_mfma_peak": round(B * c5_flop / (c5 * 1e-3) / PEAK_F32_MFMA, 4)}

    if rank == 0:
        total = B * world * args.steps
        value = total / dt
        ev = prof.get((5, 128), [])
        c5_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))
        nprod = {"fp16x3": 3, "bf16x6": 6, "bf16x3": 3, "bf16": 1, "f32": 1}[args.impl]
        peak = PEAK_F32_MFMA if args.impl == "f32" else PEAK_BF16_MFMA
        alg = B * c5_flop / (c5_ms * 1e-3) if ev else None       # algorithmic FLOP/s of the launch
        executed = alg * nprod if ev else None                               # MFMA FLOP/s actually executed
        k32 = args.impl == "fp16x3"                                             # conv_mfma_k32.hip (16x16x32 MFMA)
        kname = ("conv_mfma_f32_kernel<5, 128" if args.impl == "f32" else
                 ("conv_k32_kernel<5, 128, false, 2, %s" % ("true, false, 256" if fused else "false, false, 512")) if k32 else
                 # (bf16 storage: csrc/conv_b16k.hip <KS, COUT, MODE>, MODE 1 = the fused form, 0 = plain)
                 ("conv_b16k_kernel<5, 128, %d" % (1 if fused else 0))
                 if args.impl == "bf16" else
                 "conv_mfma_split16_kernel<5, 128, %d, false, %s" % ({"fp16x3": 2, "bf16x6": 3, "bf16x3": 2}[args.impl],
                                                                     "true" if args.impl == "fp16x3" else "false"))
        per_kernel = {(f"conv{k[0]}x{k[0]}_c{k[1]}" if k[0] != "pair" else "conv3x3_5x5_pair_c128"):
                      round(sum(a.elapsed_time(b) for a, b in v) / args.steps, 3)
                      for k, v in sorted(prof.items(), key=lambda kv: str(kv[0]))}
        traffic, traffic_src = pmc_traffic(kname, "bf16" if args.impl == "bf16" else "eval") if B == 4096 and not args.seqs else (None, None)
        dtype = {"fp16x3": "f32 as 2 power-of-two-scaled fp16 planes x 3 MFMA products, fp32 accumulate (fp32-grade)",
                 "bf16x6": "f32 as 3 bf16 planes x 6 MFMA products, fp32 accumulate (fp32-equivalent)",
                 "f32": "f32", "bf16x3": "bf16x3 (reduced: ~16 significand bits)",
                 "bf16": "bf16 activation storage + bf16 MFMA operands, fp32 accumulate (reduced precision: 3e-2)"}[args.impl]
        res = {
            "metric": "SR samples/sec (4x4->%dx%d)" % (side, side), "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": ("tactileSRSeqs (T=8) eval forward 4x4x24->100x100, batch=%d/GPU, fp32 in/out (BASELINE configs[4] shape)" % B)
                       if args.seqs else
                       ("tactileSR_model eval forward 4x4->40x40, batch=%d/GPU, fp32 in/out (BASELINE configs[1])" % B),
                       "batch_per_gpu": B, "scale_factor": model.scale_factor, "seqsCnt": model.seqsCnt,
                       "parallelism": f"replicas x{world}", "conv_impl": args.impl, "fwd_GFLOP_per_sample": round(fwd_flop / 1e9, 3),
                       "dist_world_size": dist.get_world_size() if world > 1 else 1,
                       "dist_backend": ("rccl(nccl)" if dist.get_backend() == "nccl" else dist.get_backend()) if world > 1 else None},
            "roofline": {"bound": "mfma", "kernel": kname + ("> (5x5 128->128 conv+BN+ReLU with half of the 1x1 confusion fused, 55% of all FLOPs)"
                                                             if fused else "> (5x5 128->128 conv+BN+ReLU, 54% of all FLOPs)"),
                         "achieved": round(alg / 1e12, 2) if alg else None, "peak": peak / 1e12,
                         "unit": "TFLOP/s", "frac": round(alg / peak, 4) if alg else None,
                         "algorithmic_flop_per_launch": B * c5_flop,
                         "mfma_products_per_mac": nprod,
                         "executed_tflops": round(executed / 1e12, 2) if executed else None,
                         "mfma_pipe_util": round(executed / peak, 4) if executed else None,
                         "algorithmic_vs_f32_mfma_peak": round(alg / PEAK_F32_MFMA, 4) if alg else None,
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": B * (256 * side * side * (2 if args.impl == "bf16" else 4)) + 128 * 128 * 25 * 4,
                         "avg_launch_ms": round(c5_ms, 3), "launches_timed": len(ev)},
            "whole_step": {"algorithmic_tflops": round(value / world * fwd_flop / 1e12, 2),
                           "frac_of_peak": round(value / world * fwd_flop / peak, 4),
                           "algorithmic_vs_f32_mfma_peak": round(value / world * fwd_flop / PEAK_F32_MFMA, 4),
                           "layerwise_GBps": None if args.seqs else round(value / world * LAYER_BYTES_PER_SAMPLE / 1e9, 1),
                           "ms_per_step_by_kernel": per_kernel},
        }
        if f32_ref is not None:
            res["f32_mfma_path"] = f32_ref
        if world == 1 and not args.no_cpu_baseline:
            res.update(cpu_seqs_baseline(model) if args.seqs else cpu_baseline_and_psnr(model, dev))
        return res
    return None


def run_train(args, world, rank, dev):
    """One step = Trainer_tactileSR.train_cal_loss + zero_grad/backward/Adam (reference
    train/tactileSR_train.py:41-51, cpu/trainer.py:346-362) on a per-GPU shard of `--batch` frames;
    N>1 adds the bucketed RCCL all-reduce of the 18.33 MB gradient, issued from inside backward (tactilesr_amd.ddp)."""
    from tactilesr_amd import ddp, optim
    from tactilesr_amd.train import tactileSR_train as TR
    import tactilesr_amd
    if world > 1:
        import torch.distributed as dist
    torch.manual_seed(42)
    model, cin_lr, side, cfg_over, fwd_flop = make_model(args)
    model = model.to(dev).train()
    model.train_impl = args.impl if args.impl_given else "fp16x3"       # explicit: no environment variable selects arithmetic
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-2)
    sync = ddp.GradSync(model, broadcast_buffers=args.broadcast_buffers) if world > 1 else None
    if sync:
        sync.broadcast_parameters(0)
    B = args.batch if args.batch != 4096 else (256 if args.seqs else 2048)
    g = torch.Generator().manual_seed(42 + rank)          # per-rank data seed
    batch = ((torch.rand(B, cin_lr, 4, 4, generator=g) * 8).to(dev), (torch.rand(B, 1, 100, 100, generator=g) * 250).to(dev))
    conf = TR.default_config()
    conf.update(cfg_over)
    eng = model.train_engine()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        TR.train_one_iter(model, opt, batch, conf, sync)
    barrier()
    eng.profile = {}             # HIP-event brackets per launch family inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ld = TR.train_one_iter(model, opt, batch, conf, sync)
    barrier()
    dt = time.perf_counter() - t0
    prof, eng.profile = eng.profile, None
    losses_all = [float(ld["total_loss"].detach())]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        lt = [torch.zeros(1, device=dev, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(lt, ld["total_loss"].detach().reshape(1).float())
        losses_all = [float(x) for x in lt]
    if rank == 0:
        value = B * world * args.steps / dt
        impl = eng.impl
        nprod, peak = {"bf16x6": (6, PEAK_BF16_MFMA), "fp16x3": (3, PEAK_BF16_MFMA),
                       "bf16": (1, PEAK_BF16_MFMA), "bf16op": (1, PEAK_BF16_MFMA)}.get(impl, (1, PEAK_F32_MFMA))
        # fwd + dgrad + wgrad, minus the dgrad of the T+1 stem convs (3->64, 3x3) whose input needs no gradient
        train_flop = 3 * fwd_flop - (model.seqsCnt + 1) * 2 * side * side * 27 * 64
        # dominant kernel of the step: the 5x5 128->128 weight-gradient launches (6 per step)
        wg = prof.get(("wgrad", 5, 128, 128), [])
        wg_ms = sum(a.elapsed_time(b) for a, b in wg) / max(1, len(wg))
        wg_flop = B * 2 * side * side * 128 * 128 * 25
        wg_alg = wg_flop / (wg_ms * 1e-3) if wg else None
        fam = {}
        for k, v in prof.items():
            fam.setdefault(k[0], 0.0)
            fam[k[0]] += sum(a.elapsed_time(b) for a, b in v) / args.steps
        # the 5x5 128->128 convolution launches of the step (forward + dgrad: the same kernel, 12 launches)
        cv = prof.get(("fwd", 5, 128, 128), []) + prof.get(("dgrad", 5, 128, 128), [])
        cv_ms = sum(a.elapsed_time(b) for a, b in cv) / max(1, len(cv))
        cv_alg = wg_flop / (cv_ms * 1e-3) if cv else None
        wname = ("wgrad_mfma_f32_kernel<5" if impl == "f32" else
                 "wgrad_k32_kernel<5, 1, 128, 128" if impl == "fp16x3" else
                 "wgrad_b16k_kernel<5, 1, 128, 128" if impl == "bf16" else     # (bf16 tensors: csrc/wgrad_b16k.hip)
                 "wgrad_tr16_kernel<5, 1, 128, 128")
        traffic, traffic_src = pmc_traffic(wname, "trainbf16" if impl == "bf16" else "train") if B == 2048 and not args.seqs else (None, None)
        res = {
            "metric": "SR train samples/sec (4x4->%dx%d)" % (side, side), "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x6": "f32 as 3 bf16 planes x 6 MFMA products, fp32 accumulate (fp32-equivalent)",
                      "fp16x3": "f32 as 2 scaled fp16 planes x 3 MFMA products, fp32 accumulate (fp32-grade)",
                      "bf16": "bf16 activation / gradient STORAGE + bf16 MFMA operands, fp32 accumulate / master weights / BN "
                              "statistics / weight gradients / Adam (reduced precision: checked against the bf16-emulating oracle)",
                      "bf16op": "bf16 MFMA operands on fp32 tensors (A/B only)"}.get(impl, "f32"),
            "data": "synthetic",
            "config": {"workload": ("tactileSRSeqs (T=8, 100x100) " if args.seqs else "TactileSR ") +
                       "train step (train_cal_loss + backward + Adam L2), fp32 params, %s activations, batch/GPU=%d "
                       "(BASELINE configs[%s" % ("bf16" if impl == "bf16" else "fp32", B, "4] shape)" if args.seqs else
                                                   (("2] / [3] per-GPU batch)" if impl == "bf16" else "3] per-GPU batch)") if B == 8192
                                                    else "3] is 8192/GPU: pass --batch 8192)")),
                       "batch_per_gpu": B, "parallelism": f"dp{world}",
                       "grad_allreduce_MB": round(sum(p.numel() for p in model.parameters()) * 4 / 1e6, 2),
                       "grad_buckets": len(eng.arena.buckets) if eng.arena is not None else None,
                       "dist_world_size": dist.get_world_size() if world > 1 else 1,
                       "dist_backend": ("rccl(nccl)" if dist.get_backend() == "nccl" else dist.get_backend()) if world > 1 else None,
                       "conv_impl": impl, "train_GFLOP_per_sample": round(train_flop / 1e9, 3)},
            "roofline": {"bound": "mfma", "kernel": wname + ",...> (weight gradient of the 5x5 128->128 convs, 6 launches/step)",
                         "achieved": round(wg_alg / 1e12, 2) if wg_alg else None, "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": round(wg_alg / peak, 4) if wg_alg else None,
                         "algorithmic_flop_per_launch": wg_flop, "mfma_products_per_mac": nprod,
                         "mfma_pipe_util": round(wg_alg * nprod / peak, 4) if wg_alg else None,
                         "avg_launch_ms": round(wg_ms, 3), "launches_timed": len(wg),
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": B * (2 * 128 * side * side * (2 if impl == "bf16" else 4)) + 128 * 128 * 25 * 4},
            "roofline_conv": {"bound": "mfma", "kernel": "5x5 128->128 convolution launches of the step (forward + dgrad, 12 per step)",
                              "achieved": round(cv_alg / 1e12, 2) if cv_alg else None, "peak": peak / 1e12, "unit": "TFLOP/s",
                              "frac": round(cv_alg / peak, 4) if cv_alg else None, "avg_launch_ms": round(cv_ms, 3),
                              "launches_timed": len(cv), "mfma_products_per_mac": nprod},
            "whole_step": {"algorithmic_tflops": round(value / world * train_flop / 1e12, 2),
                           "frac_of_peak": round(value / world * train_flop / peak, 4),
                           "mfma_pipe_util": round(value / world * train_flop * nprod / peak, 4),
                           "algorithmic_vs_f32_mfma_peak": round(value / world * train_flop / PEAK_F32_MFMA, 4),
                           "ms_per_step_by_family": {k: round(v, 2) for k, v in sorted(fam.items())},
                           "ms_per_launch": {"%s_k%d_co%d_ci%d" % k: round(sum(a.elapsed_time(b) for a, b in v) / max(1, len(v)), 3)
                                             for k, v in sorted(prof.items())}},
            "loss": float(ld["total_loss"].detach()),
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_seqs_baseline(model, train=True)["cpu_baseline"] if args.seqs else cpu_train_baseline()
        if sync is not None:
            # where in backward each gradient bucket left and how long the step then waited for the wire: the exposed
            # communication of the LAST timed step, on this rank (device clock when the tensors are on the GPU)
            res["ddp"] = sync.stats()
            res["comm_wait_ms"] = res["ddp"]["comm_wait_ms"]
        res["loss_all_ranks"] = losses_all
        return res
    return None


def cpu_train_baseline(steps=3):
    """BASELINE configs[0]: the reference trainer's step (train_cal_loss + zero_grad/backward/Adam-L2,
    train/tactileSR_train.py:41-51, cpu/trainer.py:346-362) as the CPU oracle restates it, B=32 fp32, on the host cores."""
    from oracle import tactilesr_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = O.random_state_dict(O.tactilesr_state_shapes(), 42)
    g = torch.Generator().manual_seed(42)
    LR, HR = torch.rand(32, 3, 4, 4, generator=g) * 8, torch.rand(32, 1, 100, 100, generator=g) * 250
    state = {}
    O.train_one_iter(sd, state, 1, LR, HR)          # warm-up
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        O.train_one_iter(sd, state, i + 2, LR, HR)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"value": round(32 / ts[len(ts) // 2], 2), "unit": "samples/s", "cores": torch.get_num_threads(),
            "cpu_model": cpu_model_string(), "kind": "port",
            "sample": "oracle trainer step (train_cal_loss + backward + Adam L2), B=32 fp32, median of %d "
                      "(BASELINE configs[0])" % steps}


def tpsf_traffic(B):
    """HBM bytes per tpsf_fwd_mfma_kernel launch from the committed PMC passes (same B only)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_tpsf_pmc_summary.json")))
    if not files or B != 8192:
        return None
    d = json.load(open(files[-1]))
    from tactilesr_amd import build as _b
    took = d.get("_meta", {}).get("files") or {}
    if any(took.get(f) != v for f, v in _b.kernel_source_state("tpsf_fwd_mfma_kernel").items()):
        return None                      # kernel source changed since the pass: stale
    for k, v in d.items():
        if "tpsf_fwd" in k and "hbm_bytes_per_launch" in v:
            return v["hbm_bytes_per_launch"]
    return None


def run_tpsf(args, world, rank, dev):
    """tPSFNet (reference model/tPSFNet.py:78-141): one step = Trainer_tPSF.train_cal_loss + backward + Adam
    (train/tPSFNet_train.py:180-190) on `--batch` samples (default 8192, BASELINE configs[2]); the forward-only
    rate (the dataset-generator use, data/SRdataset/depth2tactile.py:104-160) is reported beside it.  Single GPU
    or independent replicas (no gradient exchange is timed here; the MLP has 0.54 M parameters)."""
    import tactilesr_amd
    from tactilesr_amd import optim
    from tactilesr_amd.train import tPSFNet_train as TP
    if world > 1:
        import torch.distributed as dist
    torch.manual_seed(42)
    net = tactilesr_amd.tPSFNet(gama=1.4, perception_scale=None).to(dev).train()
    opt = optim.Adam(net.parameters(), lr=1e-3, weight_decay=0.0)
    B = args.batch if args.batch != 4096 else 8192
    g = torch.Generator().manual_seed(42 + rank)
    batch = ((torch.rand(B, 3, 4, 4, generator=g) * 800).to(dev), (torch.rand(B, 100, 100, generator=g) * 10).to(dev))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        loss, _ = TP.train_cal_loss(net, batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def fwd():
        with torch.no_grad():
            return net(batch[0] / 100.0, batch[1].unsqueeze(1))

    for _ in range(args.warmup):
        step()
        fwd()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(args.steps):
        fwd()
    barrier()
    dtf = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt, dtf], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dtf = float(t[0]), float(t[1])
    if rank == 0:
        value = B * world * args.steps / dt
        vf = B * world * args.steps / dtf
        fwd_bytes = 4 * (10000 + 10000 + 9801 + 16 + 48 + 3)      # depth in; HR, psf, LR_degrade out
        return {
            "metric": "tPSFNet train samples/sec (depth 100x100 -> HR 100x100 + 4x4 taxels)", "value": round(value, 1),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "tPSFNet train step (MLP + separable PSF conv fwd, analytic bwd, Adam), batch/GPU=%d "
                                   "(BASELINE configs[2] shape)" % B, "batch_per_gpu": B, "parallelism": f"replicas x{world}"},
            "forward_only": {"samples_per_s": round(vf, 1), "ms_per_batch": round(dtf / args.steps * 1e3, 3)},
            "roofline": {"bound": "hbm", "achieved": round(vf / world * fwd_bytes / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(vf / world * fwd_bytes / 8e12, 4), "traffic": tpsf_traffic(B),
                         "traffic_unit": "bytes/launch (PMC, profiles/rNN_tpsf_pmc_summary.json; null when the kernel sources changed since)",
                         "algorithmic_bytes_per_launch": B * fwd_bytes,
                         "kernel": "tpsf_fwd_mfma_kernel (+ the MLP launches: forward-only rate x %d algorithmic bytes/sample)" % fwd_bytes},
            "loss": float(loss.detach()),
            **({"cpu_baseline": cpu_tpsf_baseline()} if world == 1 and not args.no_cpu_baseline else {}),
        }
    return None


def cpu_tpsf_baseline(n=32):
    """Trainer_tPSF.train_cal_loss + backward (train/tPSFNet_train.py:180-190) as the CPU oracle restates the
    reference's per-sample python loop, on a bounded sample of the same synthetic workload."""
    from oracle import tactilesr_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = O.random_state_dict(O.tpsf_state_shapes(), 42)
    g = torch.Generator().manual_seed(42)
    LR, depth = torch.rand(n, 3, 4, 4, generator=g) * 800, torch.rand(n, 100, 100, generator=g) * 10
    geom = O.tpsf_geometry()
    ts = []
    for _ in range(3):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        t0 = time.perf_counter()
        loss = O.tpsf_train_cal_loss(leaves, LR, depth, geom=geom)
        torch.autograd.grad(loss, list(leaves.values()))
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"value": round(n / ts[1], 2), "unit": "samples/s", "cores": torch.get_num_threads(),
            "cpu_model": cpu_model_string(), "kind": "port",
            "sample": "oracle tPSFNet train_cal_loss + backward (per-sample direct 99x99 conv), B=%d fp32, median of 3" % n}


def cpu_seqs_baseline(model, train=False, n=4):
    """tactileSRSeqs shape (scale_factor=25, seqsCnt=8; 102 GFLOP per sample forward): the CPU oracle on a bounded
    sample of n frames -- eval forward, or the trainer step (loss + backward + Adam L2) when `train`."""
    from oracle import tactilesr_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    sf, T = model.scale_factor, model.seqsCnt
    g = torch.Generator().manual_seed(42)
    LR = torch.rand(n, 3 * T, 4, 4, generator=g) * 8
    HR = torch.rand(n, 1, 100, 100, generator=g) * 250
    ts = []
    state = {}
    for i in range(3):
        t0 = time.perf_counter()
        if train:
            O.train_one_iter(sd, state, i + 1, LR, HR, seqsCnt=T, scale_factor=sf)
        else:
            with torch.no_grad():
                O.tactilesr_forward(sd, LR, sf)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"cpu_baseline": {"value": round(n / ts[1], 2), "unit": "samples/s", "cores": torch.get_num_threads(),
                             "cpu_model": cpu_model_string(), "kind": "port",
                             "sample": "oracle %s, tactileSRSeqs shape (sf=%d, T=%d), B=%d fp32, median of 3"
                                       % ("trainer step (loss + backward + Adam L2)" if train else "eval forward", sf, T, n)}}


def cpu_baseline_and_psnr(model, dev):
    """Oracle (port of the reference's CPU path) timed on the host cores on a bounded sample, B=32 fp32: the eval
    forward (the CPU counterpart of the headline metric) and, under `train_step`, BASELINE configs[0] proper -- the
    reference trainer's step; the oracle is also the checker for 'PSNR vs ref' of the HIP output on the same frames."""
    from oracle import tactilesr_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(42)
    LR = torch.rand(32, 3, 4, 4, generator=g) * 8
    with torch.no_grad():
        ref = O.tactilesr_forward(sd, LR)      # warm-up
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ref = O.tactilesr_forward(sd, LR)
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    y = model(LR.to(dev)).cpu()
    err = float((y - ref).abs().max() / ref.abs().max())
    psnr = [float(O.calculation_psnr(y[i, 0], ref[i, 0], 250.0 / 10)) for i in range(y.shape[0])
            if float(((y[i] - ref[i]) ** 2).sum()) > 0]
    step = cpu_train_baseline()
    return {
        "cpu_baseline": {"value": round(32 / med, 2), "unit": "samples/s", "cores": torch.get_num_threads(),
                         "cpu_model": cpu_model_string(), "kind": "port",
                         "sample": "oracle eval forward, B=32 fp32, median of 5 (the CPU counterpart of this metric)",
                         "train_step": step},
        "parity": {"max_rel_err_vs_oracle": err, "psnr_vs_ref_db_min": round(min(psnr), 2) if psnr else None,
                   "frames": 32},
    }


if __name__ == "__main__":
    main()
